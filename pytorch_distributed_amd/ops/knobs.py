"""Tuning knobs of the native engines, resolved ONCE per engine / step program -- at construction, never inside a step.

Three sources, later ones win:

1. the defaults below (each measured on MI355X; the comment says where);
2. `Options.engine_knobs`, a dict (`engine_knobs.fc1_blocks=128` on the command line, utils/options.py);
3. the environment variable `PDA_<NAME>` of the same knob (what bench.py's pre-flight, the multi-process tests and
   the sweep scripts under tools/probes/ set for the child processes they start).

This module is the only place of the package that reads `PDA_*` tuning variables.  What used to be ablation switches
for code paths that lost their measurement (forked weight gradients, the optimizer co-launched with a weight-gradient
GEMM, gather-form data gradients on the bf16 engine, fc1's atomically accumulated forward, the unfused conv stacks as
an opt-out) are gone together with those paths; README.md ("Measured and removed") keeps the numbers.
"""
import os

DEFAULTS = {
    # split-K targets: workgroups per launch the weight-gradient / fc1-forward GEMMs are cut for
    # (tools/probes/x3_launch_size_sweep.sh, profiles/r2_x3_launch_size_sweep.txt)
    "wgrad_blocks": 256,
    "pair_blocks": 512,
    "fc1_blocks": 256,
    "x3_wgrad_mult": 2,         # fp32-class engine: multiplier on wgrad_blocks (more, shorter slices: 3.51k -> 3.77k)
    "x3_pair_mult": 1,          # ... and on pair_blocks for the conv2 + conv3 launch
    # batch from which the per-image conv2 + conv3 weight-gradient kernel is used (bf16 engine): at 4096 it is
    # worth +5.8 % (1224 -> 1295 updates/s), at 1024 it loses (3528 -> 3247)
    "wgrad_stack_min": 4096,
    # graph replays the host may run ahead of the GPU (profiles/r3_node_learner_loop.txt)
    "max_inflight_replays": 3,
    # data parallel: gradient wire format of the fc tail (fp32 | bf16 | measure), the all-reduce (auto | peer | rccl),
    # how the step is captured (auto | whole | split; bench.py / the learner set it from parallel/preflight.py), and
    # the peer kernel's workgroup budget (0 = its own plan; shared-card rehearsals of 8 ranks need <= 64).
    # "measure": the candidates are TIMED on the real fc-tail bucket when the step program is built -- the peer kernel
    # with the plain fp32 wire (8 bytes per element), the packed fp32 wire (5.33 bytes, still exact), a bf16 wire
    # (bf16 engines only: the fp32-class engine keeps exact fp32 sums) and, with dp_allreduce=auto on an RCCL group,
    # the process group's all-reduce -- and the fastest is used
    # (parallel/peer_allreduce.py:choose_grad_allreduce; the timings go into bench.py's JSON as "dp_wire_choice").
    "dp_grad_dtype": "measure",
    "dp_allreduce": "auto",
    "dp_graph": "auto",
    "peer_workgroups": 0,
    # debugging: roctx ranges around the learner phases of an eager (non-graph) run; the learner loop's own
    # time-attribution profile (learner_loop_profile.json, agents/learners.py:_LoopProfile)
    "roctx": 0,
    "learner_prof": 0,
}

_CHOICES = {"dp_grad_dtype": ("fp32", "bf16", "measure"), "dp_allreduce": ("auto", "peer", "rccl"),
            "dp_graph": ("auto", "whole", "split")}


class Knobs(object):
    def __init__(self, values):
        self.__dict__.update(values)

    def as_dict(self):
        return dict(self.__dict__)

    def __repr__(self):
        return "Knobs(%s)" % ", ".join("%s=%r" % kv for kv in sorted(self.__dict__.items()))


def _coerce(name, value):
    default = DEFAULTS[name]
    if isinstance(default, int):
        value = int(value)
        if value < 0:
            raise ValueError("engine knob %s must be >= 0, not %r" % (name, value))
        return value
    value = str(value).lower()
    if name in _CHOICES and value not in _CHOICES[name]:
        raise ValueError("engine knob %s must be one of %s, not %r" % (name, "|".join(_CHOICES[name]), value))
    return value


def resolve(args=None, **overrides):
    """-> Knobs.  args: an Options object (its `engine_knobs` dict) or None; overrides: explicit values (tests)."""
    values = dict(DEFAULTS)
    given = dict(getattr(args, "engine_knobs", None) or {})
    for name, value in given.items():
        if name not in DEFAULTS:
            raise ValueError("unknown engine knob %r (known: %s)" % (name, ", ".join(sorted(DEFAULTS))))
        values[name] = _coerce(name, value)
    for name in DEFAULTS:
        env = os.environ.get("PDA_" + name.upper())
        if env not in (None, ""):
            values[name] = _coerce(name, env)
    for name, value in overrides.items():
        if name not in DEFAULTS:
            raise ValueError("unknown engine knob %r" % (name,))
        values[name] = _coerce(name, value)
    return Knobs(values)
