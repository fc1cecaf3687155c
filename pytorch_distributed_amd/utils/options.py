"""Configuration: CONFIGS table + Params/EnvParams/MemoryParams/ModelParams/
AgentParams/Options, same names and fields as the reference (utils/options.py:
10-175) so "edit options.py, run main.py" still works -- but every object is an
ordinary picklable *instance* (the reference builds the nested params as class
attributes at import, utils/options.py:171-175), and any field can be overridden
without editing the file:

    Options(config=3, overrides={"num_actors": 8, "agent_params.nstep": 5})
    PDA_OVERRIDES="num_actors=2,agent_params.steps=200" python main.py
    python main.py config=4 agent_params.batch_size=4096

Extra fields (absent in the reference) configure the MI355X-native paths:
device selection, HBM replay, PER alpha/beta, batched actor inference,
data-parallel learner ranks.
"""
import ast
import os

import numpy as np
import torch
import torch.nn as nn

from .random_process import OrnsteinUhlenbeckProcess

CONFIGS = [
    # agent_type, env_type,    game,        memory_type,  model_type
    ["dqn",  "atari",     "pong",      "shared",     "dqn-cnn"],   # 0 reference row (ALE; synthetic frames if ALE is absent)
    ["dqn",  "synthetic", "pong",      "shared",     "dqn-cnn"],   # 1 plumbing: 2 actors, nstep=1, CPU learner
    ["dqn",  "synthetic", "pong",      "device",     "dqn-cnn"],   # 2 8 actors, nstep=5, one MI355X, batched actor inference
    ["ddpg", "pendulum",  "pendulum",  "shared",     "ddpg-mlp"],  # 3 Distributed DDPG, 8 actors, n-step critic targets
    ["dqn",  "synthetic", "pong",      "device-per", "dqn-cnn"],   # 4 Ape-X: 256 actors, 2M HBM replay, DP learners
    ["dqn",  "synthetic", "pong",      "device-per", "dqn-cnn"],   # 5 large-batch (4096) prioritized DQN
    ["dqn",  "chain",     "chain",     "shared",     "dqn-mlp"],   # 6 tiny learnable task for sanity tests
    ["ddpg", "pendulum",  "pendulum",  "device",     "ddpg-mlp"],  # 7 DDPG with HBM replay + native kernels
    ["dqn",  "synthetic", "minipong",  "device-dedup", "dqn-cnn"], # 8 row 2 on the built-in pixel Pong (envs/minipong.py)
    ["dqn",  "synthetic", "pong",      "device-dedup-per", "dqn-cnn"],   # 9 row 4 on frame-deduplicated replay (memories/dedup_memory.py)
]

# per-config presets applied after the defaults (dotted keys reach nested params)
PRESETS = {
    1: {"num_actors": 2, "agent_params.nstep": 1, "device": "cpu"},
    2: {"num_actors": 8, "agent_params.nstep": 5, "actor_inference": "batched"},
    # 3 = the reference's DDPG settings, untouched (lr 1e-4, 5-step targets, the actor loss's gradient left in the
    # critic, no actor/learner coupling).  On the built-in Pendulum they do not learn -- with the fp32 torch core as
    # little as with the native one: see row 7 and BASELINE.md "DDPG recipe".
    3: {"num_actors": 8, "agent_params.nstep": 5},
    4: {"num_actors": 256, "actor_inference": "batched", "memory_params.memory_size": 2 * 1024 * 1024,
        "memory_params.enable_per": True, "agent_params.nstep": 5},
    # 9 = row 4 with every frame stored once: 2 M transitions in 18.5 GB of planes instead of 113 GB of frame stacks,
    # a seventh of the ingest bytes (whole job, 256 envs: 948 k env frames/s + 8.1 k updates/s against 856 k + 7.8 k)
    9: {"num_actors": 256, "actor_inference": "batched", "memory_params.memory_size": 2 * 1024 * 1024,
        "memory_params.enable_per": True, "agent_params.nstep": 5},
    5: {"num_actors": 256, "actor_inference": "batched", "memory_params.memory_size": 1024 * 1024,
        "memory_params.enable_per": True, "agent_params.batch_size": 4096},
    6: {"num_actors": 2, "agent_params.nstep": 3, "agent_params.learn_start": 200,
        "agent_params.batch_size": 32, "agent_params.target_model_update": 50,
        "agent_params.lr": 1e-3, "memory_params.memory_size": 5000},
    # 7 = the DDPG recipe that holds on Pendulum (>= -200 over the last 10 evaluations on 4 of 4 seeds, in the
    # synchronous probe tools/probes/ddpg_recipe.py and in the asynchronous job; BASELINE.md "DDPG recipe" has the
    # ablations).  Each deviation from row 3 is needed, none of the others tried is:
    #   lr 1e-3 (reference 1e-4)       with 1e-4 / 3e-4 the policy is still a saturated bang-bang after 40 k updates
    #   nstep 1 (reference 5)          uncorrected 3- and 5-step returns under OU noise: saturated actor, 0 of 2 seeds
    #   compat_ddpg_grad_leak False    the reference adds d(-Q(s,pi(s)))/d(critic) to the critic's gradient
    #                                  (ddpg_learner.py:68-86: no zero_grad in between): 0 of 4 seeds learn with it
    #   replay ratio 1 +- 0, both ways one update per env step (summed over actors), counted from learn_start: a learner
    #                                  that is 20x faster than its actors overfits the first transitions and saturates
    #                                  the tanh actor for good; one that is slower leaves the ratio to the machine
    #   actor_sync_freq 50 (ref. 400)  per actor: with 8 actors 400 steps are 3 200 updates of policy lag (1 of 2 seeds)
    #   ou_stationary_start            N actors starting their OU noise at 0 fill the first learn_start transitions
    #                                  with |a| << 1 (3 of 4 seeds without, 4 of 4 with)
    #   actor_preact_l2 0.01           + 0.01 mean |z|^2 on the actor's pre-tanh output z.  Everything above makes
    #                                  learning LIKELY (3 of 4 .. 4 of 4 seeds on the slow CPU job, 1 of 4 .. 2 of 5 with
    #                                  the 12 k updates/s native learner): a tanh actor that the first, meaningless
    #                                  critic gradients push into saturation has dQ/da (1 - a^2) ~ 0 and never comes
    #                                  back.  The penalty's gradient does not pass through tanh': 8 of 8 seeds.
    # Not needed (measured): reward scaling, separate actor/critic rates, a larger tau, critic-only warm-up (hurts).
    # 8 = the reference's README Pong setting (8 actors, n-step 5, every hyper-parameter as shipped) on MiniPong
    # (/root/reference/utils/options.py:112-141), with two additions that the reference's own speed made unnecessary:
    #   num_envs_per_actor 16          the learner here is ~27x faster than the reference's; the 8 actor processes
    #                                  keep up by stepping 16 envs each behind one batched inference request per env
    #                                  (an actor process with 4 envs spends 63 % of its time waiting for answers);
    #                                  the replay is the frame-deduplicated one (a seventh of the ingest bytes).
    #                                  +18 after 12.3 / 12.4 / 12.7 s on 3 of 3 seeds (4 envs, full stacks: 9-46 s)
    #   max_replay_ratio 1/16          one update per 16 env steps = 8 samples per transition at batch 128, the replay
    #                                  intensity of the classic DQN recipe (batch 32 every 4 steps) and about the
    #                                  reference's own emergent ratio.  Uncoupled, the native learner runs 0.3 updates
    #                                  per env step and the evaluation score peaks at +14 and wanders between -8 and +14
    #                                  (profiles/r3_minipong_wallclock.txt); coupled it reaches +20 in ~10 s.
    8: {"num_actors": 8, "agent_params.nstep": 5, "actor_inference": "batched", "num_envs_per_actor": 16,
        "agent_params.max_replay_ratio": 0.0625},
    7: {"num_actors": 8, "agent_params.nstep": 1, "agent_params.lr": 1e-3,
        "agent_params.compat_ddpg_grad_leak": False, "agent_params.max_replay_ratio": 1.0,
        "agent_params.min_replay_ratio": 1.0, "agent_params.actor_sync_freq": 50,
        "agent_params.ou_stationary_start": True, "agent_params.actor_preact_l2": 0.01,
        "memory_params.memory_size": 1024 * 1024},
}


def _have_gpu():
    # device_count() does not initialise the HIP runtime in the calling process (is_available() does): the
    # launcher's parent builds Options and must stay GPU-free, because it starts every worker as a child
    return torch.cuda.device_count() > 0


def _default_device():
    return "cuda" if _have_gpu() else "cpu"


def _parse_value(text):
    if not isinstance(text, str):
        return text
    low = text.strip().lower()
    if low in ("none", "null"):
        return None
    if low in ("inf", "+inf"):
        return float("inf")
    try:
        return ast.literal_eval(text)
    except (ValueError, SyntaxError):
        return text


def parse_overrides(items):
    """["a=1", "b.c=x"] or "a=1,b.c=x" -> {"a": 1, "b.c": "x"}"""
    if items is None:
        return {}
    if isinstance(items, dict):
        return dict(items)
    if isinstance(items, str):
        items = [s for s in items.split(",") if s.strip()]
    out = {}
    for item in items:
        if "=" not in item:
            raise ValueError("override %r is not KEY=VALUE" % (item,))
        k, v = item.split("=", 1)
        out[k.strip()] = _parse_value(v.strip())
    return out


class Params(object):
    def __init__(self, config=None, overrides=None):
        overrides = overrides or {}
        # training signature
        self.machine = overrides.get("machine", "mi355x")      # "machine_id"
        self.timestamp = overrides.get("timestamp", "26111000")  # "yymmdd##"
        # training configuration
        self.mode = overrides.get("mode", 1)                   # 1(train) | 2(test model_file)
        self.config = overrides.get("config", 0 if config is None else config)
        self.gpu_ind = overrides.get("gpu_ind", 0)             # first learner uses cuda:gpu_ind

        (self.agent_type, self.env_type, self.game,
         self.memory_type, self.model_type) = CONFIGS[self.config]

        self.seed = 100
        self.render = False
        self.visualize = True

        self.num_envs_per_actor = 1
        self.num_actors = 8
        self.num_learners = 1

        # --- MI355X-native extensions -------------------------------------
        self.device = _default_device()     # learner / inference device: "cuda" | "cpu"
        self.actor_inference = "local"      # "local": per-actor model copy | "batched": GPU inference server
        self.inference_placement = "auto"   # batched inference runs in a "process" of its own | inside the "learner" ranks
                                            # (each rank serves 1/N of the env workers); auto: learner with the native core
        self.inference_window_us = 40       # native batched inference: a partial batch waits this long for the other live
                                            # envs' requests (0: answer at once); measured in profiles/r2_node_*.json
        self.resume = False                 # continue from ./models/MACHINE_TIMESTAMP.resume.pth (optimizer, target, counters)
        self.resume_freq = 0                # also write the resume sidecar every this many learner steps (0: only at the end)
        self.save_resume = True
        self.native_nstep = True            # ring-fed memories: n-step assembly + record packing in C (K11)
        self.weight_sync = "auto"           # learner -> inference server weights: "shm" | "collective" (RCCL/gloo broadcast)
        self.actor_nice = 5                 # niceness added to actor processes (0 = leave alone)
        self.compute = "auto"               # "auto" | "native" (HIP kernels, MFMA) | "torch" (fp32 eager)
        # arithmetic of the native Nature-CNN engine.  "fp32": every MFMA product is a split-bf16 x3 product
        # (a_hi*b_hi + a_lo*b_hi + a_hi*b_lo, fp32 accumulation; 16-bit significands, gradients within 1e-3 of an
        # fp32 PyTorch step -- the reference learner is fp32, core/single_processes/dqn_learner.py:63-87);
        # "bf16": plain bf16 operands (fastest; conv gradients within ~10 % of fp32)
        self.precision = "fp32"
        self.param_sync_freq = 25           # learner publishes weights every this many learner steps ...
        self.param_publish_interval = None  # ... but at most once per this many seconds (None: 0.02 with per-actor model
                                            # copies, min(1, evaluator_freq / 5) when inference is batched on the GPU)
        self.supervise = True               # restart dead actors, stop everything if a learner dies
        self.inject_fault = None            # "actor:1@50" kills actor process 1 at its 50th step, "learner:1@20" learner
                                            # rank 1 after its 20th update, "hang-learner:0@20" makes rank 0 stop
                                            # beating there (tests of the supervisor)
        self.watchdog_learner_s = 120.0     # a learner / inference process silent for this long is hung: the job stops
        self.watchdog_actor_s = 120.0       # a silent actor is terminated and restarted (0: no watchdog)
        self.shutdown_grace_s = 20.0        # time the remaining processes get once the job is over, before terminate()
        self.stop_agree_every = 64          # several learner ranks: they agree on leaving the loop every this many steps
                                            # (the same reduce carries the peer all-reduce's error word)
        self.dp_allreduce_on_error = "rccl" # a peer all-reduce wait timed out on some rank: "rccl" = every rank switches to
                                            # the process group's all-reduce and re-syncs from rank 0 | "stop" = the job
                                            # ends with a non-zero exit code.  It never silently continues.
        self.ingest = "ring"                # "ring": lock-free shm ring into device replay | "direct"
        self.dist_backend = None            # process-group back-end of the learner ranks (None: "nccl" = RCCL on GPUs, gloo on CPU)
        self.allow_shared_gpu = False       # num_learners > visible GPUs: refuse (False) or let ranks share a card (rehearsal)
        self.force_dp = False               # run the data-parallel step program even with one learner (rehearsal)
        self.engine_knobs = {}              # tuning knobs of the native engines (ops/knobs.py: split-K targets, DP wire
                                            # format / all-reduce / graph mode, ...); `engine_knobs.<name>=<value>`
        self.graph_steps = 10               # native DQN learner, HBM replay, one rank: updates per hipGraph replay (the
                                            # learner runs up to graph_steps - 1 updates ahead of its step counter).
                                            # 5 / 10 / 25 / 50: 10.49 / 10.59 / 10.56 / 10.46 k updates/s alone (bf16
                                            # 13.35 / 13.65 / 13.74 / 13.56), 9.79 / 9.89 / 9.81 k inside the job

        # prefix for saving models&logs
        self.root_dir = overrides.get("root_dir", os.getcwd())
        self._refresh_paths()

    def _refresh_paths(self):
        self.refs = self.machine + "_" + self.timestamp
        # NOTE: evaluator saves the current model to model_name
        self.model_name = self.root_dir + "/models/" + self.refs + ".pth"
        # NOTE: pretrained model_file is loaded for finetuning when not None
        if not hasattr(self, "model_file"):
            self.model_file = None
        if self.mode == 2:
            self.model_file = self.model_name   # mode=2 tests the current training's checkpoint
            self.visualize = False
        self.log_dir = self.root_dir + "/logs/" + self.refs + "/"


class EnvParams(Params):
    def __init__(self, config=None, overrides=None):
        super().__init__(config, overrides)
        if "mlp" in self.model_type:    # low-dim inputs, no preprocessing
            self.state_cha = 1          # equals hist_len
            self.state_hei = 1
            self.state_wid = None       # depends on the env
        elif "cnn" in self.model_type:  # raw image inputs
            self.state_cha = 4
            self.state_hei = 84
            self.state_wid = 84
        self.early_stop = None
        if self.env_type in ("atari", "synthetic"):
            self.early_stop = 12500
        elif self.env_type == "pendulum":
            self.early_stop = 200
        elif self.env_type in ("chain", "counting"):
            self.early_stop = 50
        # synthetic env knobs
        self.synthetic_episode_len = 1000   # mean steps per synthetic episode
        self.synthetic_num_actions = 6      # Pong's minimal action set


class MemoryParams(Params):
    def __init__(self, config=None, overrides=None):
        super().__init__(config, overrides)
        self.memory_size = 50000
        # dtype for states (the reference passes tensor *types* because dtypes
        # did not pickle through mp at the time; both are kept here)
        if "mlp" in self.model_type:
            self.tensortype = torch.FloatTensor
            self.state_dtype = "float32"
        else:                           # images stay uint8 end to end
            self.tensortype = torch.ByteTensor
            self.state_dtype = "uint8"
        # prioritized experience replay (finished here; a TODO in the reference)
        self.enable_per = "per" in self.memory_type
        self.priority_exponent = 0.5    # alpha (rainbow 0.5, ape-x 0.6)
        self.priority_weight = 0.4      # beta
        self.priority_eps = 1e-6
        # device replay / ingestion
        self.replay_device = _default_device() if "device" in self.memory_type else "cpu"
        self.ingest_ring_slots = 4096   # transitions buffered in the shm ring
        self.ingest_chunk = 256         # max transitions moved to HBM per drain
        self.ingest_device_slabs = 32   # device-side chunks in flight between their H2D copy and the unpack kernel
        # frame-deduplicated replay (memory_type "device-dedup[-per]", memories/dedup_memory.py)
        self.dedup_plane_cap = 96       # planes per run record (a run restart re-sends nstep + window + 1 planes)
        self.dedup_plane_slack = 1.25   # planes kept per transition row of the ring
        self.dedup_ring_mb = 64         # shm ring between the actors and the learner
        self.dedup_chunk_runs = 8       # run records per H2D copy (<= 16)
        self.dedup_max_age_ms = 250     # a run that is not full goes out after this long


class ModelParams(Params):
    def __init__(self, config=None, overrides=None):
        super().__init__(config, overrides)
        self.init = "default"           # "default" (what the reference actually runs) | "orthogonal"


class AgentParams(Params):
    def __init__(self, config=None, overrides=None):
        super().__init__(config, overrides)
        # criteria and optimizer
        self.value_criteria = nn.MSELoss()
        self.optim = torch.optim.Adam
        self.num_tasks = 1
        self.steps = 500000             # max #learner iterations
        self.gamma = 0.99
        self.lr = 1e-4
        self.lr_decay = False
        self.weight_decay = 0.0
        self.logger_freq = 15           # secs
        self.tester_nepisodes = 50
        self.evaluator_nepisodes = 2
        self.nstep = 5
        self.max_replay_ratio = 0.0     # cap on learner updates per actor env step (0: none, as in the reference)
        self.min_replay_ratio = 0.0     # floor on the same ratio: actors wait for the learner (0: none, as in the reference)
        self.poll_interval = 5.0        # logger/evaluator wake-up period (secs; the reference hard-codes 5)
        self.stop_at_eval_reward = None # end the job once an evaluation's average reward reaches this (None: run all `steps`)
        if self.agent_type == "dqn":
            self.clip_grad = np.inf
            self.actor_sync_freq = 100
            self.actor_freq = 250
            self.learner_freq = 100
            self.evaluator_freq = 30
            self.learn_start = 5000
            self.batch_size = 128
            self.target_model_update = 250
            self.enable_double = False
            self.eps = 0.4
            self.eps_alpha = 7
        elif self.agent_type == "ddpg":
            self.clip_grad = 40.0
            self.actor_sync_freq = 400
            self.actor_freq = 2500
            self.learner_freq = 1000
            self.evaluator_freq = 60
            self.learn_start = 250
            self.batch_size = 64
            self.target_model_update = 1e-3
            self.random_process = OrnsteinUhlenbeckProcess
            # the reference lets the actor loss deposit gradients into the critic
            # (ddpg_learner.py:66-86); False freezes the critic during the actor loss
            self.compat_ddpg_grad_leak = True
            self.ou_stationary_start = False # DDPG actors: OU noise starts at its stationary spread, not at 0 (see agents/actors.py)
            self.actor_preact_l2 = 0.0       # DDPG: + this * mean(|pre-tanh actor output|^2) in the actor loss (0: as the reference)
            self.actor_warmup_steps = 0      # DDPG: learner updates during which only the critic is trained (0: as the reference)


_NESTED = ("env_params", "memory_params", "model_params", "agent_params")


class Options(Params):
    def __init__(self, config=None, overrides=None, argv=None):
        ov = parse_overrides(os.environ.get("PDA_OVERRIDES"))
        ov.update(parse_overrides(argv))
        ov.update(parse_overrides(overrides))
        if config is None and "config" in ov:
            config = ov["config"]
        top = {k: v for k, v in ov.items() if "." not in k}
        super().__init__(config, top)
        self.env_params = EnvParams(self.config, top)
        self.memory_params = MemoryParams(self.config, top)
        self.model_params = ModelParams(self.config, top)
        self.agent_params = AgentParams(self.config, top)
        merged = dict(PRESETS.get(self.config, {}))
        merged.update(ov)
        self.apply(merged)

    def apply(self, overrides):
        """Set fields; un-dotted keys also propagate into every nested params
        object that carries the same field (they all inherit Params)."""
        for key, val in overrides.items():
            if key == "config":
                continue
            if "." in key:
                head, field = key.split(".", 1)
                if head == "engine_knobs":      # engine_knobs.fc1_blocks=128 (validated by ops/knobs.py:resolve)
                    self.engine_knobs = dict(self.engine_knobs or {}, **{field: val})
                    continue
                if head not in _NESTED:
                    raise KeyError("unknown params group %r" % head)
                setattr(getattr(self, head), field, val)
            else:
                setattr(self, key, val)
                for name in _NESTED:
                    sub = getattr(self, name)
                    if hasattr(sub, key):
                        setattr(sub, key, val)
        for obj in (self,) + tuple(getattr(self, n) for n in _NESTED):
            obj._refresh_paths()
        if "model_file" in overrides:
            self.model_file = overrides["model_file"]
        mp = self.memory_params
        if "memory_params.enable_per" not in overrides:
            mp.enable_per = "per" in self.memory_type
        if mp.replay_device == "cuda" and not _have_gpu():
            mp.replay_device = "cpu"
        if self.device == "cuda" and not _have_gpu():
            self.device = "cpu"
        return self
