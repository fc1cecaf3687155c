"""Learner processes (reference core/single_processes/dqn_learner.py:9-103 and
ddpg_learner.py:9-108; same signatures).

What changed relative to the reference:
* each learner owns its model on its own device and is one rank of a
  synchronous data-parallel group (flat-gradient all-reduce, parallel/comm.py)
  instead of a Hogwild writer on one shared CUDA model;
* rank 0 publishes weights to the shared `global_model` every
  `param_sync_freq` steps (utils/param_store.py);
* prioritized replay is live: importance-weighted loss and |TD| write-back;
* two step back-ends: "torch" (fp32 eager; CPU plumbing and the oracle) and
  "native" (ops/dqn_engine.py, ops/mlp_engine.py: HIP kernels, MFMA,
  HBM replay, captured step graphs) picked by `args.compute`.

`DqnTorchCore` / `DdpgTorchCore` hold the per-step arithmetic so tests and the
benchmark can drive it without processes.
"""
import os
import time

import numpy as np
import torch
import torch.nn as nn

from ..parallel.comm import SingleComm, make_comm
from ..utils.helpers import update_target_model
from ..utils.param_store import AsyncParamPublisher, sync_params


def _clip_value_(params, clip):
    if clip is None or not np.isfinite(clip):
        return
    grads = [p.grad for p in params if p.grad is not None]
    if grads:
        torch._foreach_clamp_min_(grads, -float(clip))
        torch._foreach_clamp_max_(grads, float(clip))


class _TorchCore(object):
    def __init__(self, args, model, target_model, memory, comm=None):
        self.args, self.ap = args, args.agent_params
        self.model, self.target_model, self.memory = model, target_model, memory
        self.comm = comm or SingleComm()
        self.device = next(model.parameters()).device
        update_target_model(model, target_model)      # hard copy to start
        self.model.flatten_()
        self.optimizer = self.ap.optim(self.model.parameters(), lr=self.ap.lr,
                                       weight_decay=self.ap.weight_decay)
        self.model.train()
        self.step_count = 0
        self.last_losses = {}

    def _batch(self):
        batch, inds, weights = self.memory.sample_with_priorities(self.ap.batch_size)
        s0, a, r, g, s1, t = (x.to(self.device, non_blocking=True) for x in batch)
        return s0, a, r, g, s1, t, inds, weights.to(self.device)

    def _weighted_mse(self, pred, target, weights):
        if self.memory.enable_per:
            return (weights * (pred - target) ** 2).mean()
        return self.ap.value_criteria(pred, target)

    def _finish(self, td_abs, inds, clip_params=()):
        """All-reduce, THEN clip: every back-end clips the rank-averaged gradient (= the gradient of the large
        batch), so the torch oracle and the native engine agree in data-parallel runs with a finite clip."""
        if self.comm.world_size > 1:
            self.comm.all_reduce_mean_(self.model.flat_grads)
        for params in clip_params:
            _clip_value_(params, self.ap.clip_grad)
        self.optimizer.step()
        update_target_model(self.model, self.target_model, self.ap.target_model_update, self.step_count)
        if self.memory.enable_per:
            self.memory.update_priorities(inds, td_abs)
        self.step_count += 1

    def flat_params(self):
        return self.model.flat_params

    def loss_values(self, lagged=False):
        return {k: float(v) for k, v in self.last_losses.items()}

    # ---- true resume (the reference can only warm-start from the plain state_dict) ----------
    def resume_state(self):
        st = self.optimizer.state_dict()["state"]
        keys = sorted(st)
        out = {"kind": "torch", "params": self.model.flat_params.detach().cpu().clone(),
               "target": self.target_model.flat_params.detach().cpu().clone(), "step_count": self.step_count}
        if keys:
            out["exp_avg"] = torch.cat([st[k]["exp_avg"].reshape(-1).cpu() for k in keys])
            out["exp_avg_sq"] = torch.cat([st[k]["exp_avg_sq"].reshape(-1).cpu() for k in keys])
            out["opt_step"] = int(st[keys[0]]["step"])
        return out

    def load_resume_state(self, d):
        with torch.no_grad():
            self.model.flat_params.copy_(d["params"])
            self.target_model.flat_params.copy_(d["target"])
        self.step_count = int(d["step_count"])
        if "exp_avg" in d:
            off = 0
            for p in self.model.parameters():        # materialise the optimizer state, then overwrite it
                n = p.numel()
                state = self.optimizer.state[p]
                state["step"] = torch.tensor(float(d["opt_step"]))
                state["exp_avg"] = d["exp_avg"][off:off + n].view_as(p).to(p.device).clone()
                state["exp_avg_sq"] = d["exp_avg_sq"][off:off + n].view_as(p).to(p.device).clone()
                off += n


class DqnTorchCore(_TorchCore):
    def step(self):
        s0, a, r, g, s1, t, inds, w = self._batch()
        self.model.zero_flat_grad()
        with torch.no_grad():
            if self.ap.enable_double:
                next_a = self.model(s1).argmax(dim=1, keepdim=True)
                next_q = self.target_model(s1).gather(1, next_a)
            else:
                next_q = self.target_model(s1).max(dim=1, keepdim=True)[0]
            target = r + g * next_q * (1 - t)
        pred = self.model(s0).gather(1, a.long())
        loss = self._weighted_mse(pred, target, w)
        loss.backward()
        self.last_losses = {"critic_loss": loss.detach()}
        self._finish((pred.detach() - target).abs(), inds, clip_params=[list(self.model.critic.parameters())])


class DdpgTorchCore(_TorchCore):
    def step(self):
        s0, a, r, g, s1, t, inds, w = self._batch()
        self.model.zero_flat_grad()
        leak = getattr(self.ap, "compat_ddpg_grad_leak", True)
        # actor loss: -Q(s, pi(s)).  The reference lets this backward also leave
        # gradients in the critic, which then add to the critic loss gradients.
        if not leak:
            for p in self.model.critic.parameters():
                p.requires_grad_(False)
        lam = float(getattr(self.ap, "actor_preact_l2", 0.0) or 0.0)
        if lam > 0:
            # + lam * mean(z^2) on the actor's pre-tanh output z: once a tanh policy saturates, d(-Q)/dz = dQ/da (1 - a^2)
            # is ~0 and it can never come back; this term's gradient does not pass through tanh' (options.py, preset 7)
            z = self.model.actor[:-1](s0.float().flatten(1))
            action = torch.tanh(z)
            q_pi = self.model.forward_critic(s0, action)
            actor_loss = -q_pi.mean() + lam * (z ** 2).sum(dim=1).mean()
        else:
            _, q_pi = self.model(s0)
            actor_loss = -q_pi.mean()
        self._ddpg_updates = getattr(self, "_ddpg_updates", 0) + 1
        if self._ddpg_updates >= int(getattr(self.ap, "actor_warmup_steps", 0)):    # critic warm-up: see options
            actor_loss.backward()
        if not leak:
            for p in self.model.critic.parameters():
                p.requires_grad_(True)
        # (single rank: the reference clips the actor's gradients here, before the critic loss adds to the critic's
        # (ddpg_learner.py:74); clipping is elementwise and the two parameter sets are disjoint, so clipping both
        # after the critic backward -- and, with several ranks, after the all-reduce -- is the same arithmetic)
        with torch.no_grad():
            _, next_q = self.target_model(s1)
            target = r + g * next_q * (1 - t)
        pred = self.model.forward_critic(s0, a)
        critic_loss = self._weighted_mse(pred, target, w)
        critic_loss.backward()
        self.last_losses = {"actor_loss": actor_loss.detach(), "critic_loss": critic_loss.detach()}
        self._finish((pred.detach() - target).abs(), inds, clip_params=[list(self.model.actor.parameters()),
                                                                         list(self.model.critic.parameters())])


def _pick_compute(args, device):
    mode = getattr(args, "compute", "auto")
    if mode == "torch" or device.type != "cuda":
        return "torch"
    from .. import ops
    if ops.native_available():
        return "native"
    if mode == "native":
        raise RuntimeError("compute='native' requested but the HIP extension is not loadable: " + ops.native_error())
    return "torch"


def make_core(kind, args, model, target_model, memory, comm, device):
    compute = _pick_compute(args, device)
    if compute == "native":
        if kind == "dqn" and args.model_type == "dqn-cnn":
            from ..ops.dqn_engine import DqnNativeCore
            return DqnNativeCore(args, model, target_model, memory, comm)
        if args.model_type in ("ddpg-mlp", "dqn-mlp"):
            from ..ops.mlp_engine import make_mlp_native_core
            core = make_mlp_native_core(kind, args, model, target_model, memory, comm)
            if core is not None:
                return core
    return (DqnTorchCore if kind == "dqn" else DdpgTorchCore)(args, model, target_model, memory, comm)


def native_resume_state(core):
    """Resume blob of a native (HIP engine) core: fp32 masters, target, Adam moments (kernel layout),
    the device-side step counter."""
    e = core.engine
    return {"kind": "native", "params": e.online.flat.detach().cpu().clone(),
            "target": e.target.flat.detach().cpu().clone(), "exp_avg": e.adam.m.cpu().clone(),
            "exp_avg_sq": e.adam.v.cpu().clone(), "opt_step": int(e.adam.step_dev.item()),
            "step_count": int(getattr(e, "step_count", core.step_count))}


def native_load_resume_state(core, d):
    e = core.engine
    with torch.no_grad():
        e.online.flat.copy_(d["params"])
        e.target.flat.copy_(d["target"])
        e.adam.m.copy_(d["exp_avg"])
        e.adam.v.copy_(d["exp_avg_sq"])
        e.adam.step_dev.fill_(int(d["opt_step"]))
    e.online.refresh()
    e.target.refresh()
    if hasattr(e, "step_count"):
        e.step_count = int(d["step_count"])
    core.step_count = int(d["step_count"])


def _resume_path(args):
    return args.model_name[:-4] + ".resume.pth" if args.model_name.endswith(".pth") else args.model_name + ".resume"


def save_learner_resume(args, core, learner_step, actor_step):
    from ..utils.checkpoint import save_resume
    blob = core.resume_state() if hasattr(core, "resume_state") else native_resume_state(core)
    blob["learner_step"], blob["actor_step"] = int(learner_step), int(actor_step)
    save_resume(_resume_path(args), **blob)


def load_learner_resume(args, core):
    """Returns (learner_step, actor_step) restored from the sidecar, or None."""
    import os
    from ..utils.checkpoint import load_resume
    path = _resume_path(args)
    if not os.path.exists(path):
        return None
    d = load_resume(path)
    native = not hasattr(core, "resume_state")
    if (d.get("kind") == "native") != native:
        print("[learner] resume file %s was written by the %s step back-end; only weights are restored"
              % (path, d.get("kind")))
        print("[learner] ... and the target network; the optimizer moments restart from zero")
        with torch.no_grad():
            core.flat_params().copy_(d["params"])
            if native:
                core.engine.target.flat.copy_(d["target"])
            else:
                core.target_model.flat_params.copy_(d["target"])
        if native:
            core.engine.online.refresh()
            core.engine.target.refresh()
            core.engine.step_count = int(d.get("step_count", d["learner_step"]))
        core.step_count = int(d.get("step_count", d["learner_step"]))
        return int(d["learner_step"]), int(d["actor_step"])
    if native:
        native_load_resume_state(core, d)
    else:
        core.load_resume_state(d)
    return int(d["learner_step"]), int(d["actor_step"])


def inference_slot_range(rank, world, num_slots):
    """Mailbox slots [lo, hi) served by learner rank `rank` of `world`: contiguous, sizes differing by at most one.
    Every rank hosts the native inference loop for ITS share of the env workers (reading its own live shadow
    weights), so no rank carries the whole serving load and becomes the straggler of the synchronous step."""
    base, rem = divmod(int(num_slots), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def _host_inference(args, core, model, mailbox, device, slot_range=None):
    engine = getattr(core, "engine", None)
    if device.type != "cuda" or engine is None or not hasattr(engine, "online"):
        raise RuntimeError("inference_placement='learner' needs the native HIP learner core on a GPU")
    from ..ops.infer import make_native_server
    server = make_native_server(args, model, mailbox, device, high_priority=True, weights=engine.online,
                                slot_range=slot_range)
    if server is None:
        raise RuntimeError("no native inference program for model_type %r" % args.model_type)
    server.start()
    return server


class _LoopProfile(object):
    """Knob learner_prof (PDA_LEARNER_PROF=1): where the learner loop's wall time goes inside the whole job -- host time in drain() /
    core.step() / the rest of an iteration, and how often the GPU had ALREADY run dry when the next step was handed
    to it (host-bound) -- written to <log_dir>/learner_loop_profile.json at the end.  Off by default: the stream
    query costs a few microseconds per iteration."""

    def __init__(self, device):
        self.device = device
        self.t = {"drain": 0.0, "step": 0.0, "rest": 0.0}
        self.iters = 0
        self.starved = 0
        self.t0 = None
        self.mark = None
        self.pairs, self.replays, self.syncs = [], 0, 0

    def begin(self):
        now = time.perf_counter()
        if self.mark is not None:
            self.t["rest"] += now - self.mark
        if self.t0 is None:
            self.t0 = now
        self.mark = now

    def lap(self, name):
        now = time.perf_counter()
        self.t[name] += now - self.mark
        self.mark = now

    def before_step(self, core=None):
        self.iters += 1
        if self.device.type != "cuda":
            return
        self.queries = getattr(self, "queries", 0)
        if self.iters % 16 == 0:    # (sampled: the query costs a few microseconds)
            self.queries += 1
            if torch.cuda.current_stream(self.device).query():
                self.starved += 1   # everything queued so far has finished: the GPU is waiting for the host
        # GPU-side timing of the step graphs themselves: bursts of 8 consecutive replays, an event in front of and
        # behind each (replay time, and the stream time BETWEEN two replays: unpack kernels, size updates, idling)
        prog = getattr(core, "program", None)
        self._pending_evt = None
        if prog is not None and getattr(prog, "ahead", 0) == 0 and len(self.pairs) < 4000:
            self.replays += 1
            if 200 <= self.replays < 3200:  # a continuous stretch: every replay and every gap of ~1.8 s of learning
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                chunks = int(getattr(getattr(core, "memory", None), "chunks_unpacked", 0))
                self._pending_evt = (e0, e1, self.replays, int(getattr(prog, "graph_steps", 1)), chunks, self.syncs)

    def after_step(self):
        if getattr(self, "_pending_evt", None) is not None:
            self._pending_evt[1].record()
            self.pairs.append(self._pending_evt)
            self._pending_evt = None

    def dump(self, args, rank, steps):
        import json
        wall = time.perf_counter() - (self.t0 or time.perf_counter())
        out = {"rank": rank, "steps": steps, "iterations": self.iters, "wall_s": round(wall, 3),
               "host_seconds": {k: round(v, 3) for k, v in self.t.items()},
               "host_us_per_iteration": {k: round(1e6 * v / max(self.iters, 1), 2) for k, v in self.t.items()},
               "gpu_idle_at_submit_fraction": round(self.starved / max(getattr(self, "queries", 0), 1), 4)}
        if self.pairs:
            torch.cuda.synchronize(self.device)
            rep = [1e3 * p[0].elapsed_time(p[1]) / p[3] for p in self.pairs[8:]]
            # the same gaps by what the host put between the two replays: nothing / unpack kernel(s) of ingested
            # chunks / a host synchronisation (loss read-out), after which the GPU waits for the host to catch up
            kinds = {"plain": [], "with_unpack": [], "after_host_sync": []}
            for p, q in zip(self.pairs[8:-1], self.pairs[9:]):
                if q[2] != p[2] + 1:
                    continue
                g = 1e3 * p[1].elapsed_time(q[0])
                kinds["after_host_sync" if q[5] != p[5] else ("with_unpack" if q[4] != p[4] else "plain")].append(g)
            out["gpu_us_between_replays_by_kind"] = {
                k: {"n": len(v), "mean": round(sum(v) / len(v), 2), "median": round(sorted(v)[len(v) // 2], 2)}
                for k, v in kinds.items() if v}
            gaps = [1e3 * p[1].elapsed_time(q[0]) for p, q in zip(self.pairs[8:-1], self.pairs[9:]) if q[2] == p[2] + 1]
            rep.sort()
            gaps.sort()
            if rep and gaps:
                out["gpu_us_per_update_inside_replay"] = {"median": round(rep[len(rep) // 2], 2),
                                                          "p10": round(rep[len(rep) // 10], 2),
                                                          "p90": round(rep[9 * len(rep) // 10], 2)}
                out["gpu_us_between_replays"] = {"median": round(gaps[len(gaps) // 2], 2),
                                                 "p10": round(gaps[len(gaps) // 10], 2),
                                                 "p90": round(gaps[9 * len(gaps) // 10], 2),
                                                 "mean": round(sum(gaps) / len(gaps), 2)}
                out["updates_per_replay"] = self.pairs[-1][3]
                P = self.pairs[8:]
                span = 1e3 * P[0][0].elapsed_time(P[-1][1])
                ins = sum(1e3 * p[0].elapsed_time(p[1]) for p in P)
                gp = [(1e3 * p[1].elapsed_time(q[0]), q[4] - p[4], q[5] - p[5]) for p, q in zip(P[:-1], P[1:])]
                out["stretch"] = {"replays": len(P), "span_us": round(span, 1), "inside_us": round(ins, 1),
                                  "gaps_us": round(sum(g[0] for g in gp), 1),
                                  "us_per_update": round(span / (len(P) * P[0][3]), 2),
                                  "inside_us_per_update_mean": round(ins / (len(P) * P[0][3]), 2),
                                  "gaps_over_100us": len([g for g in gp if g[0] > 100]),
                                  "largest_gaps_us_chunks_syncs": [(round(g[0], 1), g[1], g[2])
                                                                   for g in sorted(gp, reverse=True)[:12]],
                                  "largest_inside_us_per_update": [round(x, 1) for x in sorted(
                                      (1e3 * p[0].elapsed_time(p[1]) / p[3] for p in P), reverse=True)[:12]]}
        with open(os.path.join(args.log_dir, "learner_loop_profile%s.json" % ("" if rank == 0 else "_rank%d" % rank)),
                  "w") as f:
            json.dump(out, f)
        print("[learner %d] loop profile: %s" % (rank, json.dumps(out)), flush=True)


def _maybe_peer_fault(args, core, rank, step):
    """inject_fault="peer-withhold:R@S": from its S-th update on, learner rank R's peer all-reduce stops storing its
    gradient slices into its peers (and every rank's waits give up after ~2 ms instead of a second), which is what a
    dead xGMI link or a wedged peer looks like to them.  Test hook for the error-word check in the loop below."""
    spec = getattr(args, "inject_fault", None)
    if not spec or not spec.startswith("peer-withhold:"):
        return
    who, when = spec[len("peer-withhold:"):].split("@")
    prog = getattr(core, "program", None)
    if step != int(when) or prog is None or not getattr(prog, "peer", False):
        return
    for ar in (prog.ar_tail, prog.ar_head):
        if hasattr(ar, "set_fault"):
            ar.set_fault(withhold=(int(who) == rank), spin_limit=1 << 12)


def _verify_learner_ranks(args, comm, rank, members):
    """num_learners=N must mean N distinct ranks in ONE collective group, on N distinct GPUs when on GPUs: the same
    proof bench.py prints (MIN / MAX / SUM of the rank ids, all-gather of rank / device / pid; parallel/comm.py
    verify_ranks), written to <log_dir>/learner_ranks.json by rank 0.  Ranks that wrapped onto one GPU (more
    learners than devices) are refused unless `allow_shared_gpu` is set (one-GPU rehearsals)."""
    import json
    facts = comm.verify_ranks()
    bad = None
    world = args.num_learners
    if facts["rccl_ranks"] != members or not facts["ranks_distinct"]:
        bad = "num_learners=%d: expected %d group members, the group has %d (distinct: %s)" % (
            world, members, facts["rccl_ranks"], facts["ranks_distinct"])
    elif (args.device == "cuda" and not getattr(args, "allow_shared_gpu", False)
          and len(set((r["device_index"], r["device_uuid_hash"]) for r in facts["rank_devices"][:world])) < world):
        bad = ("num_learners=%d learner ranks share a GPU (%d visible); set allow_shared_gpu=True for a rehearsal"
               % (world, torch.cuda.device_count()))
    if rank == 0:
        os.makedirs(args.log_dir, exist_ok=True)
        with open(os.path.join(args.log_dir, "learner_ranks.json"), "w") as f:
            json.dump(facts, f)
        print("[learner] %d ranks verified over %s: ids min/max/sum %s, devices %s"
              % (facts["rccl_ranks"], facts["collective_backend"], facts["rank_id_min_max_sum"],
                 [(r["rank"], r["device_index"]) for r in facts["rank_devices"]]), flush=True)
    if bad:
        raise RuntimeError(bad)


def _report_rank_agreement(args, comm, core, rank, step):
    """After the last step every rank must hold the same parameters (synchronous DP): max |rank r - rank 0|."""
    import json
    try:
        mine = core.flat_params().detach().float().clone()
        ref = mine.clone()
        comm.broadcast_(ref, src=0)
        diff = (mine - ref).abs().max().reshape(1)
        diff = float(comm.all_reduce_sum_(diff).item())         # >= 0 on every rank: the sum is 0 iff all are
        if rank == 0:
            with open(os.path.join(args.log_dir, "learner_rank_agreement.json"), "w") as f:
                json.dump({"steps": step, "param_max_abs_diff_summed_over_ranks": diff}, f)
            print("[learner] after %d steps: sum over ranks of max |param(rank r) - param(rank 0)| = %.3g" % (step, diff), flush=True)
    except Exception as exc:        # a report, never a reason to fail the shutdown
        print("[learner] rank agreement report failed: %r" % (exc,), flush=True)


def _learner_loop(process_ind, args, global_logs, learner_logs, model_prototype,
                  global_memory, global_model, global_optimizer, kind):
    print("---------------------------->", process_ind, "learner")
    ap = args.agent_params
    rank = process_ind - args.num_actors - 1
    world = args.num_learners
    if args.device == "cuda":
        ngpu = torch.cuda.device_count()        # (does not initialise the GPU)
        from ..ops.knobs import resolve as resolve_knobs
        if world > 1 and getattr(args, "compute", "auto") != "torch" and resolve_knobs(args).dp_graph == "auto":
            # may the step graph contain the process group's collectives on this node?  Asked of a throw-away child
            # BEFORE this process touches the GPU (parallel/preflight.py); only matters when the peer kernel is out
            from ..parallel.preflight import choose_dp_graph_mode
            os.environ.update({"RANK": str(rank), "WORLD_SIZE": str(world), "MASTER_ADDR": "127.0.0.1",
                               "LOCAL_RANK": str((args.gpu_ind + rank) % max(ngpu, 1)),
                               "MASTER_PORT": str(int(getattr(args, "dist_port", 29533)) + 101)})
            args.engine_knobs = dict(getattr(args, "engine_knobs", None) or {},
                                     dp_graph=choose_dp_graph_mode(port_offset=0))
        local_device = torch.device("cuda:%d" % ((args.gpu_ind + rank) % max(ngpu, 1)))
        torch.cuda.set_device(local_device)
        # a GPU learner's host work is launches and small copies: no intra-op thread pool (its workers keep spinning
        # after every parallel region -- the 6.75 MB snapshot copy of the weight publisher -- on cores the actors and
        # the runtime's own threads need)
        torch.set_num_threads(1)
    else:
        local_device = torch.device("cpu")
        torch.set_num_threads(max(1, min(8, torch.get_num_threads())))
    torch.manual_seed(args.seed + 1000 * rank)
    np.random.seed(args.seed + 1000 * rank)

    from ..parallel.weight_sync import WeightPublisher, resolve_weight_sync
    collective_sync = resolve_weight_sync(args) == "collective"
    comm = make_comm(rank, world, local_device, port=getattr(args, "dist_port", None),
                     extra_ranks=1 if collective_sync else 0, backend=getattr(args, "dist_backend", None))

    if world > 1:       # (with collective weight sync the inference rank is the group's last member and joins in)
        _verify_learner_ranks(args, comm, rank, world + (1 if collective_sync else 0))

    def build():
        return model_prototype(args.model_params, args.norm_val, args.state_shape,
                               args.action_space, args.action_shape).to(local_device)
    local_model, local_target_model = build(), build()
    if local_device.type == "cpu" and sum(p.numel() for p in local_model.parameters()) < 1000000:
        # small MLPs: intra-op threads only add barriers, and on a box whose cores the actors keep busy every such
        # barrier waits for a descheduled thread (measured: 2-16 updates/s with 8 threads, ~700/s with one)
        torch.set_num_threads(1)
    sync_params(local_model, global_model)
    if world > 1:       # all ranks start from rank 0's weights
        comm.broadcast_(local_model.flatten_(), src=0)

    if hasattr(global_memory, "attach"):    # device replay: allocate this rank's HBM shard
        global_memory.attach(local_device, rank=rank, world_size=world)

    torch.set_grad_enabled(True)
    core = make_core(kind, args, local_model, local_target_model, global_memory, comm, local_device)

    # batched actor inference hosted here: a native loop thread on a high-priority stream that
    # reads this learner's live bf16 shadow weights (no weight hand-off for the actors at all)
    # With several learner ranks EVERY rank serves its share of the mailbox slots: rank 0 alone would run at a reduced
    # rate and, the step being synchronous, hold all the others to it.
    hosted = None
    if args.actor_inference == "batched" and getattr(global_model, "mailbox", None) is not None:
        from ..parallel.inference_server import resolve_placement
        if resolve_placement(args) == "learner":
            lo, hi = inference_slot_range(rank, world, global_model.mailbox.num_slots)
            if hi > lo:
                hosted = _host_inference(args, core, local_model, global_model.mailbox, local_device,
                                         slot_range=(lo, hi))
                print("[learner %d] hosts batched inference for mailbox slots [%d, %d)" % (rank, lo, hi), flush=True)

    # weight hand-off (rank 0): seqlocked host buffer for evaluator / local actors (asynchronous D2H),
    # plus a collective broadcast to the inference rank when it owns another GPU
    step = 0
    restored = None
    if getattr(args, "resume", False):     # before the first publish: actors / evaluator must never see pre-resume weights
        restored = load_learner_resume(args, core)      # every rank loads the same sidecar
        if restored is not None:
            step = restored[0]
            if rank == 0:
                global_logs.learner_step.value = restored[0]
                global_logs.actor_step.value = restored[1]
                print("[learner] resumed from %s at learner step %d" % (_resume_path(args), step))
    # How often a snapshot goes to the seqlocked host buffer.  Its readers are the evaluator and -- only with
    # actor_inference="local" -- the actors; batched inference hosted in the learners reads the live weights.  Every
    # snapshot's 6.75 MB device-to-host copy stretches the ~2.5 step-graph replays it overlaps 2.8x (measured inside
    # the job: 8.3 k updates/s at the old 50 snapshots/s, 9.5 k with none), so nobody gets them more often than needed.
    pub_interval = getattr(args, "param_publish_interval", None)
    if pub_interval is None:
        pub_interval = 0.02 if args.actor_inference != "batched" else min(1.0, 0.2 * float(ap.evaluator_freq))
    shm_pub = AsyncParamPublisher(global_model, local_device, min_interval=float(pub_interval)) if rank == 0 else None
    if shm_pub is not None and restored is not None:
        shm_pub.publish(core.flat_params(), force=True)
    bcast_pub = None
    if rank == 0 and collective_sync:
        bcast_pub = WeightPublisher(comm, core.flat_params().numel(), local_device)
        bcast_pub.publish(core.flat_params())
    resume_freq = int(getattr(args, "resume_freq", 0))
    # the reference has no actor/learner rate coupling: the replay ratio is whatever the
    # two speeds give.  With a learner that is 15x faster that ratio explodes, so it can be capped:
    # at most `max_replay_ratio` learner updates per env step taken by the actors (0 = unlimited)
    max_ratio = float(getattr(ap, "max_replay_ratio", 0.0) or 0.0)
    started = False
    # Leaving the loop is a COLLECTIVE decision when there are several ranks: a rank that saw `stop` and left while
    # its peers entered the next step's all-reduce would leave them blocked in it.  Every `agree_every` steps (and on
    # every pass while waiting for data) the ranks MIN-reduce "keep going"; in between nobody looks at the flag.
    agree_every = max(1, int(getattr(args, "stop_agree_every", 64)))
    comm_errors, failed_comm = 0, False
    global_logs.beat("learner", process_ind)
    last_beat = time.time()
    from .actors import _maybe_fault
    from ..ops.knobs import resolve as _resolve_knobs
    prof = _LoopProfile(local_device) if _resolve_knobs(args).learner_prof else None
    while step < ap.steps:
        if prof is not None and started:
            prof.begin()
        now = time.time()
        if now - last_beat > 1.0:           # sign of life for the supervisor's watchdog (launcher.py)
            global_logs.beat("learner", process_ind)
            last_beat = now
        if world > 1:
            if not started or step % agree_every == 0:
                # one MAX-reduce tells every rank the worst news: 0 carry on, 1 stop, 2 a peer all-reduce wait ran
                # out on some rank (its error word; without this check the garbage sums would just feed Adam)
                prog = getattr(core, "program", None)
                err = prog.comm_error() if (started and hasattr(prog, "comm_error")) else 0
                code = comm.agree_max(2 if err else (1 if global_logs.stop.value else 0))
                if code == 2:
                    comm_errors += 1
                    policy = str(getattr(args, "dp_allreduce_on_error", "rccl"))
                    print("[learner %d] peer all-reduce: a wait timed out (local error word 0x%x) at step %d -> %s"
                          % (rank, err, step, "falling back to the process group's all-reduce" if policy == "rccl"
                             else "stopping the job"), flush=True)
                    if policy != "rccl":
                        global_logs.stop.value = True
                        failed_comm = True
                        break
                    prog.fall_back_to_collectives()
                elif code:
                    break
        elif global_logs.stop.value:
            break
        if hasattr(global_memory, "drain"):
            global_logs.frames_ingested.add(rank, global_memory.drain())
        if prof is not None and started:
            prof.lap("drain")
        if not started:
            ready = global_memory.size > ap.learn_start
            started = comm.all_agree(ready)
            if not started:
                time.sleep(0.002 if hasattr(global_memory, "drain") else 1.0)
                continue
        if max_ratio > 0:       # optional coupling of the learner to the data rate (absent in the reference)
            # (env steps taken before learning could start do not count: they would be paid back as one burst of
            # updates on the first few hundred transitions -- which is what saturates the DDPG actor for good)
            behind = step - max_ratio * max(0, global_logs.actor_step.value - ap.learn_start)
            if comm.all_agree(behind <= 0) is False:
                time.sleep(0.0005)
                continue
        if prof is not None:
            prof.before_step(core)
            prof.mark = time.perf_counter()
        core.step()
        if prof is not None:
            prof.after_step()
            prof.lap("step")
        step += 1
        _maybe_fault(args, "learner", rank, step)
        _maybe_peer_fault(args, core, rank, step)
        if rank == 0:
            with global_logs.learner_step.get_lock():
                global_logs.learner_step.value += 1
            if step % args.param_sync_freq == 0 or step == ap.steps:
                if shm_pub.publish(core.flat_params()) and bcast_pub is not None:
                    bcast_pub.publish(core.flat_params())
            if resume_freq and step % resume_freq == 0:
                save_learner_resume(args, core, step, global_logs.actor_step.value)
            if step % ap.learner_freq == 0:
                if prof is not None:
                    prof.syncs += 1
                for name, val in core.loss_values(lagged=True).items():
                    getattr(learner_logs, name).value += val
                learner_logs.loss_counter.value += 1
    if prof is not None:
        prof.dump(args, rank, step)
    if rank == 0:
        if getattr(args, "save_resume", True):
            save_learner_resume(args, core, step, global_logs.actor_step.value)
        shm_pub.publish(core.flat_params(), force=True)
        shm_pub.close()
        if bcast_pub is not None:
            bcast_pub.publish(core.flat_params())
            bcast_pub.close()
    if world > 1 and not collective_sync:
        _report_rank_agreement(args, comm, core, rank, step)
    if rank == 0 and hasattr(global_memory, "stats") and getattr(global_memory, "attached", False):
        import json
        with open(os.path.join(args.log_dir, "replay_stats.json"), "w") as f:      # (frame-deduplicated replay)
            json.dump(global_memory.stats(), f)
    if hasattr(global_memory, "close"):
        global_memory.close()
    if hosted is not None:
        hosted.stop()
        global_model.mailbox.closed[0] = 1
        st = hosted.stats()
        print("[inference@learner] %d requests in %d batches (mean %.1f per batch, %.0f us per batch)"
              % (st["requests"], st["batches"], st["mean_batch"], st["mean_batch_us"]), flush=True)
        from ..parallel.inference_server import write_inference_stats
        write_inference_stats(args, st, "learner",
                              "inference_stats.json" if rank == 0 else "inference_stats_rank%d.json" % rank)
        hosted.close()
    if rank == 0 and world > 1:
        import json
        prog = getattr(core, "program", None)
        with open(os.path.join(args.log_dir, "learner_comm.json"), "w") as f:
            json.dump({"peer_allreduce_timeouts": comm_errors, "stopped_on_comm_error": failed_comm,
                       "allreduce": getattr(prog, "allreduce_kind", getattr(comm, "backend", None)),
                       "dp_graph": getattr(prog, "dp_graph_mode", None)}, f)
    comm.close()
    if failed_comm:
        raise SystemExit(3)     # a non-zero exit code: the supervisor reports the job as failed


def dqn_learner(process_ind, args, global_logs, learner_logs, model_prototype,
                global_memory, global_model, global_optimizer=None):
    _learner_loop(process_ind, args, global_logs, learner_logs, model_prototype,
                  global_memory, global_model, global_optimizer, "dqn")


def ddpg_learner(process_ind, args, global_logs, learner_logs, model_prototype,
                 global_memory, global_model, global_optimizer=None):
    _learner_loop(process_ind, args, global_logs, learner_logs, model_prototype,
                  global_memory, global_model, global_optimizer, "ddpg")
