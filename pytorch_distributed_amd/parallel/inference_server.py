"""Batched actor inference: one GPU forward serves every env worker (kernel K10).

The reference gives each actor process a private CUDA model and runs batch-1
forwards with two `.item()` syncs per env step (dqn_actor.py:23-30,84;
dqn_cnn_model.py:58-78).  Here actors are pure-CPU env loops that post their
observation into a shared-memory mailbox slot; the server process gathers every
pending slot, runs ONE batched forward on the MI355X (HIP engine: uint8 frames
in, MFMA convs, epsilon-greedy/argmax/max-Q epilogue with a per-actor epsilon
vector), and posts actions + q-values back.

Mailbox protocol per slot i (all in one shared tensor block, 64-bit sequence
words accessed with acquire/release atomics from parallel/csrc/shm_ring.cpp):
  actor : write obs[i]; req[i] = k (release); wait resp[i] >= k; read out[i]
  server: see req[i] > resp[i]; read obs[i]; ...forward...; write out[i];
          resp[i] = req_seen (release)
"""
import time

import numpy as np
import torch

from . import hostlib
from ..utils.param_store import sync_params


class InferenceMailbox(object):
    def __init__(self, num_slots, state_shape, state_dtype, action_shape, discrete=True):
        self.num_slots = int(num_slots)
        self.discrete = discrete
        self.obs = torch.zeros((self.num_slots,) + tuple(state_shape), dtype=state_dtype).share_memory_()
        self.seq = torch.zeros(self.num_slots, 16, dtype=torch.int64).share_memory_()  # [:,0]=req [:,8]=resp (separate lines)
        self.out = torch.zeros(self.num_slots, int(action_shape) + 2, dtype=torch.float32).share_memory_()
        self.eps = torch.zeros(self.num_slots, dtype=torch.float32).share_memory_()
        self.closed = torch.zeros(1, dtype=torch.int64).share_memory_()     # set by the server on exit
        self.action_shape = int(action_shape)
        self._np = None
        self._count = {}
        self._obs_ptr = {}
        self._ids = None

    def _views(self):
        if self._np is None:
            self._np = (self.obs.numpy(), self.out.numpy(), self.seq.data_ptr())
        return self._np

    def __getstate__(self):
        d = self.__dict__.copy()
        d["_np"] = None
        d["_count"] = {}
        d["_obs_ptr"] = {}
        d["_ids"] = None
        return d

    def set_epsilon(self, slot, eps):
        self.eps[slot] = float(eps)

    def _req_ptr(self, slot):
        return self._views()[2] + slot * 128

    def _resp_ptr(self, slot):
        return self._views()[2] + slot * 128 + 64

    # ---- actor side --------------------------------------------------------
    def send(self, slot, state):
        """Post one observation without waiting (pair with wait()): an actor that drives several
        envs steps env B while the server answers env A.  Returns the request number."""
        L = hostlib.lib()
        obs, _, seq = self._views()
        src = state if (type(state) is np.ndarray and state.dtype == obs.dtype and state.flags.c_contiguous) \
            else np.ascontiguousarray(state, dtype=obs.dtype)
        k = self._count.get(slot)
        if k is None:
            k = int(L.pda_atomic_load(self._req_ptr(slot)))
        k += 1
        self._count[slot] = k
        dst = self._obs_ptr.get(slot)
        if dst is None:         # (per process: the address of this slot's observation in the shared mapping)
            dst = self._obs_ptr[slot] = obs[slot].ctypes.data
        L.pda_mailbox_send(seq, slot, k, dst, src.ctypes.data, src.nbytes)
        return k

    def wait(self, slot, k, timeout_s=120.0):
        """Block until request k of `slot` is answered: a short spin, then a futex sleep
        (parallel/csrc/mailbox.h); the GIL is released throughout."""
        L = hostlib.lib()
        _, out, seq = self._views()
        got = L.pda_mailbox_wait(seq, slot, k, 4000, 2000)
        deadline = None
        while got < k:
            if int(self.closed[0]) != 0:      # job is shutting down: hand back a no-op action
                a = self.action_shape
                return (np.zeros((1, 1), dtype=np.int64) if self.discrete else np.zeros((1, a))), 0.0, 0.0
            now = time.time()
            if deadline is None:
                deadline = now + timeout_s
            elif now > deadline:
                raise TimeoutError("inference server did not answer slot %d" % slot)
            got = L.pda_mailbox_wait(seq, slot, k, 0, 2000)
        a = self.action_shape
        row = out[slot]
        if self.discrete:
            action = np.array([[int(row[0])]])
        else:
            action = row[:a].astype(np.float64).reshape(1, a)
        return action, float(row[a]), float(row[a + 1])

    def request(self, slot, state, timeout_s=120.0):
        """Post one observation and block until the server answers."""
        return self.wait(slot, self.send(slot, state), timeout_s)

    # ---- server side (host fallback; the GPU path is ops/infer.py) -----------------------
    def pending(self):
        L = hostlib.lib()
        if self._ids is None:
            self._ids = np.zeros(self.num_slots, dtype=np.int32)
            self._reqs = np.zeros(self.num_slots, dtype=np.uint64)
        n = L.pda_mailbox_collect(self._views()[2], self.num_slots, self._ids.ctypes.data,
                                  self._reqs.ctypes.data, self.num_slots)
        return self._ids[:n].copy(), self._reqs[:n].copy()

    def respond(self, ids, req_seen, actions, qvalues, max_qvalues):
        L = hostlib.lib()
        _, out, seq = self._views()
        a = self.action_shape
        out[ids, :a] = np.asarray(actions, dtype=np.float32).reshape(len(ids), a)
        out[ids, a] = qvalues
        out[ids, a + 1] = max_qvalues
        ids = np.ascontiguousarray(ids, dtype=np.int32)
        req_seen = np.ascontiguousarray(req_seen, dtype=np.uint64)
        L.pda_mailbox_post(seq, ids.ctypes.data, req_seen.ctypes.data, len(ids))


class BatchedPolicy(object):
    """Device-side action selection for a batch of observations.  Uses the HIP
    engine when available, the torch module otherwise."""

    def __init__(self, args, model, device):
        self.args, self.model, self.device = args, model, device
        self.kind = args.agent_type
        self.engine = None
        self.gen = torch.Generator(device=device)
        self.gen.manual_seed(args.seed + 4242)
        if device.type == "cuda" and getattr(args, "compute", "auto") != "torch":
            from .. import ops
            if ops.native_available() and args.model_type == "dqn-cnn":
                from ..ops.dqn_engine import DqnCnnInference
                self.engine = DqnCnnInference(model, max_batch=max(args.num_actors * max(1, int(args.num_envs_per_actor)), 1))
            elif ops.native_available() and args.model_type == "ddpg-mlp":
                from ..ops.mlp_engine import DdpgActorInference
                self.engine = DdpgActorInference(model, max_batch=max(args.num_actors * max(1, int(args.num_envs_per_actor)), 1))

    def refresh(self):
        if self.engine is not None:
            self.engine.refresh_weights()

    @torch.no_grad()
    def act(self, obs, eps):
        """obs (n, ...) on device; eps (n,) on device.  Returns host arrays
        (actions (n, A), q (n,), maxq (n,))."""
        n = obs.shape[0]
        if self.kind == "dqn":
            if self.engine is not None:
                a, q, mq = self.engine.act(obs, eps)
            else:
                qv = self.model(obs)
                mq, greedy = qv.max(dim=1)
                explore = torch.rand(n, device=self.device, generator=self.gen) < eps
                rand_a = torch.randint(0, qv.shape[1], (n,), device=self.device, generator=self.gen)
                a = torch.where(explore, rand_a, greedy)
                q = qv.gather(1, a[:, None])[:, 0]
            return a.float().cpu().numpy().reshape(n, 1), q.cpu().numpy(), mq.cpu().numpy()
        if self.engine is not None:
            act = self.engine.act(obs)
        else:
            act = self.model.forward_actor(obs)
        z = np.zeros(n, dtype=np.float32)
        return act.float().cpu().numpy(), z, z


def write_inference_stats(args, st, where, name="inference_stats.json"):
    """Leave the serving counters where the parent (bench.py --node, bench/node_throughput.py) can read them."""
    import json
    import os
    try:
        os.makedirs(args.log_dir, exist_ok=True)
        keep = {k: st[k] for k in ("requests", "batches", "mean_batch", "max_batch", "mean_batch_us", "mean_window_us",
                                   "live_slots", "registered") if k in st}
        keep["where"] = where
        keep["window_us"] = int(getattr(args, "inference_window_us", 0))
        with open(os.path.join(args.log_dir, name), "w") as f:
            json.dump(keep, f)
    except OSError:
        pass


def resolve_placement(args):
    """Where the batched policy runs: "learner" = a native loop thread inside EVERY learner rank, each serving its
    share of the mailbox slots from the rank's live weights on a high-priority stream (agents/learners.py);
    "process" = a process of its own.  auto: "learner" whenever the learners run the native core."""
    place = getattr(args, "inference_placement", "auto")
    if place != "auto":
        return place
    return "learner" if native_learner_expected(args) else "process"


def native_learner_expected(args):
    """True when the learner processes of this job will run a native (HIP) core with live shadow weights that a
    hosted inference loop can read: GPU device, compute not forced to torch, the library built, a model family
    with a native serving program.  Decided from the options alone (the launcher's parent never touches the GPU),
    so the launcher and every learner rank reach the same answer."""
    import os
    if getattr(args, "device", "cpu") != "cuda" or getattr(args, "compute", "auto") == "torch":
        return False
    if args.model_type not in ("dqn-cnn", "ddpg-mlp"):
        return False
    if str(getattr(args, "weight_sync", "auto")) == "collective":
        return False            # an explicit request for the broadcast path implies a serving process of its own
    from .. import ops
    return os.path.exists(ops.LIB_PATH)


def _native_server(args, model, mailbox, device):
    if device.type != "cuda" or getattr(args, "compute", "auto") == "torch":
        return None
    from .. import ops
    if not ops.native_available():
        return None
    from ..ops.infer import make_native_server
    return make_native_server(args, model, mailbox, device)


def inference_server(process_ind, args, global_logs, model_prototype, global_model):
    print("---------------------------->", process_ind, "inference server")
    from .weight_sync import WeightReceiver, inference_device, resolve_weight_sync
    mailbox = global_model.mailbox
    device = inference_device(args)
    if device.type == "cuda":
        torch.cuda.set_device(device)
    model = model_prototype(args.model_params, args.norm_val, args.state_shape,
                            args.action_space, args.action_shape).to(device)
    version = sync_params(model, global_model)
    model.eval()
    torch.set_grad_enabled(False)
    receiver = None
    if resolve_weight_sync(args) == "collective":
        # join the learners' process group as rank L; weights then arrive by broadcast from rank 0
        from .comm import make_comm
        comm = make_comm(args.num_learners, args.num_learners, device, port=getattr(args, "dist_port", None),
                         extra_ranks=1, backend=getattr(args, "dist_backend", None))
        if args.num_learners > 1:
            comm.verify_ranks()     # the learners' membership check is a collective of the whole group (agents/learners.py)
        receiver = WeightReceiver(comm, model.flatten_(with_grad=False).numel(), device)
    try:
        native = _native_server(args, model, mailbox, device)
        if native is not None:
            _serve_native(args, global_logs, global_model, native, version, receiver=receiver,
                          process_ind=process_ind)
        else:
            policy = BatchedPolicy(args, model, device)
            policy.refresh()
            _serve(args, global_logs, global_model, mailbox, model, policy, device, version, receiver=receiver,
                   process_ind=process_ind)
    finally:
        mailbox.closed[0] = 1
        if receiver is not None:
            receiver.join(20)
            print("[inference] %d weight broadcasts received" % receiver.version, flush=True)
            comm.close()


def _serve_native(args, global_logs, global_model, native, version, poll_s=0.02, receiver=None, process_ind=0):
    """The C++ loop thread answers the actors; this thread only re-syncs the weights."""
    native.start()
    try:
        while global_logs.running(args.agent_params.steps):
            time.sleep(poll_s)
            global_logs.beat("inference", process_ind)     # watchdog (launcher.Supervisor)
            if receiver is not None:
                native.refresh_from_receiver(receiver)
                continue
            v = global_model.param_version.value
            if v != version and v % 2 == 0:
                version = native.refresh_from(global_model)
    finally:
        native.stop()
        st = native.stats()
        print("[inference] %d requests in %d batches (mean %.1f / max %d per batch, %.0f us per batch, "
              "zero-copy mailbox: %s)" % (st["requests"], st["batches"], st["mean_batch"], st["max_batch"],
                                          st["mean_batch_us"], bool(st["registered"])), flush=True)
        write_inference_stats(args, st, "process")
        native.close()


def _serve(args, global_logs, global_model, mailbox, model, policy, device, version, receiver=None, process_ind=0):
    """Python serving loop (CPU runs, model families without a native program)."""
    stats = {"requests": 0, "batches": 0, "busy": 0.0, "max_batch": 0}
    try:
        _serve_loop(args, global_logs, global_model, mailbox, model, policy, device, version, receiver, process_ind,
                    stats)
    finally:
        st = {"requests": stats["requests"], "batches": stats["batches"], "max_batch": stats["max_batch"],
              "mean_batch": stats["requests"] / max(stats["batches"], 1),
              "mean_batch_us": 1e6 * stats["busy"] / max(stats["batches"], 1), "registered": 0}
        write_inference_stats(args, st, "process-python")


def _serve_loop(args, global_logs, global_model, mailbox, model, policy, device, version, receiver, process_ind, stats):
    eps_dev = None
    served, last_sync = 0, 0
    idle = 0
    last_beat = 0.0
    while global_logs.running(args.agent_params.steps):
        if time.time() - last_beat > 1.0:
            last_beat = time.time()
            global_logs.beat("inference", process_ind)     # watchdog (launcher.Supervisor)
        ids, req_seen = mailbox.pending()
        if len(ids) == 0:
            idle += 1
            time.sleep(2e-5 if idle < 200 else 5e-4)
            continue
        idle = 0
        t0 = time.time()
        obs = mailbox.obs[torch.as_tensor(ids)]
        if device.type == "cuda":
            obs = obs.pin_memory().to(device, non_blocking=True)
        if eps_dev is None or served - last_sync >= args.agent_params.actor_sync_freq:
            eps_dev = mailbox.eps.to(device)
        eps = eps_dev[torch.as_tensor(ids, device=device)]
        actions, q, mq = policy.act(obs, eps)
        mailbox.respond(ids, req_seen, actions, q, mq)
        served += len(ids)
        stats["requests"] += len(ids)
        stats["batches"] += 1
        stats["busy"] += time.time() - t0
        stats["max_batch"] = max(stats["max_batch"], len(ids))
        if served - last_sync >= args.agent_params.actor_sync_freq:
            if receiver is not None:
                if receiver.poll_into(model.flat_params):
                    policy.refresh()
            else:
                v = global_model.param_version.value
                if v != version and v % 2 == 0:
                    version = sync_params(model, global_model)
                    policy.refresh()
            last_sync = served
