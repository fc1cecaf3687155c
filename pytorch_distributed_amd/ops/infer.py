"""Python face of the native batched-inference loop (csrc/infer_server.hip, kernel K10).

`NativeInferenceServer(mailbox, weights...)` builds the per-bucket hipGraphs of the
policy network on a stream of its own and can then either be polled from Python
(`serve_once()`, used by tests) or run its own host thread (`start()` / `stop()`):
collect pending mailbox slots -> DMA their observations out of the host-registered
shared-memory block -> replay the graph -> post the answers, with no Python in the loop.

The network weights are the bf16 shadow buffers of a `CnnWeights` / `MlpWeights`
object (ops/dqn_engine.py, ops/mlp_engine.py).  Hosted in the inference-server
process they belong to a private model copy that `refresh_from(global_model)` re-syncs
on the loop's stream; hosted in the learner process (`inference_placement="learner"`)
they are the learner's own live shadow weights, so actors always act on the newest
parameters and no weight publish is needed for them at all.
"""
import ctypes

import torch

from . import lib

_vp = ctypes.c_void_p


class _InferConfig(ctypes.Structure):      # mirrors InferConfig in csrc/infer_server.hip
    _fields_ = [("kind", ctypes.c_int), ("nslots", ctypes.c_int),
                ("seq", _vp), ("obs", _vp), ("obs_stride", ctypes.c_long),
                ("out", _vp), ("out_ld", ctypes.c_int), ("eps", _vp),
                ("C", ctypes.c_int), ("H", ctypes.c_int), ("W", ctypes.c_int), ("HID", ctypes.c_int),
                ("A", ctypes.c_int), ("inv_norm", ctypes.c_float),
                ("D", ctypes.c_int), ("Dp", ctypes.c_int), ("H1", ctypes.c_int), ("H1p", ctypes.c_int),
                ("H2", ctypes.c_int), ("H2p", ctypes.c_int), ("Ap", ctypes.c_int),
                ("w", _vp * 5), ("b", _vp * 5),
                ("seed", ctypes.c_ulonglong), ("high_priority", ctypes.c_int), ("max_bucket", ctypes.c_int),
                ("window_us", ctypes.c_int)]


class _Stats(ctypes.Structure):
    _fields_ = [(n, ctypes.c_ulonglong) for n in ("batches", "requests", "busy_ns", "max_batch", "registered",
                                                  "idle_polls", "window_ns", "live_slots")]


_ready = False


def _L():
    global _ready
    L = lib()
    if not _ready:
        L.pda_infer_create.argtypes = [ctypes.POINTER(_InferConfig)]
        L.pda_infer_create.restype = _vp
        for name in ("pda_infer_serve_once", "pda_infer_start"):
            getattr(L, name).argtypes = [_vp]
            getattr(L, name).restype = ctypes.c_int
        L.pda_infer_stream.argtypes = [_vp]
        L.pda_infer_stream.restype = _vp
        for name in ("pda_infer_stop", "pda_infer_destroy"):
            getattr(L, name).argtypes = [_vp]
            getattr(L, name).restype = None
        L.pda_infer_stats.argtypes = [_vp, ctypes.POINTER(_Stats)]
        L.pda_infer_stats.restype = None
        _ready = True
    return L


class NativeInferenceServer(object):
    def __init__(self, mailbox, device, seed=0, high_priority=False, window_us=0, slot_range=None):
        """slot_range (lo, hi): serve only these mailbox slots (several servers, e.g. one per learner rank, then
        split the env workers between them); default all."""
        self.window_us = int(window_us)
        lo, hi = slot_range if slot_range is not None else (0, mailbox.num_slots)
        if not 0 <= lo < hi <= mailbox.num_slots:
            raise ValueError("slot_range %r outside the mailbox's %d slots" % (slot_range, mailbox.num_slots))
        self.slot_lo, self.slot_hi = int(lo), int(hi)
        self.mailbox = mailbox
        self.device = torch.device(device)
        self.seed = int(seed)
        self.high_priority = bool(high_priority)
        self.handle = None
        self.weights = None
        self.model = None
        self._stream = None

    # ---- construction ---------------------------------------------------------------------
    def _base_config(self, kind):
        mb = self.mailbox
        cfg = _InferConfig()
        cfg.kind = kind
        lo = self.slot_lo           # the server sees slots [lo, hi) as its slots 0 .. hi - lo - 1
        cfg.nslots = self.slot_hi - lo
        cfg.seq = mb.seq.data_ptr() + lo * mb.seq.stride(0) * mb.seq.element_size()
        cfg.obs_stride = mb.obs[0].numel() * mb.obs.element_size()
        cfg.obs = mb.obs.data_ptr() + lo * cfg.obs_stride
        cfg.out_ld = mb.out.shape[1]
        cfg.out = mb.out.data_ptr() + lo * cfg.out_ld * mb.out.element_size()
        cfg.eps = mb.eps.data_ptr() + lo * mb.eps.element_size()
        cfg.seed = (self.seed * 2654435761 + 97) & 0xFFFFFFFFFFFF
        cfg.high_priority = 1 if self.high_priority else 0
        cfg.max_bucket = 0
        cfg.window_us = self.window_us
        return cfg

    def _create(self, cfg):
        with torch.cuda.device(self.device):
            torch.cuda.synchronize(self.device)         # weights fully written before the warm-up passes read them
            self.handle = _L().pda_infer_create(ctypes.byref(cfg))
        if not self.handle:
            raise RuntimeError("native inference server could not be built (see stderr)")
        self._stream = torch.cuda.ExternalStream(int(_L().pda_infer_stream(self.handle)), device=self.device)
        import atexit
        atexit.register(self.stop)      # the loop thread must not outlive the mailbox mapping
        return self

    def build_cnn(self, weights, geom, norm_val, model=None):
        """weights: ops.dqn_engine.CnnWeights on this device (kept alive here)."""
        assert self.mailbox.obs.dtype == torch.uint8
        self.weights, self.model = weights, model
        cfg = self._base_config(0)
        cfg.C, cfg.H, cfg.W, cfg.HID, cfg.A = geom.C, geom.H, geom.W, geom.HID, geom.A
        cfg.inv_norm = 1.0 / float(norm_val)
        for i in range(5):
            cfg.w[i] = weights.w[i].data_ptr()
            cfg.b[i] = weights.b[i].data_ptr()
        return self._create(cfg)

    def build_ddpg_actor(self, weights, model=None):
        """weights: ops.mlp_engine.MlpWeights holding actor.0/2/4 shadows."""
        assert self.mailbox.obs.dtype == torch.float32
        self.weights, self.model = weights, model
        cfg = self._base_config(1)
        H1, D, Dp = weights.shape["actor.0"]
        H2, _, H1p = weights.shape["actor.2"]
        A, _, H2p = weights.shape["actor.4"]
        cfg.D, cfg.Dp, cfg.H1, cfg.H1p, cfg.H2, cfg.H2p = D, Dp, H1, H1p, H2, H2p
        cfg.A, cfg.Ap = A, max(8, (A + 7) // 8 * 8)
        assert self.mailbox.obs[0].numel() == D
        for i, name in enumerate(("actor.0", "actor.2", "actor.4")):
            cfg.w[i] = weights.w[name].data_ptr()
            cfg.b[i] = weights.b[name].data_ptr()
        return self._create(cfg)

    # ---- weights ------------------------------------------------------------------------------
    def refresh_from(self, global_model):
        """Pull a snapshot of the shared parameters into the private model and rebuild the bf16
        shadows ON THE LOOP'S STREAM, i.e. ordered between two inference batches."""
        from ..utils.param_store import sync_params
        with torch.cuda.stream(self._stream):
            version = sync_params(self.model, global_model)
            self.weights.refresh()
        return version

    def refresh_from_receiver(self, receiver):
        """Same, with the snapshot that arrived by collective broadcast (parallel/weight_sync.py)."""
        with torch.cuda.stream(self._stream):
            if receiver.poll_into(self.model.flat_params):
                self.weights.refresh()
                return True
        return False

    # ---- serving ------------------------------------------------------------------------------
    def serve_once(self):
        with torch.cuda.device(self.device):
            n = int(_L().pda_infer_serve_once(self.handle))
        if n < 0:
            raise RuntimeError("native inference server: GPU error (see stderr)")
        return n

    def start(self):
        with torch.cuda.device(self.device):
            _L().pda_infer_start(self.handle)

    def stop(self):
        if self.handle:
            _L().pda_infer_stop(self.handle)

    def stats(self):
        st = _Stats()
        _L().pda_infer_stats(self.handle, ctypes.byref(st))
        d = {n: int(getattr(st, n)) for n, _ in _Stats._fields_}
        d["mean_batch"] = d["requests"] / max(d["batches"], 1)
        d["mean_batch_us"] = d["busy_ns"] / 1e3 / max(d["batches"], 1)
        d["mean_window_us"] = d["window_ns"] / 1e3 / max(d["batches"], 1)
        return d

    def close(self):
        if self.handle:
            _L().pda_infer_destroy(self.handle)
            self.handle = None


def make_native_server(args, model, mailbox, device, high_priority=False, weights=None, slot_range=None):
    """Server for the configured model family, or None when the family has no native program
    (the caller then serves from Python)."""
    window_us = int(getattr(args, "inference_window_us", 0))
    if args.model_type == "dqn-cnn":
        from .dqn_engine import CnnGeometry, CnnWeights
        geom = CnnGeometry(model.input_dims, model.output_dims, model.critic[1][0].out_features)
        w = weights if weights is not None else CnnWeights(model, geom, with_grad=False)
        return NativeInferenceServer(mailbox, device, seed=args.seed + 4242, high_priority=high_priority,
                                     window_us=window_us, slot_range=slot_range).build_cnn(w, geom, model.norm_val,
                                                                                           model)
    if args.model_type == "ddpg-mlp":
        from .mlp_engine import DDPG_LAYERS, MlpWeights
        w = weights if weights is not None else MlpWeights(model, DDPG_LAYERS, with_grad=False,
                                                           fp32_only=("critic.1.2",))
        return NativeInferenceServer(mailbox, device, seed=args.seed + 4242, high_priority=high_priority,
                                     window_us=window_us, slot_range=slot_range).build_ddpg_actor(w, model)
    return None
