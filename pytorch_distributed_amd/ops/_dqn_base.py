"""ctypes argument blocks and prototypes shared by the DQN engine modules (csrc/rl_kernels.hip)."""
import ctypes

import torch

from . import lib

_vp = ctypes.c_void_p
BF16 = torch.bfloat16
HID_SUPPORTED = (256, 512)


def _p(t):
    return t.data_ptr() if t is not None else 0     # (tensors and nn.Planes / _PlaneView alike)


def _stream():
    return _vp(torch.cuda.current_stream().cuda_stream)


class _HeadArgs(ctypes.Structure):     # mirrors DqnHeadArgs in csrc/rl_kernels.hip
    _fields_ = [("h1pre_on0", _vp), ("h1pre_tg1", _vp), ("h1pre_on1", _vp),
                ("b4_on", _vp), ("b4_tg", _vp),
                ("w5_on", _vp), ("b5_on", _vp), ("w5_tg", _vp), ("b5_tg", _vp),
                ("tails", _vp), ("tail_ld", ctypes.c_int), ("off_action", ctypes.c_int),
                ("off_reward", ctypes.c_int), ("off_gamma", ctypes.c_int), ("off_term", ctypes.c_int),
                ("idx", _vp), ("isw", _vp),
                ("h1_on0", _vp), ("dh1", _vp), ("dq", _vp), ("dq_ld", ctypes.c_int),
                ("q_out", _vp), ("td_out", _vp), ("loss_sum", _vp),
                ("leaf_sum", _vp), ("leaf_min", _vp), ("prio_eps", ctypes.c_float), ("prio_alpha", ctypes.c_float),
                ("B", ctypes.c_int), ("HID", ctypes.c_int), ("A", ctypes.c_int),
                ("rng_counter", _vp), ("opt_step", _vp),
                ("h1_plane", ctypes.c_long), ("dh1_plane", ctypes.c_long), ("dq_plane", ctypes.c_long),
                ("h1_slices", ctypes.c_int), ("h1_slice_stride", ctypes.c_long),
                ("h1_f32", _vp), ("dh1_f32", _vp), ("dq_f32", _vp)]


class _ActArgs(ctypes.Structure):      # mirrors ActHeadArgs
    _fields_ = [("h1pre", _vp), ("b4", _vp), ("w5", _vp), ("b5", _vp),
                ("eps", _vp), ("out", _vp), ("out_ld", ctypes.c_int),
                ("q_out", _vp), ("B", ctypes.c_int), ("HID", ctypes.c_int), ("A", ctypes.c_int),
                ("seed", ctypes.c_ulonglong), ("counter", _vp)]


_lib_ready = False


def _L():
    global _lib_ready
    L = lib()
    if not _lib_ready:
        L.pda_dqn_head.argtypes = [ctypes.POINTER(_HeadArgs), _vp]
        L.pda_dqn_head.restype = None
        L.pda_act_head.argtypes = [ctypes.POINTER(_ActArgs), _vp]
        L.pda_act_head.restype = None
        L.pda_uniform_idx.argtypes = [_vp, _vp, ctypes.c_int, _vp, ctypes.c_ulonglong, _vp, _vp]
        L.pda_uniform_idx.restype = None
        L.pda_uniform_f32.argtypes = [_vp, ctypes.c_int, ctypes.c_ulonglong, _vp, _vp]
        L.pda_uniform_f32.restype = None
        _lib_ready = True
    return L


class _Range(object):
    """roctx range around a learner phase when the `roctx` knob is set (visible in rocprofv3 --marker-trace of an
    eager, non-graph run); a no-op otherwise."""
    enabled = False     # set by DqnStepProgram from its knobs

    def __init__(self, name):
        self.name = name

    def __enter__(self):
        if self.enabled:
            torch.cuda.nvtx.range_push(self.name)

    def __exit__(self, *exc):
        if self.enabled:
            torch.cuda.nvtx.range_pop()


