"""Hand-written gfx950 HIP kernels and their Python bindings.

The kernels live in ops/csrc/*.hip and are compiled by ops/build.py into ONE
shared library, `ops/libpda_hip.so`, with a plain C ABI (raw device pointers +
hipStream_t), loaded here through ctypes.  There is no torch C++ extension
layer and no hipify step: the sources are HIP from the first line, compile in
seconds without torch headers, and every launch goes on the stream torch
currently has active, so whole learner steps can be captured into a hipGraph
with torch.cuda.graph().

On a box with a GPU the native path is mandatory for compute="native"/"auto":
if the library is missing the ops raise instead of silently running eager
PyTorch.
"""
import ctypes
import os

_LIB = None
_ERR = None
_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libpda_hip.so")


def _load():
    global _LIB, _ERR
    if _LIB is not None or _ERR is not None:
        return _LIB
    if not os.path.exists(LIB_PATH):
        _ERR = "%s not built (run `python -m pytorch_distributed_amd.ops.build`)" % LIB_PATH
        return None
    try:
        _LIB = ctypes.CDLL(LIB_PATH)
    except OSError as exc:      # e.g. no libamdhip64 on a CPU-only box
        _ERR = "cannot load %s: %s" % (LIB_PATH, exc)
    return _LIB


def lib():
    """The loaded library; raises if it is unavailable."""
    handle = _load()
    if handle is None:
        raise RuntimeError(_ERR)
    return handle


def native_error():
    _load()
    return _ERR or ""


def native_available():
    """True when the HIP library loads and a GPU is visible."""
    import torch
    if not torch.cuda.is_available():
        return False
    return _load() is not None


def require_native():
    import torch
    if torch.cuda.is_available() and _load() is None:
        raise RuntimeError("GPU present but HIP kernels unavailable: " + _ERR)
    return _LIB
