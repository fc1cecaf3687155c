"""Loader for the host-side C++ runtime (`parallel/libpda_host.so`): the
lock-free shared-memory ring and the atomics behind the inference mailbox.
Pure host code (g++), so it builds and loads on a CPU-only box; built on first
use if the in-tree .so is missing.
"""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(_HERE, "csrc", "shm_ring.cpp")
HEADERS = [os.path.join(_HERE, "csrc", "mailbox.h"), os.path.join(_HERE, "csrc", "frame_runs.h")]
LIB_PATH = os.path.join(_HERE, "libpda_host.so")
_LIB = None


LAST_REPORT = {}       # "shm_ring.cpp" -> "compiled" | "reused" for the most recent build()


def build(force=False, sanitize=None):
    out = LIB_PATH if sanitize is None else LIB_PATH.replace(".so", "_%s.so" % sanitize)
    force = force or os.environ.get("PDA_BUILD_FORCE", "0") == "1"
    if not force and os.path.exists(out) and os.path.getmtime(out) >= max(os.path.getmtime(f) for f in [SRC] + HEADERS):
        LAST_REPORT.setdefault(os.path.basename(SRC), "reused")    # (lib() re-checks after an explicit build)
        return out
    LAST_REPORT[os.path.basename(SRC)] = "compiled"
    cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", "-pthread", "-o", out + ".tmp%d" % os.getpid(), SRC]
    if sanitize:
        cmd[1:1] = ["-fsanitize=%s" % sanitize, "-g"]
    subprocess.check_call(cmd)
    os.replace(out + ".tmp%d" % os.getpid(), out)
    return out


def lib():
    global _LIB
    if _LIB is None:
        L = ctypes.CDLL(build())
        u64, u32, vp, i32 = ctypes.c_uint64, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_int
        L.pda_ring_bytes.restype = u64
        L.pda_ring_bytes.argtypes = [u32, u32]
        L.pda_ring_init.restype = i32
        L.pda_ring_init.argtypes = [vp, u64, u32, u32]
        L.pda_ring_valid.restype = i32
        L.pda_ring_valid.argtypes = [vp]
        L.pda_ring_push.restype = i32
        L.pda_ring_push.argtypes = [vp, vp, u32, vp, u32, vp, u32, vp, u32]
        L.pda_ring_pop_many.restype = i32
        L.pda_ring_pop_many.argtypes = [vp, vp, u32, u32]
        L.pda_ring_set_stuck_ms.restype = None
        L.pda_ring_set_stuck_ms.argtypes = [vp, u32]
        L.pda_ring_claim_and_abandon.restype = ctypes.c_int64
        L.pda_ring_claim_and_abandon.argtypes = [vp]
        for name in ("pda_ring_pushed", "pda_ring_popped", "pda_ring_full_events", "pda_ring_backlog",
                     "pda_ring_skipped"):
            getattr(L, name).restype = u64
            getattr(L, name).argtypes = [vp]
        L.pda_atomic_load.restype = u64
        L.pda_atomic_load.argtypes = [vp]
        L.pda_atomic_store.restype = None
        L.pda_atomic_store.argtypes = [vp, u64]
        L.pda_atomic_add.restype = u64
        L.pda_atomic_add.argtypes = [vp, u64]
        L.pda_atomic_wait_ge.restype = u64
        L.pda_atomic_wait_ge.argtypes = [vp, u64, u64]
        L.pda_pump_create.restype = vp
        L.pda_pump_create.argtypes = [vp, i32, vp, u32, u32, u32]
        L.pda_pump_take.restype = i32
        L.pda_pump_take.argtypes = [vp, ctypes.POINTER(ctypes.c_int)]
        L.pda_pump_release.restype = None
        L.pda_pump_release.argtypes = [vp, i32]
        L.pda_pump_moved.restype = u64
        L.pda_pump_moved.argtypes = [vp]
        L.pda_pump_destroy.restype = None
        L.pda_pump_destroy.argtypes = [vp]
        dbl = ctypes.c_double
        L.pda_nstep_create.restype = vp
        L.pda_nstep_create.argtypes = [i32, dbl, i32, u32, u32, i32, i32, u32]
        L.pda_nstep_create_dedup.restype = vp
        L.pda_nstep_create_dedup.argtypes = [i32, dbl, i32, u32, i32, i32, i32, i32, i32, u32]
        L.pda_run_record_size.restype = u64
        L.pda_run_record_size.argtypes = [i32, i32, i32, u32, i32]
        L.pda_nstep_ship.restype = i32
        L.pda_nstep_ship.argtypes = [vp, vp, u32]
        L.pda_nstep_run_stats.restype = None
        L.pda_nstep_run_stats.argtypes = [vp, vp]
        L.pda_nstep_destroy.restype = None
        L.pda_nstep_destroy.argtypes = [vp]
        L.pda_nstep_emitted.restype = u64
        L.pda_nstep_emitted.argtypes = [vp]
        L.pda_nstep_reset.restype = None
        L.pda_nstep_reset.argtypes = [vp, vp]
        L.pda_nstep_push.restype = i32
        L.pda_nstep_push.argtypes = [vp, vp, vp, dbl, vp, i32, dbl, dbl, i32, u32]
        L.pda_nstep_flush.restype = i32
        L.pda_nstep_flush.argtypes = [vp, vp, dbl, i32, u32]
        L.pda_mailbox_request.restype = u64
        L.pda_mailbox_request.argtypes = [vp, i32, u64, vp, vp, u64, u32, u32]
        L.pda_mailbox_send.restype = None
        L.pda_mailbox_send.argtypes = [vp, i32, u64, vp, vp, u64]
        L.pda_mailbox_wait.restype = u64
        L.pda_mailbox_wait.argtypes = [vp, i32, u64, u32, u32]
        L.pda_mailbox_collect.restype = i32
        L.pda_mailbox_collect.argtypes = [vp, i32, vp, vp, i32]
        L.pda_mailbox_post.restype = None
        L.pda_mailbox_post.argtypes = [vp, vp, vp, i32]
        _LIB = L
    return _LIB
