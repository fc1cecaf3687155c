"""HBM replay data movement (kernel K9): unpack a chunk of packed transition
records (as popped from the shm ring and DMA'd to the device) into the ring
buffers, and gather sampled rows.  Pure-bandwidth kernels: 16-byte vector
loads/stores, one workgroup per record.  The learner's hot path never runs the
state gather at all -- conv1's loader reads uint8 frames through the sampled
index vector -- `gather_states` exists for the torch back-end and for tests.
"""
import ctypes

import torch

from . import lib, native_available


def _p(t):
    return ctypes.c_void_p(t.data_ptr())


def _stream(device):
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def unpack_records(records, n, positions, state0s, state1s, tails, state_bytes, state_stride, tail_floats,
                   pos_start=0, leaves=None):
    """records (>=n, record_bytes) uint8 on the replay device; positions (n,)
    int32 ring slots, or None: record r goes to slot (pos_start + r) mod len(state0s), computed
    inside the kernel.  Writes state0s[pos], state1s[pos] (raw bytes) and tails[pos, :tail_floats].
    leaves: as unpack_runs (GPU only)."""
    if n == 0:
        return
    dev = state0s.device
    m = int(state0s.shape[0])
    if dev.type == "cuda":
        if not native_available():
            from . import native_error
            raise RuntimeError("unpack_records on a GPU needs the HIP library: " + native_error())
        lib().pda_replay_unpack(_p(records), ctypes.c_long(int(records.stride(0))), int(n),
                                _p(positions) if positions is not None else ctypes.c_void_p(0),
                                ctypes.c_long(int(pos_start)), ctypes.c_long(m),
                                _p(state0s), _p(state1s), int(state_bytes), int(state_stride),
                                _p(tails), int(tails.stride(0)), int(tail_floats),
                                _p(leaves[0]) if leaves else ctypes.c_void_p(0),
                                _p(leaves[1]) if leaves else ctypes.c_void_p(0),
                                ctypes.c_int(int(leaves[2]) if leaves else 0),
                                ctypes.c_float(float(leaves[3]) if leaves else 0.0),
                                ctypes.c_float(float(leaves[4]) if leaves else 1.0), _stream(dev))
        return
    if positions is None:
        positions = (torch.arange(n, device=dev) + int(pos_start)) % m
    pos = positions[:n].long()
    rec = records[:n]
    s0 = rec[:, :state_bytes]
    s1 = rec[:, state_stride:state_stride + state_bytes]
    state0s.view(torch.uint8).reshape(state0s.shape[0], -1)[pos] = s0
    state1s.view(torch.uint8).reshape(state1s.shape[0], -1)[pos] = s1
    tail = rec[:, 2 * state_stride:2 * state_stride + 4 * tail_floats].contiguous().view(torch.float32)
    tails[pos, :tail_floats] = tail


def unpack_runs(records, n_runs, codec, counts, plane_base, trans_pos, planes, tails, ref0, refN, window, window_vals,
                leaves=None):
    """Frame-deduplicated replay (parallel/csrc/frame_runs.h): copy the planes of n_runs run records to
    planes[plane_base[r] ...] and their transitions to ring rows trans_pos[r] ... (mod len(tails)): tails, and the two
    frame windows as plane indices ref0 / refN.  counts: [(num_planes, num_transitions)] per run, read by the host
    from the records' headers; window (int32[4], may be None) receives window_vals = (rows, first row, ring length).
    leaves = (tree sum, tree min, index of the priority in a tail, eps, alpha): (GPU only) the kernel also writes every
    new row's leaf (|priority| + eps)^alpha into both trees; the caller repairs the levels above
    (DeviceSumMinTree.repair_range)."""
    if n_runs == 0:
        return
    dev = planes.device
    m = int(tails.shape[0])
    if dev.type == "cuda":
        if not native_available():
            from . import native_error
            raise RuntimeError("unpack_runs on a GPU needs the HIP library: " + native_error())
        L = lib()
        if not getattr(L, "_unpack_runs_ready", False):
            i32, vp, lng = ctypes.c_int, ctypes.c_void_p, ctypes.c_long
            L.pda_replay_unpack_runs.argtypes = [vp, lng, i32, vp, vp, i32, vp, i32, i32, i32, i32, vp, i32, vp, vp, vp,
                                                 i32, i32, i32, vp, vp, i32, ctypes.c_float, ctypes.c_float, vp]
            L.pda_replay_unpack_runs.restype = i32
            L._unpack_runs_ready = True
        pb = (ctypes.c_int * n_runs)(*[int(v) for v in plane_base])
        tp = (ctypes.c_int * n_runs)(*[int(v) for v in trans_pos])
        rc = L.pda_replay_unpack_runs(_p(records), int(records.stride(0)), int(n_runs), pb, tp, m, _p(planes),
                                      int(planes.stride(0)), codec.plane_cap, codec.trans_cap, codec.tail_floats,
                                      _p(tails), int(tails.stride(0)), _p(ref0), _p(refN),
                                      _p(window) if window is not None else ctypes.c_void_p(0),
                                      int(window_vals[0]), int(window_vals[1]), int(window_vals[2]),
                                      _p(leaves[0]) if leaves else ctypes.c_void_p(0),
                                      _p(leaves[1]) if leaves else ctypes.c_void_p(0),
                                      int(leaves[2]) if leaves else 0, float(leaves[3]) if leaves else 0.0,
                                      float(leaves[4]) if leaves else 1.0, _stream(dev))
        if rc != 0:
            raise ValueError("pda_replay_unpack_runs refused its arguments (%d runs, plane stride %d)"
                             % (n_runs, planes.stride(0)))
        return
    rec = records[:n_runs].numpy()
    for r in range(n_runs):
        pl, tl, o0, oN = codec.parse(rec[r])
        n_p, n_t = counts[r]
        assert (n_p, n_t) == (len(pl), len(tl))
        planes[plane_base[r]:plane_base[r] + n_p] = torch.from_numpy(pl.copy())
        pos = (torch.arange(n_t) + int(trans_pos[r])) % m
        tails[pos, :codec.tail_floats] = torch.from_numpy(tl)
        ref0[pos] = torch.from_numpy(o0) + int(plane_base[r])
        refN[pos] = torch.from_numpy(oN) + int(plane_base[r])
    if window is not None:
        window[:3] = torch.tensor(list(window_vals[:3]), dtype=torch.int32)


def gather_states(states, idx, out=None):
    """states (M, ...) -> (B, ...) rows at idx."""
    dev = states.device
    B = int(idx.numel())
    if out is None:
        out = torch.empty((B,) + tuple(states.shape[1:]), dtype=states.dtype, device=dev)
    if dev.type == "cuda" and native_available():
        row_bytes = states[0].numel() * states.element_size()
        if row_bytes % 16 == 0:
            lib().pda_replay_gather(_p(states), _p(idx.to(torch.int32)), B, int(row_bytes), _p(out), _stream(dev))
            return out
    out.copy_(states[idx.long()])
    return out
