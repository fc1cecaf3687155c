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
                   pos_start=0):
    """records (>=n, record_bytes) uint8 on the replay device; positions (n,)
    int32 ring slots, or None: record r goes to slot (pos_start + r) mod len(state0s), computed
    inside the kernel.  Writes state0s[pos], state1s[pos] (raw bytes) and tails[pos, :tail_floats]."""
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
                                _p(tails), int(tails.stride(0)), int(tail_floats), _stream(dev))
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
