"""ShmRing: Python face of the lock-free MPMC ring (csrc/shm_ring.cpp).

The ring's memory is a torch uint8 tensor in shared memory, so the object
pickles across spawn like any shared tensor; each process re-resolves the base
pointer lazily.  Records are fixed-size byte blobs; `TransitionCodec` defines
the transition layout used between actors and device replay:

    [state0 | pad16][stateN | pad16][action f32*A][reward f32*T][gamma f32*T]
    [terminal f32*T][priority f32][pad to 16]
"""
import time

import numpy as np
import torch

from . import hostlib


def _round_up(v, a):
    return (v + a - 1) // a * a


class TransitionCodec(object):
    def __init__(self, state_shape, state_dtype, action_shape, num_tasks):
        self.state_shape = tuple(state_shape)
        self.state_np = np.uint8 if state_dtype in (torch.uint8, "uint8") else np.float32
        self.state_bytes = int(np.prod(self.state_shape)) * np.dtype(self.state_np).itemsize
        self.state_stride = _round_up(self.state_bytes, 16)
        self.action_shape = int(action_shape)
        self.num_tasks = int(num_tasks)
        self.tail_floats = self.action_shape + 3 * self.num_tasks + 1
        self.tail_offset = 2 * self.state_stride
        self.record_bytes = _round_up(self.tail_offset + 4 * self.tail_floats, 16)

    def tail(self, action, reward, gamma, terminal, priority):
        t = np.empty(self.tail_floats, dtype=np.float32)
        a, k = self.action_shape, self.num_tasks
        t[:a] = np.asarray(action, dtype=np.float32).reshape(-1)
        t[a:a + k] = np.asarray(reward, dtype=np.float32).reshape(-1)
        t[a + k:a + 2 * k] = np.asarray(gamma, dtype=np.float32).reshape(-1)
        t[a + 2 * k:a + 3 * k] = float(terminal)
        t[a + 3 * k] = float(priority)
        return t


class RunCodec(object):
    """Layout of a frame-deduplicated run record (csrc/frame_runs.h): header | transition table | planes.
    A state [C, ...] is `window` = C planes of plane_bytes each."""

    HEADER = 64
    MAGIC = 0x52554E31

    def __init__(self, state_shape, action_shape, num_tasks, plane_cap=96, trans_cap=None):
        self.state_shape = tuple(state_shape)
        if len(self.state_shape) < 2:
            raise ValueError("frame-deduplicated replay needs [C, ...] frame stacks, not states of shape %r"
                             % (self.state_shape,))
        self.window = int(self.state_shape[0])
        self.plane_bytes = int(np.prod(self.state_shape[1:]))
        if self.plane_bytes % 16:
            raise ValueError("frame-deduplicated replay needs planes of a multiple of 16 bytes (got %d)" % self.plane_bytes)
        self.plane_stride = self.plane_bytes
        self.tail_floats = int(action_shape) + 3 * int(num_tasks) + 1
        self.trans_stride = _round_up(4 * self.tail_floats + 8, 16)
        self.plane_cap = int(plane_cap)
        self.trans_cap = int(trans_cap or plane_cap)
        self.planes_offset = self.HEADER + self.trans_cap * self.trans_stride
        self.record_bytes = self.planes_offset + self.plane_cap * self.plane_stride

    def header(self, rec):
        """(num_planes, num_transitions) of one record (uint8 numpy row)."""
        h = rec[:16].view(np.uint32)
        if int(h[0]) != self.MAGIC:
            raise ValueError("not a run record (magic %#x)" % int(h[0]))
        return int(h[1]), int(h[2])

    def parse(self, rec):
        """-> (planes uint8 [np, plane_bytes], tails f32 [nt, tail_floats], off0 int32 [nt], offN int32 [nt])"""
        n_p, n_t = self.header(rec)
        table = rec[self.HEADER:self.HEADER + n_t * self.trans_stride].reshape(n_t, self.trans_stride)
        tails = table[:, :4 * self.tail_floats].copy().view(np.float32)
        offs = table[:, 4 * self.tail_floats:4 * self.tail_floats + 8].copy().view(np.int32)
        planes = rec[self.planes_offset:self.planes_offset + n_p * self.plane_stride].reshape(n_p, self.plane_stride)
        return planes, tails, offs[:, 0].copy(), offs[:, 1].copy()

    def pack(self, planes, tails, off0, offN):
        """The inverse of parse (Python producers, tests): -> uint8 [record_bytes]."""
        n_p, n_t = len(planes), len(tails)
        assert n_p <= self.plane_cap and n_t <= self.trans_cap
        rec = np.zeros(self.record_bytes, dtype=np.uint8)
        rec[:16].view(np.uint32)[:] = [self.MAGIC, n_p, n_t, self.window]
        table = rec[self.HEADER:self.HEADER + n_t * self.trans_stride].reshape(n_t, self.trans_stride)
        table[:, :4 * self.tail_floats] = np.ascontiguousarray(tails, dtype=np.float32).reshape(n_t, -1).view(np.uint8)
        offs = np.stack([np.asarray(off0, dtype=np.int32), np.asarray(offN, dtype=np.int32)], axis=1)
        table[:, 4 * self.tail_floats:4 * self.tail_floats + 8] = offs.view(np.uint8)
        rec[self.planes_offset:self.planes_offset + n_p * self.plane_stride] = \
            np.ascontiguousarray(planes, dtype=np.uint8).reshape(-1)
        return rec


class ShmRing(object):
    def __init__(self, num_slots, payload_bytes):
        L = hostlib.lib()
        self.num_slots = int(num_slots)
        self.payload_bytes = int(payload_bytes)
        nbytes = int(L.pda_ring_bytes(self.num_slots, self.payload_bytes)) + 64
        self.region = torch.zeros(nbytes, dtype=torch.uint8).share_memory_()
        self._base = None
        rc = L.pda_ring_init(self.base, nbytes - (self.base - self.region.data_ptr()),
                             self.num_slots, self.payload_bytes)
        if rc != 0:
            raise RuntimeError("pda_ring_init failed: %d" % rc)

    @property
    def base(self):
        if self._base is None:
            self._base = _round_up(self.region.data_ptr(), 64)
        return self._base

    def __getstate__(self):
        d = self.__dict__.copy()
        d["_base"] = None
        return d

    def push_segments(self, segs, timeout_s=30.0):
        """segs: up to 4 contiguous numpy arrays forming one record.  Blocks
        (sleep back-off) while the ring is full: that is the back-pressure when
        the learner ingests slower than actors produce."""
        L = hostlib.lib()
        args = []
        for s in segs:
            args += [s.ctypes.data, s.nbytes]
        while len(args) < 8:
            args += [None, 0]
        deadline = None
        delay = 1e-5
        while True:
            rc = L.pda_ring_push(self.base, *args)
            if rc == 1:
                return True
            if rc < 0:
                raise ValueError("record larger than ring payload")
            if deadline is None:
                deadline = time.time() + timeout_s
            elif time.time() > deadline:
                return False
            time.sleep(delay)
            delay = min(delay * 2, 2e-3)

    def pop_many(self, dst, max_records):
        """dst: uint8 tensor/array (max_records, stride) contiguous; returns count."""
        L = hostlib.lib()
        if torch.is_tensor(dst):
            ptr, stride = dst.data_ptr(), dst.stride(0)
        else:
            ptr, stride = dst.ctypes.data, dst.strides[0]
        return int(L.pda_ring_pop_many(self.base, ptr, stride, int(max_records)))

    def stats(self):
        L = hostlib.lib()
        return {"pushed": int(L.pda_ring_pushed(self.base)), "popped": int(L.pda_ring_popped(self.base)),
                "full_events": int(L.pda_ring_full_events(self.base)), "backlog": int(L.pda_ring_backlog(self.base)),
                "skipped": int(L.pda_ring_skipped(self.base))}
