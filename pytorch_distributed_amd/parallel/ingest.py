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
