"""Flat-buffer optimizer / target-update kernels (K13-K15): python bindings.

`FlatAdam` drives csrc/rl_kernels.hip:adam_kernel: ONE launch clips the gradient
by value, applies Adam to every parameter of a model's flat fp32 buffer and
re-emits the bf16 "shadow" weights in the layouts the MFMA kernels read.
Gradients and moments of shadowed weights are kept in that kernel layout.  Hyper-parameters and the step counter live
in device memory so the launch can sit inside a captured hipGraph.
`polyak_` is the fused hard/soft target update (reference utils/helpers.py:19-25
loops over parameters in Python).
"""
import ctypes

import numpy as np
import torch

from . import lib

OPT_CHUNK = 2048
OPT_ROW_LDS = 4352      # csrc/rl_kernels.hip: LDS floats for one weight row in the transposition path
_vp = ctypes.c_void_p


def _p(t):
    return _vp(t.data_ptr()) if t is not None else _vp(0)


def _stream():
    return _vp(torch.cuda.current_stream().cuda_stream)


class SegmentTable(object):
    """Device tables describing each parameter tensor of a flat buffer.
    entries: dicts with start, count, cols, C, P, shadow_off, shadow_ld, shadow(bool), tr_off, tr_ld."""

    def __init__(self, entries, device):
        rows, chunks = [], []
        for si, e in enumerate(entries):
            rows.append([e["start"], e["count"], e.get("cols", e["count"]), e.get("C", 1),
                         e.get("P", e.get("cols", e["count"])), e.get("shadow_off", 0), e.get("shadow_ld", 0),
                         (1 if e.get("shadow", False) else 0) | (2 if e.get("fine", False) else 0)
                         | (4 if e.get("medium", False) else 0),
                         e.get("tr_off", 0), e.get("tr_ld", 0)])     # transposed shadow copy (csrc/optim.h SegEntry)
            cols, C, P = rows[-1][2], rows[-1][3], rows[-1][4]
            row_mode = bool(e.get("shadow", False)) and C > 1 and (P | 1) * C <= OPT_ROW_LDS
            step = cols if row_mode else (64 if e.get("fine", False) else 256 if e.get("medium", False) else OPT_CHUNK)
            for first in range(0, e["count"], step):       # one row per block | flat chunks
                chunks.append([si, first])
        self.segs = torch.tensor(rows, dtype=torch.int32).to(device)
        self.chunks = torch.tensor(chunks, dtype=torch.int32).to(device)
        self.nchunks = len(chunks)
        self._chunk_list = chunks

    def subset(self, keep):
        """A view of this table restricted to the segments for which keep(segment_index) is true
        (same segment rows, fewer chunks)."""
        sub = object.__new__(SegmentTable)
        sub.segs = self.segs
        chunks = [c for c in self._chunk_list if keep(c[0])]
        sub._chunk_list = chunks
        sub.chunks = torch.tensor(chunks if chunks else [[0, 0]], dtype=torch.int32).to(self.segs.device)
        sub.nchunks = len(chunks)
        return sub


def plain_table(numel, device):
    return SegmentTable([{"start": 0, "count": int(numel)}], device)


class AdamExtras(ctypes.Structure):     # mirrors AdamExtras in csrc/rl_kernels.hip
    _fields_ = [("slab", _vp), ("slabinfo", _vp), ("next_idx", _vp), ("next_w", _vp), ("next_B", ctypes.c_int),
                ("size", _vp), ("seed", ctypes.c_ulonglong), ("counter", _vp), ("shadow_plane", ctypes.c_long),
                ("tinfo", _vp), ("tgt", _vp), ("tgt_shadow", _vp), ("tau", ctypes.c_float),
                ("rep_sum", _vp), ("rep_min", _vp), ("rep_idx", _vp), ("rep_n", ctypes.c_int),
                ("rep_parent_off", ctypes.c_int), ("rep_count", _vp), ("rep_top_off", ctypes.c_int),
                ("rep_top_n", ctypes.c_int), ("rep_bar", _vp),
                ("ps_idx", _vp), ("ps_w", _vp), ("ps_B", ctypes.c_int), ("ps_depth", ctypes.c_int),
                ("ps_leaves", ctypes.c_int), ("ps_beta", ctypes.c_float), ("ps_offs", ctypes.c_int * 8),
                ("ps_done", _vp), ("ps_count", _vp)]


class FlatAdam(object):
    def __init__(self, flat_params, flat_grads, table, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0,
                 clip=np.inf, grad_scale=1.0, shadow=None, shadow_plane=0):
        """shadow_plane: element offset from the shadow's hi plane to its lo plane (fp32-precision engines keep
        split-bf16 shadow weights); 0 = plain bf16 shadow."""
        dev = flat_params.device
        self.p, self.g, self.table, self.shadow = flat_params, flat_grads, table, shadow
        self.shadow_plane = int(shadow_plane)
        self._plane_extras = None
        if self.shadow_plane:
            self._plane_extras = AdamExtras()
            self._plane_extras.shadow_plane = self.shadow_plane
        self.m = torch.zeros_like(flat_params)
        self.v = torch.zeros_like(flat_params)
        clip = float(clip) if np.isfinite(clip) else 3.0e38
        self.hyper = torch.tensor([lr, betas[0], betas[1], eps, weight_decay, clip, grad_scale],
                                  dtype=torch.float32).to(dev)
        self.step_dev = torch.zeros(1, dtype=torch.int32, device=dev)
        L = lib()
        L.pda_adam.argtypes = [_vp] * 7 + [ctypes.c_int, _vp, _vp, _vp]
        L.pda_adam.restype = None
        L.pda_adam_x.argtypes = [_vp] * 7 + [ctypes.c_int, _vp, _vp, ctypes.POINTER(AdamExtras), _vp]
        L.pda_adam_x.restype = None
        L.pda_tick.argtypes = [_vp, _vp, _vp]
        L.pda_tick.restype = None

    def set_lr(self, lr):
        self.hyper[0] = float(lr)

    def step(self, tick=False, table=None, extras=None):
        """tick=True bumps the step counter with an extra tiny launch; engines whose
        loss kernel already bumps it (dqn_head / td_vector) pass False.  `table` restricts the
        launch to a subset of the segments (SegmentTable.subset) -- the DP step updates the fc
        tail while the conv gradients are still being all-reduced."""
        if tick:
            lib().pda_tick(_p(self.step_dev), _vp(0), _stream())
        t = table if table is not None else self.table
        if t.nchunks == 0:
            return
        if extras is None:
            extras = self._plane_extras
        elif self.shadow_plane:
            extras.shadow_plane = self.shadow_plane
        if extras is not None:      # slab reduction and/or next-step sampler folded into this launch
            lib().pda_adam_x(_p(self.p), _p(self.g), _p(self.m), _p(self.v), _p(self.shadow), _p(t.segs),
                             _p(t.chunks), t.nchunks, _p(self.hyper), _p(self.step_dev), ctypes.byref(extras),
                             _stream())
            return
        lib().pda_adam(_p(self.p), _p(self.g), _p(self.m), _p(self.v), _p(self.shadow), _p(t.segs),
                       _p(t.chunks), t.nchunks, _p(self.hyper), _p(self.step_dev), _stream())

    def set_grad_scale(self, scale):
        """Gradients are multiplied by this before clipping (1/world_size after a SUM all-reduce)."""
        self.hyper[6] = float(scale)


def polyak_(tgt_flat, src_flat, tau, table=None, shadow=None, shadow_plane=0):
    """tgt = (1-tau) tgt + tau src over flat fp32 buffers; tau >= 1 is a hard copy.
    With a table + shadow the target's bf16 shadow weights are re-emitted too (shadow_plane: see FlatAdam)."""
    L = lib()
    if table is None:
        L.pda_polyak_flat.argtypes = [_vp, _vp, ctypes.c_long, ctypes.c_float, _vp]
        L.pda_polyak_flat.restype = None
        if tau >= 1.0:
            tgt_flat.copy_(src_flat)
        else:
            L.pda_polyak_flat(_p(tgt_flat), _p(src_flat), int(tgt_flat.numel()), float(tau), _stream())
        return
    L.pda_polyak.argtypes = [_vp] * 5 + [ctypes.c_int, ctypes.c_float, ctypes.c_long, _vp]
    L.pda_polyak.restype = None
    L.pda_polyak(_p(tgt_flat), _p(src_flat), _p(shadow), _p(table.segs), _p(table.chunks), table.nchunks,
                 float(tau), int(shadow_plane), _stream())


def polyak_params_(dst_params, src_params, tau):
    """Soft update of parameter lists living on a GPU.  Returns False when the
    native library is unavailable (caller falls back to torch._foreach)."""
    from . import native_available
    if not native_available():
        return False
    for d, s in zip(dst_params, src_params):
        if not (d.is_contiguous() and s.is_contiguous() and d.dtype == torch.float32 and s.dtype == torch.float32):
            return False
    # parameters of a flattened model are adjacent views: merge them into runs so a
    # whole model is one launch
    runs = []
    for d, s in zip(dst_params, src_params):
        dp, sp, n = d.data_ptr(), s.data_ptr(), d.numel()
        if runs and runs[-1][0] + 4 * runs[-1][2] == dp and runs[-1][1] + 4 * runs[-1][2] == sp:
            runs[-1][2] += n
        else:
            runs.append([dp, sp, n])
    L = lib()
    L.pda_polyak_flat.argtypes = [_vp, _vp, ctypes.c_long, ctypes.c_float, _vp]
    L.pda_polyak_flat.restype = None
    for dp, sp, n in runs:
        L.pda_polyak_flat(_vp(dp), _vp(sp), n, float(tau), _stream())
    return True


class LaggedReadout(object):
    """Read a small device accumulator (the learner's loss sums) out WITHOUT a host synchronisation: each call
    enqueues an asynchronous copy of the accumulator into pinned memory (and its reset, stream-ordered behind the
    copy) and returns the values of the PREVIOUS call's copy, which has long landed.  The learner loop reads its
    loss every `learner_freq` updates; a blocking read there drains the whole launch queue and leaves the GPU idle
    until the host has caught up again (~250 us per read-out inside the job, profiles/r3_node_learner_loop.txt).
    The first call has nothing older to return and blocks once."""

    def __init__(self, acc):
        self.acc = acc
        self.host = [torch.zeros(acc.numel(), dtype=acc.dtype).pin_memory() for _ in range(2)]
        self.evt = [torch.cuda.Event(), torch.cuda.Event()]
        self.steps = [0, 0]
        self.k = 0
        self.last = None

    def read(self, steps):
        """-> (accumulator values as a host tensor, number of steps they cover), one call behind."""
        i = self.k & 1
        self.host[i].copy_(self.acc.reshape(-1), non_blocking=True)
        self.acc.zero_()
        self.evt[i].record()
        self.steps[i] = int(steps)
        self.k += 1
        j = i ^ 1
        if self.k == 1:
            self.evt[i].synchronize()
            j = i
        else:
            self.evt[j].synchronize()       # recorded a whole read-out period ago: returns at once
        return self.host[j].clone(), max(self.steps[j], 1)
