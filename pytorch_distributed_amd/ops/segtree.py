"""Device-resident sum/min segment tree for prioritized replay (kernels K7/K8).

Finishes what the reference left as dead code (utils/segment_tree.py:13-69) and
re-shapes it for CDNA4: instead of a binary tree walked one dependent load per
level (21 levels for 2M leaves), the tree has RADIX 64 -- one child per lane of
a 64-wide wavefront.  A sample is one wave: at each level the 64 lanes load the
64 children of the current node in one coalesced 256 B read, run a wave-level
inclusive scan (DPP/shuffles), ballot for the first lane whose prefix reaches the
target mass and descend.  2M leaves are 4 levels (64^4 = 16.7M) instead of 21.
Priority updates write the leaves, then repair one level per launch with a
64-lane wave reduction per touched node (idempotent on duplicates, no atomics,
deterministic).

`DeviceSumMinTree` holds both trees in one fp32 buffer.  On CPU tensors the same
API runs a torch/numpy fallback with identical layout and results (used by the
no-GPU tests and as the oracle's second opinion next to utils.SumMinTree).
"""
import ctypes

import numpy as np
import torch

from . import lib, native_available

RADIX = 64


def level_sizes(num_leaves):
    """Padded node counts per level, leaves first; the last level has <= 64 nodes."""
    sizes = [max(RADIX, (num_leaves + RADIX - 1) // RADIX * RADIX)]
    while sizes[-1] > RADIX:
        n = (sizes[-1] + RADIX - 1) // RADIX
        sizes.append(n if n <= RADIX else (n + RADIX - 1) // RADIX * RADIX)
    return sizes


class DeviceSumMinTree(object):
    def __init__(self, num_leaves, device):
        self.num_leaves = int(num_leaves)
        self.device = torch.device(device)
        self.sizes = level_sizes(self.num_leaves)
        self.offsets = [0]
        for s in self.sizes[:-1]:
            self.offsets.append(self.offsets[-1] + s)
        self.total_nodes = self.offsets[-1] + RADIX     # top level padded to 64
        self.depth = len(self.sizes)
        self.sum = torch.zeros(self.total_nodes, dtype=torch.float32, device=self.device)
        self.min = torch.full((self.total_nodes,), float("inf"), dtype=torch.float32, device=self.device)
        self.native = self.device.type == "cuda" and native_available()
        if self.device.type == "cuda" and not self.native:
            from . import native_error
            raise RuntimeError("DeviceSumMinTree on a GPU needs the HIP library: " + native_error())
        self._offs_t = torch.tensor(self.offsets + [0] * (8 - len(self.offsets)), dtype=torch.int32)
        assert self.depth <= 8

    # ---- level helpers (fallback) -----------------------------------------
    def _repair_cpu(self, leaf_inds):
        idx = torch.unique(leaf_inds.long())
        for lvl in range(1, self.depth):
            idx = torch.unique(idx // RADIX)
            child0 = self.offsets[lvl - 1] + idx * RADIX
            gather = child0[:, None] + torch.arange(RADIX)[None, :]
            self.sum[self.offsets[lvl] + idx] = self.sum[gather].sum(1)
            self.min[self.offsets[lvl] + idx] = self.min[gather].min(1)[0]

    # ---- API -----------------------------------------------------------------
    def update(self, leaf_inds, values):
        """leaf_inds (n,) int, values (n,) fp32 already in priority space."""
        n = int(leaf_inds.numel())
        if n == 0:
            return
        leaf_inds = leaf_inds.to(self.device, torch.int32).contiguous()
        values = values.to(self.device, torch.float32).contiguous().reshape(-1)
        if not self.native:
            self.sum[leaf_inds.long()] = values
            self.min[leaf_inds.long()] = values
            self._repair_cpu(leaf_inds)
            return
        L = lib()
        stream = torch.cuda.current_stream(self.device).cuda_stream
        L.pda_segtree_set_leaves(_p(self.sum), _p(self.min), _p(leaf_inds), _p(values), n, _s(stream))
        self.repair(leaf_inds, n)

    def clear(self, leaf_inds):
        """Empty leaves: no mass in the sum tree, +inf in the min tree (as never-written leaves are)."""
        if int(leaf_inds.numel()) == 0:
            return
        leaf_inds = leaf_inds.to(self.device, torch.int32).contiguous()
        li = leaf_inds.long()
        self.sum[li] = 0.0
        self.min[li] = float("inf")
        if not self.native:
            self._repair_cpu(leaf_inds)
        else:
            self.repair(leaf_inds)

    def repair(self, leaf_inds, n=None, first_level=1, stop_level=None):
        """Recompute the ancestors of the given leaves (their values were already written, e.g. by the fused TD
        kernel) on levels first_level .. stop_level - 1 (default: all of them, 1 .. depth - 1)."""
        n = int(leaf_inds.numel()) if n is None else n
        if not self.native:
            self._repair_cpu(leaf_inds.cpu())
            return
        L = lib()
        stream = torch.cuda.current_stream(self.device).cuda_stream
        for lvl in range(first_level, self.depth if stop_level is None else stop_level):
            L.pda_segtree_repair_level(_p(self.sum), _p(self.min), _p(leaf_inds), n,
                                       self.offsets[lvl - 1], self.offsets[lvl], lvl, _s(stream))

    def repair_range(self, start, count):
        """Repair above `count` consecutive leaves (mod num_leaves) from `start` whose values were just written:
        native trees in one single-workgroup launch, up to the level the sampling kernel re-derives (lazy_top)."""
        count = int(min(count, self.num_leaves))
        if count <= 0:
            return
        if not self.native:
            self._repair_cpu((torch.arange(count) + int(start)) % self.num_leaves)
            return
        L = lib()
        if not getattr(L, "_repair_range_ready", False):
            L.pda_segtree_repair_range.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.POINTER(ctypes.c_int),
                                                   ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
            L.pda_segtree_repair_range.restype = None
            L._repair_range_ready = True
        offs = (ctypes.c_int * 8)(*([int(o) for o in self.offsets] + [0] * (8 - self.depth)))
        stop = self.depth - 1 if self.lazy_top else self.depth
        L.pda_segtree_repair_range(_p(self.sum), _p(self.min), offs, stop, int(start) % self.num_leaves, count,
                                   self.num_leaves, _s(torch.cuda.current_stream(self.device).cuda_stream))

    @property
    def lazy_top(self):
        """True when the sampling kernel re-derives the top level from the level below it (native trees of four or
        more levels, > 262 k leaves): a priority update then only has to repair levels 1 .. depth - 2
        (repair(stop_level=depth - 1)); the stored top level is refreshed by the next sample and total() / min_value()
        read the level below.  (Smaller trees are completed inside the learner's optimizer launch -- csrc/optim.h
        AdamExtras::rep_top_n -- or by repair_range, and sampled from their stored top level.)"""
        return self.native and self.depth >= 4

    def _summary_level(self):
        lvl = self.depth - 2 if self.lazy_top else self.depth - 1
        return self.offsets[lvl], (self.sizes[lvl] if lvl < self.depth - 1 else RADIX)

    def total(self):
        off, n = self._summary_level()
        return self.sum[off:off + n].sum()

    def min_value(self):
        off, n = self._summary_level()
        return self.min[off:off + n].min()

    def get(self, leaf_inds):
        return self.sum[leaf_inds.long()]

    def _ntop(self):
        return int(self.sizes[-1]) if self.lazy_top else 0

    def sample_rng(self, B, beta, seed, counter, out_idx, out_weight):
        """sample() with the B stratum uniforms drawn inside the kernel from (seed, *counter) -- the counter hash of
        pda_uniform_f32, so the draw equals sample(pda_uniform_f32(...)) -- native only, one launch."""
        assert self.native
        L = lib()
        stream = torch.cuda.current_stream(self.device).cuda_stream
        offs = (ctypes.c_int * 8)(*([int(o) for o in self.offsets] + [0] * (8 - self.depth)))
        L.pda_segtree_sample_rng.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.POINTER(ctypes.c_int), ctypes.c_int,
                                             ctypes.c_int, ctypes.c_int, ctypes.c_float, ctypes.c_void_p,
                                             ctypes.c_void_p, ctypes.c_ulonglong, ctypes.c_void_p, ctypes.c_int,
                                             ctypes.c_void_p]
        L.pda_segtree_sample_rng.restype = None
        L.pda_segtree_sample_rng(_p(self.sum), _p(self.min), offs, self.depth, self.num_leaves, int(B),
                                 ctypes.c_float(beta), _p(out_idx), _p(out_weight), int(seed), _p(counter),
                                 self._ntop(), _s(stream))
        return out_idx, out_weight

    def sample(self, uniforms, beta, out_idx=None, out_weight=None):
        """Stratified draw: sample i targets mass (i + uniforms[i]) / B * total.
        Returns (idx int32 (B,), importance weight fp32 (B,) = (p_min/p_i)^beta)."""
        B = int(uniforms.numel())
        if out_idx is None:
            out_idx = torch.empty(B, dtype=torch.int32, device=self.device)
        if out_weight is None:
            out_weight = torch.empty(B, dtype=torch.float32, device=self.device)
        if self.native:
            L = lib()
            stream = torch.cuda.current_stream(self.device).cuda_stream
            offs = (ctypes.c_int * 8)(*([int(o) for o in self.offsets] + [0] * (8 - self.depth)))
            L.pda_segtree_sample(_p(self.sum), _p(self.min), offs, self.depth, self.num_leaves,
                                 _p(uniforms), B, ctypes.c_float(beta), _p(out_idx), _p(out_weight), self._ntop(),
                                 _s(stream))
            return out_idx, out_weight
        total = self.total().double()
        v = ((torch.arange(B, dtype=torch.float64) + uniforms.double().cpu()) / B * total).float()
        node = torch.zeros(B, dtype=torch.long)
        for lvl in range(self.depth - 1, -1, -1):
            base = self.offsets[lvl] + node * RADIX
            kids = self.sum[base[:, None] + torch.arange(RADIX)[None, :]]
            csum = torch.cumsum(kids, 1)
            hit = (csum >= v[:, None]) & (kids > 0)
            any_hit = hit.any(1)
            first = torch.where(any_hit, hit.float().argmax(1),
                                (kids > 0).float().cumsum(1).argmax(1))     # last positive child
            excl = torch.gather(csum - kids, 1, first[:, None])[:, 0]
            v = v - excl
            node = node * RADIX + first
        idx = node.clamp_(max=self.num_leaves - 1)
        p = self.sum[idx]
        w = (self.min_value() / p.clamp_min(1e-30)) ** beta
        out_idx.copy_(idx.to(torch.int32))
        out_weight.copy_(w)
        return out_idx, out_weight


def _p(t):
    return ctypes.c_void_p(t.data_ptr())


def _s(stream):
    return ctypes.c_void_p(stream)
