"""Nature-CNN DQN on hand-written gfx950 kernels: learner step and batched actor
inference (kernels K1-K4, K10, K12-K15 of SURVEY.md section 2.3).

The reference learner step (core/single_processes/dqn_learner.py:50-101) is
~40 stock PyTorch launches behind a lock-held CPU gather, a uint8->fp32 cast on
the host and a 28.9 MB unpinned H2D copy.  Here one step is a fixed sequence of
~20 HIP kernels on data that never leaves HBM, captured once into a hipGraph:

  sample      uniform counter-hash draw, or radix-64 sum-tree descent (PER)
  forward     conv1 reads uint8 replay frames *through the index vector* (no
              gather copy, /255 folded into the epilogue); online(s0), target(sN)
              and -- for double DQN -- online(sN) run as problem groups of the
              SAME launches; activations bf16 NHWC; fc1 split-K into fp32
  head        fc2 + n-step TD target + (importance-weighted) MSE + dL/dq + fc2
              dgrad + relu mask + PER leaf priorities, one wave64 per sample
  backward    MFMA wgrad/dgrad GEMMs; split-K partial tiles go to scratch slabs
              with plain coalesced stores (fp32 atomics run memory-side on
              MI355X: 134 us per conv wgrad in the first profile) and ONE reduce
              launch sums them into the flat gradient buffer
  (DP)        RCCL all-reduce of the flat gradient: the fc tail (95 % of the
              bytes) is reduced on a side stream while the conv backward runs
  update      one fused clip+Adam launch that also re-emits the bf16 shadow
              weights; fused hard/soft target update; sum/min-tree repair

fp32 master weights live in the model's own flat buffer, so state_dict(),
checkpoints and weight publishing see ordinary PyTorch parameters.
"""
import ctypes
import os

import numpy as np
import torch

from . import lib, nn
from .knobs import resolve as resolve_knobs
from .optim import AdamExtras, FlatAdam, SegmentTable, polyak_

from ._dqn_base import BF16, HID_SUPPORTED, _ActArgs, _HeadArgs, _L, _Range, _p, _stream, _vp      # noqa: F401
from .cnn_layers import (CnnGeometry, CnnWeights, _Acts, _PlaneView, _ksplit, cnn_forward,   # noqa: F401
                         fc1_slabs)


class DqnCnnEngine(object):
    """One learner's device state and step program."""

    def __init__(self, model, target_model, batch_size, lr=1e-4, weight_decay=0.0, clip=np.inf,
                 double=False, target_update=250, per=None, seed=0, tail_offsets=(0, 1, 2, 3), precision="bf16",
                 knobs=None):
        """knobs: ops.knobs.Knobs (split-K targets etc.; None = resolved from defaults / environment here).
        precision: "bf16" = bf16 MFMA operands (per-image fused conv kernels); "fp32" = every product a
        split-bf16 x3 product with fp32 accumulation (csrc/gemm_x3.hip), the reference learner's precision."""
        self.geom = g = CnnGeometry(model.input_dims, model.output_dims, model.critic[1][0].out_features)
        if precision not in ("bf16", "fp32"):
            raise ValueError("precision must be 'bf16' or 'fp32', not %r" % (precision,))
        self.precision = precision
        self.x3 = x3 = precision == "fp32"
        self.knobs = knobs if knobs is not None else resolve_knobs()
        self.B = int(batch_size)
        self.norm_val = float(model.norm_val)
        self.double = bool(double)
        self.target_update = target_update
        self.per = per                      # None or dict(eps, alpha, beta)
        self.online = CnnWeights(model, g, with_grad=True, split=x3)
        self.target = CnnWeights(target_model, g, with_grad=False, split=x3)
        self.device = dev = self.online.flat.device
        self.grads = model.flat_grads
        self.grads.zero_()
        self.gw, self.gb = self.online.grad_views(self.grads)
        self.adam = FlatAdam(self.online.flat, self.grads, self.online.table, lr=lr, weight_decay=weight_decay,
                             clip=clip, shadow=self.online.shadow, shadow_plane=self.online.shadow_plane)
        self.groups = 3 if self.double else 2
        self.acts = _Acts(g, self.B, self.groups, dev, backward=True, split=x3)
        self.acts.fc1_blocks = self.knobs.fc1_blocks
        # fc1 forward split-K as slabs summed by the head kernel: makes two runs of a step program bit-identical
        # (the last source of run-to-run differences was the arrival order of fc1's fp32 atomics)
        fc1_slabs(self.acts, g, self.B, self.groups, dev)
        self.deterministic = True
        # conv2 + conv3 weight gradients as two problem groups of ONE launch
        self.pair_wgrad = bool(self.acts.stack_dgrad) or x3
        # large batches: conv2 + conv3 weight gradients per image with register accumulation (csrc/nature_wgrad_stack.hip)
        self.stack_wgrad = bool(self.acts.stack_dgrad) and self.B >= self.knobs.wgrad_stack_min and not x3
        # fc1's data gradient in the launch of the two fc weight gradients (small batches: all three under-fill the GPU)
        self.fc_duo = self.B <= 512
        # conv1's weight gradient as a tail of the dgrad stack launch (da1 still in LDS)
        self.fuse_w1 = bool(self.acts.stack_dgrad) and nn.conv1_specialised(g.H, g.W, g.C, 8, 8, 4, 32) and not x3
        # the same for the fp32-precision dgrad stack (two workgroups = two slab slices per image: small batches)
        self.x3_fuse_w1 = x3 and nn.x3_cnn_dgrad_stack_supported(g.H, g.W, g.C)
        self._plan_wgrads()
        self._fc_pair = None
        self.idx = torch.zeros(self.B, dtype=torch.int32, device=dev)
        self.isw = torch.ones(self.B, dtype=torch.float32, device=dev)
        self.uniforms = torch.zeros(self.B, dtype=torch.float32, device=dev)
        self.td = torch.zeros(self.B, dtype=torch.float32, device=dev)
        self.q = torch.zeros(self.B, g.A, dtype=torch.float32, device=dev)
        self.loss_acc = torch.zeros(self.B, dtype=torch.float32, device=dev)     # per-sample sums (no atomics)
        self.loss_steps = 0
        self.size_dev = torch.zeros(4, dtype=torch.int32, device=dev)    # {rows, first row, ring length, -}: csrc/common.h window_pick
        self.counter = torch.zeros(1, dtype=torch.int64, device=dev)
        self.seed = int(seed) & 0xFFFFFFFFFFFF
        self.tail_offsets = tuple(int(o) for o in tail_offsets)
        self.replay = None
        self.step_count = 0
        # the tail bucket [fc1.weight .. end] is ready right after the two fc wgrads
        self.tail_start = self.online.off["critic.1.0"][0]
        # (Single rank: every kernel of the step runs on ONE stream.  Forking the weight gradients -- or the fc tail's
        # optimizer update -- onto a side stream as a branch of the step's hipGraph was measured slower three times
        # (4828 vs 5240, 9.1k vs 10.6k, 11.1k vs 13.5k updates/s): a cross-stream edge inside a graph costs more
        # than the overlapped work returns at these kernel sizes.  Those paths are gone.)
        self._target_copy(1.0)      # hard copy to start

    def _target_copy(self, tau):
        t = self.target
        polyak_(t.flat, self.online.flat, float(tau), t.table, t.shadow, t.shadow_plane)

    def _plan_wgrads(self):
        """Split-K plan of the three conv weight gradients: slab sizes + the one reduce launch."""
        g, B, dev = self.geom, self.B, self.device
        plans = []      # (layer, O, K, red)
        # fp32 precision: the reduction carries the operand copies (gemm.h SrcSplit): x2 against uint8 frames, else x3
        mult = (1, 1, 1)        # reductions are in logical units for the split-operand kernels too (gemm_body_x3)
        for li, (O, K, red) in enumerate([(32, g.K1, mult[0] * B * g.H1 * g.W1), (64, g.K2, mult[1] * B * g.H2 * g.W2),
                                          (64, g.K3, mult[2] * B * g.P3)]):
            tiles = ((O + 63) // 64 if O > 32 else 1) * ((K + 63) // 64)
            # (fp32 precision: the reductions are 2-3 x longer; more, shorter slices measured 3.51k -> 3.77k updates/s)
            wg_blocks = self.knobs.wgrad_blocks * (self.knobs.x3_wgrad_mult if self.x3 else 1)
            pair_blocks = self.knobs.pair_blocks * (self.knobs.x3_pair_mult if self.x3 else 1)
            ks = _ksplit(red, tiles, wg_blocks, 128)
            if li > 0 and self.stack_wgrad:
                ks, ns_override = 0, nn.cnn_wgrad_stack_slices(B)      # per-image kernel: one slice per workgroup
            elif li > 0 and self.pair_wgrad:
                # conv2 + conv3 go out as two problem groups of one launch with a common slice length, sized so
                # that the launch as a whole is ~knobs.pair_blocks workgroups
                work = mult[1] * (8 * (B * g.H2 * g.W2) + 9 * (B * g.P3))
                ks = nn.round_ksplit(max(128, (work + pair_blocks - 1) // pair_blocks))
            ns = nn.num_slices(red, ks) if ks > 0 else ns_override
            if li == 0 and nn.conv1_specialised(g.H, g.W, g.C, 8, 8, 4, 32):
                ns = nn.conv1_wgrad_slices(B)       # whole-image kernel: one slice per workgroup
                if self.fuse_w1 and not self.x3:
                    ns = nn.cnn_dgrad_stack_slices(B)   # computed by the dgrad stack launch itself
                if self.x3_fuse_w1:
                    # fp32-precision dgrad stack: one slice per workgroup (two per image up to 256 images; above, 256
                    # workgroups walk the batch and keep one accumulator each)
                    ns = nn.x3_cnn_dgrad_stack_slices(B)
            plans.append((li, O, K, ks, ns))
        self.wg_plan = plans
        total = sum(ns * (O * K + O) for _, O, K, _, ns in plans)
        self.slab = torch.zeros(total, dtype=torch.float32, device=dev)
        jobs, off = [], 0
        self.wg_slabs = []
        for li, O, K, ks, ns in plans:
            w = self.slab[off:off + ns * O * K]
            off += ns * O * K
            b = self.slab[off:off + ns * O]
            off += ns * O
            self.wg_slabs.append((w, b, ks, O * K, O))
            jobs.append((w, self.gw[li], O * K, ns, O * K))
            jobs.append((b, self.gb[li], O, ns, O))
        self.reduce = nn.ReducePlan(jobs)
        # the same sums, folded into the optimizer launch (single-rank steps): per parameter segment
        # (weight li -> segment 2*li, bias -> 2*li + 1) where its split-K slab starts and how many slices
        info = torch.zeros(2 * len(self.online.ORDER), 4, dtype=torch.int32)
        base = self.slab.data_ptr()
        for li, O, K, ks, ns in plans:
            w, b, _, wstride, bstride = self.wg_slabs[li]
            info[2 * li] = torch.tensor([(w.data_ptr() - base) // 4, ns, wstride, 0], dtype=torch.int32)
            info[2 * li + 1] = torch.tensor([(b.data_ptr() - base) // 4, ns, bstride, 0], dtype=torch.int32)
        self.slabinfo = info.to(dev)

    # ---- data sources --------------------------------------------------------------------------
    def bind_replay(self, state0s, state1s, tails, tree=None, window=None):
        """HBM replay: uint8 [M,C,H,W] x2 and fp32 tails [M][tail_ld] on this device -- or, frame-deduplicated
        (memories/dedup_memory.py), two nn.FrameWindows over one plane store plus the memory's sampling window
        (int32[4] on the device, kept by its unpack kernel), which then replaces this engine's own size word."""
        if isinstance(state0s, nn.FrameWindows):
            g = self.geom
            if not (nn.cnn_fwd_stack_supported(g.H, g.W, g.C) or nn.conv1_specialised(g.H, g.W, g.C, 8, 8, 4, 32)):
                raise ValueError("frame-deduplicated replay needs the Nature-CNN kernels (4 x 84 x 84 frame stacks)")
            assert state0s.store.is_cuda and state0s.window == g.C and window is not None and window.numel() >= 3
            assert state0s.stride == g.H * g.W, "a plane of the store is one H x W frame"
            self.size_dev = window
        else:
            assert state0s.dtype == torch.uint8 and state0s.is_cuda
        assert tails.dtype == torch.float32
        self.replay = (state0s, state1s, tails, tree)
        return self

    def make_staging(self, tail_floats):
        """Staging buffers for replay kept on the host (memory_type "shared")."""
        g, dev = self.geom, self.device
        self.stage_s0 = torch.zeros(self.B, g.C, g.H, g.W, dtype=torch.uint8, device=dev)
        self.stage_s1 = torch.zeros_like(self.stage_s0)
        self.stage_tails = torch.zeros(self.B, tail_floats, dtype=torch.float32, device=dev)
        return self

    # ---- step pieces (each only enqueues kernels on the current stream) ------------------------------
    def sample(self):
        s0, s1, tails, tree = self.replay
        L = _L()
        if tree is not None and self.per is not None:
            if getattr(tree, "native", False):      # stratum uniforms drawn inside the descent kernel: one launch
                tree.sample_rng(self.B, float(self.per["beta"]), self.seed, self.counter, self.idx, self.isw)
            else:
                L.pda_uniform_f32(_p(self.uniforms), self.B, self.seed, _p(self.counter), _stream())
                tree.sample(self.uniforms, float(self.per["beta"]), self.idx, self.isw)
        else:
            L.pda_uniform_idx(_p(self.idx), _p(self.isw), self.B, _p(self.size_dev), self.seed, _p(self.counter),
                              _stream())

    def _sources(self, staged):
        if staged:
            return self.stage_s0, self.stage_s1, self.stage_tails, None, None
        s0, s1, tails, tree = self.replay
        return s0, s1, tails, self.idx, (tree if self.per is not None else None)

    def forward_head(self, staged=False):
        g, B, a = self.geom, self.B, self.acts
        s0, s1, tails, idx, tree = self._sources(staged)
        nets = [self.online, self.target] + ([self.online] if self.double else [])
        frames = [s0, s1] + ([s1] if self.double else [])
        cnn_forward(g, nets, frames, [idx] * len(nets), a, B, self.norm_val)
        on, tg = self.online, self.target
        oa, orw, og, ot = self.tail_offsets
        args = _HeadArgs(
            h1pre_on0=_p(a.h1pre[0]), h1pre_tg1=_p(a.h1pre[1]), h1pre_on1=_p(a.h1pre[2]) if self.double else 0,
            b4_on=_p(on.b[3]), b4_tg=_p(tg.b[3]),
            w5_on=_p(on.w[4]), b5_on=_p(on.b[4]), w5_tg=_p(tg.w[4]), b5_tg=_p(tg.b[4]),
            tails=_p(tails), tail_ld=int(tails.stride(0)), off_action=oa, off_reward=orw, off_gamma=og, off_term=ot,
            idx=_p(idx), isw=_p(self.isw) if (self.per is not None) else 0,
            h1_on0=_p(a.h1), dh1=_p(a.dh1), dq=_p(a.dq), dq_ld=a.dq_ld,
            q_out=_p(self.q), td_out=_p(self.td), loss_sum=_p(self.loss_acc),
            leaf_sum=_p(tree.sum) if tree is not None else 0, leaf_min=_p(tree.min) if tree is not None else 0,
            prio_eps=float(self.per["eps"]) if self.per else 0.0,
            prio_alpha=float(self.per["alpha"]) if self.per else 1.0,
            B=B, HID=g.HID, A=g.A, rng_counter=_p(self.counter), opt_step=_p(self.adam.step_dev),
            h1_plane=getattr(a.h1, "plane", 0), dh1_plane=getattr(a.dh1, "plane", 0),
            dq_plane=getattr(a.dq, "plane", 0),
            h1_slices=a.h1_slices, h1_slice_stride=B * g.HID)
        _L().pda_dqn_head(ctypes.byref(args), _stream())

    def backward_fc(self):
        g, B, a = self.geom, self.B, self.acts
        # one slice each, written straight into the gradient buffer (kernel layout); both in one launch
        if self._fc_pair is None:
            self._fc_pair = (nn.X3LinearWgradPair if self.x3 else nn.LinearWgradPair)(
                [(a.dh1, a.a3[0], self.gw[3], self.gb[3], g.HID, g.F, g.HID, g.F, g.F),
                 (a.dq, a.h1, self.gw[4], self.gb[4], g.A, g.HID, a.dq_ld, g.HID, g.HID)], B)
        if self.fc_duo:
            # ... and fc1's data gradient rides in the same launch (backward_conv then starts at the conv stack)
            on = self.online
            self._fc_pair.run_with_dgrad(a.dh1, on.w[3], a.a3[0].hi if self.x3 else a.a3[0], a.da3, B, g.HID, g.F,
                                         g.HID, g.F, g.F, g.F, act=nn.ACT_RELU)
        else:
            self._fc_pair.run()

    def fused_extras(self, staged, slabs=True):
        """AdamExtras: (single-rank steps) slab sums in the optimizer launch, plus (device replay,
        uniform sampling) the draw of the next step's replay indices."""
        x = AdamExtras()
        if slabs:
            x.slab, x.slabinfo = self.slab.data_ptr(), self.slabinfo.data_ptr()
        if self.presamples(staged) and not self._per_bound():
            x.next_idx, x.next_w, x.next_B = self.idx.data_ptr(), self.isw.data_ptr(), self.B
            x.size, x.seed, x.counter = self.size_dev.data_ptr(), self.seed, self.counter.data_ptr()
        tree = self.replay[3] if (self.replay is not None and not staged and self.per is not None) else None
        if tree is not None and getattr(tree, "native", False) and tree.depth >= 2:
            # prioritized replay: the first repair level of the sum / min tree rides in the optimizer launch (the
            # leaves were written by the head kernel).  Trees of up to three levels (<= 262 k leaves) are complete when
            # that launch ends (the last repair block recomputes the top level); larger ones get the levels in between
            # from post_update() and their top level re-derived by the next step's sampling kernel (tree.lazy_top)
            x.rep_sum, x.rep_min, x.rep_idx = tree.sum.data_ptr(), tree.min.data_ptr(), self.idx.data_ptr()
            x.rep_n, x.rep_parent_off = self.B, int(tree.offsets[1])
            if self._tree_in_optimizer_launch(tree):
                if getattr(self, "_rep_sync", None) is None:
                    # arrivals | sampler release flag | sampler arrivals | barrier of the repair blocks
                    self._rep_sync = torch.zeros(4, dtype=torch.int32, device=self.device)
                x.rep_count, x.rep_bar = self._rep_sync.data_ptr(), self._rep_sync.data_ptr() + 12
                x.ps_depth, x.ps_leaves = tree.depth, tree.num_leaves
                for i, o in enumerate(tree.offsets):
                    x.ps_offs[i] = int(o)
                if tree.depth >= 3:
                    x.rep_top_off, x.rep_top_n = int(tree.offsets[tree.depth - 1]), int(tree.sizes[tree.depth - 1])
                if self.presamples(staged):
                    # ... and the next step's prioritized draw behind the repair, in the same launch
                    x.ps_idx, x.ps_w, x.ps_B = self.idx.data_ptr(), self.isw.data_ptr(), self.B
                    x.ps_beta = float(self.per["beta"])
                    x.ps_done, x.ps_count = self._rep_sync.data_ptr() + 4, self._rep_sync.data_ptr() + 8
                    x.seed, x.counter = self.seed, self.counter.data_ptr()
        return x

    def _per_bound(self):
        return self.replay is not None and self.replay[3] is not None and self.per is not None

    def _tree_in_optimizer_launch(self, tree):
        """Whole-tree repair + the next step's draw as blocks of the optimizer launch: trees of up to four levels and
        batches whose repair and sampler blocks (B / 4 each) are resident together with the optimizer's own -- at batch
        4096 they are not, and the blocks that wait for each other serialise the launch (1.61 -> 1.67 ms per step)."""
        return bool(getattr(tree, "native", False)) and 2 <= tree.depth <= 4 and self.B <= 512

    def presamples(self, staged):
        """True when the optimizer launch of a step also draws the next step's replay rows: uniform sampling from
        device replay, or prioritized sampling from a native tree of up to four levels (<= 16.7 M leaves), which that
        launch also repairs completely (fused_extras)."""
        if staged or self.replay is None:
            return False
        if not self._per_bound():
            return True
        return self._tree_in_optimizer_launch(self.replay[3])

    def backward_conv(self, staged=False, reduce=True):
        """reduce: sum the split-K slabs with a launch of their own (data parallel: the all-reduce needs the sums);
        False when the optimizer launch sums them (fused_extras)."""
        g, B, a = self.geom, self.B, self.acts
        s0, _, _, idx, _ = self._sources(staged)
        on = self.online
        M3, M2 = B * g.P3, B * g.H2 * g.W2
        (w1s, b1s, ks1, st1, bst1), (w2s, b2s, ks2, st2, bst2), (w3s, b3s, ks3, st3, bst3) = self.wg_slabs

        if self.x3:
            # fp32 precision: gather-form data gradients and split-K weight gradients on split-bf16 planes
            if not self.fc_duo:
                nn.x3_linear_dgrad(a.dh1, on.w[3], a.a3[0].hi, a.da3, B, g.HID, g.F, g.HID, g.F, g.F, g.F,
                                   act=nn.ACT_RELU)
            if nn.x3_cnn_dgrad_stack_supported(g.H, g.W, g.C):
                # both conv data gradients in one launch (per image, every plane LDS-resident)
                nn.x3_cnn_dgrad_stack(a.da3, a.a2[0].hi, a.a1[0].hi, on.wT[2], on.wT[1], a.da2, a.da1, B,
                                      w1=(s0, idx, w1s, b1s, 1.0 / self.norm_val) if self.x3_fuse_w1 else None)
            else:
                nn.x3_conv_dgrad(a.da3, on.w[2], a.a2[0].hi, a.da2, B, g.H2, g.W2, 64, 3, 3, 1, 64, act=nn.ACT_RELU)
                nn.x3_conv_dgrad(a.da2, on.w[1], a.a1[0].hi, a.da1, B, g.H1, g.W1, 32, 4, 4, 2, 64, act=nn.ACT_RELU)
            geoms = [(g.H1, g.W1, 32, 4, 4, 2), (g.H2, g.W2, 64, 3, 3, 1)]

            if self.x3_fuse_w1:
                pass                # conv1's weight gradient was computed by the dgrad stack launch
            elif nn.conv1_specialised(g.H, g.W, g.C, 8, 8, 4, 32):
                nn.x3_conv1_wgrad_u8_whole(a.da1, s0, idx, w1s, b1s, B, g.H, g.W, g.C, 1.0 / self.norm_val, st1, bst1)
            else:
                nn.x3_conv_wgrad_u8(a.da1, s0, idx, w1s, b1s, B, g.H, g.W, g.C, 8, 8, 4, 32, 1.0 / self.norm_val,
                                    ksplit=ks1, slice_stride=st1, bslice_stride=bst1)
            if self.pair_wgrad:
                assert ks2 == ks3
                nn.x3_conv_wgrad([a.da2, a.da3], [a.a1[0], a.a2[0]], [w2s, w3s], [b2s, b3s], B, geoms, 64, ks2,
                                 [st2, st3], [bst2, bst3])
            else:
                nn.x3_conv_wgrad([a.da3], [a.a2[0]], [w3s], [b3s], B, geoms[1:], 64, ks3, [st3], [bst3])
                nn.x3_conv_wgrad([a.da2], [a.a1[0]], [w2s], [b2s], B, geoms[:1], 64, ks2, [st2], [bst2])
            if reduce:
                self.reduce.run()
            return
        if not self.fc_duo:
            nn.linear_dgrad(a.dh1, on.w[3], a.a3[0], a.da3, B, g.HID, g.F, g.HID, g.F, g.F, g.F, act=nn.ACT_RELU)
        if a.stack_dgrad:
            # both conv data gradients in one launch (per-image, LDS-resident), then the two weight gradients
            nn.cnn_dgrad_stack(a.da3, a.a2[0], a.a1[0], on.w[2], on.w[1], a.da2, a.da1, B,
                               w1=(s0, idx, w1s, b1s, 1.0 / self.norm_val) if self.fuse_w1 else None)
            if self.stack_wgrad:
                nn.cnn_wgrad_stack(a.a1[0], a.a2[0], a.da2, a.da3, w2s, b2s, w3s, b3s, B)
            elif self.pair_wgrad:
                assert ks2 == ks3
                pair = ([a.da2, a.da3], [a.a1[0], a.a2[0]], [w2s, w3s], [b2s, b3s], B,
                        [(g.H1, g.W1, 32, 4, 4, 2), (g.H2, g.W2, 64, 3, 3, 1)], 64, ks2, [st2, st3], [bst2, bst3])
                nn.conv_wgrad_pair(*pair)
            else:
                nn.conv_wgrad(a.da3, a.a2[0], w3s, b3s, B, g.H2, g.W2, 64, 3, 3, 1, 64, ksplit=ks3, slice_stride=st3,
                              bslice_stride=bst3)
                nn.conv_wgrad(a.da2, a.a1[0], w2s, b2s, B, g.H1, g.W1, 32, 4, 4, 2, 64, ksplit=ks2, slice_stride=st2,
                              bslice_stride=bst2)
        else:
            self._backward_conv_layerwise(a, on, B, M3, M2, (w3s, b3s, ks3, st3, bst3), (w2s, b2s, ks2, st2, bst2))
        if not (a.stack_dgrad and self.fuse_w1):
            nn.conv_wgrad_u8(a.da1, s0, idx, w1s, b1s, B, g.H, g.W, g.C, 8, 8, 4, 32, 1.0 / self.norm_val,
                             ksplit=ks1, slice_stride=st1, bslice_stride=bst1)
        if reduce:
            self.reduce.run()

    def _backward_conv_layerwise(self, a, on, B, M3, M2, s3, s2):
        """bf16 engine, generic geometries: conv3 / conv2 weight and data gradients as separate launches (data
        gradient = dcol GEMM + col2im)."""
        g = self.geom
        (w3s, b3s, ks3, st3, bst3), (w2s, b2s, ks2, st2, bst2) = s3, s2
        nn.conv_wgrad(a.da3, a.a2[0], w3s, b3s, B, g.H2, g.W2, 64, 3, 3, 1, 64, ksplit=ks3, slice_stride=st3,
                      bslice_stride=bst3)
        nn.linear_dgrad(a.da3, on.w[2], None, a.dcol3, M3, 64, g.K3, 64, g.K3, 0, g.K3)
        nn.col2im(a.dcol3, a.a2[0], a.da2, B, g.H2, g.W2, 64, 3, 3, 1, act=nn.ACT_RELU)
        nn.conv_wgrad(a.da2, a.a1[0], w2s, b2s, B, g.H1, g.W1, 32, 4, 4, 2, 64, ksplit=ks2, slice_stride=st2,
                      bslice_stride=bst2)
        nn.linear_dgrad(a.da2, on.w[1], None, a.dcol2, M2, 64, g.K2, 64, g.K2, 0, g.K2)
        nn.col2im(a.dcol2, a.a1[0], a.da1, B, g.H1, g.W1, 32, 4, 4, 2, act=nn.ACT_RELU)

    def grads_torch_layout(self):
        """The flat gradient re-ordered to PyTorch's parameter layout (tests, debugging)."""
        return self.online.to_torch_layout(self.grads)

    def update(self, staged=False, tick=False, extras=None):
        """tick: bump Adam's step counter here (the head kernel does it in a full step)."""
        self.adam.step(tick=tick, extras=extras)
        fused = extras is not None and bool(extras.rep_sum)
        self.post_update(staged, repair_from=(0 if extras.rep_count else 2) if fused else 1)

    def post_update(self, staged=False, repair_from=1):
        """What follows the optimizer: soft target update, sum/min-tree repair of the sampled leaves (repair_from = 0:
        the optimizer launch repaired the whole tree; 2: only level 1, and the sampling kernel re-derives the top)."""
        if self.target_update < 1.0:
            self._target_copy(self.target_update)
        if not staged and self.per is not None and self.replay[3] is not None:
            tree = self.replay[3]
            if repair_from == 0:
                pass                # the optimizer launch repaired the whole tree (csrc/optim.h)
            elif repair_from == 2:
                tree.repair(self.idx, self.B, first_level=2, stop_level=tree.depth - 1)
            else:
                tree.repair(self.idx, self.B)

    def hard_target_update(self):
        self._target_copy(1.0)

    def set_size(self, n):
        self.size_dev[:1].fill_(int(n))

    def mean_loss(self, lagged=False):
        """Mean loss since the last call (one host sync).  lagged: no synchronisation -- the value returned is the
        mean over the period BEFORE the last call (ops/optim.py LaggedReadout; what the learner loop logs)."""
        if lagged and self.loss_acc.is_cuda:
            if getattr(self, "_loss_readout", None) is None:
                from .optim import LaggedReadout
                self._loss_readout = LaggedReadout(self.loss_acc)
            vals, n = self._loss_readout.read(self.loss_steps)
            self.loss_steps = 0
            return float(vals.sum()) / n
        n = max(self.loss_steps, 1)
        val = float(self.loss_acc.sum().item()) / n
        self.loss_acc.zero_()
        self.loss_steps = 0
        return val


class DqnNativeCore(object):
    """Learner-loop adapter (agents/learners.py): same surface as DqnTorchCore."""

    def __init__(self, args, model, target_model, memory, comm=None, use_graph=True, force_dp=False):
        ap, mp_ = args.agent_params, args.memory_params
        self.args, self.ap, self.memory, self.comm = args, ap, memory, comm
        self.model, self.target_model = model, target_model
        per = None
        if memory.enable_per:
            per = {"eps": memory.priority_eps, "alpha": memory.priority_exponent, "beta": memory.priority_weight}
        a, k = int(memory.action_shape), int(memory.reward_shape)
        rank = comm.rank if comm is not None else 0
        self.engine = DqnCnnEngine(model, target_model, ap.batch_size, lr=ap.lr, weight_decay=ap.weight_decay,
                                   clip=ap.clip_grad, double=ap.enable_double,
                                   target_update=ap.target_model_update, per=per,
                                   seed=args.seed * 7919 + 104729 * rank + 13, tail_offsets=(0, a, a + k, a + 2 * k),
                                   precision=getattr(args, "precision", "bf16"), knobs=resolve_knobs(args))
        self.on_device = hasattr(memory, "tails") and getattr(memory, "attached", False) and memory.tails.is_cuda
        self.dedup = bool(getattr(memory, "dedup", False))
        if self.on_device and self.dedup:
            s0, s_n = memory.frame_windows()
            self.engine.bind_replay(s0, s_n, memory.tails, memory.tree, window=memory.window_dev)
        elif self.on_device:
            self.engine.bind_replay(memory.state0s, memory.state1s, memory.tails, memory.tree)
        else:
            self.engine.make_staging(a + 3 * k + 1)
        self.program = DqnStepProgram(self.engine, comm, use_graph=use_graph, staged=not self.on_device,
                                      force_dp=force_dp or bool(getattr(args, "force_dp", False)),
                                      graph_steps=int(getattr(args, "graph_steps", 1)))
        self._last_size = -1
        self.step_count = 0

    def step(self):
        e = self.engine
        if self.on_device:
            n = self.memory.size
            if n != self._last_size and not self.dedup:     # (deduplicated replay publishes its own sampling window)
                e.set_size(n)
                self._last_size = n
            self.program.step()
        else:
            batch, inds, weights = self.memory.sample_with_priorities(self.ap.batch_size)
            s0, a, r, g, s1, t = batch
            e.stage_s0.copy_(s0.to(torch.uint8), non_blocking=True)
            e.stage_s1.copy_(s1.to(torch.uint8), non_blocking=True)
            tails = torch.cat([a.float().reshape(e.B, -1), r.float().reshape(e.B, -1), g.float().reshape(e.B, -1),
                               t.float().reshape(e.B, -1)], dim=1)
            e.stage_tails[:, :tails.shape[1]].copy_(tails, non_blocking=True)
            e.isw.copy_(weights.reshape(-1).float(), non_blocking=True)
            self.program.step()
            if self.memory.enable_per and inds is not None:
                self.memory.update_priorities(inds, e.td.abs().cpu())
        self.step_count += 1

    def flat_params(self):
        return self.engine.online.flat

    def loss_values(self, lagged=False):
        return {"critic_loss": self.engine.mean_loss(lagged=lagged)}


class DqnCnnInference(object):
    """Batched actor-side forward + epsilon-greedy epilogue (kernel K10)."""

    def __init__(self, model, max_batch, seed=0):
        self.geom = g = CnnGeometry(model.input_dims, model.output_dims, model.critic[1][0].out_features)
        self.model = model
        self.norm_val = float(model.norm_val)
        self.weights = CnnWeights(model, g, with_grad=False)
        self.device = dev = self.weights.flat.device
        self.max_batch = int(max_batch)
        self.acts = _Acts(g, self.max_batch, 1, dev, backward=False)
        self.out = torch.zeros(self.max_batch, 4, dtype=torch.float32, device=dev)
        self.q = torch.zeros(self.max_batch, g.A, dtype=torch.float32, device=dev)
        self.counter = torch.zeros(1, dtype=torch.int64, device=dev)
        self.seed = (int(seed) * 2654435761 + 97) & 0xFFFFFFFFFFFF

    def refresh_weights(self):
        if self.weights.flat.data_ptr() != self.model.flat_params.data_ptr():
            self.weights = CnnWeights(self.model, self.geom, with_grad=False)
        else:
            self.weights.refresh()

    @torch.no_grad()
    def act(self, obs, eps=None):
        """obs uint8 (n,C,H,W) on the device; eps fp32 (n,) or None (greedy).
        Returns device tensors (action int64 (n,), q[action] (n,), max q (n,))."""
        n = int(obs.shape[0])
        assert n <= self.max_batch and obs.dtype == torch.uint8 and obs.is_contiguous()
        g = self.geom
        cnn_forward(g, [self.weights], [obs], [None], self.acts, n, self.norm_val)
        w = self.weights
        args = _ActArgs(h1pre=_p(self.acts.h1pre[0]), b4=_p(w.b[3]), w5=_p(w.w[4]), b5=_p(w.b[4]),
                        eps=_p(eps) if eps is not None else 0, out=_p(self.out), out_ld=4, q_out=_p(self.q),
                        B=n, HID=g.HID, A=g.A, seed=self.seed, counter=_p(self.counter))
        _L().pda_act_head(ctypes.byref(args), _stream())
        self.counter += 1
        out = self.out[:n]
        return out[:, 0].long(), out[:, 1].clone(), out[:, 2].clone()

    def q_values(self, n):
        return self.q[:n]



from .dqn_program import DqnStepProgram      # noqa: E402,F401  (DqnNativeCore builds one; re-exported for callers)
