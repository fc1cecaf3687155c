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


def _ksplit(red, tiles, target_blocks=768, minimum=64):
    """Split-K slice length (multiple of 32) so that tiles * slices ~ target_blocks."""
    slices = max(1, target_blocks // max(tiles, 1))
    ks = (red + slices - 1) // slices
    return nn.round_ksplit(max(minimum, ks))


class CnnGeometry(object):
    def __init__(self, input_dims, num_actions, hidden=512):
        C, H, W = input_dims
        self.C, self.H, self.W = C, H, W
        self.H1, self.W1 = nn.conv_out(H, 8, 4), nn.conv_out(W, 8, 4)
        self.H2, self.W2 = nn.conv_out(self.H1, 4, 2), nn.conv_out(self.W1, 4, 2)
        self.H3, self.W3 = nn.conv_out(self.H2, 3, 1), nn.conv_out(self.W2, 3, 1)
        self.P3 = self.H3 * self.W3
        self.F = 64 * self.P3
        self.K1, self.K2, self.K3 = C * 64, 32 * 16, 64 * 9
        self.A, self.HID = int(num_actions), int(hidden)
        if W % 4 or (C * H * W) % 4:
            raise ValueError("uint8 conv1 loader needs 4-byte aligned rows (W %% 4 == 0)")
        if self.HID not in HID_SUPPORTED or self.A > 64:
            raise ValueError("unsupported head shape")


class _PlaneView(object):
    """hi-plane tensor view + element offset of the matching lo plane (what nn.x3_* take as an operand)."""

    def __init__(self, hi, lo):
        self.hi, self.lo = hi, lo
        self.plane = (lo.data_ptr() - hi.data_ptr()) // hi.element_size()

    def data_ptr(self):
        return self.hi.data_ptr()

    def value(self):
        return self.hi.float() + self.lo.float()


class CnnWeights(object):
    """Views of one network's flat fp32 parameters + its bf16 shadow weights in
    kernel layout (conv: [O][(ky,kx,c)], conv1: [O][(c,ky,kx)], fc1: [N][(p,c)])."""

    ORDER = ("critic.0.0", "critic.0.2", "critic.0.4", "critic.1.0", "critic.1.2")

    def __init__(self, model, geom, with_grad, split=False):
        """split: the shadow keeps a second plane with the bf16 residuals w - bf16(w) (fp32-precision engine)."""
        self.model, self.g = model, geom
        self.split = bool(split)
        self.flat = model.flatten_(with_grad=with_grad)
        dev = self.flat.device
        segs = {name: (off, shape) for name, off, shape in model.param_segments()}
        g = geom
        shadow_shapes = [(32, g.K1, 1, g.K1), (64, g.K2, 32, 16), (64, g.K3, 64, 9), (g.HID, g.F, 64, g.P3)]
        main = sum(r * c for r, c, _, _ in shadow_shapes)
        # conv2 / conv3 also keep a transposed copy [(tap, c)][o] behind the main shadows (csrc/optim.h SegEntry): the
        # data-gradient kernels read their weight fragments from it with plain 16-byte loads
        # (fp32-precision engines only: the bf16 data-gradient kernel stages its weights through LDS, and the extra
        # scattered shadow writes cost its optimizer launch 1.4 us -- 13.45k vs 13.2k updates/s)
        tr_offs, total = {}, main
        if self.split:
            tr_offs = {1: main, 2: main + shadow_shapes[1][0] * shadow_shapes[1][1]}
            total = tr_offs[2] + shadow_shapes[2][0] * shadow_shapes[2][1]
        self.shadow = torch.zeros(total * (2 if self.split else 1), dtype=BF16, device=dev)
        self.shadow_plane = total if self.split else 0
        entries, self.w, self.b, self.off, self.wT = [], [], [], {}, {}
        soff = 0
        for li, layer in enumerate(self.ORDER):
            woff, wshape = segs[layer + ".weight"]
            boff, bshape = segs[layer + ".bias"]
            wn, bn = int(np.prod(wshape)), int(np.prod(bshape))
            self.off[layer] = (woff, wn, boff, bn)
            # conv gradients arrive as split-K slabs that the optimizer may sum itself: one element per thread
            ent = {"start": woff, "count": wn, "cols": wn // wshape[0], "fine": li < 3}
            if li < 4:
                rows, cols, C, P = shadow_shapes[li]
                assert rows == wshape[0] and cols == wn // wshape[0], (layer, wshape, rows, cols)
                ent.update(C=C, P=P, shadow_off=soff, shadow_ld=cols, shadow=True)
                wv = self.shadow[soff:soff + rows * cols].view(rows, cols)
                if self.split:      # hi-plane view + the offset of its lo plane, as the x3 kernels take operands
                    wv = _PlaneView(wv, self.shadow[total + soff:total + soff + rows * cols].view(rows, cols))
                self.w.append(wv)
                if li in tr_offs:
                    ent.update(tr_off=tr_offs[li], tr_ld=rows)
                    tv = self.shadow[tr_offs[li]:tr_offs[li] + rows * cols].view(cols, rows)
                    if self.split:
                        tv = _PlaneView(tv, self.shadow[total + tr_offs[li]:total + tr_offs[li] + rows * cols].view(cols, rows))
                    self.wT[li] = tv
                soff += rows * cols
            else:
                self.w.append(self.flat[woff:woff + wn].view(wshape))       # fc2 stays fp32
            entries.append(ent)
            entries.append({"start": boff, "count": bn, "fine": li < 3})
            self.b.append(self.flat[boff:boff + bn])
        self.table = SegmentTable(entries, dev)
        self.refresh()

    def refresh(self):
        """Rebuild the bf16 shadows from the fp32 master weights (after a load)."""
        polyak_(self.flat, self.flat, 1.0, self.table, self.shadow, self.shadow_plane)

    def _convert(self, vec, to_kernel):
        out = vec.clone()
        shapes = [(32, self.g.K1, 1, self.g.K1), (64, self.g.K2, 32, 16), (64, self.g.K3, 64, 9),
                  (self.g.HID, self.g.F, 64, self.g.P3)]
        for layer, (rows, cols, C, P) in zip(self.ORDER[:4], shapes):
            woff, wn, _, _ = self.off[layer]
            blk = vec[woff:woff + wn]
            if to_kernel:       # [row][c*P+p] -> [row][p*C+c]
                out[woff:woff + wn] = blk.view(rows, C, P).permute(0, 2, 1).reshape(-1)
            else:
                out[woff:woff + wn] = blk.view(rows, P, C).permute(0, 2, 1).reshape(-1)
        return out

    def to_kernel_layout(self, vec):
        """Flat vector in PyTorch parameter layout -> the layout gradients/moments use here."""
        return self._convert(vec, True)

    def to_torch_layout(self, vec):
        return self._convert(vec, False)

    def grad_views(self, flat_grads):
        gw, gb = [], []
        for layer in self.ORDER:
            woff, wn, boff, bn = self.off[layer]
            gw.append(flat_grads[woff:woff + wn])
            gb.append(flat_grads[boff:boff + bn])
        return gw, gb


class _Acts(object):
    def __init__(self, geom, B, groups, device, backward, split=False):
        g = geom
        self.split = bool(split)

        def z(*s, dtype=BF16):
            if self.split and dtype == BF16:        # hi + lo planes of a logical fp32 tensor
                return nn.Planes(*s, device=device)
            return torch.zeros(*s, dtype=dtype, device=device)
        self.a1 = [z(B * g.H1 * g.W1, 32) for _ in range(groups)]
        self.a2 = [z(B * g.H2 * g.W2, 64) for _ in range(groups)]
        self.a3 = [z(B * g.P3, 64) for _ in range(groups)]
        self.h1pre = [z(B, g.HID, dtype=torch.float32) for _ in range(groups)]
        self.h1_slices, self.h1_ksplit = 0, 0       # > 0: fc1's split-K sums arrive as slabs (fc1_slabs below)
        self.fc1_blocks = 256                       # split-K target of fc1's forward (knobs.fc1_blocks)
        if backward:
            self.h1 = z(B, g.HID)
            self.dh1 = z(B, g.HID)
            self.dq_ld = max(16, (g.A + 7) // 8 * 8)
            self.dq = z(B, self.dq_ld)
            self.da3 = z(B * g.P3, 64)
            self.da2 = z(B * g.H2 * g.W2, 64)
            self.da1 = z(B * g.H1 * g.W1, 32)
            self.stack_dgrad = nn.cnn_dgrad_stack_supported(g.H, g.W, g.C) and not self.split
            # bf16 engine, geometries the per-image stack does not cover: dcol GEMM + col2im.  (The gather form
            # was ~4 % slower end to end at batch 128 and 4096 on the bf16 engine and is gone from it; the
            # fp32-class engine has only the gather form, which needs even conv1 output sides.)
            if self.split and (g.H1 % 2 or g.W1 % 2):
                raise ValueError("the fp32-precision engine needs even conv1 output sides (gather-form dgrad)")
            if not self.split and not self.stack_dgrad:
                self.dcol3 = z(B * g.P3, g.K3)
                self.dcol2 = z(B * g.H2 * g.W2, g.K2)


def cnn_forward(geom, nets, frames, idxs, acts, B, norm_val, keep_first_only=True):
    """Run len(nets) (<= 3) networks as problem groups of the same launches.
    Leaves fc1's pre-activation split-K sums in acts.h1pre[g] (which must be zero)."""
    g, n = geom, len(nets)
    if getattr(acts, "split", False):
        if nn.x3_cnn_fwd_stack_supported(g.H, g.W, g.C):
            # one launch, one workgroup per image and network, a1 / a2 (both planes) LDS-resident
            keep = [True] + [False] * (n - 1) if keep_first_only else [True] * n
            nn.x3_cnn_fwd_stack(frames, idxs, [w.w[0] for w in nets], [w.b[0] for w in nets],
                                [w.w[1] for w in nets], [w.b[1] for w in nets], [w.w[2] for w in nets],
                                [w.b[2] for w in nets], [acts.a1[i] if keep[i] else None for i in range(n)],
                                [acts.a2[i] if keep[i] else None for i in range(n)], acts.a3[:n], B, g.H, g.W, g.C,
                                1.0 / float(norm_val))
        else:
            # layer-wise implicit GEMMs on split-bf16 planes (csrc/gemm_x3.hip)
            nn.x3_conv_fwd_u8(frames, idxs, [w.w[0] for w in nets], [w.b[0] for w in nets], acts.a1[:n], B,
                              g.H, g.W, g.C, 8, 8, 4, 32, 1.0 / float(norm_val))
            nn.x3_conv_fwd(acts.a1[:n], [w.w[1] for w in nets], [w.b[1] for w in nets], acts.a2[:n], B, g.H1, g.W1,
                           32, 4, 4, 2, 64)
            nn.x3_conv_fwd(acts.a2[:n], [w.w[2] for w in nets], [w.b[2] for w in nets], acts.a3[:n], B, g.H2, g.W2,
                           64, 3, 3, 1, 64)
        if acts.h1_slices:
            nn.x3_linear_fwd(acts.a3[:n], [w.w[3] for w in nets], acts.h1pre[:n], B, g.HID, g.F, g.F, g.F, g.HID,
                             ksplit=acts.h1_ksplit, slice_stride=B * g.HID)
            return
        tiles = ((B + 63) // 64) * (g.HID // 64) * n
        nn.x3_linear_fwd(acts.a3[:n], [w.w[3] for w in nets], acts.h1pre[:n], B, g.HID, g.F, g.F, g.F, g.HID,
                         ksplit=_ksplit(g.F, tiles, acts.fc1_blocks, 128))
        return
    if nn.cnn_fwd_stack_supported(g.H, g.W, g.C):
        # one launch, one workgroup per image; only group 0 (the trained network on s0) keeps a1 / a2
        keep = [True] + [False] * (n - 1) if keep_first_only else [True] * n
        nn.cnn_fwd_stack(frames, idxs, [w.w[0] for w in nets], [w.b[0] for w in nets],
                         [w.w[1] for w in nets], [w.b[1] for w in nets], [w.w[2] for w in nets],
                         [w.b[2] for w in nets], [acts.a1[i] if keep[i] else None for i in range(n)],
                         [acts.a2[i] if keep[i] else None for i in range(n)], acts.a3[:n], B, g.H, g.W, g.C,
                         1.0 / float(norm_val))
    else:
        nn.conv_fwd_u8(frames, idxs, [w.w[0] for w in nets], [w.b[0] for w in nets], acts.a1[:n], B,
                       g.H, g.W, g.C, 8, 8, 4, 32, 1.0 / float(norm_val))
        nn.conv_fwd(acts.a1[:n], [w.w[1] for w in nets], [w.b[1] for w in nets], acts.a2[:n], B, g.H1, g.W1, 32, 4, 4, 2, 64)
        nn.conv_fwd(acts.a2[:n], [w.w[2] for w in nets], [w.b[2] for w in nets], acts.a3[:n], B, g.H2, g.W2, 64, 3, 3, 1, 64)
    if acts.h1_slices:
        nn.linear_fwd_slab(acts.a3[:n], [w.w[3] for w in nets], acts.h1pre[:n], B, g.HID, g.F, g.F, g.F, g.HID,
                           B * g.HID, acts.h1_ksplit)
        return
    tiles = ((B + 63) // 64) * (g.HID // 64) * n
    nn.linear_fwd(acts.a3[:n], [w.w[3] for w in nets], None, None, acts.h1pre[:n], B, g.HID, g.F, g.F, g.F,
                  0, g.HID, ksplit=_ksplit(g.F, tiles, acts.fc1_blocks, 128))


def fc1_slabs(acts, geom, B, groups, device):
    """Switch fc1's forward split-K from fp32 atomics into one buffer to slabs [slices][B][HID] per network that
    the head kernel adds in slice order: bit-reproducible steps, no memory-side atomics, nothing to zero."""
    g = geom
    mult = 1        # (split operands: the x3 GEMM body reduces over the logical K, csrc/gemm.h gemm_body_x3)
    tiles = ((B + 63) // 64) * (g.HID // 64) * groups
    ks = _ksplit(mult * g.F, tiles, acts.fc1_blocks, 128)
    acts.h1_ksplit = ks
    acts.h1_slices = nn.num_slices(mult * g.F, ks)
    acts.h1pre = [torch.zeros(acts.h1_slices, B, g.HID, dtype=torch.float32, device=device) for _ in range(groups)]


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
        self.x3_fuse_w1 = x3 and nn.x3_cnn_dgrad_stack_supported(g.H, g.W, g.C) and self.B <= 256
        self._plan_wgrads()
        self._fc_pair = None
        self.idx = torch.zeros(self.B, dtype=torch.int32, device=dev)
        self.isw = torch.ones(self.B, dtype=torch.float32, device=dev)
        self.uniforms = torch.zeros(self.B, dtype=torch.float32, device=dev)
        self.td = torch.zeros(self.B, dtype=torch.float32, device=dev)
        self.q = torch.zeros(self.B, g.A, dtype=torch.float32, device=dev)
        self.loss_acc = torch.zeros(self.B, dtype=torch.float32, device=dev)     # per-sample sums (no atomics)
        self.loss_steps = 0
        self.size_dev = torch.zeros(1, dtype=torch.int32, device=dev)
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
                    ns = 2 * B                          # fp32-precision dgrad stack: one slice per workgroup, two per image
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
    def bind_replay(self, state0s, state1s, tails, tree=None):
        """HBM replay: uint8 [M,C,H,W] x2 and fp32 tails [M][tail_ld] on this device."""
        assert state0s.dtype == torch.uint8 and state0s.is_cuda and tails.dtype == torch.float32
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
        if self.presamples(staged):
            x.next_idx, x.next_w, x.next_B = self.idx.data_ptr(), self.isw.data_ptr(), self.B
            x.size, x.seed, x.counter = self.size_dev.data_ptr(), self.seed, self.counter.data_ptr()
        return x

    def presamples(self, staged):
        return (not staged) and self.replay is not None and not (self.replay[3] is not None and self.per is not None)

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
        self.post_update(staged)

    def post_update(self, staged=False):
        """What follows the optimizer: soft target update, sum/min-tree repair of the sampled leaves."""
        if self.target_update < 1.0:
            self._target_copy(self.target_update)
        if not staged and self.per is not None and self.replay[3] is not None:
            self.replay[3].repair(self.idx, self.B)

    def hard_target_update(self):
        self._target_copy(1.0)

    def set_size(self, n):
        self.size_dev.fill_(int(n))

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


class DqnStepProgram(object):
    """Sequences an engine's step and captures it into hipGraphs.

    One rank: one graph per step.  Data parallel (one process per GPU, RCCL): the step is cut
    where the collectives go, and both all-reduces run on a side stream under other work:

      main   [A: sample, forward, head, fc wgrad] [B: conv backward ........] [Adam(fc tail)] [Adam(conv), target, tree]
      comm                                 all-reduce(fc tail, 95 % of bytes)   all-reduce(conv grads)

    The all-reduce is a SUM; the 1/world_size is Adam's grad_scale, so no scaling launch exists."""

    def __init__(self, engine, comm=None, use_graph=True, staged=False, force_dp=False, graph_steps=1):
        self.e, self.comm, self.staged = engine, comm, staged
        self.world = comm.world_size if comm is not None else 1
        self.dp = comm is not None and (self.world > 1 or force_dp)
        self.use_graph = use_graph
        self.graphs = None
        self.dp_graph_mode = None
        self.calls = 0
        # Updates per graph replay.  Between two replays the GPU idles ~8 us (graph completion + next launch: the
        # kernels inside a graph run back to back, tools/step_gaps.py), 9 % of a 0.09 ms step; a graph of K steps
        # pays it once per K.  The program then runs AHEAD of its caller: the first step() of a group performs K
        # updates, the next K - 1 calls return at once.  Only for the HBM-replay programs (single rank, or data
        # parallel with the collectives captured in the graph), and K is cut
        # down to a divisor of the hard target-update period so that those updates still fall between replays.
        self.knobs = knobs = engine.knobs
        _Range.enabled = bool(knobs.roctx)
        k = int(graph_steps or 1)
        if staged or not use_graph:
            k = 1
        tu = engine.target_update
        if k > 1 and tu >= 1.0:
            while k > 1 and int(tu) % k != 0:
                k -= 1
        self.graph_steps, self.ahead = max(1, k), 0
        # How far the host may run ahead of the GPU, in graph replays.  Unbounded, the launch queue fills up and --
        # with the inference loop's thread and the ingestion copies driving the same device -- every launch then
        # waits inside the runtime: the learner fell from 8.1 k to 4.9 k updates/s inside the job when its one
        # periodic host synchronisation (the loss read-out) was removed.  Three replays in flight keep the GPU fed
        # (a replay is ~0.5 ms of work, a launch ~20 us) and the queue short.
        self.max_inflight = max(1, int(knobs.max_inflight_replays))
        self._inflight = []
        # single rank: the split-K slab sums and the next step's index draw ride in the optimizer launch
        self.fused_tail = not self.dp
        self.presampled = False
        self.extras = engine.fused_extras(staged) if self.fused_tail else None
        self.dp_extras = None
        if self.dp and engine.presamples(staged):
            self.dp_extras = engine.fused_extras(staged, slabs=False)
        first_tail_seg = 2 * engine.online.ORDER.index("critic.1.0")     # (weight, bias) segment pairs per layer
        self.tail_table = engine.online.table.subset(lambda s: s >= first_tail_seg)
        self.head_table = engine.online.table.subset(lambda s: s < first_tail_seg)
        # (Measured and removed: the fc tail's optimizer update co-launched with the conv2/conv3 weight-gradient
        # GEMM -- 11.76k vs 12.51k updates/s at batch 128, the Adam workgroups' HBM streams slow the GEMM's operand
        # loads by more than the overlap returns -- and the same update forked onto a side stream under the conv
        # backward: 9.1k vs 10.6k fp32-class, 11.1k vs 13.5k bf16.)
        if self.dp:
            e = engine
            e.adam.set_grad_scale(1.0 / self.world)
            self.comm_stream = torch.cuda.Stream(device=e.device)
            self.fc_done, self.conv_done = torch.cuda.Event(), torch.cuda.Event()
            self.tail_done, self.head_done = torch.cuda.Event(), torch.cuda.Event()
            self.tail = e.grads[e.tail_start:]
            self.head = e.grads[:e.tail_start]
            # the fc tail is 95 % of the gradient bytes (6.4 MB fp32): optionally all-reduced as bf16 -- half the
            # bytes on the xGMI links, which at ~0.1 ms of compute per step is what bounds data-parallel scaling.
            # The all-reduce itself: the one-kernel xGMI peer all-reduce (parallel/peer_allreduce.py: every rank maps
            # every peer's hand-off region through hipIpc; two hops over all links, replayable inside the step graph)
            # when it passes its self-test on every rank, else the process group's (RCCL ring / gloo).
            # Knobs dp_allreduce = auto | peer | rccl, dp_grad_dtype = measure | fp32 | bf16 (ops/knobs.py): with
            # "measure" the candidates are timed here, on the real bucket, and the fastest wins.
            from ..parallel.peer_allreduce import choose_grad_allreduce, make_grad_allreduce
            kind, wg = knobs.dp_allreduce, knobs.peer_workgroups
            if not hasattr(comm, "backend") or not self.tail.is_cuda:
                kind = "rccl"           # FakeComm / CPU tensors: the communicator's own reduce
            self.comm_report, self._spare_allreduces = {"us": {}, "chosen": None}, []
            if knobs.dp_grad_dtype == "measure" and kind != "rccl":
                wires = ("fp32",) if e.x3 else ("fp32", "bf16")
                self.ar_tail, self.comm_report, self._spare_allreduces = choose_grad_allreduce(
                    comm, self.tail, kind, wires, wg)
                head_kind = "rccl" if self.ar_tail.kind == "rccl" else kind
                self.grad_dtype = "bf16" if self.ar_tail.kind == "peer-bf16" else "fp32"
            else:
                self.grad_dtype = "fp32" if knobs.dp_grad_dtype == "measure" else knobs.dp_grad_dtype
                self.ar_tail = make_grad_allreduce(comm, self.tail, kind, wire=self.grad_dtype, workgroups=wg)
                head_kind = kind
                self.comm_report["chosen"] = self.ar_tail.kind
            self.ar_head = make_grad_allreduce(comm, self.head, head_kind, workgroups=wg)
            self.allreduce_kind = self.ar_tail.kind
            self.peer = self.allreduce_kind.startswith("peer")
            if getattr(comm, "backend", None) != "nccl" and not self.peer:
                self.grad_dtype = "fp32"        # host back-ends (gloo rehearsals, tests) reduce the fp32 buffer
            # (process-group all-reduce with a bf16 wire: two conversion launches around the collective -- ~20 us per
            # step in the 1-rank rehearsal, 9.2k vs 7.8k updates/s -- so it only pays where the wire saves more)
            self.tail16 = (torch.empty_like(self.tail, dtype=BF16)
                           if self.grad_dtype == "bf16" and not self.peer else None)

    def _part_a(self):
        if not self.staged and not (self.presampled and self.e.presamples(self.staged)):
            with _Range("learner/sample"):
                self.e.sample()
        with _Range("learner/forward+td_head"):
            self.e.forward_head(self.staged)
        with _Range("learner/backward_fc"):
            self.e.backward_fc()

    def _part_b(self):
        with _Range("learner/backward_conv"):
            self.e.backward_conv(self.staged)

    def _part_c(self):
        with _Range("learner/adam+target+tree"):
            self.e.update(self.staged)

    def _whole(self):
        e = self.e
        if not self.staged and not (self.presampled and e.presamples(self.staged)):
            with _Range("learner/sample"):
                e.sample()
        with _Range("learner/forward+td_head"):
            e.forward_head(self.staged)
        with _Range("learner/backward_fc"):
            e.backward_fc()
        with _Range("learner/backward_conv"):
            e.backward_conv(self.staged, reduce=False)
        with _Range("learner/adam+target+tree"):
            e.update(self.staged, extras=self.extras)      # + slab sums + the next step's index draw
        self.presampled = True

    def _capture(self, fns):
        graphs = []
        for fn in fns:
            gr = torch.cuda.CUDAGraph()
            # thread_local: RCCL's watchdog thread may touch the runtime while this thread captures
            with torch.cuda.graph(gr, capture_error_mode="thread_local"):
                fn()
            graphs.append(gr)
        return graphs

    def _run_dp(self, a, b, probe=None):
        """One data-parallel step on the current stream + the comm stream.  Works eagerly, with
        a/b replaying captured graphs, and -- because every cross-stream edge is an event --
        inside a stream capture, where the comm stream becomes a forked branch of ONE graph.
        probe: dict of timing events (measure_comm) recorded at the hand-over points of an EAGER step."""
        e, cur = self.e, torch.cuda.current_stream()

        def mark(name, stream):
            if probe is not None:
                probe[name].record(stream)

        mark("step0", cur)
        a()
        self.fc_done.record(cur)
        with torch.cuda.stream(self.comm_stream):
            self.comm_stream.wait_event(self.fc_done)
            mark("tail0", self.comm_stream)
            if self.tail16 is not None:
                self.tail16.copy_(self.tail)
                self.comm.all_reduce_sum_(self.tail16)
                self.tail.copy_(self.tail16)
            else:
                self.ar_tail.all_reduce_sum_(self.tail)     # (peer kernel: bf16 wire conversion happens in it)
            mark("tail1", self.comm_stream)
            self.tail_done.record(self.comm_stream)
        b()
        self.conv_done.record(cur)
        with torch.cuda.stream(self.comm_stream):
            self.comm_stream.wait_event(self.conv_done)
            mark("head0", self.comm_stream)
            self.ar_head.all_reduce_sum_(self.head)
            mark("head1", self.comm_stream)
            self.head_done.record(self.comm_stream)
        mark("wait_tail", cur)
        cur.wait_event(self.tail_done)
        e.adam.step(table=self.tail_table)          # fc tail update runs under the conv all-reduce
        mark("wait_head", cur)
        cur.wait_event(self.head_done)
        e.adam.step(table=self.head_table, extras=self.dp_extras)     # + next step's index draw
        e.post_update(self.staged)
        mark("step1", cur)
        self.presampled = self.dp_extras is not None

    def measure_comm(self, steps=20):
        """Where a data-parallel step's communication goes, from `steps` EAGER steps with timing events at the
        hand-over points (these steps are real updates; call it outside any timed region).  -> dict, us per step:
          allreduce_tail_us / allreduce_head_us   the two gradient buckets' all-reduces, on the comm stream
          exposed_tail_us / exposed_head_us       how long the main stream had nothing to do but wait for each
                                                  (0 = the all-reduce was hidden under the conv backward / the fc
                                                  tail's optimizer update)
          step_us                                 the whole eager step (launch-bound: slower than the graph replay)"""
        if not self.dp:
            return {}
        names = ("step0", "tail0", "tail1", "head0", "head1", "wait_tail", "wait_head", "step1")
        acc = dict.fromkeys(("allreduce_tail_us", "allreduce_head_us", "exposed_tail_us", "exposed_head_us",
                             "step_us"), 0.0)
        dev = self.e.device
        for _ in range(int(steps)):
            probe = {n: torch.cuda.Event(enable_timing=True) for n in names}
            self._run_dp(self._part_a, self._part_b, probe=probe)
            torch.cuda.synchronize(dev)
            acc["allreduce_tail_us"] += 1e3 * probe["tail0"].elapsed_time(probe["tail1"])
            acc["allreduce_head_us"] += 1e3 * probe["head0"].elapsed_time(probe["head1"])
            acc["exposed_tail_us"] += max(0.0, 1e3 * probe["wait_tail"].elapsed_time(probe["tail1"]))
            acc["exposed_head_us"] += max(0.0, 1e3 * probe["wait_head"].elapsed_time(probe["head1"]))
            acc["step_us"] += 1e3 * probe["step0"].elapsed_time(probe["step1"])
            self.e.loss_steps += 1
            self.e.step_count += 1
        out = {k: round(v / max(int(steps), 1), 2) for k, v in acc.items()}
        out["steps"] = int(steps)
        out["note"] = ("eager steps with events; the exposed figures are upper bounds for the graph-replayed step, "
                       "whose compute side is shorter")
        return out

    def _dp_whole(self):
        self._run_dp(self._part_a, self._part_b)

    # ---- fail-safe: the peer kernel's waits are bounded; a wait that ran out leaves an error word ---------------------
    def comm_error(self):
        """0, or the error word of a peer all-reduce whose wait timed out on THIS rank (synchronises the device).
        The learner loop MAX-reduces it over the ranks every `stop_agree_every` steps (agents/learners.py), so a
        timed-out wait never feeds Adam for longer than that."""
        if not self.dp:
            return 0
        err = 0
        for ar in (self.ar_tail, self.ar_head):
            if hasattr(ar, "error"):
                err = max(err, abs(int(ar.error())))
        return err

    def fall_back_to_collectives(self):
        """Collective: every rank leaves the peer kernel for the process group's all-reduce (RCCL / gloo).  The
        updates since the failed wait used incomplete gradient sums, differently on each rank, so rank 0's
        parameters, target network and optimizer state are re-broadcast before the step graphs are rebuilt."""
        from ..parallel.peer_allreduce import _RcclAllReduce
        e, comm = self.e, self.comm
        torch.cuda.synchronize(e.device)
        for name in ("ar_tail", "ar_head"):
            ar = getattr(self, name)
            if hasattr(ar, "clear_error"):
                ar.clear_error()
            ar.close()
            setattr(self, name, _RcclAllReduce(comm))
        for ar in getattr(self, "_spare_allreduces", []):
            ar.close()
        self._spare_allreduces = []
        self.allreduce_kind = self.ar_tail.kind
        self.peer = False
        self.grad_dtype = "fp32"
        self.tail16 = None
        step = e.adam.step_dev.to(torch.float32).reshape(-1)
        for buf in (e.online.flat, e.target.flat, e.adam.m, e.adam.v, step):
            comm.broadcast_(buf, src=0)
        e.adam.step_dev.copy_(step.reshape(e.adam.step_dev.shape).to(e.adam.step_dev.dtype))
        e.online.refresh()
        e.target.refresh()
        self.graphs, self.calls, self.ahead = None, 0, 0
        self.presampled = False
        torch.cuda.synchronize(e.device)

    def _capture_dp(self):
        """Preferred: the whole step, collectives included, as ONE graph (one host call per
        step: 4899 vs 3810 updates/s in the 1-rank RCCL rehearsal).  RCCL's intra-node
        collectives are plain kernels and capture like any other launch; host-driven back-ends
        (gloo) cannot be captured, and a refused capture poisons the process, so the choice is
        made up front: knob dp_graph = whole | split (PDA_DP_GRAPH), or by back-end.  bench.py sets the variable from
        a pre-flight child process (parallel/preflight.py) that tries the capture first.
        Every rank takes the same decision (MIN over ranks)."""
        mode = self.knobs.dp_graph
        if self.peer and mode != "split":
            mode = "whole"      # the peer all-reduce is a plain kernel launch: nothing host-driven is left in the step
        if mode == "auto":
            # RCCL inside a graph only when a pre-flight child has shown that this node captures and replays it
            # (parallel/preflight.py sets PDA_DP_GRAPH=whole): a refused capture poisons the process
            mode = "split"
        whole = self.comm.all_agree(mode == "whole")
        if whole:
            self.dp_graph_mode = "whole"

            def k_steps():
                for _ in range(self.graph_steps):
                    self._dp_whole()
            return self._capture([k_steps])
        self.dp_graph_mode = "split"
        self.graph_steps = 1            # collectives stay host-driven: one step per pair of graphs
        return self._capture([self._part_a, self._part_b])

    def _throttle(self):
        """Block (host only) until at most max_inflight - 1 earlier replays are still queued or running."""
        ev = None
        if len(self._inflight) >= self.max_inflight:
            ev = self._inflight.pop(0)
            ev.synchronize()
        return ev if ev is not None else torch.cuda.Event()

    def _launched(self, ev):
        ev.record()
        self._inflight.append(ev)

    def step(self):
        e = self.e
        if self.ahead > 0:              # this update already ran as part of the last multi-step replay
            self.ahead -= 1
            self.calls += 1
            return
        ev = self._throttle() if (self.use_graph and e.device.type == "cuda") else None
        try:
            self._step_launch()
        finally:
            if ev is not None:
                self._launched(ev)

    def _step_launch(self):
        e = self.e
        first = (not self.use_graph) or self.calls == 0
        done = 1
        # the first step runs eagerly: it loads every code object (and lets RCCL set up its
        # channels), so nothing is lazily initialised while a stream is capturing
        if not self.dp:
            if first:
                self._whole()
            else:
                if self.graphs is None:
                    def k_steps():
                        for _ in range(self.graph_steps):
                            self._whole()
                    self.graphs = self._capture([k_steps])
                self.graphs[0].replay()
                done = self.graph_steps
                self.ahead = done - 1
        else:
            if first:
                self._dp_whole()
            else:
                if self.graphs is None:
                    self.graphs = self._capture_dp()
                if len(self.graphs) == 1:
                    self.graphs[0].replay()
                    done = self.graph_steps
                    self.ahead = done - 1
                else:
                    self._run_dp(self.graphs[0].replay, self.graphs[1].replay)
        self.calls += 1
        e.loss_steps += done
        # (with K updates per replay the period is a multiple of K and the first step ran alone, so a due
        # update is always the last one of the replay)
        e.step_count += done - 1
        if e.target_update >= 1.0 and e.step_count % int(e.target_update) == 0:
            e.hard_target_update()
        e.step_count += 1


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
        if self.on_device:
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
            if n != self._last_size:
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
