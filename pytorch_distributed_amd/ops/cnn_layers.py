"""Nature-CNN building blocks shared by the DQN learner engine and the batched actor inference: layer geometry,
the flat-parameter views with their bf16 (and split-bf16) shadow weights in kernel layout, the activation / gradient
buffers, and the forward pass as problem groups of the same launches (kernels K1-K4 of SURVEY.md section 2.3).
A CNN of another geometry gets a module like this one next to it; the engine and the step program do not change."""
import numpy as np
import torch

from . import nn
from ._dqn_base import BF16, HID_SUPPORTED
from .optim import SegmentTable, polyak_


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
    if device.type == "cuda" and nn.gemm_large_tiles(B, g.HID, groups):
        # large batches: 128 x 128 tiles, two workgroups per CU -- with one, too few loads are in flight per CU
        # (batch 4096: 47 -> 43 us bf16, 99 -> 92 us split-bf16 with two slices; four: 58 / 117 us)
        big_tiles = ((B + 127) // 128) * ((g.HID + 127) // 128) * groups
        ks = _ksplit(mult * g.F, big_tiles, 2 * acts.fc1_blocks, 128)
    acts.h1_ksplit = ks
    acts.h1_slices = nn.num_slices(mult * g.F, ks)
    acts.h1pre = [torch.zeros(acts.h1_slices, B, g.HID, dtype=torch.float32, device=device) for _ in range(groups)]


