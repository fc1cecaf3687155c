"""fp32-precision MFMA family (csrc/gemm_x3.hip: split-bf16 operands, three bf16 products per logical product,
fp32 accumulation) against fp64 PyTorch references on UNROUNDED fp32 inputs.

The reference learner is fp32 (core/single_processes/dqn_learner.py:63-87); these kernels must reproduce fp32
results to ~1e-5 relative, where the plain bf16 family (test_gpu_gemm.py) needs inputs pre-rounded to bf16 and
2e-2 tolerances."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def nn():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from pytorch_distributed_amd import ops
    ops.require_native()
    from pytorch_distributed_amd.ops import nn as _nn
    return _nn


def rel(a, b):
    """max |a - b| / max |b|"""
    return ((a.double() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-30)).item()


TOL = 1e-4          # 16-bit significands: ~2^-16 per product, averaged down by the fp32 accumulation


def planes_nhwc(nn, x):
    """fp32 [B, C, H, W] -> Planes over NHWC rows [B*H*W, C]."""
    B, C, H, W = x.shape
    return nn.Planes.from_float(x.permute(0, 2, 3, 1).reshape(B * H * W, C).contiguous())


def planes_conv_w(nn, w):
    """fp32 [O, C, KH, KW] -> Planes [O, KH*KW*C] in (ky, kx, c) order."""
    return nn.Planes.from_float(w.permute(0, 2, 3, 1).reshape(w.shape[0], -1).contiguous())


def from_nhwc(v, B, H, W, C):
    return v.reshape(B, H, W, C).permute(0, 3, 1, 2)


def test_planes_hold_16_bits(nn):
    x = torch.randn(4096, device="cuda")
    p = nn.Planes.from_float(x)
    assert rel(p.value(), x) < 2 ** -16
    assert rel(x.bfloat16().float(), x) > 2 ** -10          # what one plane alone loses


@pytest.mark.parametrize("batch,use_idx", [(16, True), (5, False)])
def test_x3_conv1_fwd_u8(nn, batch, use_idx):
    torch.manual_seed(2)
    mem = torch.randint(0, 256, (64, 4, 84, 84), dtype=torch.uint8, device="cuda")
    idx = torch.randint(0, 64, (batch,), dtype=torch.int32, device="cuda") if use_idx else None
    w = torch.randn(32, 4, 8, 8, device="cuda") / 16
    b = torch.randn(32, device="cuda")
    out = nn.Planes(batch * 400, 32, device="cuda")
    nn.x3_conv_fwd_u8([mem], [idx], [nn.Planes.from_float(w.reshape(32, -1))], [b], [out], batch, 84, 84, 4, 8, 8, 4,
                      32, 1.0 / 255.0)
    x = (mem[idx.long()] if use_idx else mem[:batch]).double() / 255.0
    ref = F.relu(F.conv2d(x, w.double(), b.double(), stride=4))
    assert rel(from_nhwc(out.value(), batch, 20, 20, 32), ref) < TOL


@pytest.mark.parametrize("geom", [(20, 20, 32, 4, 4, 2, 64), (9, 9, 64, 3, 3, 1, 64)])
def test_x3_conv_fwd_and_backward(nn, geom):
    H, W, C, KH, KW, S, O = geom
    torch.manual_seed(3)
    batch = 16
    OH, OW = nn.conv_out(H, KH, S), nn.conv_out(W, KW, S)
    x = torch.relu(torch.randn(batch, C, H, W, device="cuda"))
    w = torch.randn(O, C, KH, KW, device="cuda") / (C * KH * KW) ** 0.5
    b = torch.randn(O, device="cuda")
    dy = torch.randn(batch, O, OH, OW, device="cuda")
    xd, wd, bd = (t.double().requires_grad_(True) for t in (x, w, b))
    pre = F.conv2d(xd, wd, bd, stride=S)
    pre.backward(dy.double())
    # forward (two problem groups)
    outs = [nn.Planes(batch * OH * OW, O, device="cuda") for _ in range(2)]
    nn.x3_conv_fwd([planes_nhwc(nn, x), planes_nhwc(nn, x * 0.5)], [planes_conv_w(nn, w)] * 2, [b, b], outs, batch,
                   H, W, C, KH, KW, S, O)
    assert rel(from_nhwc(outs[0].value(), batch, OH, OW, O), F.relu(pre.detach())) < TOL
    assert rel(from_nhwc(outs[1].value(), batch, OH, OW, O),
               F.relu(F.conv2d(xd.detach() * 0.5, wd.detach(), bd.detach(), stride=S))) < TOL
    # weight / bias gradient through split-K slabs
    K = C * KH * KW
    red = 3 * batch * OH * OW
    ns = nn.num_slices(red, 512)
    slab = torch.full((ns, O * K), 7.0, device="cuda")
    bslab = torch.full((ns, O), 7.0, device="cuda")
    nn.x3_conv_wgrad([planes_nhwc(nn, dy)], [planes_nhwc(nn, x)], [slab], [bslab], batch, [(H, W, C, KH, KW, S)], O,
                     512, [O * K], [O])
    gw = slab.sum(0).reshape(O, KH, KW, C).permute(0, 3, 1, 2)
    assert rel(gw, wd.grad) < TOL
    assert rel(bslab.sum(0), bd.grad) < TOL
    # data gradient (gather form), masked by the relu of the layer input
    dx = nn.Planes(batch * H * W, C, device="cuda")
    dx.full.fill_(9.0)
    nn.x3_conv_dgrad(planes_nhwc(nn, dy), planes_conv_w(nn, w), planes_nhwc(nn, x).hi, dx, batch, H, W, C, KH, KW, S, O,
                     act=nn.ACT_RELU)
    assert rel(from_nhwc(dx.value(), batch, H, W, C), xd.grad * (xd.detach() > 0)) < TOL


def test_x3_conv_wgrad_pair_and_u8(nn):
    torch.manual_seed(7)
    batch, ks = 12, 768
    geoms = [(20, 20, 32, 4, 4, 2), (9, 9, 64, 3, 3, 1)]
    dys, xs, slabs, bslabs, refs = [], [], [], [], []
    for (H, W, C, KH, KW, S) in geoms:
        OH, OW = nn.conv_out(H, KH, S), nn.conv_out(W, KW, S)
        K = C * KH * KW
        x = torch.relu(torch.randn(batch, C, H, W, device="cuda"))
        dy = torch.randn(batch, 64, OH, OW, device="cuda")
        wd = torch.zeros(64, C, KH, KW, device="cuda", dtype=torch.double, requires_grad=True)
        F.conv2d(x.double(), wd, None, stride=S).backward(dy.double())
        ns = nn.num_slices(3 * batch * OH * OW, ks)
        dys.append(planes_nhwc(nn, dy))
        xs.append(planes_nhwc(nn, x))
        slabs.append(torch.full((ns, 64 * K), 7.0, device="cuda"))
        bslabs.append(torch.full((ns, 64), 7.0, device="cuda"))
        refs.append((wd.grad, dy.double().sum((0, 2, 3)), KH, KW, C))
    nn.x3_conv_wgrad(dys, xs, slabs, bslabs, batch, geoms, 64, ks, [s.shape[1] for s in slabs], [64, 64])
    for slab, bslab, (gw, gb, KH, KW, C) in zip(slabs, bslabs, refs):
        assert rel(slab.sum(0).reshape(64, KH, KW, C).permute(0, 3, 1, 2), gw) < TOL
        assert rel(bslab.sum(0), gb) < TOL
    # conv1 from uint8 frames through an index vector
    mem = torch.randint(0, 256, (40, 4, 84, 84), dtype=torch.uint8, device="cuda")
    idx = torch.randint(0, 40, (batch,), dtype=torch.int32, device="cuda")
    dy = torch.randn(batch, 32, 20, 20, device="cuda")
    wd = torch.zeros(32, 4, 8, 8, device="cuda", dtype=torch.double, requires_grad=True)
    F.conv2d(mem[idx.long()].double() / 255.0, wd, None, stride=4).backward(dy.double())
    ns = nn.num_slices(2 * batch * 400, 512)
    slab = torch.full((ns, 32 * 256), 7.0, device="cuda")
    bslab = torch.full((ns, 32), 7.0, device="cuda")
    nn.x3_conv_wgrad_u8(planes_nhwc(nn, dy), mem, idx, slab, bslab, batch, 84, 84, 4, 8, 8, 4, 32, 1.0 / 255.0,
                        ksplit=512, slice_stride=32 * 256, bslice_stride=32)
    assert rel(slab.sum(0).reshape(32, 4, 8, 8), wd.grad) < TOL
    assert rel(bslab.sum(0), dy.double().sum((0, 2, 3))) < TOL


def test_x3_linear_family(nn):
    torch.manual_seed(4)
    B, Nout, Kin = 128, 512, 3136
    x = torch.relu(torch.randn(B, Kin, device="cuda"))
    w = torch.randn(Nout, Kin, device="cuda") / Kin ** 0.5
    dy = torch.randn(B, Nout, device="cuda")
    xp, wp, dyp = nn.Planes.from_float(x), nn.Planes.from_float(w), nn.Planes.from_float(dy)
    # forward, split-K atomics into fp32 (two groups)
    accs = [torch.zeros(B, Nout, device="cuda") for _ in range(2)]
    nn.x3_linear_fwd([xp, xp], [wp, wp], accs, B, Nout, Kin, Kin, Kin, Nout, ksplit=1024)
    ref = x.double() @ w.double().t()
    assert rel(accs[0], ref) < TOL and rel(accs[1], ref) < TOL
    # data gradient with relu mask
    dx = nn.Planes(B, Kin, device="cuda")
    nn.x3_linear_dgrad(dyp, wp, xp.hi, dx, B, Nout, Kin, Nout, Kin, Kin, Kin, act=nn.ACT_RELU)
    assert rel(dx.value(), (dy.double() @ w.double()) * (x > 0)) < TOL
    # the two fc weight gradients, alone and together with the data gradient
    dq = torch.zeros(B, 16, device="cuda")
    dq[:, :6] = torch.randn(B, 6, device="cuda")
    h = torch.relu(torch.randn(B, 512, device="cuda"))
    dqp, hp = nn.Planes.from_float(dq), nn.Planes.from_float(h)
    for duo in (False, True):
        gw, gb = torch.full((Nout, Kin), 7.0, device="cuda"), torch.full((Nout,), 7.0, device="cuda")
        gw2, gb2 = torch.full((6, 512), 7.0, device="cuda"), torch.full((6,), 7.0, device="cuda")
        pair = nn.X3LinearWgradPair([(dyp, xp, gw, gb, Nout, Kin, Nout, Kin, Kin),
                                     (dqp, hp, gw2, gb2, 6, 512, 16, 512, 512)], B)
        if duo:
            dx.full.fill_(3.0)
            pair.run_with_dgrad(dyp, wp, xp.hi, dx, B, Nout, Kin, Nout, Kin, Kin, Kin, act=nn.ACT_RELU)
            assert rel(dx.value(), (dy.double() @ w.double()) * (x > 0)) < TOL
        else:
            pair.run()
        assert rel(gw, dy.double().t() @ x.double()) < TOL
        assert rel(gb, dy.double().sum(0)) < TOL
        assert rel(gw2, dq[:, :6].double().t() @ h.double()) < TOL
        assert rel(gb2, dq[:, :6].double().sum(0)) < TOL
