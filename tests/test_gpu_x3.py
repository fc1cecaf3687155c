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


class _View(object):
    """hi-plane view of a shared split buffer + the offset to its lo plane (what nn.x3_* take as an operand)."""

    def __init__(self, hi, plane):
        self.hi, self.plane = hi, plane

    def data_ptr(self):
        return self.hi.data_ptr()


@pytest.mark.parametrize("batch,use_idx", [(9, True), (3, False), (520, True)])
def test_x3_cnn_fwd_stack(nn, batch, use_idx):
    """conv1 -> conv2 -> conv3 in one launch on split planes (cnn_fwd_stack_x3_kernel), two networks, the second
    without kept a1 / a2, against fp64 torch.  520 images per network: 128 workgroups per network walk the batch
    (four or five images each, the next image's frames requested under conv3)."""
    if not nn.x3_cnn_fwd_stack_supported(84, 84, 4):
        pytest.skip("fused fp32-precision forward switched off")
    torch.manual_seed(5)
    mem = torch.randint(0, 256, (32, 4, 84, 84), dtype=torch.uint8, device="cuda")
    idx = torch.randint(0, 32, (batch,), dtype=torch.int32, device="cuda") if use_idx else None
    nets = []
    for n in range(2):
        ws = [torch.randn(32, 4, 8, 8, device="cuda") / 16, torch.randn(64, 32, 4, 4, device="cuda") / 22,
              torch.randn(64, 64, 3, 3, device="cuda") / 24]
        bs = [torch.randn(32, device="cuda") / 4, torch.randn(64, device="cuda") / 4, torch.randn(64, device="cuda") / 4]
        flat = [ws[0].reshape(32, -1), ws[1].permute(0, 2, 3, 1).reshape(64, -1), ws[2].permute(0, 2, 3, 1).reshape(64, -1)]
        total = sum(f.numel() for f in flat)
        buf = torch.zeros(2, total, dtype=torch.bfloat16, device="cuda")       # one shadow buffer: one plane offset
        views, off = [], 0
        for f in flat:
            hi = f.contiguous().to(torch.bfloat16)
            buf[0, off:off + f.numel()] = hi.reshape(-1)
            buf[1, off:off + f.numel()] = (f.contiguous() - hi.float()).to(torch.bfloat16).reshape(-1)
            views.append(_View(buf[0, off:off + f.numel()].view(f.shape), total))
            off += f.numel()
        nets.append((ws, bs, views, buf))
    a1 = [nn.Planes(batch * 400, 32, device="cuda"), None]
    a2 = [nn.Planes(batch * 81, 64, device="cuda"), None]
    a3 = [nn.Planes(batch * 49, 64, device="cuda") for _ in range(2)]
    nn.x3_cnn_fwd_stack([mem, mem], [idx, idx], [n[2][0] for n in nets], [n[1][0] for n in nets],
                        [n[2][1] for n in nets], [n[1][1] for n in nets], [n[2][2] for n in nets],
                        [n[1][2] for n in nets], a1, a2, a3, batch, 84, 84, 4, 1.0 / 255.0)
    x = (mem[idx.long()] if use_idx else mem[:batch]).double() / 255.0
    for n, (ws, bs, _, _) in enumerate(nets):
        r1 = F.relu(F.conv2d(x, ws[0].double(), bs[0].double(), stride=4))
        r2 = F.relu(F.conv2d(r1, ws[1].double(), bs[1].double(), stride=2))
        r3 = F.relu(F.conv2d(r2, ws[2].double(), bs[2].double(), stride=1))
        if n == 0:
            assert rel(from_nhwc(a1[0].value(), batch, 20, 20, 32), r1) < TOL
            assert rel(from_nhwc(a2[0].value(), batch, 9, 9, 64), r2) < TOL
        assert rel(from_nhwc(a3[n].value(), batch, 7, 7, 64), r3) < TOL


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
    red = batch * OH * OW           # logical reduction length (the kernel stages hi and lo tiles per k-tile)
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


@pytest.mark.parametrize("batch", [7, 130, 300, 1100])
def test_x3_cnn_dgrad_stack(nn, batch):
    """da3 -> da2 -> da1 in one launch on split planes (cnn_dgrad_stack_x3_kernel) against fp64 torch autograd of
    relu(conv3(relu(conv2(a1)))) with random a1 >= 0 (about half of it exactly 0: the relu' gate of the last stage).
    Batches 300 / 1100 take the one-workgroup-per-image variants (more images than CUs): one image per workgroup without
    the conv1 tail, 256 workgroups walking the batch with it (one or two / four or five images each, the next image's
    inputs and conv3 weights requested under the tail)."""
    if not nn.x3_cnn_dgrad_stack_supported(84, 84, 4):
        pytest.skip("fused fp32-precision backward switched off")
    torch.manual_seed(11)
    a1 = torch.relu(torch.randn(batch, 32, 20, 20, device="cuda"))
    w2 = torch.randn(64, 32, 4, 4, device="cuda") / 22
    w3 = torch.randn(64, 64, 3, 3, device="cuda") / 24
    b2, b3 = torch.randn(64, device="cuda") / 4, torch.randn(64, device="cuda") / 4
    da3 = torch.randn(batch, 64, 7, 7, device="cuda")
    a1d = a1.double().requires_grad_(True)
    a2d = F.relu(F.conv2d(a1d, w2.double(), b2.double(), stride=2))
    a2d.retain_grad()
    pre3 = F.conv2d(a2d, w3.double(), b3.double(), stride=1)
    pre3.backward(da3.double())
    ref_da2 = a2d.grad * (a2d.detach() > 0)         # gradient w.r.t. conv2's pre-activation
    ref_da1 = a1d.grad * (a1d.detach() > 0)         # ... conv1's pre-activation
    # the kernel reads the TRANSPOSED shadows [(tap, c)][o]
    flat = [w2.permute(2, 3, 1, 0).reshape(-1, 64).contiguous(), w3.permute(2, 3, 1, 0).reshape(-1, 64).contiguous()]
    total = sum(f.numel() for f in flat)
    buf = torch.zeros(2, total, dtype=torch.bfloat16, device="cuda")
    views, off = [], 0
    for f in flat:
        hi = f.to(torch.bfloat16)
        buf[0, off:off + f.numel()] = hi.reshape(-1)
        buf[1, off:off + f.numel()] = (f - hi.float()).to(torch.bfloat16).reshape(-1)
        views.append(_View(buf[0, off:off + f.numel()].view(f.shape), total))
        off += f.numel()
    a2_hi = planes_nhwc(nn, a2d.detach().float()).hi
    a1_hi = planes_nhwc(nn, a1).hi
    da2 = nn.Planes(batch * 81, 64, device="cuda")
    da1 = nn.Planes(batch * 400, 32, device="cuda")
    da2.full.fill_(9.0)
    da1.full.fill_(9.0)
    nn.x3_cnn_dgrad_stack(planes_nhwc(nn, da3), a2_hi, a1_hi, views[1], views[0], da2, da1, batch)
    assert rel(from_nhwc(da2.value(), batch, 9, 9, 64), ref_da2) < TOL
    assert rel(from_nhwc(da1.value(), batch, 20, 20, 32), ref_da1) < TOL
    # the same launch with conv1's weight gradient as its tail, frames through an index
    mem = torch.randint(0, 256, (40, 4, 84, 84), dtype=torch.uint8, device="cuda")
    idx = torch.randint(0, 40, (batch,), dtype=torch.int32, device="cuda")
    ns = nn.x3_cnn_dgrad_stack_slices(batch)        # one slab slice per workgroup: two per image, or 256 walking the batch
    assert ns == (2 * batch if batch <= 256 else 256)
    gw = torch.full((ns, 32 * 256), 7.0, device="cuda")
    gb = torch.full((ns, 32), 7.0, device="cuda")
    da2.full.fill_(9.0)
    da1.full.fill_(9.0)
    nn.x3_cnn_dgrad_stack(planes_nhwc(nn, da3), a2_hi, a1_hi, views[1], views[0], da2, da1, batch,
                          w1=(mem, idx, gw, gb, 1.0 / 255.0))
    assert rel(from_nhwc(da1.value(), batch, 20, 20, 32), ref_da1) < TOL
    x = mem[idx.long()].double() / 255.0
    cols = F.unfold(x, kernel_size=8, stride=4)                     # [B, 256 (c, ky, kx), 400]
    ref_gw = torch.einsum("bop,bkp->ok", ref_da1.reshape(batch, 32, 400), cols)
    assert rel(gw.sum(0).view(32, 256), ref_gw) < TOL
    assert rel(gb.sum(0), ref_da1.sum(dim=(0, 2, 3))) < TOL


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
        ns = nn.num_slices(batch * OH * OW, ks)
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
    ns = nn.num_slices(batch * 400, 512)
    slab = torch.full((ns, 32 * 256), 7.0, device="cuda")
    bslab = torch.full((ns, 32), 7.0, device="cuda")
    nn.x3_conv_wgrad_u8(planes_nhwc(nn, dy), mem, idx, slab, bslab, batch, 84, 84, 4, 8, 8, 4, 32, 1.0 / 255.0,
                        ksplit=512, slice_stride=32 * 256, bslice_stride=32)
    assert rel(slab.sum(0).reshape(32, 4, 8, 8), wd.grad) < TOL
    assert rel(bslab.sum(0), dy.double().sum((0, 2, 3))) < TOL
    # the same gradient by the whole-image kernel (frame stack in LDS, dy hi / lo tiles, one slab slice per workgroup)
    ns = nn.conv1_wgrad_slices(batch)
    slab = torch.full((ns, 32 * 256), 7.0, device="cuda")
    bslab = torch.full((ns, 32), 7.0, device="cuda")
    nn.x3_conv1_wgrad_u8_whole(planes_nhwc(nn, dy), mem, idx, slab, bslab, batch, 84, 84, 4, 1.0 / 255.0, 32 * 256, 32)
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


@pytest.mark.parametrize("mode", [0, 1, 2, 3])
@pytest.mark.parametrize("M,N,K", [(4096, 512, 3136), (300, 200, 328)])
def test_x3_linear_large_tiles(nn, mode, M, N, K):
    """The 128-row tiles (k-tile 32) of the split-operand fc1 forward / data gradient: batch 4096's shapes, where mode 0
    picks them, and a ragged shape, against fp64 torch."""
    nn.gemm_tile_mode(mode)
    try:
        torch.manual_seed(6)
        x = torch.relu(torch.randn(M, K, device="cuda"))
        ws = [torch.randn(N, K, device="cuda") / K ** 0.5 for _ in range(2)]
        dy = torch.randn(M, N, device="cuda")
        xp, wp, dyp = nn.Planes.from_float(x), [nn.Planes.from_float(w) for w in ws], nn.Planes.from_float(dy)
        slabs = [torch.full((1, M, N), 7.0, device="cuda") for _ in range(2)]
        nn.x3_linear_fwd([xp, xp], wp, slabs, M, N, K, K, K, N, ksplit=K, slice_stride=M * N)
        for w, s in zip(ws, slabs):
            assert rel(s[0], x.double() @ w.double().t()) < TOL
        dx = nn.Planes(M, K, device="cuda")
        dx.full.fill_(3.0)
        nn.x3_linear_dgrad(dyp, wp[0], xp.hi, dx, M, N, K, N, K, K, K, act=nn.ACT_RELU)
        assert rel(dx.value(), (dy.double() @ ws[0].double()) * (x > 0)) < TOL
    finally:
        nn.gemm_tile_mode(0)


# ---- the engine at fp32 precision ------------------------------------------------------------------------------
def _models(A=6, seed=0):
    from pytorch_distributed_amd.models.dqn_cnn_model import DqnCnnModel
    from pytorch_distributed_amd.utils.options import Options
    torch.manual_seed(seed)
    opt = Options(config=2)
    m = DqnCnnModel(opt.model_params, 255.0, (4, 84, 84), A, 1).cuda()
    t = DqnCnnModel(opt.model_params, 255.0, (4, 84, 84), A, 1).cuda()
    return m, t


def _batch(B, A, seed=1):
    g = torch.Generator(device="cuda").manual_seed(seed)
    s0 = torch.randint(0, 256, (B, 4, 84, 84), dtype=torch.uint8, device="cuda", generator=g)
    s1 = torch.randint(0, 256, (B, 4, 84, 84), dtype=torch.uint8, device="cuda", generator=g)
    tails = torch.zeros(B, 5, device="cuda")
    tails[:, 0] = torch.randint(0, A, (B,), device="cuda", generator=g).float()
    tails[:, 1] = torch.randn(B, device="cuda", generator=g)
    tails[:, 2] = 0.99 ** 5
    tails[:, 3] = (torch.rand(B, device="cuda", generator=g) < 0.1).float()
    return s0, s1, tails


def _torch_step(model, target, s0, s1, tails, double):
    """The reference learner's arithmetic (core/single_processes/dqn_learner.py:67-80), in fp64."""
    import copy
    model, target = copy.deepcopy(model).double(), copy.deepcopy(target).double()

    def fwd(net, s):        # DqnCnnModel.forward in fp64 (the module's own forward casts its input to fp32)
        x = net.critic[0](s.double() / 255.0)
        return net.critic[1](x.flatten(1))
    a, r, g, t = (tails[:, i:i + 1].double() for i in range(4))
    with torch.no_grad():
        if double:
            nq = fwd(target, s1).gather(1, fwd(model, s1).argmax(1, keepdim=True))
        else:
            nq = fwd(target, s1).max(1, keepdim=True)[0]
        y = r + g * nq * (1 - t)
    q = fwd(model, s0)
    pred = q.gather(1, a.long())
    loss = ((pred - y) ** 2).mean()
    loss.backward()
    return model, q.detach(), (pred - y).detach().reshape(-1), loss.item()


@pytest.mark.parametrize("B,double,A", [(128, False, 6), (32, True, 18)])
def test_engine_fp32_precision_gradients_match_fp64_torch(nn, B, double, A):
    """Per-layer gradients of the fp32-precision engine within 1e-3 (relative L2) of the reference learner's
    step -- the bf16 engine is allowed 6-12 % here (tests/test_gpu_dqn_engine.py)."""
    from pytorch_distributed_amd.ops import dqn_engine as eng
    model, target = _models(A)
    e = eng.DqnCnnEngine(model, target, B, double=double, target_update=10 ** 9, precision="fp32")
    assert e.x3 and e.online.shadow_plane > 0
    with torch.no_grad():
        for p in target.parameters():
            p.add_(0.01 * torch.randn_like(p))
    e.target.refresh()
    # split shadows carry 16 significand bits of the masters
    w4 = model.critic[1][0].weight.detach()
    got = e.online.w[3].value()
    want = w4.reshape(512, 64, 49).permute(0, 2, 1).reshape(512, -1)
    assert rel(got, want) < 2 ** -16
    s0, s1, tails = _batch(B, A)
    e.make_staging(5)
    e.stage_s0.copy_(s0); e.stage_s1.copy_(s1); e.stage_tails.copy_(tails)
    e.forward_head(staged=True)
    e.backward_fc()
    e.backward_conv(staged=True)
    torch.cuda.synchronize()
    ref, q_ref, td_ref, loss_ref = _torch_step(model, target, s0, s1, tails, double)
    assert rel(e.q, q_ref) < 1e-4
    assert rel(e.td, td_ref) < 1e-4
    assert abs(e.loss_acc.sum().item() - loss_ref) < 1e-4 * max(1.0, abs(loss_ref))
    grads = e.grads_torch_layout()
    off, errs = 0, {}
    for name, p in ref.named_parameters():
        got = grads[off:off + p.numel()].view_as(p)
        off += p.numel()
        errs[name] = ((got.double() - p.grad).norm() / p.grad.norm().clamp_min(1e-30)).item()
    print("fp32-precision grad rel errs vs fp64:", {k: float("%.2g" % v) for k, v in errs.items()})
    # What an fp32 PyTorch step (the reference learner's own arithmetic) deviates from the same fp64 step by.  ReLU's
    # derivative is discontinuous: a unit whose pre-activation lies within an implementation's rounding error of zero
    # passes its whole gradient in one implementation and nothing in the other, so ANY two finite-precision steps
    # disagree by ~sqrt(fraction of such units) on the conv gradients.
    import copy
    m32 = copy.deepcopy(model)
    m32.zero_grad(set_to_none=True)
    a_, r_, g_, t_ = (tails[:, i:i + 1] for i in range(4))
    with torch.no_grad():
        nq = (target(s1).gather(1, m32(s1).argmax(1, keepdim=True)) if double else target(s1).max(1, keepdim=True)[0])
        y32 = r_ + g_ * nq * (1 - t_)
    ((m32(s0).gather(1, a_.long()) - y32) ** 2).mean().backward()
    errs32 = {n: ((p32.grad.double() - p.grad).norm() / p.grad.norm()).item()
              for (n, p32), (_, p) in zip(m32.named_parameters(), ref.named_parameters())}
    print("fp32 torch grad rel errs vs fp64:    ", {k: float("%.2g" % v) for k, v in errs32.items()})
    for name, err in errs.items():
        limit = 5e-3 if name.startswith("critic.0") else 1e-4      # conv layers: mask flips; fc layers: arithmetic
        assert err < limit, "%s: rel err %g (all: %s)" % (name, err, errs)

    # The arithmetic alone, with the discontinuity taken out: the same fp64 step run with the ENGINE's ReLU masks
    # (units the two implementations disagree on are counted, and must be a handful per million).
    masks = {"a1": e.acts.a1[0].hi.view(B, 20, 20, 32).permute(0, 3, 1, 2) > 0,
             "a2": e.acts.a2[0].hi.view(B, 9, 9, 64).permute(0, 3, 1, 2) > 0,
             "a3": e.acts.a3[0].hi.view(B, 7, 7, 64).permute(0, 3, 1, 2) > 0,
             "h1": e.acts.h1.hi > 0}
    m64 = copy.deepcopy(model).double()
    c, f = m64.critic[0], m64.critic[1]
    flips, total = 0, 0

    def masked(pre, mask):
        nonlocal flips, total
        flips += int(((pre > 0) != mask).sum().item())
        total += mask.numel()
        return pre * mask.double()
    x = masked(c[0](s0.double() / 255.0), masks["a1"])
    x = masked(c[2](x), masks["a2"])
    x = masked(c[4](x), masks["a3"])
    h = masked(f[0](x.flatten(1)), masks["h1"])
    q = f[2](h)
    y = (q_ref.gather(1, a_.long()) - td_ref[:, None]).detach()        # the fp64 step's own TD target
    ((q.gather(1, a_.long()) - y) ** 2).mean().backward()
    errs_m = {}
    off = 0
    for name, p in m64.named_parameters():
        got = grads[off:off + p.numel()].view_as(p)
        off += p.numel()
        errs_m[name] = ((got.double() - p.grad).norm() / p.grad.norm().clamp_min(1e-30)).item()
    print("fp32-precision grad rel errs vs fp64 with the engine's ReLU masks:",
          {k: float("%.2g" % v) for k, v in errs_m.items()}, "| mask disagreements: %d of %d units" % (flips, total))
    assert flips <= 1e-4 * total
    for name, err in errs_m.items():
        assert err < 1e-4, "%s: rel err %g with identical masks (all: %s)" % (name, err, errs_m)

    # How far from true fp32 is "fp32-equivalent"?  The same masked step in plain fp32 PyTorch (the reference learner's
    # arithmetic: 24-bit significands) against the same fp64 step, layer by layer with identical masks: the engine
    # (16-bit operand significands, fp32 accumulation; 2^8 = 256x coarser operands) must stay within 128x of fp32's
    # own distance from fp64 (measured: 15-20x on every weight gradient, 16-31x on the biases, 67x on the 18-action fc2 bias).
    m32m = copy.deepcopy(model).float()
    c32, f32 = m32m.critic[0], m32m.critic[1]
    x = c32[0](s0.float() / 255.0) * masks["a1"].float()
    x = c32[2](x) * masks["a2"].float()
    x = c32[4](x) * masks["a3"].float()
    h = (f32[0](x.flatten(1)) * masks["h1"].float())
    ((f32[2](h).gather(1, a_.long()) - y.float()) ** 2).mean().backward()
    ratios = {}
    for (name, p32), (_, p64) in zip(m32m.named_parameters(), m64.named_parameters()):
        e32 = ((p32.grad.double() - p64.grad).norm() / p64.grad.norm().clamp_min(1e-30)).item()
        ratios[name] = errs_m[name] / max(e32, 1e-9)
    print("engine error / fp32-torch error (both vs fp64, identical masks):",
          {k: float("%.3g" % v) for k, v in ratios.items()})
    for name in ratios:
        e32 = errs_m[name] / ratios[name]
        assert errs_m[name] <= 128.0 * max(e32, 1e-9), \
            "%s: engine %.3g vs fp32 torch %.3g from fp64 (ratio %.1f)" % (name, errs_m[name], e32, ratios[name])


def test_engine_fp32_precision_adam_keeps_split_shadows(nn):
    from pytorch_distributed_amd.ops import dqn_engine as eng
    model, target = _models()
    e = eng.DqnCnnEngine(model, target, 32, lr=1e-3, target_update=2, precision="fp32")
    torch.manual_seed(3)
    for _ in range(3):
        e.grads.copy_(e.online.to_kernel_layout(torch.randn_like(e.grads) * 0.1))
        e.update(staged=True, tick=True)
    torch.cuda.synchronize()
    from pytorch_distributed_amd.ops import nn as pnn
    w2 = model.critic[0][2].weight.detach()
    want = w2.permute(0, 2, 3, 1).reshape(64, -1)
    assert torch.equal(e.online.w[1].hi, pnn.pack_conv_weight(w2))
    assert rel(e.online.w[1].value(), want) < 2 ** -16
    # the transposed copies [(tap, c)][o] the data-gradient stack reads follow every optimizer path bit for bit
    for li in (1, 2):
        assert torch.equal(e.online.wT[li].hi, e.online.w[li].hi.t().contiguous())
        assert torch.equal(e.online.wT[li].lo, e.online.w[li].lo.t().contiguous())
    e.hard_target_update()
    assert torch.equal(e.target.flat, e.online.flat)
    assert torch.equal(e.target.shadow, e.online.shadow)
    # ... also the vector row path (gradients given directly, as in a data-parallel step after the reduce launch)
    # and a refresh from the master weights
    e.adam.step(tick=True)
    e.online.flat.mul_(1.25)
    e.online.refresh()
    torch.cuda.synchronize()
    for li in (1, 2):
        assert torch.equal(e.online.wT[li].hi, e.online.w[li].hi.t().contiguous())
        assert torch.equal(e.online.wT[li].lo, e.online.w[li].lo.t().contiguous())


def test_engine_fp32_precision_steps_from_hbm_replay_in_a_graph(nn):
    from pytorch_distributed_amd.ops import dqn_engine as eng
    B, mem_n = 64, 512
    s0, s1, tails = _batch(B, 6)
    tails[:, 3] = 1.0           # terminal: pure regression onto the rewards
    g = torch.Generator(device="cuda").manual_seed(2)
    frames = torch.randint(0, 256, (mem_n, 4, 84, 84), dtype=torch.uint8, device="cuda", generator=g)
    rt = tails[torch.arange(mem_n, device="cuda") % B].contiguous()
    model, target = _models(seed=7)
    e = eng.DqnCnnEngine(model, target, B, lr=3e-4, target_update=1000, seed=11, precision="fp32")
    e.bind_replay(frames, frames.flip(0).contiguous(), rt, None)
    e.set_size(mem_n)
    prog = eng.DqnStepProgram(e, None, use_graph=True, staged=False, graph_steps=5)
    prog.step()
    first = e.mean_loss()
    for _ in range(200):
        prog.step()
    e.mean_loss()
    for _ in range(10):
        prog.step()
    last = e.mean_loss()
    torch.cuda.synchronize()
    assert prog.graphs is not None and prog.graph_steps == 5
    assert last == last and last < 0.5 * first, (first, last)
