"""HIP MFMA GEMM family vs plain PyTorch fp32 references (run on a real MI355X).
Inputs are rounded to bf16 first so the only difference is accumulation order."""
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


@pytest.fixture
def generic_conv1(nn):
    """Route the Nature conv1 geometry through the generic implicit-GEMM kernels for one test."""
    yield nn.force_generic_conv1
    nn.force_generic_conv1(False)


def bf(x):
    return x.to(torch.bfloat16).float()


def close(a, b, rtol=2e-2, atol=2e-2):
    err = (a.float() - b.float()).abs().max().item()
    ref = b.float().abs().max().item()
    assert err <= atol + rtol * ref, "max err %g vs ref max %g" % (err, ref)


def test_tr_probe_mapping(nn):
    got = nn.tr_probe()
    for lane in range(64):
        g, i = lane >> 4, lane & 15
        for j in range(8):
            assert int(got[lane, j]) == (8 * g + j) * 72 + i, (lane, j, got[lane].tolist())


@pytest.mark.parametrize("M,N,K", [(128, 512, 3136), (64, 320, 32), (37, 200, 320), (4096, 64, 576)])
def test_linear_fwd(nn, M, N, K):
    torch.manual_seed(0)
    x = bf(torch.randn(M, K, device="cuda"))
    w = bf(torch.randn(N, K, device="cuda") / K ** 0.5)
    b = torch.randn(N, device="cuda")
    out = torch.zeros(M, N, dtype=torch.bfloat16, device="cuda")
    out32 = torch.zeros(M, N, device="cuda")
    nn.linear_fwd([x.bfloat16()], [w.bfloat16()], [b], [out], [out32], M, N, K, K, K, N, N, act=nn.ACT_RELU)
    ref = F.relu(x @ w.t() + b)
    close(out32, ref, 1e-3, 1e-3)
    close(out, ref)
    acc = torch.zeros(M, N, device="cuda")
    nn.linear_fwd([x.bfloat16()], [w.bfloat16()], None, None, [acc], M, N, K, K, K, 0, N, ksplit=256)
    close(acc, x @ w.t(), 1e-3, 1e-3)


@pytest.mark.parametrize("M,groups,ksplit", [(128, 2, 448), (640, 2, 3136), (200, 1, 1600)])
def test_linear_fwd_slab_both_block_orders(nn, M, groups, ksplit):
    """fc1's split-K forward as slabs: with >= 8 (slice, group) units the launch runs in XCD order, with fewer (large
    batches: one slice) in plain block order -- the slices must add up to x w^T either way."""
    torch.manual_seed(4)
    N, K = 512, 3136
    ns = nn.num_slices(K, ksplit)
    xs = [bf(torch.randn(M, K, device="cuda")) for _ in range(groups)]
    ws = [bf(torch.randn(N, K, device="cuda") / K ** 0.5) for _ in range(groups)]
    outs = [torch.full((ns, M, N), 7.0, device="cuda") for _ in range(groups)]
    nn.linear_fwd_slab([x.bfloat16() for x in xs], [w.bfloat16() for w in ws], outs, M, N, K, K, K, N, M * N, ksplit)
    for x, w, o in zip(xs, ws, outs):
        close(o.sum(0), x @ w.t(), 1e-3, 1e-3)


@pytest.fixture
def tile_mode(nn):
    """Force a tile variant of the GEMM entry points for one test (csrc/gemm.hip pda_gemm_tile_mode)."""
    yield nn.gemm_tile_mode
    nn.gemm_tile_mode(0)


@pytest.mark.parametrize("mode", [0, 1, 2, 3])
@pytest.mark.parametrize("M,N,K", [(8192, 512, 3136), (300, 200, 328)])
def test_linear_fwd_slab_and_dgrad_large_tiles(nn, tile_mode, mode, M, N, K):
    """128 x 128 tiles (k-tile 64 / 32) of fc1's forward and data gradient -- batch 4096's shapes, where mode 0 picks
    them itself, and a ragged shape (no side a multiple of 128) -- against the 64 x 64 tiles and fp32 torch."""
    tile_mode(mode)
    torch.manual_seed(5)
    xs = [bf(torch.relu(torch.randn(M, K, device="cuda"))) for _ in range(2)]
    ws = [bf(torch.randn(N, K, device="cuda") / K ** 0.5) for _ in range(2)]
    outs = [torch.full((1, M, N), 7.0, device="cuda") for _ in range(2)]
    nn.linear_fwd_slab([x.bfloat16() for x in xs], [w.bfloat16() for w in ws], outs, M, N, K, K, K, N, M * N, K)
    for x, w, o in zip(xs, ws, outs):
        close(o[0], x @ w.t(), 1e-3, 1e-3)
    # data gradient of the same layer: dx[M][K] = (dy[M][N] w[N][K]) * (x > 0)
    dy = bf(torch.randn(M, N, device="cuda"))
    dx = torch.full((M, K), 3.0, dtype=torch.bfloat16, device="cuda")
    nn.linear_dgrad(dy.bfloat16(), ws[0].bfloat16(), xs[0].bfloat16(), dx, M, N, K, N, K, K, K, act=nn.ACT_RELU)
    close(dx, (dy @ ws[0]) * (xs[0] > 0))


def test_linear_fwd_grouped_tanh(nn):
    torch.manual_seed(1)
    M, N, K = 64, 96, 64
    xs = [bf(torch.randn(M, K, device="cuda")) for _ in range(3)]
    ws = [bf(torch.randn(N, K, device="cuda") / 8) for _ in range(3)]
    bs = [torch.randn(N, device="cuda") for _ in range(3)]
    outs = [torch.zeros(M, N, device="cuda") for _ in range(3)]
    nn.linear_fwd([x.bfloat16() for x in xs], [w.bfloat16() for w in ws], bs, None, outs, M, N, K, K, K, 0, N,
                  act=nn.ACT_TANH)
    for x, w, b, o in zip(xs, ws, bs, outs):
        close(o, torch.tanh(x @ w.t() + b), 1e-3, 1e-3)


@pytest.mark.parametrize("generic", [False, True])
@pytest.mark.parametrize("batch,use_idx", [(32, True), (5, False), (600, True)])
def test_conv1_fwd_u8(nn, batch, use_idx, generic, generic_conv1):
    """generic: the implicit-GEMM loader of gemm.h; otherwise the whole-image kernel of nature_conv1.hip
    (600 images > 512 workgroups: the per-block image loop)."""
    generic_conv1(generic)
    assert nn.conv1_specialised(84, 84, 4, 8, 8, 4, 32) == (not generic)
    torch.manual_seed(2)
    mem = torch.randint(0, 256, (64, 4, 84, 84), dtype=torch.uint8, device="cuda")
    idx = torch.randint(0, 64, (batch,), dtype=torch.int32, device="cuda") if use_idx else None
    w = bf(torch.randn(32, 4, 8, 8, device="cuda") / 16)
    b = torch.randn(32, device="cuda")
    out = torch.zeros(batch * 400, 32, dtype=torch.bfloat16, device="cuda")
    nn.conv_fwd_u8([mem], [idx], [w.reshape(32, -1).bfloat16().contiguous()], [b], [out], batch, 84, 84, 4, 8, 8, 4, 32,
                   1.0 / 255.0)
    x = (mem[idx.long()] if use_idx else mem[:batch]).float() / 255.0
    ref = F.relu(F.conv2d(x, w, b, stride=4))
    close(nn.from_nhwc(out, batch, 20, 20, 32), ref)


@pytest.mark.parametrize("geom", [(20, 20, 32, 4, 4, 2, 64), (9, 9, 64, 3, 3, 1, 64)])
def test_conv_fwd_nhwc(nn, geom):
    H, W, C, KH, KW, S, O = geom
    torch.manual_seed(3)
    batch = 16
    x = bf(torch.randn(batch, C, H, W, device="cuda"))
    w = bf(torch.randn(O, C, KH, KW, device="cuda") / (C * KH * KW) ** 0.5)
    b = torch.randn(O, device="cuda")
    OH, OW = nn.conv_out(H, KH, S), nn.conv_out(W, KW, S)
    outs = [torch.zeros(batch * OH * OW, O, dtype=torch.bfloat16, device="cuda") for _ in range(2)]
    nn.conv_fwd([nn.nhwc(x), nn.nhwc(x * 0.5)], [nn.pack_conv_weight(w)] * 2, [b, b], outs, batch, H, W, C, KH, KW, S, O)
    close(nn.from_nhwc(outs[0], batch, OH, OW, O), F.relu(F.conv2d(x, w, b, stride=S)))
    close(nn.from_nhwc(outs[1], batch, OH, OW, O), F.relu(F.conv2d(x * 0.5, w, b, stride=S)))


def test_linear_dgrad_and_wgrad(nn):
    torch.manual_seed(4)
    B, Nout, C, P = 128, 512, 64, 49
    Kin = C * P
    dy = bf(torch.randn(B, Nout, device="cuda"))
    w = bf(torch.randn(Nout, Kin, device="cuda") / Kin ** 0.5)      # kernel layout: columns (p, c)
    x = bf(torch.randn(B, Kin, device="cuda"))
    y = bf(torch.randn(B, Kin, device="cuda"))
    dx = torch.zeros(B, Kin, dtype=torch.bfloat16, device="cuda")
    nn.linear_dgrad(dy.bfloat16(), w.bfloat16(), y.bfloat16(), dx, B, Nout, Kin, Nout, Kin, Kin, Kin, act=nn.ACT_RELU)
    close(dx, (dy @ w) * (y > 0))
    nn.linear_dgrad(dy.bfloat16(), w.bfloat16(), y.bfloat16(), dx, B, Nout, Kin, Nout, Kin, Kin, Kin, act=nn.ACT_TANH)
    close(dx, (dy @ w) * (1 - y * y))
    gw = torch.full((Nout, Kin), 7.0, device="cuda")       # plain stores: stale contents must vanish
    gb = torch.full((Nout,), 7.0, device="cuda")
    nn.linear_wgrad(dy.bfloat16(), x.bfloat16(), gw, gb, B, Nout, Kin, Nout, Kin, Kin)
    close(gw, dy.t() @ x, 1e-3, 1e-3)
    close(gb, dy.sum(0), 1e-3, 1e-3)
    # small-M instance (fc2: 6 actions padded to 16 columns), split-K over the batch
    dq = torch.zeros(B, 16, device="cuda")
    dq[:, :6] = bf(torch.randn(B, 6, device="cuda"))
    h = bf(torch.randn(B, 512, device="cuda"))
    ns = nn.num_slices(B, 64)
    slab = torch.full((ns, 6 * 512), 7.0, device="cuda")
    bslab = torch.full((ns, 6), 7.0, device="cuda")
    nn.linear_wgrad(dq.bfloat16(), h.bfloat16(), slab, bslab, B, 6, 512, 16, 512, 512, ksplit=64,
                    slice_stride=6 * 512, bslice_stride=6)
    gw2 = torch.zeros(6, 512, device="cuda")
    gb2 = torch.zeros(6, device="cuda")
    nn.ReducePlan([(slab, gw2, 6 * 512, ns, 6 * 512), (bslab, gb2, 6, ns, 6)]).run()
    close(gw2, dq[:, :6].t() @ h, 1e-3, 1e-3)
    close(gb2, dq[:, :6].sum(0), 1e-3, 1e-3)


@pytest.mark.parametrize("geom", [(20, 20, 32, 4, 4, 2, 64), (9, 9, 64, 3, 3, 1, 64)])
def test_conv_backward_nhwc(nn, geom):
    H, W, C, KH, KW, S, O = geom
    torch.manual_seed(5)
    batch = 16
    OH, OW = nn.conv_out(H, KH, S), nn.conv_out(W, KW, S)
    x = bf(torch.relu(torch.randn(batch, C, H, W, device="cuda"))).requires_grad_(True)
    w = bf(torch.randn(O, C, KH, KW, device="cuda") / (C * KH * KW) ** 0.5).requires_grad_(True)
    b = torch.zeros(O, device="cuda", requires_grad=True)
    dy = bf(torch.randn(batch, O, OH, OW, device="cuda"))
    F.conv2d(x, w, b, stride=S).backward(dy)
    # weight / bias gradient
    K = C * KH * KW
    ns = nn.num_slices(batch * OH * OW, 256)
    slab = torch.full((ns, O * K), 7.0, device="cuda")
    bslab = torch.full((ns, O), 7.0, device="cuda")
    nn.conv_wgrad(nn.nhwc(dy), nn.nhwc(x.detach()), slab, bslab, batch, H, W, C, KH, KW, S, O, ksplit=256,
                  slice_stride=O * K, bslice_stride=O)
    gw = torch.zeros(O, KH, KW, C, device="cuda")            # kernel layout (ky, kx, c)
    gb = torch.zeros(O, device="cuda")
    nn.ReducePlan([(slab, gw, O * K, ns, O * K), (bslab, gb, O, ns, O)]).run()
    close(gw.permute(0, 3, 1, 2), w.grad, 2e-3, 2e-3)
    close(gb, b.grad, 2e-3, 2e-3)
    # data gradient = dcol GEMM + col2im (masked by relu of the layer input)
    M = batch * OH * OW
    dcol = torch.zeros(M, K, dtype=torch.bfloat16, device="cuda")
    nn.linear_dgrad(nn.nhwc(dy), nn.pack_conv_weight(w), None, dcol, M, O, K, O, K, 0, K)
    dx = torch.zeros(batch * H * W, C, dtype=torch.bfloat16, device="cuda")
    nn.col2im(dcol, nn.nhwc(x.detach()), dx, batch, H, W, C, KH, KW, S, act=nn.ACT_RELU)
    close(nn.from_nhwc(dx, batch, H, W, C), x.grad * (x.detach() > 0), 3e-2, 3e-2)
    # the fused gather-form data gradient gives the same thing in one launch
    dx2 = torch.full((batch * H * W, C), 9.0, dtype=torch.bfloat16, device="cuda")
    nn.conv_dgrad(nn.nhwc(dy), nn.pack_conv_weight(w), nn.nhwc(x.detach()), dx2, batch, H, W, C, KH, KW, S, O,
                  act=nn.ACT_RELU)
    close(nn.from_nhwc(dx2, batch, H, W, C), x.grad * (x.detach() > 0), 3e-2, 3e-2)


def test_conv_wgrad_pair_equals_separate_launches(nn):
    """conv2 + conv3 weight gradients as two problem groups of one launch (common split-K slice length)
    write the same slabs, bit for bit, as the two separate launches."""
    torch.manual_seed(12)
    batch, ks = 24, 256
    geoms = [(20, 20, 32, 4, 4, 2), (9, 9, 64, 3, 3, 1)]
    dys, xs, sep, pair = [], [], [], []
    for (H, W, C, KH, KW, S) in geoms:
        OH, OW = nn.conv_out(H, KH, S), nn.conv_out(W, KW, S)
        K = C * KH * KW
        ns = nn.num_slices(batch * OH * OW, ks)
        dys.append(bf(torch.randn(batch * OH * OW, 64, device="cuda")).bfloat16())
        xs.append(bf(torch.relu(torch.randn(batch * H * W, C, device="cuda"))).bfloat16())
        sep.append((torch.full((ns, 64 * K), 7.0, device="cuda"), torch.full((ns, 64), 7.0, device="cuda"), K))
        pair.append((torch.full((ns, 64 * K), 9.0, device="cuda"), torch.full((ns, 64), 9.0, device="cuda"), K))
    for (H, W, C, KH, KW, S), dy, x, (slab, bslab, K) in zip(geoms, dys, xs, sep):
        nn.conv_wgrad(dy, x, slab, bslab, batch, H, W, C, KH, KW, S, 64, ksplit=ks, slice_stride=64 * K,
                      bslice_stride=64)
    nn.conv_wgrad_pair(dys, xs, [p[0] for p in pair], [p[1] for p in pair], batch, geoms, 64, ks,
                       [64 * p[2] for p in pair], [64, 64])
    torch.cuda.synchronize()
    for (s, bs, _), (p, bp, _) in zip(sep, pair):
        assert torch.equal(s, p) and torch.equal(bs, bp)


@pytest.mark.parametrize("batch", [3, 300])
def test_cnn_wgrad_stack_matches_torch(nn, batch):
    """conv2 + conv3 weight / bias gradients per image with register accumulation == fp32 torch conv backward
    (300 images > 256 workgroups exercises the per-workgroup image loop and its prefetch)."""
    torch.manual_seed(13)
    a1 = bf(torch.relu(torch.randn(batch, 32, 20, 20, device="cuda"))).requires_grad_(True)
    a2 = bf(torch.relu(torch.randn(batch, 64, 9, 9, device="cuda"))).requires_grad_(True)
    w2 = torch.zeros(64, 32, 4, 4, device="cuda", requires_grad=True)
    w3 = torch.zeros(64, 64, 3, 3, device="cuda", requires_grad=True)
    b2 = torch.zeros(64, device="cuda", requires_grad=True)
    b3 = torch.zeros(64, device="cuda", requires_grad=True)
    da2 = bf(torch.randn(batch, 64, 9, 9, device="cuda"))
    da3 = bf(torch.randn(batch, 64, 7, 7, device="cuda"))
    F.conv2d(a1, w2, b2, stride=2).backward(da2)
    F.conv2d(a2, w3, b3, stride=1).backward(da3)
    ns = nn.cnn_wgrad_stack_slices(batch)
    s2, sb2 = torch.full((ns, 64 * 512), 7.0, device="cuda"), torch.full((ns, 64), 7.0, device="cuda")
    s3, sb3 = torch.full((ns, 64 * 576), 7.0, device="cuda"), torch.full((ns, 64), 7.0, device="cuda")
    nn.cnn_wgrad_stack(nn.nhwc(a1.detach()), nn.nhwc(a2.detach()), nn.nhwc(da2), nn.nhwc(da3), s2, sb2, s3, sb3, batch)
    gw2, gb2 = torch.zeros(64, 4, 4, 32, device="cuda"), torch.zeros(64, device="cuda")
    gw3, gb3 = torch.zeros(64, 3, 3, 64, device="cuda"), torch.zeros(64, device="cuda")
    nn.ReducePlan([(s2, gw2, 64 * 512, ns, 64 * 512), (sb2, gb2, 64, ns, 64),
                   (s3, gw3, 64 * 576, ns, 64 * 576), (sb3, gb3, 64, ns, 64)]).run()
    tol = 3e-3 * max(1.0, batch / 8) ** 0.5
    close(gw2.permute(0, 3, 1, 2), w2.grad, tol, 3e-3)
    close(gw3.permute(0, 3, 1, 2), w3.grad, tol, 3e-3)
    close(gb2, b2.grad, tol, 3e-3)
    close(gb3, b3.grad, tol, 3e-3)


@pytest.mark.parametrize("batch", [5, 300])
def test_cnn_dgrad_stack_with_conv1_wgrad_tail(nn, batch):
    """The dgrad stack launch with the conv1 weight-gradient tail: da2 / da1 identical to the plain launch, and
    the conv1 gradient (from da1 still in LDS + the frame patch) == fp32 torch."""
    torch.manual_seed(14)
    mem = torch.randint(0, 256, (64, 4, 84, 84), dtype=torch.uint8, device="cuda")
    idx = torch.randint(0, 64, (batch,), dtype=torch.int32, device="cuda")
    a1 = bf(torch.relu(torch.randn(batch * 400, 32, device="cuda"))).bfloat16()
    a2 = bf(torch.relu(torch.randn(batch * 81, 64, device="cuda"))).bfloat16()
    da3 = bf(torch.randn(batch * 49, 64, device="cuda")).bfloat16()
    w2 = bf(torch.randn(64, 512, device="cuda") / 22).bfloat16()
    w3 = bf(torch.randn(64, 576, device="cuda") / 24).bfloat16()
    ref2, ref1 = torch.zeros_like(a2), torch.zeros_like(a1)
    nn.cnn_dgrad_stack(da3, a2, a1, w3, w2, ref2, ref1, batch)
    ns = nn.cnn_dgrad_stack_slices(batch)
    slab, bslab = torch.full((ns, 32 * 256), 7.0, device="cuda"), torch.full((ns, 32), 7.0, device="cuda")
    da2, da1 = torch.zeros_like(a2), torch.zeros_like(a1)
    nn.cnn_dgrad_stack(da3, a2, a1, w3, w2, da2, da1, batch, w1=(mem, idx, slab, bslab, 1.0 / 255.0))
    torch.cuda.synchronize()
    assert torch.equal(da2, ref2) and torch.equal(da1, ref1)
    gw, gb = torch.zeros(32, 256, device="cuda"), torch.zeros(32, device="cuda")
    nn.ReducePlan([(slab, gw, 32 * 256, ns, 32 * 256), (bslab, gb, 32, ns, 32)]).run()
    x = mem[idx.long()].float() / 255.0
    w = torch.zeros(32, 4, 8, 8, device="cuda", requires_grad=True)
    b = torch.zeros(32, device="cuda", requires_grad=True)
    dy = da1.float().view(batch, 20, 20, 32).permute(0, 3, 1, 2).contiguous()
    F.conv2d(x, w, b, stride=4).backward(dy)
    tol = 2e-3 * max(1.0, batch / 8) ** 0.5 * float(dy.abs().max())
    close(gw.reshape(32, 4, 8, 8), w.grad, tol, 3e-3)
    close(gb, b.grad, tol, 3e-3)


@pytest.mark.parametrize("generic", [False, True])
@pytest.mark.parametrize("batch", [8, 300])
def test_conv1_wgrad_u8(nn, batch, generic, generic_conv1):
    generic_conv1(generic)
    torch.manual_seed(6)
    mem = torch.randint(0, 256, (32, 4, 84, 84), dtype=torch.uint8, device="cuda")
    idx = torch.randint(0, 32, (batch,), dtype=torch.int32, device="cuda")
    x = mem[idx.long()].float() / 255.0
    w = torch.zeros(32, 4, 8, 8, device="cuda", requires_grad=True)
    b = torch.zeros(32, device="cuda", requires_grad=True)
    dy = bf(torch.randn(batch, 32, 20, 20, device="cuda"))
    F.conv2d(x, w, b, stride=4).backward(dy)
    ns = nn.num_slices(batch * 400, 512) if generic else nn.conv1_wgrad_slices(batch)
    slab = torch.full((ns, 32 * 256), 7.0, device="cuda")
    bslab = torch.full((ns, 32), 7.0, device="cuda")
    nn.conv_wgrad_u8(nn.nhwc(dy), mem, idx, slab, bslab, batch, 84, 84, 4, 8, 8, 4, 32, 1.0 / 255.0, ksplit=512,
                     slice_stride=32 * 256, bslice_stride=32)
    gw = torch.zeros(32, 256, device="cuda")
    gb = torch.zeros(32, device="cuda")
    nn.ReducePlan([(slab, gw, 32 * 256, ns, 32 * 256), (bslab, gb, 32, ns, 32)]).run()
    tol = 2e-3 * max(1.0, batch / 8) ** 0.5
    close(gw.reshape(32, 4, 8, 8), w.grad, tol, 2e-3)
    close(gb, b.grad, tol, 2e-3)


@pytest.mark.parametrize("batch,groups", [(5, 1), (128, 2), (300, 3)])
def test_cnn_fwd_stack_matches_layerwise(nn, batch, groups):
    """conv1 -> conv2 -> conv3 in ONE launch (intermediates in LDS) == fp32 torch convs, and the a1 / a2
    copy-outs of the kept group match too; 300 images > 256 workgroups exercises the per-block image loop."""
    if not nn.cnn_fwd_stack_supported(84, 84, 4):
        pytest.skip("fused forward not available")
    torch.manual_seed(9)
    mem = torch.randint(0, 256, (64, 4, 84, 84), dtype=torch.uint8, device="cuda")
    idxs, ws, outs = [], [], []
    for gi in range(groups):
        idx = torch.randint(0, 64, (batch,), dtype=torch.int32, device="cuda")
        w1 = bf(torch.randn(32, 4, 8, 8, device="cuda") / 16)
        w2 = bf(torch.randn(64, 32, 4, 4, device="cuda") / 22)
        w3 = bf(torch.randn(64, 64, 3, 3, device="cuda") / 24)
        b1, b2, b3 = (torch.randn(n, device="cuda") * 0.1 for n in (32, 64, 64))
        idxs.append(idx)
        ws.append((w1, b1, w2, b2, w3, b3))
        keep = gi == 0
        outs.append((torch.zeros(batch * 400, 32, dtype=torch.bfloat16, device="cuda") if keep else None,
                     torch.zeros(batch * 81, 64, dtype=torch.bfloat16, device="cuda") if keep else None,
                     torch.zeros(batch * 49, 64, dtype=torch.bfloat16, device="cuda")))
    nn.cnn_fwd_stack([mem] * groups, idxs,
                     [w[0].reshape(32, -1).bfloat16().contiguous() for w in ws], [w[1] for w in ws],
                     [nn.pack_conv_weight(w[2]) for w in ws], [w[3] for w in ws],
                     [nn.pack_conv_weight(w[4]) for w in ws], [w[5] for w in ws],
                     [o[0] for o in outs], [o[1] for o in outs], [o[2] for o in outs], batch, 84, 84, 4, 1.0 / 255.0)
    for gi in range(groups):
        w1, b1, w2, b2, w3, b3 = ws[gi]
        x = mem[idxs[gi].long()].float() / 255.0
        r1 = bf(F.relu(F.conv2d(x, w1, b1, stride=4)))
        r2 = bf(F.relu(F.conv2d(r1, w2, b2, stride=2)))
        r3 = F.relu(F.conv2d(r2, w3, b3, stride=1))
        close(nn.from_nhwc(outs[gi][2], batch, 7, 7, 64), r3, 3e-2, 3e-2)
        if outs[gi][0] is not None:
            close(nn.from_nhwc(outs[gi][0], batch, 20, 20, 32), r1)
            close(nn.from_nhwc(outs[gi][1], batch, 9, 9, 64), r2, 3e-2, 3e-2)


@pytest.mark.parametrize("batch", [3, 128, 600])
def test_cnn_dgrad_stack_matches_torch(nn, batch):
    """da3 -> conv3^T, relu'(a2) -> da2 -> conv2^T, relu'(a1) -> da1 in ONE launch == torch autograd."""
    if not nn.cnn_dgrad_stack_supported(84, 84, 4):
        pytest.skip("fused backward not available")
    torch.manual_seed(11)
    a1 = bf(torch.relu(torch.randn(batch, 32, 20, 20, device="cuda"))).requires_grad_(True)
    w2 = bf(torch.randn(64, 32, 4, 4, device="cuda") / 22)
    w3 = bf(torch.randn(64, 64, 3, 3, device="cuda") / 24)
    z2 = F.conv2d(a1, w2, None, stride=2) + 0.1
    a2 = torch.relu(z2)
    a2b = bf(a2.detach())
    da3 = bf(torch.randn(batch, 64, 7, 7, device="cuda"))
    # reference with the same bf16 rounding of the intermediate gradient
    a2_leaf = a2b.clone().requires_grad_(True)
    F.conv2d(a2_leaf, w3, None, stride=1).backward(da3)
    da2_ref = bf(a2_leaf.grad * (a2b > 0))
    a1_leaf = a1.detach().clone().requires_grad_(True)
    F.conv2d(a1_leaf, w2, None, stride=2).backward(da2_ref)
    da1_ref = a1_leaf.grad * (a1.detach() > 0)
    da2 = torch.full((batch * 81, 64), 7.0, dtype=torch.bfloat16, device="cuda")
    da1 = torch.full((batch * 400, 32), 7.0, dtype=torch.bfloat16, device="cuda")
    nn.cnn_dgrad_stack(nn.nhwc(da3), nn.nhwc(a2b), nn.nhwc(a1.detach()), nn.pack_conv_weight(w3),
                       nn.pack_conv_weight(w2), da2, da1, batch)
    close(nn.from_nhwc(da2, batch, 9, 9, 64), da2_ref, 3e-2, 3e-2)
    close(nn.from_nhwc(da1, batch, 20, 20, 32), da1_ref, 3e-2, 3e-2)
