"""ctypes bindings for the MFMA GEMM family (csrc/gemm.hip) and small helpers.

All functions enqueue on torch's current HIP stream and allocate nothing, so
they are safe inside torch.cuda.graph() capture.  Tensors are passed as raw
device pointers; shapes/pitches are explicit because activations are bf16 NHWC
buffers owned by the engines, not autograd tensors.

Layouts
  activations   bf16, NHWC flattened to [rows = (b, y, x)][channels]
  conv weights  bf16 "shadow" copies [O][(ky, kx, c)]  (conv1: [O][(c, ky, kx)] = PyTorch's)
  fc weights    bf16 [N][K], K permuted to the NHWC flatten order where needed
  gradients     fp32, plain stores in the same (kernel) layout as the bf16 weights; split-K
                launches write partial slabs that ReducePlan sums
"""
import ctypes

import torch

from . import lib

ACT_NONE, ACT_RELU, ACT_TANH = 0, 1, 2
BF16 = torch.bfloat16
GEMM_MAX_GROUPS = 4     # csrc/gemm.h: problem groups per launch

_vp = ctypes.c_void_p


def _ptr(t):
    return _vp(t.data_ptr()) if t is not None else _vp(0)


def _arr(ts):
    return (_vp * len(ts))(*[(t.data_ptr() if t is not None else 0) for t in ts])


def _stream():
    return _vp(torch.cuda.current_stream().cuda_stream)


class FrameWindows(object):
    """Frame stacks read as windows of a plane store (frame-deduplicated replay, memories/dedup_memory.py):
    image b of a batch is the `window` consecutive planes that start at plane ref[idx[b]] of `store`
    (uint8 [planes, plane_bytes]).  Accepted wherever the Nature-CNN kernels take `frames` (csrc/nature_cnn.h
    frame_index); the generic implicit-GEMM conv1 kernels do not read windows."""

    def __init__(self, store, ref, window):
        assert store.dtype == torch.uint8 and store.dim() == 2 and ref.dtype == torch.int32
        assert store.stride(0) % 16 == 0
        self.store, self.ref, self.window = store, ref, int(window)
        self.stride = int(store.stride(0))

    def data_ptr(self):
        return self.store.data_ptr()


def _windows(frames):
    """-> (ref tensors or None per entry, plane stride or 0) for a list of frame sources."""
    refs = [f.ref if isinstance(f, FrameWindows) else None for f in frames]
    strides = {f.stride for f in frames if isinstance(f, FrameWindows)}
    if not strides:
        return refs, 0
    if len(strides) != 1 or any(r is None for r in refs):
        raise ValueError("frame sources of one launch must all be windows of stores with one plane stride, or none")
    return refs, strides.pop()


def _no_windows(frames, what):
    if any(isinstance(f, FrameWindows) for f in (frames if isinstance(frames, (list, tuple)) else [frames])):
        raise ValueError("%s reads full frame stacks only (frame-deduplicated replay needs the Nature-CNN kernels)" % what)


_configured = False


def _L():
    global _configured
    L = lib()
    if not _configured:
        c_int, c_long, c_float = ctypes.c_int, ctypes.c_long, ctypes.c_float
        L.pda_conv_fwd_u8.argtypes = [_vp] * 5 + [c_int] * 9 + [c_float, c_int, _vp]
        L.pda_conv_fwd.argtypes = [_vp] * 4 + [c_int] * 10 + [_vp]
        L.pda_linear_fwd.argtypes = [_vp] * 5 + [c_int] * 4 + [c_long] * 4 + [c_int, c_int, _vp]
        L.pda_linear_dgrad.argtypes = [_vp] * 4 + [c_int] * 3 + [c_long] * 4 + [c_int, _vp]
        L.pda_linear_wgrad.argtypes = [_vp] * 4 + [c_int] * 3 + [c_long] * 5 + [c_float, c_int, _vp]
        L.pda_conv_wgrad.argtypes = [_vp] * 4 + [c_int] * 8 + [c_long, c_long, c_int, _vp]
        L.pda_conv_wgrad_u8.argtypes = [_vp] * 5 + [c_int] * 8 + [c_float, c_long, c_long, c_int, _vp]
        L.pda_wgrad_reduce.argtypes = [_vp] * 5 + [c_int, _vp]
        L.pda_col2im.argtypes = [_vp] * 3 + [c_int] * 8 + [_vp]
        L.pda_conv1_supported.argtypes = [c_int] * 7
        L.pda_conv1_supported.restype = c_int
        L.pda_conv1_fwd_u8_w.argtypes = [_vp] * 3 + [c_long] + [_vp] * 3 + [c_int] * 5 + [c_float, c_int, _vp]
        L.pda_conv1_fwd_u8_w.restype = None
        L.pda_conv1_wgrad_slices.argtypes = [c_int]
        L.pda_conv1_wgrad_slices.restype = c_int
        L.pda_conv1_wgrad_u8.argtypes = [_vp] * 5 + [c_int] * 4 + [c_float, c_long, c_long, _vp, c_long, _vp]
        L.pda_conv1_wgrad_u8.restype = None
        L.pda_cnn_fwd_stack_supported.argtypes = [c_int] * 3
        L.pda_cnn_fwd_stack_supported.restype = c_int
        L.pda_cnn_fwd_stack.argtypes = [_vp] * 3 + [c_long] + [_vp] * 9 + [c_int] * 5 + [c_float, _vp]
        L.pda_cnn_fwd_stack.restype = None
        L.pda_cnn_dgrad_stack_supported.argtypes = [c_int] * 3
        L.pda_cnn_dgrad_stack_supported.restype = c_int
        L.pda_cnn_dgrad_stack.argtypes = [_vp] * 7 + [c_int, _vp]
        L.pda_cnn_dgrad_stack.restype = None
        L.pda_linear_wgrad2.argtypes = [_vp] * 4 + [c_int] + [_vp] * 5 + [_vp]
        L.pda_linear_wgrad2.restype = None
        L.pda_conv_dgrad.argtypes = [_vp] * 4 + [c_int] * 9 + [_vp]
        L.pda_conv_dgrad.restype = None
        L.pda_tr_probe.argtypes = [_vp, _vp]
        for name in ("pda_conv_fwd_u8", "pda_conv_fwd", "pda_linear_fwd", "pda_linear_dgrad", "pda_linear_wgrad",
                     "pda_conv_wgrad", "pda_conv_wgrad_u8", "pda_col2im", "pda_tr_probe", "pda_wgrad_reduce"):
            getattr(L, name).restype = None
        _configured = True
    return L


def conv_out(h, k, s):
    return (h - k) // s + 1


FORCE_GENERIC_CONV1 = False


def force_generic_conv1(on):
    """Test hook: route the Nature conv1 geometry through the generic implicit-GEMM kernels (python and C side)."""
    global FORCE_GENERIC_CONV1
    FORCE_GENERIC_CONV1 = bool(on)
    L = _L()
    L.pda_force_generic_conv1.argtypes = [ctypes.c_int]
    L.pda_force_generic_conv1.restype = None
    L.pda_force_generic_conv1(1 if on else 0)


def gemm_tile_mode(mode):
    """Test / measurement hook of the GEMM entry points' tile choice (csrc/gemm.hip pda_gemm_tile_mode): 0 = by
    problem size (128 x 128 tiles when they still give every CU a workgroup), 1 = 64 x 64 tiles only, 2 / 3 = the
    large-tile variants whenever the shape allows."""
    L = _L()
    L.pda_gemm_tile_mode.argtypes = [ctypes.c_int]
    L.pda_gemm_tile_mode.restype = None
    L.pda_gemm_tile_mode(int(mode))


def gemm_large_tiles(M, N, units=1):
    """True when the slab / split-operand fc forward of this size runs on 128 x 128 tiles (csrc/gemm.hip)."""
    L = _L()
    L.pda_gemm_large_tiles.argtypes = [ctypes.c_int] * 4
    L.pda_gemm_large_tiles.restype = ctypes.c_int
    return bool(L.pda_gemm_large_tiles(int(M), int(N), int(units), 2))


def conv1_specialised(H, W, C, KH, KW, S, O):
    """True when the whole-image conv1 kernels (csrc/nature_conv1.hip) cover this layer.  (FORCE_GENERIC_CONV1: a
    test hook -- tests/test_gpu_gemm.py runs the Nature geometry through the generic implicit-GEMM kernels too.)"""
    return not FORCE_GENERIC_CONV1 and bool(_L().pda_conv1_supported(H, W, C, KH, KW, S, O))


def cnn_fwd_stack_supported(H, W, C):
    """True when the one-launch conv1 -> conv2 -> conv3 forward (csrc/nature_fwd_stack.hip) covers this input."""
    return conv1_specialised(H, W, C, 8, 8, 4, 32) and bool(_L().pda_cnn_fwd_stack_supported(H, W, C))


def cnn_fwd_stack(frames, idx, w1, b1, w2, b2, w3, b3, a1, a2, a3, batch, H, W, C, scale):
    """Nature-CNN conv stack forward, one workgroup per image, every intermediate in LDS.  Lists with one
    entry per problem group (<= 4); a1 / a2 entries may be None (activations not needed for a backward)."""
    n = len(frames)
    refs, stride = _windows(frames)
    _L().pda_cnn_fwd_stack(_arr(frames), _arr(idx), _arr(refs), stride, _arr(w1), _arr(b1), _arr(w2), _arr(b2),
                           _arr(w3), _arr(b3), _arr(a1), _arr(a2), _arr(a3), n, batch, H, W, C, float(scale), _stream())


def cnn_dgrad_stack_supported(H, W, C):
    return not FORCE_GENERIC_CONV1 and bool(_L().pda_cnn_dgrad_stack_supported(H, W, C))


def cnn_dgrad_stack_slices(batch):
    L = _L()
    L.pda_cnn_dgrad_stack_slices.argtypes = [ctypes.c_int]
    L.pda_cnn_dgrad_stack_slices.restype = ctypes.c_int
    return int(L.pda_cnn_dgrad_stack_slices(int(batch)))


def cnn_dgrad_stack(da3, a2, a1, w3, w2, da2, da1, batch, w1=None):
    """da2 = conv3^T(da3) * relu'(a2), da1 = conv2^T(da2) * relu'(a1) for the Nature CNN on 84x84 inputs, one
    workgroup per image with every intermediate in LDS (no dcol matrices, no col2im passes).
    w1 = (frames uint8, idx int32 or None, gw slab [slices][32*256], gb slab [slices][32], scale): the same launch
    also leaves conv1's weight / bias gradient, one slab slice per workgroup (cnn_dgrad_stack_slices(batch))."""
    L = _L()
    if w1 is None:
        L.pda_cnn_dgrad_stack(_ptr(da3), _ptr(a2), _ptr(a1), _ptr(w3), _ptr(w2), _ptr(da2), _ptr(da1), batch,
                              _stream())
        return
    if not getattr(L, "_dgw1_ready", False):
        L.pda_cnn_dgrad_stack_w1.argtypes = ([_vp] * 7 + [ctypes.c_int] + [_vp] * 4 + [ctypes.c_long, ctypes.c_long,
                                             ctypes.c_float, _vp, ctypes.c_long, _vp])
        L.pda_cnn_dgrad_stack_w1.restype = None
        L._dgw1_ready = True
    frames, idx, gw, gb, scale = w1
    (ref,), stride = _windows([frames])
    L.pda_cnn_dgrad_stack_w1(_ptr(da3), _ptr(a2), _ptr(a1), _ptr(w3), _ptr(w2), _ptr(da2), _ptr(da1), int(batch),
                             _ptr(frames), _ptr(idx), _ptr(gw), _ptr(gb), 32 * 256, 32, float(scale), _ptr(ref), stride,
                             _stream())


def cnn_wgrad_stack_slices(batch):
    L = _L()
    L.pda_cnn_wgrad_stack_slices.argtypes = [ctypes.c_int]
    L.pda_cnn_wgrad_stack_slices.restype = ctypes.c_int
    return int(L.pda_cnn_wgrad_stack_slices(int(batch)))


def cnn_wgrad_stack(a1, a2, da2, da3, gw2, gb2, gw3, gb3, batch):
    """conv2 + conv3 weight gradients of the Nature CNN (84x84 inputs) per image, accumulated in registers over a
    workgroup's images: slab slices [cnn_wgrad_stack_slices(batch)][64*512], [..][64], [..][64*576], [..][64]."""
    L = _L()
    if not getattr(L, "_wgstack_ready", False):
        L.pda_cnn_wgrad_stack.argtypes = [_vp] * 8 + [ctypes.c_long] * 4 + [ctypes.c_int, _vp]
        L.pda_cnn_wgrad_stack.restype = None
        L._wgstack_ready = True
    L.pda_cnn_wgrad_stack(_ptr(a1), _ptr(a2), _ptr(da2), _ptr(da3), _ptr(gw2), _ptr(gb2), _ptr(gw3), _ptr(gb3),
                          64 * 512, 64, 64 * 576, 64, int(batch), _stream())


def conv1_wgrad_slices(batch):
    return int(_L().pda_conv1_wgrad_slices(int(batch)))


def conv_fwd_u8(frames, idx, w, bias, out, batch, H, W, C, KH, KW, S, O, scale, act=ACT_RELU):
    """Lists (one entry per problem group, <= 3).  frames: uint8 [*, C, H, W]; idx: int32 [batch] or None."""
    assert KW == 8 and len(frames) <= 3
    if conv1_specialised(H, W, C, KH, KW, S, O):
        refs, stride = _windows(frames)
        _L().pda_conv1_fwd_u8_w(_arr(frames), _arr(idx), _arr(refs), stride, _arr(w), _arr(bias), _arr(out),
                                len(frames), batch, H, W, C, float(scale), act, _stream())
        return
    _no_windows(frames, "the generic conv1 forward")
    _L().pda_conv_fwd_u8(_arr(frames), _arr(idx), _arr(w), _arr(bias), _arr(out), len(frames), batch,
                         H, W, C, KH, KW, S, O, float(scale), act, _stream())


def conv_fwd(x, w, bias, out, batch, H, W, C, KH, KW, S, O, act=ACT_RELU):
    assert C % 8 == 0 and len(x) <= 3
    _L().pda_conv_fwd(_arr(x), _arr(w), _arr(bias), _arr(out), len(x), batch, H, W, C, KH, KW, S, O, act, _stream())


def linear_fwd(x, w, bias, out, out32, M, N, K, ldx, ldw, ldo=0, ldo32=0, act=ACT_NONE, ksplit=0):
    # K need not be a multiple of 8, but then the pad columns (up to the next multiple of 8,
    # inside ldx / ldw) of x AND w must be zero: the loaders fetch 8 k at a time
    assert ldx % 8 == 0 and ldw % 8 == 0 and ldx >= (K + 7) // 8 * 8 and ldw >= (K + 7) // 8 * 8 and len(x) <= 3
    n = len(x)
    _L().pda_linear_fwd(_arr(x), _arr(w), _arr(bias if bias else [None] * n), _arr(out if out else [None] * n),
                        _arr(out32 if out32 else [None] * n), n, M, N, K, ldx, ldw, ldo, ldo32, act, ksplit,
                        _stream())


def linear_fwd_slab(x, w, out32, M, N, K, ldx, ldw, ldo32, slice_stride, ksplit):
    """Split-K linear forward into slabs: slice y of group g -> out32[g] + y * slice_stride elements, plain stores,
    XCD-ordered launch.  num_slices(K, ksplit) slices; the consumer sums them (csrc/rl_kernels.hip: dqn_head)."""
    assert ldx % 8 == 0 and ldw % 8 == 0 and len(x) <= GEMM_MAX_GROUPS and slice_stride >= M * ldo32
    L = _L()
    if not getattr(L, "_fwd_slab_ready", False):
        c_int, c_long = ctypes.c_int, ctypes.c_long
        L.pda_linear_fwd_slab.argtypes = [_vp] * 3 + [c_int] * 4 + [c_long] * 4 + [c_int, _vp]
        L.pda_linear_fwd_slab.restype = None
        L._fwd_slab_ready = True
    L.pda_linear_fwd_slab(_arr(x), _arr(w), _arr(out32), len(x), M, N, K, ldx, ldw, ldo32, int(slice_stride),
                          int(ksplit), _stream())


def linear_dgrad(dy, w, y, dx, M, Nout, Kin, lddy, ldw, ldy, lddx, act=ACT_NONE):
    """dx[M][Kin] = (dy[M][Nout] @ w[Nout][Kin]) * act'(y[M][Kin]) (y None: no mask).
    Nout / Kin need not be multiples of 8 when the pad columns of dy (inside lddy) are zero
    and ldw covers Kin rounded up to 8."""
    assert lddy % 8 == 0 and ldw % 8 == 0 and lddy >= (Nout + 7) // 8 * 8 and ldw >= (Kin + 7) // 8 * 8
    _L().pda_linear_dgrad(_ptr(dy), _ptr(w), _ptr(y), _ptr(dx), M, Nout, Kin, lddy, ldw, ldy, lddx, act, _stream())


_BK = None


def gemm_bk():
    """k-tile of the GEMM family (csrc/gemm.h GEMM_BK): the launcher rounds split-K slices up to it."""
    global _BK
    if _BK is None:
        L = lib()
        L.pda_gemm_bk.restype = ctypes.c_int
        _BK = int(L.pda_gemm_bk())
    return _BK


def round_ksplit(ksplit):
    bk = gemm_bk()
    return (int(ksplit) + bk - 1) // bk * bk


def num_slices(red, ksplit):
    """Split-K slices a launch with this ksplit produces over a reduction of length red."""
    if ksplit <= 0:
        return 1
    ks = round_ksplit(ksplit)
    return (red + ks - 1) // ks


def linear_wgrad(dy, x, gw, gb, B, Nout, Kin, lddy, ldx, ldg, scale=1.0, ksplit=0, slice_stride=0, bslice_stride=0):
    """Slice y of gw[Nout][Kin] = dy[B][Nout]^T @ x[B][Kin] over its share of B -> gw + y*slice_stride,
    column sums of dy -> gb + y*bslice_stride (plain stores, GEMM layout).  ksplit=0: one slice."""
    assert lddy % 8 == 0 and ldx % 8 == 0
    assert ksplit == 0 or slice_stride >= Nout * ldg
    _L().pda_linear_wgrad(_ptr(dy), _ptr(x), _ptr(gw), _ptr(gb), B, Nout, Kin, lddy, ldx, ldg,
                          slice_stride, bslice_stride, float(scale), ksplit, _stream())


class LinearWgradPair(object):
    """Two single-slice linear weight gradients over the same batch as the two problem groups of one
    launch.  jobs: [(dy, x, gw, gb, Nout, Kin, lddy, ldx, ldg)] x 2."""

    def __init__(self, jobs, B):
        assert len(jobs) == 2
        self.keep, self.B = jobs, int(B)
        self.dy = (_vp * 2)(*[j[0].data_ptr() for j in jobs])
        self.x = (_vp * 2)(*[j[1].data_ptr() for j in jobs])
        self.gw = (_vp * 2)(*[j[2].data_ptr() for j in jobs])
        self.gb = (_vp * 2)(*[j[3].data_ptr() for j in jobs])
        self.nout = (ctypes.c_int * 2)(*[int(j[4]) for j in jobs])
        self.kin = (ctypes.c_int * 2)(*[int(j[5]) for j in jobs])
        self.lddy = (ctypes.c_long * 2)(*[int(j[6]) for j in jobs])
        self.ldx = (ctypes.c_long * 2)(*[int(j[7]) for j in jobs])
        self.ldg = (ctypes.c_long * 2)(*[int(j[8]) for j in jobs])
        for j in jobs:
            assert j[6] % 8 == 0 and j[7] % 8 == 0

    def run(self):
        _L().pda_linear_wgrad2(self.dy, self.x, self.gw, self.gb, self.B, self.nout, self.kin, self.lddy,
                               self.ldx, self.ldg, _stream())

    def run_with_dgrad(self, dy, w, y, dx, M, Nout, Kin, lddy, ldw, ldy, lddx, act=ACT_NONE):
        """The same two weight gradients plus linear_dgrad(dy, w, y, dx, ...) in ONE launch
        (csrc/gemm.h:gemm_duo_kernel): the three are independent and each under-fills the GPU."""
        L = _L()
        if not getattr(L, "_duo_ready", False):
            c_int, c_long = ctypes.c_int, ctypes.c_long
            L.pda_fc_backward_duo.argtypes = ([_vp] * 4 + [c_int] * 3 + [c_long] * 4 + [c_int] + [_vp] * 4 + [c_int] +
                                              [_vp] * 5 + [_vp])
            L.pda_fc_backward_duo.restype = None
            L._duo_ready = True
        assert lddy % 8 == 0 and ldw % 8 == 0
        L.pda_fc_backward_duo(_ptr(dy), _ptr(w), _ptr(y), _ptr(dx), M, Nout, Kin, lddy, ldw, ldy, lddx, act,
                              self.dy, self.x, self.gw, self.gb, self.B, self.nout, self.kin, self.lddy,
                              self.ldx, self.ldg, _stream())


def conv_wgrad(dy, x, gw, gb, batch, H, W, C, KH, KW, S, O, ksplit=0, slice_stride=0, bslice_stride=0):
    """gw[O][(ky,kx,c)] slices (see linear_wgrad); x is NHWC bf16."""
    assert ksplit == 0 or slice_stride >= O * C * KH * KW
    _L().pda_conv_wgrad(_ptr(dy), _ptr(x), _ptr(gw), _ptr(gb), batch, H, W, C, KH, KW, S, O,
                        slice_stride, bslice_stride, ksplit, _stream())


def conv_wgrad_pair(dys, xs, gws, gbs, batch, geoms, O, ksplit, slice_strides, bslice_strides):
    """Two conv weight gradients in one launch (csrc/gemm.hip:pda_conv_wgrad_pair).
    geoms: [(H, W, C, KH, KW, S)] x 2; both problems use the split-K slice length `ksplit`."""
    import ctypes as _ct
    L = _L()
    if not getattr(L, "_pair_ready", False):
        L.pda_conv_wgrad_pair.argtypes = [_vp] * 4 + [_ct.c_int] + [_vp] * 6 + [_ct.c_int, _vp, _vp, _ct.c_int, _vp]
        L.pda_conv_wgrad_pair.restype = None
        L._pair_ready = True
    cols = list(zip(*geoms))
    iarr = [(_ct.c_int * 2)(*[int(v) for v in col]) for col in cols]
    ss = (_ct.c_long * 2)(*[int(v) for v in slice_strides])
    bs = (_ct.c_long * 2)(*[int(v) for v in bslice_strides])
    L.pda_conv_wgrad_pair(_arr(dys), _arr(xs), _arr(gws), _arr(gbs), int(batch), iarr[0], iarr[1], iarr[2], iarr[3],
                          iarr[4], iarr[5], int(O), ss, bs, int(ksplit), _stream())


def conv_wgrad_u8(dy, frames, idx, gw, gb, batch, H, W, C, KH, KW, S, O, scale, ksplit=0, slice_stride=0,
                  bslice_stride=0):
    assert ksplit == 0 or slice_stride >= O * C * KH * KW
    if conv1_specialised(H, W, C, KH, KW, S, O):
        # one slice per workgroup (conv1_wgrad_slices(batch) of them); ksplit is not used
        assert conv1_wgrad_slices(batch) == 1 or slice_stride >= O * C * KH * KW
        (ref,), stride = _windows([frames])
        _L().pda_conv1_wgrad_u8(_ptr(dy), _ptr(frames), _ptr(idx), _ptr(gw), _ptr(gb), batch, H, W, C,
                                float(scale), slice_stride, bslice_stride, _ptr(ref), stride, _stream())
        return
    _no_windows(frames, "the generic conv1 weight gradient")
    _L().pda_conv_wgrad_u8(_ptr(dy), _ptr(frames), _ptr(idx), _ptr(gw), _ptr(gb), batch, H, W, C, KH, KW, S, O,
                           float(scale), slice_stride, bslice_stride, ksplit, _stream())


class ReducePlan(object):
    """Host-side argument block for pda_wgrad_reduce: dst[e] = sum_s slab[s*stride + e] for up to 8 jobs."""

    def __init__(self, jobs):
        """jobs: [(slab tensor, dst tensor, count, nslices, stride)]"""
        assert 0 < len(jobs) <= 8
        n = len(jobs)
        self.keep = jobs
        self.slabs = (_vp * n)(*[j[0].data_ptr() for j in jobs])
        self.dsts = (_vp * n)(*[j[1].data_ptr() for j in jobs])
        self.counts = (ctypes.c_int * n)(*[int(j[2]) for j in jobs])
        self.nslices = (ctypes.c_int * n)(*[int(j[3]) for j in jobs])
        self.strides = (ctypes.c_long * n)(*[int(j[4]) for j in jobs])
        self.n = n

    def run(self):
        _L().pda_wgrad_reduce(self.slabs, self.dsts, self.counts, self.nslices, self.strides, self.n, _stream())


def conv_dgrad(dy, w, yin, dx, batch, H, W, C, KH, KW, S, O, act=ACT_RELU):
    """dx[(b,iy,ix)][c] = conv data gradient of dy[(b,oy,ox)][o] through w[O][(ky,kx,c)], times act'(yin):
    a gather-form implicit GEMM per input-pixel parity class -- no dcol matrix, no col2im pass."""
    assert S in (1, 2) and KH % S == 0 and KW % S == 0 and H % S == 0 and W % S == 0 and C % 8 == 0 and O % 8 == 0
    _L().pda_conv_dgrad(_ptr(dy), _ptr(w), _ptr(yin), _ptr(dx), batch, H, W, C, KH, KW, S, O, act, _stream())


def col2im(dcol, yin, dx, batch, H, W, C, KH, KW, S, act=ACT_RELU):
    _L().pda_col2im(_ptr(dcol), _ptr(yin), _ptr(dx), batch, H, W, C, KH, KW, S, act, _stream())


def tr_probe():
    out = torch.zeros(64 * 8, dtype=torch.int32, device="cuda")
    _L().pda_tr_probe(_ptr(out), _stream())
    return out.cpu().reshape(64, 8)


# ---- weight layout helpers (torch ops; used at init and by tests) ---------------------------------
def pack_conv_weight(w):
    """fp32 [O, C, KH, KW] -> bf16 [O, KH*KW*C] in (ky, kx, c) order."""
    return w.detach().permute(0, 2, 3, 1).reshape(w.shape[0], -1).to(BF16).contiguous()


def pack_fc_after_conv(w, C, P):
    """fp32 [N, C*P] with columns (c, p) -> bf16 [N, P*C] with columns (p, c)."""
    N = w.shape[0]
    return w.detach().reshape(N, C, P).permute(0, 2, 1).reshape(N, P * C).to(BF16).contiguous()


def nhwc(x):
    """[B, C, H, W] -> bf16 [B*H*W, C]."""
    B, C, H, W = x.shape
    return x.permute(0, 2, 3, 1).reshape(B * H * W, C).to(BF16).contiguous()


def from_nhwc(x, B, H, W, C):
    return x.float().reshape(B, H, W, C).permute(0, 3, 1, 2).contiguous()


# ---- fp32-precision ("x3") family: split-bf16 operands, csrc/gemm_x3.hip ------------------------------------
class Planes(object):
    """A logical fp32 tensor kept as two bf16 planes: hi = bf16(x) in full[0], lo = bf16(x - hi) in full[1]
    (16 significand bits together).  `plane` = elements from hi to lo, as the kernels take it."""

    def __init__(self, *shape, device=None, split=True):
        self.full = torch.zeros((2 if split else 1,) + tuple(shape), dtype=BF16, device=device)
        self.hi = self.full[0]
        self.plane = int(self.full.stride(0)) if split else 0

    @classmethod
    def from_float(cls, x):
        """Split an fp32 tensor (tests, weight packing)."""
        out = cls(*x.shape, device=x.device)
        hi = x.to(BF16)
        out.full[0].copy_(hi)
        out.full[1].copy_((x - hi.float()).to(BF16))
        return out

    def data_ptr(self):
        return self.hi.data_ptr()

    def value(self):
        """fp32 value the planes stand for."""
        if self.full.shape[0] == 1:
            return self.hi.float()
        return self.full[0].float() + self.full[1].float()

    def zero_(self):
        self.full.zero_()
        return self


def _plane(t):
    return int(getattr(t, "plane", 0))


_x3_ready = False


def _X():
    global _x3_ready
    L = _L()
    if not _x3_ready:
        c_int, c_long, c_float = ctypes.c_int, ctypes.c_long, ctypes.c_float
        L.pda_x3_conv_fwd_u8.argtypes = [_vp] * 3 + [c_long, _vp, _vp, c_long] + [c_int] * 9 + [c_float, c_int, _vp]
        L.pda_x3_conv_fwd.argtypes = [_vp, c_long, _vp, c_long, _vp, _vp, c_long] + [c_int] * 10 + [_vp]
        L.pda_x3_linear_fwd.argtypes = [_vp, c_long, _vp, c_long, _vp] + [c_int] * 4 + [c_long] * 4 + [c_int, _vp]
        L.pda_x3_linear_dgrad.argtypes = ([_vp, c_long, _vp, c_long, _vp, _vp, c_long] + [c_int] * 3 + [c_long] * 4 +
                                          [c_int, _vp])
        L.pda_x3_conv_dgrad.argtypes = [_vp, c_long, _vp, c_long, _vp, _vp, c_long] + [c_int] * 9 + [_vp]
        L.pda_x3_linear_wgrad2.argtypes = [_vp] * 6 + [c_int, c_int] + [_vp] * 5 + [_vp]
        L.pda_x3_fc_backward_duo.argtypes = ([_vp, c_long, _vp, c_long, _vp, _vp, c_long] + [c_int] * 3 + [c_long] * 4 +
                                             [c_int] + [_vp] * 6 + [c_int] + [_vp] * 5 + [_vp])
        L.pda_x3_conv_wgrad.argtypes = [_vp] * 6 + [c_int, c_int] + [_vp] * 6 + [c_int, _vp, _vp, c_int, _vp]
        L.pda_x3_conv_wgrad_u8.argtypes = ([_vp, c_long, _vp, _vp, _vp, _vp] + [c_int] * 8 +
                                           [c_float, c_long, c_long, c_int, _vp])
        L.pda_cnn_fwd_stack_x3.argtypes = ([_vp] * 3 + [c_long] + [_vp] * 6 +
                                           [c_long, _vp, c_long, _vp, c_long, _vp, c_long] + [c_int] * 5 + [c_float, _vp])
        L.pda_conv1_fwd_u8_x3_w.argtypes = ([_vp] * 3 + [c_long, _vp, c_long, _vp, _vp, c_long] + [c_int] * 5 +
                                            [c_float, c_int, _vp])
        L.pda_conv1_fwd_u8_x3_w.restype = None
        L.pda_cnn_fwd_stack_x3.restype = None
        L.pda_cnn_dgrad_stack_x3.argtypes = ([_vp, c_long, _vp, _vp, _vp, _vp, c_long, _vp, c_long, _vp, c_long, c_int] +
                                             [_vp] * 4 + [c_long, c_long, c_float, _vp, c_long, _vp])
        L.pda_cnn_dgrad_stack_x3.restype = None
        for name in ("pda_x3_conv_fwd_u8", "pda_x3_conv_fwd", "pda_x3_linear_fwd", "pda_x3_linear_dgrad",
                     "pda_x3_conv_dgrad", "pda_x3_linear_wgrad2", "pda_x3_fc_backward_duo", "pda_x3_conv_wgrad",
                     "pda_x3_conv_wgrad_u8"):
            getattr(L, name).restype = None
        _x3_ready = True
    return L


def _same_plane(ts):
    planes = {_plane(t) for t in ts if t is not None}
    assert len(planes) == 1, "problem groups of one launch share one plane offset"
    return planes.pop()


def _longs(vals):
    return (ctypes.c_long * len(vals))(*[int(v) for v in vals])


def _ints(vals):
    return (ctypes.c_int * len(vals))(*[int(v) for v in vals])


def x3_conv_fwd_u8(frames, idx, w, bias, out, batch, H, W, C, KH, KW, S, O, scale, act=ACT_RELU):
    """conv_fwd_u8 with split weights (Planes) and split outputs; frames are exact."""
    assert KW == 8 and len(frames) <= GEMM_MAX_GROUPS and (C * KH * KW) % 8 == 0
    refs, stride = _windows(frames)
    if stride:
        if not conv1_specialised(H, W, C, KH, KW, S, O):
            _no_windows(frames, "the generic conv1 forward")
        _X().pda_conv1_fwd_u8_x3_w(_arr(frames), _arr(idx), _arr(refs), stride, _arr(w), _same_plane(w), _arr(bias),
                                   _arr(out), _same_plane(out), len(frames), batch, H, W, C, float(scale), act, _stream())
        return
    _X().pda_x3_conv_fwd_u8(_arr(frames), _arr(idx), _arr(w), _same_plane(w), _arr(bias), _arr(out), _same_plane(out),
                            len(frames), batch, H, W, C, KH, KW, S, O, float(scale), act, _stream())


def x3_cnn_fwd_stack_supported(H, W, C):
    """True when the one-launch fp32-precision conv stack (cnn_fwd_stack_x3_kernel) covers this input."""
    return cnn_fwd_stack_supported(H, W, C)


def x3_cnn_fwd_stack(frames, idx, w1, b1, w2, b2, w3, b3, a1, a2, a3, batch, H, W, C, scale):
    """cnn_fwd_stack on split planes: weights (all of one network's shadow buffer: one plane offset) and activations
    as Planes; a1 / a2 entries may be None."""
    n = len(frames)
    assert n <= GEMM_MAX_GROUPS and H == 84 and W == 84 and C == 4
    wp = _same_plane(list(w1) + list(w2) + list(w3))
    kept1, kept2 = [t for t in a1 if t is not None], [t for t in a2 if t is not None]
    refs, stride = _windows(frames)
    _X().pda_cnn_fwd_stack_x3(_arr(frames), _arr(idx), _arr(refs), stride, _arr(w1), _arr(b1), _arr(w2), _arr(b2),
                              _arr(w3), _arr(b3), wp,
                              _arr(a1), _same_plane(kept1) if kept1 else 0, _arr(a2),
                              _same_plane(kept2) if kept2 else 0, _arr(a3), _same_plane(a3), n, batch, H, W, C,
                              float(scale), _stream())


def x3_cnn_dgrad_stack_supported(H, W, C):
    return not FORCE_GENERIC_CONV1 and bool(_L().pda_cnn_dgrad_stack_supported(H, W, C))


def x3_cnn_dgrad_stack_slices(batch):
    """conv1 weight-gradient slab slices x3_cnn_dgrad_stack(..., w1=...) writes: one per workgroup."""
    L = _L()
    L.pda_cnn_dgrad_stack_x3_slices.argtypes = [ctypes.c_int]
    L.pda_cnn_dgrad_stack_x3_slices.restype = ctypes.c_int
    return int(L.pda_cnn_dgrad_stack_x3_slices(int(batch)))


def x3_cnn_dgrad_stack(da3, a2_hi, a1_hi, w3t, w2t, da2, da1, batch, w1=None):
    """da3 -> da2 -> da1 on split planes in one launch (cnn_dgrad_stack_x3_kernel): da* as Planes, the weights as the
    TRANSPOSED split shadows [(tap, c)][o] the optimizer keeps (CnnWeights.wT; both in one shadow buffer: one plane
    offset), a2 / a1 the hi planes of the forward activations.
    w1 = (frames uint8, idx int32 or None, gw slab [slices][32*256], gb slab [slices][32], scale), slices =
    x3_cnn_dgrad_stack_slices(batch): the same launch also leaves conv1's weight / bias gradient, one slab slice per
    workgroup (two workgroups share an image up to 256 images; above, 256 workgroups walk the batch)."""
    assert _plane(w3t) == _plane(w2t)
    frames, idx, gw, gb, scale = w1 if w1 is not None else (None, None, None, None, 1.0)
    (ref,), stride = _windows([frames])
    _X().pda_cnn_dgrad_stack_x3(_ptr(da3), _plane(da3), _ptr(a2_hi), _ptr(a1_hi), _ptr(w3t), _ptr(w2t), _plane(w3t),
                                _ptr(da2), _plane(da2), _ptr(da1), _plane(da1), batch, _ptr(frames), _ptr(idx),
                                _ptr(gw), _ptr(gb), 32 * 256, 32, float(scale), _ptr(ref), stride, _stream())


def x3_conv_fwd(x, w, bias, out, batch, H, W, C, KH, KW, S, O, act=ACT_RELU):
    assert C % 8 == 0 and len(x) <= GEMM_MAX_GROUPS
    _X().pda_x3_conv_fwd(_arr(x), _same_plane(x), _arr(w), _same_plane(w), _arr(bias), _arr(out), _same_plane(out),
                         len(x), batch, H, W, C, KH, KW, S, O, act, _stream())


def x3_linear_fwd(x, w, out32, M, N, K, ldx, ldw, ldo32, ksplit=0, slice_stride=0):
    """Split-K partial sums over the 3 K reduction (ksplit counts it): atomically added into the fp32 out32
    (pre-zeroed), or -- slice_stride > 0 -- written as slabs, slice y at out32 + y * slice_stride."""
    assert K % 8 == 0 and ldx % 8 == 0 and ldw % 8 == 0 and len(x) <= GEMM_MAX_GROUPS
    _X().pda_x3_linear_fwd(_arr(x), _same_plane(x), _arr(w), _same_plane(w), _arr(out32), len(x), M, N, K, ldx, ldw,
                           ldo32, int(slice_stride), int(ksplit), _stream())


def x3_linear_dgrad(dy, w, y, dx, M, Nout, Kin, lddy, ldw, ldy, lddx, act=ACT_NONE):
    assert Nout % 8 == 0 and lddy % 8 == 0 and ldw % 8 == 0
    _X().pda_x3_linear_dgrad(_ptr(dy), _plane(dy), _ptr(w), _plane(w), _ptr(y), _ptr(dx), _plane(dx), M, Nout, Kin,
                             lddy, ldw, ldy, lddx, act, _stream())


def x3_conv_dgrad(dy, w, yin, dx, batch, H, W, C, KH, KW, S, O, act=ACT_RELU):
    assert S in (1, 2) and KH % S == 0 and KW % S == 0 and H % S == 0 and W % S == 0 and C % 8 == 0 and O % 8 == 0
    _X().pda_x3_conv_dgrad(_ptr(dy), _plane(dy), _ptr(w), _plane(w), _ptr(yin), _ptr(dx), _plane(dx), batch, H, W, C,
                           KH, KW, S, O, act, _stream())


class X3LinearWgradPair(object):
    """LinearWgradPair on split operands.  jobs: [(dy, x, gw, gb, Nout, Kin, lddy, ldx, ldg)] x 2 (dy, x: Planes)."""

    def __init__(self, jobs, B):
        assert len(jobs) == 2
        self.keep, self.B = jobs, int(B)
        self.dy = (_vp * 2)(*[j[0].data_ptr() for j in jobs])
        self.dyp = _longs([_plane(j[0]) for j in jobs])
        self.x = (_vp * 2)(*[j[1].data_ptr() for j in jobs])
        self.xp = _longs([_plane(j[1]) for j in jobs])
        self.gw = (_vp * 2)(*[j[2].data_ptr() for j in jobs])
        self.gb = (_vp * 2)(*[j[3].data_ptr() for j in jobs])
        self.nout, self.kin = _ints([j[4] for j in jobs]), _ints([j[5] for j in jobs])
        self.lddy, self.ldx, self.ldg = (_longs([j[i] for j in jobs]) for i in (6, 7, 8))
        for j in jobs:
            assert j[6] % 8 == 0 and j[7] % 8 == 0

    def run(self):
        _X().pda_x3_linear_wgrad2(self.dy, self.dyp, self.x, self.xp, self.gw, self.gb, 2, self.B, self.nout,
                                  self.kin, self.lddy, self.ldx, self.ldg, _stream())

    def run_with_dgrad(self, dy, w, y, dx, M, Nout, Kin, lddy, ldw, ldy, lddx, act=ACT_NONE):
        assert Nout % 8 == 0 and lddy % 8 == 0 and ldw % 8 == 0
        _X().pda_x3_fc_backward_duo(_ptr(dy), _plane(dy), _ptr(w), _plane(w), _ptr(y), _ptr(dx), _plane(dx), M, Nout,
                                    Kin, lddy, ldw, ldy, lddx, act, self.dy, self.dyp, self.x, self.xp, self.gw,
                                    self.gb, self.B, self.nout, self.kin, self.lddy, self.ldx, self.ldg, _stream())


def x3_conv_wgrad(dys, xs, gws, gbs, batch, geoms, O, ksplit, slice_strides, bslice_strides):
    """One or two conv weight gradients (split operands) in one launch; the reduction is 3 * batch * OH * OW and
    `ksplit` counts that extended reduction.  geoms: [(H, W, C, KH, KW, S)]."""
    n = len(dys)
    cols = list(zip(*geoms))
    iarr = [_ints(col) for col in cols]
    _X().pda_x3_conv_wgrad(_arr(dys), _longs([_plane(t) for t in dys]), _arr(xs), _longs([_plane(t) for t in xs]),
                           _arr(gws), _arr(gbs), n, int(batch), iarr[0], iarr[1], iarr[2], iarr[3], iarr[4], iarr[5],
                           int(O), _longs(slice_strides), _longs(bslice_strides), int(ksplit), _stream())


def x3_conv1_wgrad_u8_whole(dy, frames, idx, gw, gb, batch, H, W, C, scale, slice_stride, bslice_stride):
    """conv1 weight gradient (dy split, frames exact) by the whole-image kernel: conv1_wgrad_slices(batch) slab slices."""
    L = _L()
    L.pda_conv1_wgrad_u8_x3.argtypes = [_vp, ctypes.c_long, _vp, _vp, _vp, _vp] + [ctypes.c_int] * 4 + \
        [ctypes.c_float, ctypes.c_long, ctypes.c_long, _vp, ctypes.c_long, _vp]
    L.pda_conv1_wgrad_u8_x3.restype = None
    (ref,), stride = _windows([frames])
    L.pda_conv1_wgrad_u8_x3(_ptr(dy), _plane(dy), _ptr(frames), _ptr(idx), _ptr(gw), _ptr(gb), batch, H, W, C,
                            float(scale), slice_stride, bslice_stride, _ptr(ref), stride, _stream())


def x3_conv_wgrad_u8(dy, frames, idx, gw, gb, batch, H, W, C, KH, KW, S, O, scale, ksplit=0, slice_stride=0,
                     bslice_stride=0):
    """conv1 weight gradient: dy split, frames exact (generic implicit GEMM, split-K slabs)."""
    assert ksplit == 0 or slice_stride >= O * C * KH * KW
    _no_windows(frames, "the generic conv1 weight gradient")
    _X().pda_x3_conv_wgrad_u8(_ptr(dy), _plane(dy), _ptr(frames), _ptr(idx), _ptr(gw), _ptr(gb), batch, H, W, C, KH,
                              KW, S, O, float(scale), slice_stride, bslice_stride, int(ksplit), _stream())
