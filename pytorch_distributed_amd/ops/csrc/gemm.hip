// C-ABI entry points for the MFMA GEMM family (see gemm.h for the kernel).
// Every function takes raw device pointers and a hipStream_t and only enqueues
// work -- no allocation, no synchronisation -- so callers can capture whole
// learner steps into a hipGraph.
#include "gemm.h"
#include "api.h"
#include "optim.h"
#include <cstdlib>

namespace {

// fewer 64x64 tiles than CUs on the chip: halve the tile so that twice the workgroups share the work
bool small_grid(int M, int N, int groups) {
    long tiles = (long)((M + 63) / 64) * ((N + 63) / 64) * groups;
    return tiles < 224;
}

int g_force_generic_conv1 = 0;
int g_tile_mode = 0;

}  // namespace

// Large problems (batch 4096: fc1 forward 8192 x 512 x 3136) run on 128-row tiles.  A 64 x 64 launch is bound by the
// rate at which a CU pulls operand bytes through its L1 (822 MB in 66 us = 12.4 TB/s over the chip for that forward,
// and the split-bf16 launch moves the same bytes per second): the way up is fewer operand bytes per FLOP, half of them
// per doubling of the tile side.  Measured at batch 4096 (tools/probes/gemm_tiles_bench.py, profiles/r3_gemm_tiles.txt):
//   fc1 forward    bf16 66 -> 47 us (128 x 128, k-tile 32; k-tile 64 leaves one wave per SIMD: 52 us), two k-slices 43 us
//                  split-bf16 136 -> 99 us (128 x 128, k-tile 32: four planes of a k-tile in 64 KB), two slices 92 us
//   fc1 data grad  bf16 56 -> 61 / 92 us: kept on 64 x 64 (8 k-tiles per workgroup, then a 2-byte-per-lane epilogue
//                  that nothing overlaps at one or two workgroups per CU); split-bf16 103 -> 95 us on 128 x 64
// pda_gemm_tile_mode (test / measurement hook): 0 = the rule below, 1 = 64 x 64 tiles only, 2 / 3 = a large-tile
// variant whenever both sides reach 128 (bf16: 128 x 128; split operands: 2 = 128 x 128, 3 = 128 x 64).
PDA_EXPORT void pda_gemm_tile_mode(int mode) { g_tile_mode = mode; }
// -> 0: 64 x 64 tiles; otherwise the large-tile variant to use.  best = the call site's measured best variant (0: none),
// taken when its 128 x 128 grid still gives nearly every CU a workgroup.
PDA_EXPORT int pda_gemm_large_tiles(int M, int N, int units, int best) {
    if (g_tile_mode == 1 || M < 128 || N < 128) return 0;
    if (g_tile_mode >= 2) return g_tile_mode;
    long tiles = (long)((M + 127) / 128) * ((N + 127) / 128) * units;
    return tiles >= 224 ? best : 0;
}

// test hook (tests/test_gpu_gemm.py, ops/nn.py:force_generic_conv1): run the Nature conv1 geometry through the generic
// implicit-GEMM loaders instead of the whole-image kernels
PDA_EXPORT void pda_force_generic_conv1(int on) { g_force_generic_conv1 = on; }
PDA_EXPORT int pda_generic_conv1_forced() { return g_force_generic_conv1; }

// ---- conv1 forward: uint8 planar frames (optionally gathered through idx) ----------
// out[g][(b,oy,ox)][o] = act(scale * sum_k u8 * W[o][k] + bias[o]),  k = (c,ky,kx)
PDA_EXPORT void pda_conv_fwd_u8(const void* const* frames, const void* const* idx, const void* const* w,
                                const void* const* bias, void* const* out, int ngroups, int batch,
                                int H, int W, int C, int KH, int KW, int S, int O, float scale, int act,
                                hipStream_t stream) {
    // Nature conv1 geometry: the whole-image kernel (nature_conv1.hip), unless the test hook keeps this loader
    const bool generic_only = g_force_generic_conv1 != 0;
    if (!generic_only && ngroups <= GEMM_MAX_GROUPS && pda_conv1_supported(H, W, C, KH, KW, S, O)) {
        pda_conv1_fwd_u8(frames, idx, w, bias, out, ngroups, batch, H, W, C, scale, act, stream);
        return;
    }
    ConvGeom g = make_geom(H, W, C, KH, KW, S);
    int M = batch * g.OH * g.OW, K = C * KH * KW;
    check_fastdiv((long)M + 4 * GEMM_BK, (long)g.OH * g.OW, "conv rows");
    GemmArgs<SrcFrameU8K, SrcDenseK, EpiBiasAct> a;
    for (int i = 0; i < ngroups; ++i) {
        a.g[i].a = {(const uint8_t*)frames[i], (const int*)idx[i], (long)C * H * W, g, M, K};
        a.g[i].b = {(const bf16_t*)w[i], (long)K, O, K};
        a.g[i].e = {(bf16_t*)out[i], nullptr, (const float*)bias[i], scale, act, (long)O, 0};
        a.g[i].M = M; a.g[i].N = O; a.g[i].K = K;
    }
    if (O <= 32) launch_gemm<64, 32, 4, 1>(a, ngroups, M, O, K, 0, stream);
    else launch_gemm<64, 64, 2, 2>(a, ngroups, M, O, K, 0, stream);
}

// ---- conv forward on NHWC bf16 activations; weights [O][(ky,kx,c)] bf16 ------------------
PDA_EXPORT void pda_conv_fwd(const void* const* x, const void* const* w, const void* const* bias,
                             void* const* out, int ngroups, int batch, int H, int W, int C, int KH, int KW,
                             int S, int O, int act, hipStream_t stream) {
    ConvGeom g = make_geom(H, W, C, KH, KW, S);
    int M = batch * g.OH * g.OW, K = C * KH * KW;
    check_fastdiv((long)M + 4 * GEMM_BK, (long)g.OH * g.OW, "conv rows");
    GemmArgs<SrcIm2colK, SrcDenseK, EpiBiasAct> a;
    for (int i = 0; i < ngroups; ++i) {
        a.g[i].a = {(const bf16_t*)x[i], g, M, K};
        a.g[i].b = {(const bf16_t*)w[i], (long)K, O, K};
        a.g[i].e = {(bf16_t*)out[i], nullptr, (const float*)bias[i], 1.f, act, (long)O, 0};
        a.g[i].M = M; a.g[i].N = O; a.g[i].K = K;
    }
    if (O <= 32) launch_gemm<64, 32, 4, 1>(a, ngroups, M, O, K, 0, stream);
    else if (small_grid(M, O, ngroups)) launch_gemm<32, 64, 1, 4>(a, ngroups, M, O, K, 0, stream);
    else launch_gemm<64, 64, 2, 2>(a, ngroups, M, O, K, 0, stream);
}

// ---- linear forward: out = act(x[M][K] * w[N][K]^T + bias) ---------------------------------
// ksplit > 0: split-K partial sums are atomically added into out32 (pre-zeroed, no bias/act;
// the consumer applies them).  Otherwise bf16 and/or fp32 outputs with bias + act.
PDA_EXPORT void pda_linear_fwd(const void* const* x, const void* const* w, const void* const* bias,
                               void* const* out, void* const* out32, int ngroups, int M, int N, int K,
                               long ldx, long ldw, long ldo, long ldo32, int act, int ksplit, hipStream_t stream) {
    if (ksplit > 0) {
        GemmArgs<SrcDenseK, SrcDenseK, EpiAtomicF32> a;
        for (int i = 0; i < ngroups; ++i) {
            a.g[i].a = {(const bf16_t*)x[i], ldx, M, K};
            a.g[i].b = {(const bf16_t*)w[i], ldw, N, K};
            a.g[i].e = {(float*)out32[i], ldo32, 1.f};
            a.g[i].M = M; a.g[i].N = N; a.g[i].K = K;
        }
        launch_gemm<64, 64, 2, 2>(a, ngroups, M, N, K, ksplit, stream);
        return;
    }
    GemmArgs<SrcDenseK, SrcDenseK, EpiBiasAct> a;
    for (int i = 0; i < ngroups; ++i) {
        a.g[i].a = {(const bf16_t*)x[i], ldx, M, K};
        a.g[i].b = {(const bf16_t*)w[i], ldw, N, K};
        a.g[i].e = {(bf16_t*)(out ? out[i] : nullptr), (float*)(out32 ? out32[i] : nullptr),
                    (const float*)(bias ? bias[i] : nullptr), 1.f, act, ldo, ldo32};
        a.g[i].M = M; a.g[i].N = N; a.g[i].K = K;
    }
    if (M <= 32) launch_gemm<32, 64, 1, 4>(a, ngroups, M, N, K, 0, stream);
    else launch_gemm<64, 64, 2, 2>(a, ngroups, M, N, K, 0, stream);
}

// ---- linear forward, split-K into slabs: slice y of group g -> out32[g] + y * slice_stride (plain stores) --------
// For fc1 of the learner step: the head kernel adds the slices in order (bit-reproducible, no atomics, nothing to
// zero).  Workgroups run in XCD-aware order: all tiles of one (slice, group) unit share one XCD's L2, so the launch
// fetches its operand rows from HBM once instead of once per XCD (plain order: 18.5 MB per launch at 23 % L2 hits).
PDA_EXPORT void pda_linear_fwd_slab(const void* const* x, const void* const* w, void* const* out32, int ngroups,
                                    int M, int N, int K, long ldx, long ldw, long ldo32, long slice_stride,
                                    int ksplit, hipStream_t stream) {
    GemmArgs<SrcDenseK, SrcDenseK, EpiSlabF32> a;
    for (int i = 0; i < ngroups; ++i) {
        a.g[i].a = {(const bf16_t*)x[i], ldx, M, K};
        a.g[i].b = {(const bf16_t*)w[i], ldw, N, K};
        a.g[i].e = {(float*)out32[i], ldo32, slice_stride};
        a.g[i].M = M; a.g[i].N = N; a.g[i].K = K;
    }
    int3 grid = plan_gemm<64, 64>(a, ngroups, M, N, K, ksplit > 0 ? ksplit : K);
    int units = grid.y * grid.z;
    if (units < 8) {    // fewer (slice, group) units than XCDs (large batches: one slice): the unit-per-XCD order would
        // leave XCDs idle (batch 4096: 196 us on 2 of the 8 XCDs against 70 us in plain order)
        if (pda_gemm_large_tiles(M, N, units, 2))
            launch_gemm<128, 128, 2, 2, SrcDenseK, SrcDenseK, EpiSlabF32, 32>(a, ngroups, M, N, K, ksplit > 0 ? ksplit : K, stream);
        else
            launch_gemm<64, 64, 2, 2>(a, ngroups, M, N, K, ksplit > 0 ? ksplit : K, stream);
        return;
    }
    hipLaunchKernelGGL((gemm_xcd_kernel<64, 64, 2, 2, GEMM_BK, SrcDenseK, SrcDenseK, EpiSlabF32>),
                       dim3((units + 7) / 8 * 8 * grid.x), dim3(GEMM_THREADS), 0, stream, a, grid);
}

// ---- data gradient of a linear map: dx[M][Kin] = (dy[M][Nout] * w[Nout][Kin]) * act'(y) ------
// Also the "dcol" half of a conv data gradient (y == null: no mask).
PDA_EXPORT void pda_linear_dgrad(const void* dy, const void* w, const void* y, void* dx, int M, int Nout, int Kin,
                                 long lddy, long ldw, long ldy, long lddx, int act, hipStream_t stream) {
    GemmArgs<SrcDenseK, SrcDenseT, EpiMaskBf16> a;
    a.g[0].a = {(const bf16_t*)dy, lddy, M, Nout};
    a.g[0].b = {(const bf16_t*)w, ldw, Kin, Nout};          // element(row=kin, red=nout) = w[nout*ldw + kin]
    a.g[0].e = {(bf16_t*)dx, (const bf16_t*)y, y ? act : ACT_NONE, lddx, ldy};
    a.g[0].M = M; a.g[0].N = Kin; a.g[0].K = Nout;
    // under-filled launches (fc1 dgrad at batch 128: 98 tiles of 64x64 for 256 CUs) take the 32-row tile
    if (pda_gemm_large_tiles(M, Kin, 1, 0))      // (not chosen by the size rule: see the measurements above)
        launch_gemm<128, 128, 2, 2, SrcDenseK, SrcDenseT, EpiMaskBf16, 32>(a, 1, M, Kin, Nout, 0, stream);
    else if (M <= 32 || small_grid(M, Kin, 1)) launch_gemm<32, 64, 1, 4>(a, 1, M, Kin, Nout, 0, stream);
    else launch_gemm<64, 64, 2, 2>(a, 1, M, Kin, Nout, 0, stream);
}

// ---- conv data gradient, gather form: dx[b,iy,ix,c] (NHWC bf16) from dy (NHWC bf16) and the forward
// weight shadow w[O][(ky,kx,c)]; times act'(y_in).  One problem group per input-pixel parity class
// (S*S groups, S <= 2), no intermediate dcol matrix and no col2im pass.  Needs KH % S == KW % S == 0,
// H % S == W % S == 0, O % 8 == 0, C % 8 == 0.
PDA_EXPORT void pda_conv_dgrad(const void* dy, const void* w, const void* yin, void* dx, int batch, int H, int W,
                               int C, int KH, int KW, int S, int O, int act, hipStream_t stream) {
    ConvGeom cg = make_geom(H, W, C, KH, KW, S);
    GemmArgs<SrcDgradDyK, SrcDgradWT, EpiDgradScatter> a;
    int ng = 0, maxM = 0;
    int K = (KH / S) * (KW / S) * O;
    for (int py = 0; py < S; ++py)
        for (int px = 0; px < S; ++px) {
            DgradGeom g = make_dgrad_geom(cg, O, py, px);
            int M = batch * g.HC * g.WC;
            check_fastdiv((long)M + 64, (long)g.HC * g.WC, "conv dgrad rows");
            check_fastdiv((long)K + 4 * GEMM_BK, O, "conv dgrad reduction");
            a.g[ng].a = {(const bf16_t*)dy, g, M, K};
            a.g[ng].b = {(const bf16_t*)w, g, C, K};
            a.g[ng].e = {(bf16_t*)dx, (const bf16_t*)yin, yin ? act : ACT_NONE, g};
            a.g[ng].M = M; a.g[ng].N = C; a.g[ng].K = K;
            maxM = M > maxM ? M : maxM;
            ++ng;
        }
    if (C <= 32) launch_gemm<64, 32, 4, 1>(a, ng, maxM, C, K, 0, stream);
    else launch_gemm<64, 64, 2, 2>(a, ng, maxM, C, K, 0, stream);
}

// ---- weight gradients -----------------------------------------------------------------------------
// All three write split-K partial tiles with plain stores in GEMM layout: slice y goes to
// gw + y*slice_stride (+ bias column sums to gb + y*bslice_stride).  With one slice gw/gb can be
// the gradient buffer itself; otherwise they are scratch slabs summed by pda_wgrad_reduce.
// linear: gw[Nout][Kin] = dy[B][Nout]^T x[B][Kin]
PDA_EXPORT void pda_linear_wgrad(const void* dy, const void* x, void* gw, void* gb, int B, int Nout, int Kin,
                                 long lddy, long ldx, long ldg, long slice_stride, long bslice_stride, float scale,
                                 int ksplit, hipStream_t stream) {
    GemmArgs<SrcDenseT, SrcDenseT, EpiWgrad> a;
    a.g[0].a = {(const bf16_t*)dy, lddy, Nout, B};
    a.g[0].b = {(const bf16_t*)x, ldx, Kin, B};
    a.g[0].e = {(float*)gw, (float*)gb, ldg, slice_stride, bslice_stride, scale};
    a.g[0].M = Nout; a.g[0].N = Kin; a.g[0].K = B;
    if (Nout <= 32) launch_gemm<32, 64, 1, 4>(a, 1, Nout, Kin, B, ksplit, stream);
    else launch_gemm<64, 64, 2, 2>(a, 1, Nout, Kin, B, ksplit, stream);
}

// two linear weight gradients over the same batch in ONE launch (problem groups; single slice each):
// at batch 128 the fc2 gradient is a 4 us launch of its own for a few kFLOP
PDA_EXPORT void pda_linear_wgrad2(const void* const* dy, const void* const* x, void* const* gw, void* const* gb,
                                  int B, const int* Nout, const int* Kin, const long* lddy, const long* ldx,
                                  const long* ldg, hipStream_t stream) {
    GemmArgs<SrcDenseT, SrcDenseT, EpiWgrad> a;
    int maxM = 0, maxN = 0;
    for (int i = 0; i < 2; ++i) {
        a.g[i].a = {(const bf16_t*)dy[i], lddy[i], Nout[i], B};
        a.g[i].b = {(const bf16_t*)x[i], ldx[i], Kin[i], B};
        a.g[i].e = {(float*)gw[i], (float*)gb[i], ldg[i], 0, 0, 1.f};
        a.g[i].M = Nout[i]; a.g[i].N = Kin[i]; a.g[i].K = B;
        maxM = Nout[i] > maxM ? Nout[i] : maxM;
        maxN = Kin[i] > maxN ? Kin[i] : maxN;
    }
    launch_gemm<64, 64, 2, 2>(a, 2, maxM, maxN, B, 0, stream);
}

// every weight gradient of an MLP step in ONE launch: n <= GEMM_MAX_GROUPS problems of different shapes over
// their own row ranges (B[i] rows), single slice each.  The grid is sized for the largest; the others' surplus
// workgroups leave at once.  (mlp_engine.py: six launches of a few kFLOP each -> one.)
PDA_EXPORT void pda_linear_wgrad_n(int n, const void* const* dy, const void* const* x, void* const* gw,
                                   void* const* gb, const int* B, const int* Nout, const int* Kin,
                                   const long* lddy, const long* ldx, const long* ldg, hipStream_t stream) {
    if (n <= 0 || n > GEMM_MAX_GROUPS) return;
    GemmArgs<SrcDenseT, SrcDenseT, EpiWgrad> a;
    int maxM = 0, maxN = 0, maxK = 0;
    for (int i = 0; i < n; ++i) {
        a.g[i].a = {(const bf16_t*)dy[i], lddy[i], Nout[i], B[i]};
        a.g[i].b = {(const bf16_t*)x[i], ldx[i], Kin[i], B[i]};
        a.g[i].e = {(float*)gw[i], (float*)gb[i], ldg[i], 0, 0, 1.f};
        a.g[i].M = Nout[i]; a.g[i].N = Kin[i]; a.g[i].K = B[i];
        maxM = Nout[i] > maxM ? Nout[i] : maxM;
        maxN = Kin[i] > maxN ? Kin[i] : maxN;
        maxK = B[i] > maxK ? B[i] : maxK;
    }
    launch_gemm<64, 64, 2, 2>(a, n, maxM, maxN, maxK, 0, stream);
}

// fc backward in one launch: the data gradient of one linear layer (pda_linear_dgrad's problem, 32-row tiles) and the
// two single-slice weight gradients of pda_linear_wgrad2 -- independent of each other, all ready after the head.
PDA_EXPORT void pda_fc_backward_duo(const void* d_dy, const void* d_w, const void* d_y, void* d_dx, int M, int Nout,
                                    int Kin, long lddy, long ldw, long ldy, long lddx, int act,
                                    const void* const* dy, const void* const* x, void* const* gw, void* const* gb,
                                    int B, const int* wNout, const int* wKin, const long* wlddy, const long* wldx,
                                    const long* wldg, hipStream_t stream) {
    GemmArgs<SrcDenseK, SrcDenseT, EpiMaskBf16> a1;
    a1.g[0].a = {(const bf16_t*)d_dy, lddy, M, Nout};
    a1.g[0].b = {(const bf16_t*)d_w, ldw, Kin, Nout};
    a1.g[0].e = {(bf16_t*)d_dx, (const bf16_t*)d_y, d_y ? act : ACT_NONE, lddx, ldy};
    a1.g[0].M = M; a1.g[0].N = Kin; a1.g[0].K = Nout;
    int3 g1 = plan_gemm<32, 64>(a1, 1, M, Kin, Nout, 0);
    GemmArgs<SrcDenseT, SrcDenseT, EpiWgrad> a2;
    int maxM = 0, maxN = 0;
    for (int i = 0; i < 2; ++i) {
        a2.g[i].a = {(const bf16_t*)dy[i], wlddy[i], wNout[i], B};
        a2.g[i].b = {(const bf16_t*)x[i], wldx[i], wKin[i], B};
        a2.g[i].e = {(float*)gw[i], (float*)gb[i], wldg[i], 0, 0, 1.f};
        a2.g[i].M = wNout[i]; a2.g[i].N = wKin[i]; a2.g[i].K = B;
        maxM = wNout[i] > maxM ? wNout[i] : maxM;
        maxN = wKin[i] > maxN ? wKin[i] : maxN;
    }
    int3 g2 = plan_gemm<64, 64>(a2, 2, maxM, maxN, B, 0);
    // XCD-aware tile order: measured at batch 128, 12.55k vs 12.40k updates/s for the plain order
    const int xcd = g1.y == 1 && g2.y == 1;
    int blocks = xcd ? xcd_cols_blocks(g1, a1.tiles_n) + xcd_cols_blocks(g2, a2.tiles_n)
                     : g1.x * g1.y * g1.z + g2.x * g2.y * g2.z;
    hipLaunchKernelGGL((gemm_duo_kernel<32, 64, 1, 4, SrcDenseK, SrcDenseT, EpiMaskBf16,
                                        64, 64, 2, 2, SrcDenseT, SrcDenseT, EpiWgrad>),
                       dim3(blocks), dim3(GEMM_THREADS), 0, stream, a1, g1, a2, g2, xcd);
}

// conv, NHWC bf16 input: gw[O][(ky,kx,c)] = sum_m dy[m][o] * im2col(x)[m][k]
PDA_EXPORT void pda_conv_wgrad(const void* dy, const void* x, void* gw, void* gb, int batch, int H, int W, int C,
                               int KH, int KW, int S, int O, long slice_stride, long bslice_stride, int ksplit,
                               hipStream_t stream) {
    ConvGeom g = make_geom(H, W, C, KH, KW, S);
    int M = batch * g.OH * g.OW, K = C * KH * KW;
    check_fastdiv((long)M + 4 * GEMM_BK, (long)g.OH * g.OW, "conv rows");
    GemmArgs<SrcDenseT, SrcIm2colT, EpiWgrad> a;
    a.g[0].a = {(const bf16_t*)dy, (long)O, O, M};
    a.g[0].b = {(const bf16_t*)x, g, K, M};
    a.g[0].e = {(float*)gw, (float*)gb, (long)K, slice_stride, bslice_stride, 1.f};
    a.g[0].M = O; a.g[0].N = K; a.g[0].K = M;
    if (O <= 32) launch_gemm<32, 64, 1, 4>(a, 1, O, K, M, ksplit, stream);
    else launch_gemm<64, 64, 2, 2>(a, 1, O, K, M, ksplit, stream);
}

// Two conv weight gradients (same output-channel count) as problem groups of ONE launch with a common
// split-K slice length: at batch 128 each of conv2 / conv3 alone is ~240 workgroups, less than one per CU,
// and the two launches ran back to back (2 x 9 us); together they fill the chip once.
PDA_EXPORT void pda_conv_wgrad_pair(const void* const* dy, const void* const* x, void* const* gw, void* const* gb,
                                    int batch, const int* H, const int* W, const int* C, const int* KH,
                                    const int* KW, const int* S, int O, const long* slice_stride,
                                    const long* bslice_stride, int ksplit, hipStream_t stream) {
    GemmArgs<SrcDenseT, SrcIm2colT, EpiWgrad> a;
    int maxN = 0, maxK = 0;
    for (int i = 0; i < 2; ++i) {
        ConvGeom g = make_geom(H[i], W[i], C[i], KH[i], KW[i], S[i]);
        int M = batch * g.OH * g.OW, K = C[i] * KH[i] * KW[i];
        check_fastdiv((long)M + 4 * GEMM_BK, (long)g.OH * g.OW, "conv wgrad rows");
        a.g[i].a = {(const bf16_t*)dy[i], (long)O, O, M};
        a.g[i].b = {(const bf16_t*)x[i], g, K, M};
        a.g[i].e = {(float*)gw[i], (float*)gb[i], (long)K, slice_stride[i], bslice_stride[i], 1.f};
        a.g[i].M = O; a.g[i].N = K; a.g[i].K = M;
        if (K > maxN) maxN = K;
        if (M > maxK) maxK = M;
    }
    int3 grid = plan_gemm<64, 64>(a, 2, O, maxN, maxK, ksplit);
    int units = grid.y * grid.z;
    hipLaunchKernelGGL((gemm_xcd_kernel<64, 64, 2, 2, GEMM_BK, SrcDenseT, SrcIm2colT, EpiWgrad>),
                       dim3((units + 7) / 8 * 8 * grid.x), dim3(GEMM_THREADS), 0, stream, a, grid);
}

// conv1 from uint8 frames: gw[O][(c,ky,kx)] (== PyTorch's own order)
PDA_EXPORT void pda_conv_wgrad_u8(const void* dy, const void* frames, const void* idx, void* gw, void* gb,
                                  int batch, int H, int W, int C, int KH, int KW, int S, int O, float scale,
                                  long slice_stride, long bslice_stride, int ksplit, hipStream_t stream) {
    ConvGeom g = make_geom(H, W, C, KH, KW, S);
    int M = batch * g.OH * g.OW, K = C * KH * KW;
    check_fastdiv((long)M + 4 * GEMM_BK, (long)g.OH * g.OW, "conv rows");
    GemmArgs<SrcDenseT, SrcFrameU8T, EpiWgrad> a;
    a.g[0].a = {(const bf16_t*)dy, (long)O, O, M};
    a.g[0].b = {(const uint8_t*)frames, (const int*)idx, (long)C * H * W, g, K, M};
    a.g[0].e = {(float*)gw, (float*)gb, (long)K, slice_stride, bslice_stride, scale};
    a.g[0].M = O; a.g[0].N = K; a.g[0].K = M;
    if (O <= 32) launch_gemm<32, 64, 1, 4>(a, 1, O, K, M, ksplit, stream);
    else launch_gemm<64, 64, 2, 2>(a, 1, O, K, M, ksplit, stream);
}

// ---- sum split-K slabs into the gradient buffer: dst[e] = sum_s slab[s*stride + e], all jobs in one launch
struct ReduceJob { const float* slab; float* dst; int count; int nslices; long stride; };
struct ReduceJobs { ReduceJob j[8]; int first_block[9]; };

// 64 elements per workgroup: wave w adds slices w, w + 4, ... with 8 loads in flight per lane, the four partial sums
// meet in LDS (the same order as the optimizer launch's own slab sums, optim.h).  One thread per element walking all
// the slices -- 256 of them for conv1 at batch 128 -- is a chain of 64 dependent memory round trips: 12.9 us in the
// data-parallel step, where this launch sits between the weight gradients and the all-reduce of the conv bucket.
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(ReduceJobs jobs, int njobs) {
    __shared__ float part[4][64];
    int job = 0;
    while (job + 1 < njobs && (int)blockIdx.x >= jobs.first_block[job + 1]) ++job;
    const ReduceJob J = jobs.j[job];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int e = ((int)blockIdx.x - jobs.first_block[job]) * 64 + lane;
    const bool ok = e < J.count;
    const float* p = J.slab + (ok ? e : 0);
    float acc[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) acc[u] = 0.f;
    int sl = wave;
    for (; sl + 28 < J.nslices; sl += 32) {
#pragma unroll
        for (int u = 0; u < 8; ++u) acc[u] += p[(long)(sl + 4 * u) * J.stride];
    }
    for (; sl < J.nslices; sl += 4) acc[0] += p[(long)sl * J.stride];
    part[wave][lane] = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
    __syncthreads();
    if (wave == 0 && ok) J.dst[e] = (part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane]);
}

PDA_EXPORT void pda_wgrad_reduce(const void* const* slabs, void* const* dsts, const int* counts, const int* nslices,
                                 const long* strides, int njobs, hipStream_t stream) {
    ReduceJobs jobs;
    int blocks = 0;
    for (int i = 0; i < njobs && i < 8; ++i) {
        jobs.j[i] = {(const float*)slabs[i], (float*)dsts[i], counts[i], nslices[i], strides[i]};
        jobs.first_block[i] = blocks;
        blocks += (counts[i] + 63) / 64;
    }
    jobs.first_block[njobs] = blocks;
    if (blocks == 0) return;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(blocks), dim3(256), 0, stream, jobs, njobs);
}

// ---- col2im: fold dcol[(b,oy,ox)][(ky,kx,c)] back onto the NHWC input grid, times act'(y_in) -----
__global__ void col2im_kernel(const bf16_t* __restrict__ dcol, const bf16_t* __restrict__ yin,
                              bf16_t* __restrict__ dx, ConvGeom g, int batch, int act) {
    // one thread = 8 channels of one input pixel
    int cg = g.C / 8;
    long total = (long)batch * g.H * g.W * cg;
    long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    int c8 = (int)(t % cg);
    long pix = t / cg;
    int ix = (int)(pix % g.W);
    long r = pix / g.W;
    int iy = (int)(r % g.H);
    int b = (int)(r / g.H);
    int K = g.KH * g.KW * g.C;
    float acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = 0.f;
    for (int ky = iy % g.S; ky < g.KH; ky += g.S) {
        int oy = (iy - ky) / g.S;
        if (iy - ky < 0) break;
        if (oy >= g.OH) continue;
        for (int kx = ix % g.S; kx < g.KW; kx += g.S) {
            int ox = (ix - kx) / g.S;
            if (ix - kx < 0) break;
            if (ox >= g.OW) continue;
            const bf16_t* src = dcol + ((long)(b * g.OH + oy) * g.OW + ox) * K + (ky * g.KW + kx) * g.C + c8 * 8;
            uint4 v = *reinterpret_cast<const uint4*>(src);
            const bf16_t* e = reinterpret_cast<const bf16_t*>(&v);
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] += bf2f(e[i]);
        }
    }
    long o = pix * g.C + c8 * 8;
    uint4 yv = *reinterpret_cast<const uint4*>(yin + o);
    const bf16_t* ye = reinterpret_cast<const bf16_t*>(&yv);
    uint32_t outw[4];
#pragma unroll
    for (int i = 0; i < 8; i += 2) {
        float g0 = acc[i], g1 = acc[i + 1];
        float y0 = bf2f(ye[i]), y1 = bf2f(ye[i + 1]);
        if (act == ACT_RELU) { g0 = y0 > 0.f ? g0 : 0.f; g1 = y1 > 0.f ? g1 : 0.f; }
        else if (act == ACT_TANH) { g0 *= (1.f - y0 * y0); g1 *= (1.f - y1 * y1); }
        outw[i / 2] = pack2bf(g0, g1);
    }
    *reinterpret_cast<uint4*>(dx + o) = make_uint4(outw[0], outw[1], outw[2], outw[3]);
}

PDA_EXPORT void pda_col2im(const void* dcol, const void* yin, void* dx, int batch, int H, int W, int C, int KH,
                           int KW, int S, int act, hipStream_t stream) {
    ConvGeom g = make_geom(H, W, C, KH, KW, S);
    long total = (long)batch * H * W * (C / 8);
    int threads = 256;
    long blocks = (total + threads - 1) / threads;
    hipLaunchKernelGGL(col2im_kernel, dim3((unsigned)blocks), dim3(threads), 0, stream, (const bf16_t*)dcol,
                       (const bf16_t*)yin, (bf16_t*)dx, g, batch, act);
}

// k-tile of the GEMM family: split-K slice lengths are rounded up to a multiple of this
PDA_EXPORT int pda_gemm_bk() { return GEMM_BK; }

// ---- probe: returns what ds_read_b64_tr_b16 delivers for a known LDS image (unit-tested on the GPU) --
__global__ void tr_probe_kernel(int* out) {
    __shared__ __attribute__((aligned(16))) bf16_t lds[32 * 72];
    for (int i = threadIdx.x; i < 32 * 72; i += blockDim.x) lds[i] = (bf16_t)i;
    __syncthreads();
    int lane = threadIdx.x & 63;
    // same addressing as read_frag<true, 64>: pitch 72, r0 = 0
    int g = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
    const bf16_t* p0 = lds + (8 * g + q) * 72 + 4 * pp;
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p0));
    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p0 + 4 * 72));
    for (int i = 0; i < 4; ++i) {
        out[lane * 8 + i] = (unsigned short)lo[i];
        out[lane * 8 + 4 + i] = (unsigned short)hi[i];
    }
}
PDA_EXPORT void pda_tr_probe(void* out, hipStream_t stream) {
    hipLaunchKernelGGL(tr_probe_kernel, dim3(1), dim3(64), 0, stream, (int*)out);
}
