// fp32-precision entry points of the MFMA GEMM family: every operand is a split-bf16 pair of planes
// (gemm.h: SrcSplit), every product three bf16 MFMA products accumulated in fp32.
//
// The reference learner is fp32 end to end (core/single_processes/dqn_learner.py:63-87).  gfx950 has no
// reduced-precision fast path for fp32 inputs (no xf32) and its fp32 MFMA runs at 1/16 of the bf16 rate, while
// the bf16 learner step at batch 128 keeps the matrix cores busy ~5 % of the time: three bf16 products per
// logical product cost far less than one fp32-MFMA product and carry 16 significand bits per operand
// (gradients within 1e-4 of an fp32 PyTorch step, tests/test_gpu_x3.py).
//
// Convention: a tensor argument `t` with `t_plane` = element offset from its hi plane to its lo plane.
// Activations / activation gradients enter as A = (hi, lo, hi), weights as B = (hi, hi, lo); uint8 frames are
// exact in bf16 and enter twice against (hi, lo) of the other operand.
#ifdef PDA_X3_BK
#define GEMM_BK PDA_X3_BK
#endif
#include "gemm.h"
#include "api.h"
#include <cstdlib>

namespace {

bool small_grid(int M, int N, int groups) {
    long tiles = (long)((M + 63) / 64) * ((N + 63) / 64) * groups;
    return tiles < 224;
}

}  // namespace

// ---- conv1 forward from uint8 frames (exact in bf16: two products per pair) ----
PDA_EXPORT void pda_x3_conv_fwd_u8(const void* const* frames, const void* const* idx, const void* const* w,
                                   long w_plane, const void* const* bias, void* const* out, long out_plane,
                                   int ngroups, int batch, int H, int W, int C, int KH, int KW, int S, int O,
                                   float scale, int act, hipStream_t stream) {
    const bool generic_only = pda_generic_conv1_forced() != 0;
    if (!generic_only && ngroups <= GEMM_MAX_GROUPS && pda_conv1_supported(H, W, C, KH, KW, S, O)) {
        // whole-image kernel: the frame stack is read once into LDS, both weight planes live in registers
        pda_conv1_fwd_u8_x3(frames, idx, w, w_plane, bias, out, out_plane, ngroups, batch, H, W, C, scale, act, stream);
        return;
    }
    ConvGeom g = make_geom(H, W, C, KH, KW, S);
    int M = batch * g.OH * g.OW, K = C * KH * KW;
    check_fastdiv((long)M + 4 * GEMM_BK, (long)g.OH * g.OW, "conv rows");
    GemmArgs<SPLIT_X2(SrcFrameU8K), SPLIT_HL2(SrcDenseK), EpiBiasAct> a;
    for (int i = 0; i < ngroups; ++i) {
        a.g[i].a = {{(const uint8_t*)frames[i], (const int*)idx[i], (long)C * H * W, g, M, K}, 0, K};
        a.g[i].b = {{(const bf16_t*)w[i], (long)K, O, K}, w_plane, K};
        a.g[i].e = {(bf16_t*)out[i], nullptr, (const float*)bias[i], scale, act, (long)O, 0, out_plane};
        a.g[i].M = M; a.g[i].N = O; a.g[i].K = K;
    }
    if (O <= 32) launch_gemm<64, 32, 4, 1>(a, ngroups, M, O, K, 0, stream);
    else launch_gemm<64, 64, 2, 2>(a, ngroups, M, O, K, 0, stream);
}

// ---- conv forward on split NHWC activations ------------------------------------------------------------
PDA_EXPORT void pda_x3_conv_fwd(const void* const* x, long x_plane, const void* const* w, long w_plane,
                                const void* const* bias, void* const* out, long out_plane, int ngroups, int batch,
                                int H, int W, int C, int KH, int KW, int S, int O, int act, hipStream_t stream) {
    ConvGeom g = make_geom(H, W, C, KH, KW, S);
    int M = batch * g.OH * g.OW, K = C * KH * KW;
    check_fastdiv((long)M + 4 * GEMM_BK, (long)g.OH * g.OW, "conv rows");
    GemmArgs<SPLIT_A3(SrcIm2colK), SPLIT_B3(SrcDenseK), EpiBiasAct> a;
    for (int i = 0; i < ngroups; ++i) {
        a.g[i].a = {{(const bf16_t*)x[i], g, M, K}, x_plane, K};
        a.g[i].b = {{(const bf16_t*)w[i], (long)K, O, K}, w_plane, K};
        a.g[i].e = {(bf16_t*)out[i], nullptr, (const float*)bias[i], 1.f, act, (long)O, 0, out_plane};
        a.g[i].M = M; a.g[i].N = O; a.g[i].K = K;
    }
    if (O <= 32) launch_gemm<64, 32, 4, 1>(a, ngroups, M, O, K, 0, stream);
    else if (small_grid(M, O, ngroups)) launch_gemm<32, 64, 1, 4>(a, ngroups, M, O, K, 0, stream);
    else launch_gemm<64, 64, 2, 2>(a, ngroups, M, O, K, 0, stream);
}

// ---- linear forward, split-K: partial sums into fp32 out32 -------------------------------------------
// slice_stride == 0: atomically added into out32 (pre-zeroed).  slice_stride > 0: slice y goes to
// out32 + y * slice_stride with plain stores, XCD-ordered (deterministic; the consumer sums the slices).
PDA_EXPORT void pda_x3_linear_fwd(const void* const* x, long x_plane, const void* const* w, long w_plane,
                                  void* const* out32, int ngroups, int M, int N, int K, long ldx, long ldw,
                                  long ldo32, long slice_stride, int ksplit, hipStream_t stream) {
    typedef SPLIT_A3(SrcDenseK) SA;
    typedef SPLIT_B3(SrcDenseK) SB;
    if (slice_stride > 0) {
        GemmArgs<SA, SB, EpiSlabF32> a;
        for (int i = 0; i < ngroups; ++i) {
            a.g[i].a = {{(const bf16_t*)x[i], ldx, M, K}, x_plane, K};
            a.g[i].b = {{(const bf16_t*)w[i], ldw, N, K}, w_plane, K};
            a.g[i].e = {(float*)out32[i], ldo32, slice_stride};
            a.g[i].M = M; a.g[i].N = N; a.g[i].K = K;
        }
        int3 grid = plan_gemm<64, 64>(a, ngroups, M, N, K, ksplit > 0 ? ksplit : K);
        int units = grid.y * grid.z;
        if (units < 8) {    // fewer (slice, group) units than XCDs (large batches: one slice): the unit-per-XCD order would
            // leave XCDs idle (batch 4096: 196 us on 2 of the 8 XCDs against 70 us in plain order)
            // large problems: 128-row tiles with a k-tile of 32 (four planes of a k-tile in 64 KB of LDS), see gemm.hip
            const int big = pda_gemm_large_tiles(M, N, units, 2);
            if (big == 3) launch_gemm<128, 64, 2, 2, SA, SB, EpiSlabF32, 32>(a, ngroups, M, N, K, ksplit > 0 ? ksplit : K, stream);
            else if (big) launch_gemm<128, 128, 2, 2, SA, SB, EpiSlabF32, 32>(a, ngroups, M, N, K, ksplit > 0 ? ksplit : K, stream);
            else launch_gemm<64, 64, 2, 2>(a, ngroups, M, N, K, ksplit > 0 ? ksplit : K, stream);
            return;
        }
        hipLaunchKernelGGL((gemm_xcd_kernel<64, 64, 2, 2, GEMM_BK, SA, SB, EpiSlabF32>),
                           dim3((units + 7) / 8 * 8 * grid.x), dim3(GEMM_THREADS), 0, stream, a, grid);
        return;
    }
    GemmArgs<SA, SB, EpiAtomicF32> a;
    for (int i = 0; i < ngroups; ++i) {
        a.g[i].a = {{(const bf16_t*)x[i], ldx, M, K}, x_plane, K};
        a.g[i].b = {{(const bf16_t*)w[i], ldw, N, K}, w_plane, K};
        a.g[i].e = {(float*)out32[i], ldo32, 1.f};
        a.g[i].M = M; a.g[i].N = N; a.g[i].K = K;
    }
    launch_gemm<64, 64, 2, 2>(a, ngroups, M, N, K, ksplit > 0 ? ksplit : K, stream);
}

// ---- data gradient of a linear map --------------------------------------------------------------------
PDA_EXPORT void pda_x3_linear_dgrad(const void* dy, long dy_plane, const void* w, long w_plane, const void* y,
                                    void* dx, long dx_plane, int M, int Nout, int Kin, long lddy, long ldw, long ldy,
                                    long lddx, int act, hipStream_t stream) {
    typedef SPLIT_A3(SrcDenseK) SA;
    typedef SPLIT_B3(SrcDenseT) SB;
    GemmArgs<SA, SB, EpiMaskBf16> a;
    a.g[0].a = {{(const bf16_t*)dy, lddy, M, Nout}, dy_plane, Nout};
    a.g[0].b = {{(const bf16_t*)w, ldw, Kin, Nout}, w_plane, Nout};
    a.g[0].e = {(bf16_t*)dx, (const bf16_t*)y, y ? act : ACT_NONE, lddx, ldy, dx_plane};
    a.g[0].M = M; a.g[0].N = Kin; a.g[0].K = Nout;
    const int big = pda_gemm_large_tiles(M, Kin, 1, 3);
    if (big == 3) launch_gemm<128, 64, 2, 2, SA, SB, EpiMaskBf16, 32>(a, 1, M, Kin, Nout, 0, stream);
    else if (big) launch_gemm<128, 128, 2, 2, SA, SB, EpiMaskBf16, 32>(a, 1, M, Kin, Nout, 0, stream);
    else if (M <= 32 || small_grid(M, Kin, 1)) launch_gemm<32, 64, 1, 4>(a, 1, M, Kin, Nout, 0, stream);
    else launch_gemm<64, 64, 2, 2>(a, 1, M, Kin, Nout, 0, stream);
}

// ---- conv data gradient, gather form (see pda_conv_dgrad) -------------------------------------------
PDA_EXPORT void pda_x3_conv_dgrad(const void* dy, long dy_plane, const void* w, long w_plane, const void* yin,
                                  void* dx, long dx_plane, int batch, int H, int W, int C, int KH, int KW, int S,
                                  int O, int act, hipStream_t stream) {
    ConvGeom cg = make_geom(H, W, C, KH, KW, S);
    GemmArgs<SPLIT_A3(SrcDgradDyK), SPLIT_B3(SrcDgradWT), EpiDgradScatter> a;
    int ng = 0, maxM = 0;
    int K = (KH / S) * (KW / S) * O;
    for (int py = 0; py < S; ++py)
        for (int px = 0; px < S; ++px) {
            DgradGeom g = make_dgrad_geom(cg, O, py, px);
            int M = batch * g.HC * g.WC;
            check_fastdiv((long)M + 64, (long)g.HC * g.WC, "conv dgrad rows");
            check_fastdiv((long)K + 4 * GEMM_BK, O, "conv dgrad reduction");
            a.g[ng].a = {{(const bf16_t*)dy, g, M, K}, dy_plane, K};
            a.g[ng].b = {{(const bf16_t*)w, g, C, K}, w_plane, K};
            a.g[ng].e = {(bf16_t*)dx, (const bf16_t*)yin, yin ? act : ACT_NONE, g, dx_plane};
            a.g[ng].M = M; a.g[ng].N = C; a.g[ng].K = K;
            maxM = M > maxM ? M : maxM;
            ++ng;
        }
    if (C <= 32) launch_gemm<64, 32, 4, 1>(a, ng, maxM, C, K, 0, stream);
    else if (small_grid(maxM, C, ng)) launch_gemm<32, 64, 1, 4>(a, ng, maxM, C, K, 0, stream);
    else launch_gemm<64, 64, 2, 2>(a, ng, maxM, C, K, 0, stream);
}

// ---- weight gradients (split-K slabs, plain stores; see gemm.hip) -------------------------------------
// up to two linear weight gradients over the same batch in one launch (single slice each)
PDA_EXPORT void pda_x3_linear_wgrad2(const void* const* dy, const long* dy_plane, const void* const* x,
                                     const long* x_plane, void* const* gw, void* const* gb, int n, int B,
                                     const int* Nout, const int* Kin, const long* lddy, const long* ldx,
                                     const long* ldg, hipStream_t stream) {
    GemmArgs<SPLIT_A3(SrcDenseT), SPLIT_B3(SrcDenseT), EpiWgrad> a;
    int maxM = 0, maxN = 0;
    for (int i = 0; i < n && i < 2; ++i) {
        a.g[i].a = {{(const bf16_t*)dy[i], lddy[i], Nout[i], B}, dy_plane[i], B};
        a.g[i].b = {{(const bf16_t*)x[i], ldx[i], Kin[i], B}, x_plane[i], B};
        a.g[i].e = {(float*)gw[i], (float*)gb[i], ldg[i], 0, 0, 1.f};
        a.g[i].M = Nout[i]; a.g[i].N = Kin[i]; a.g[i].K = B;
        maxM = Nout[i] > maxM ? Nout[i] : maxM;
        maxN = Kin[i] > maxN ? Kin[i] : maxN;
    }
    launch_gemm<64, 64, 2, 2>(a, n < 2 ? n : 2, maxM, maxN, B, 0, stream);
}

// fc backward in one launch: one linear data gradient (32-row tiles) + the two weight gradients above
PDA_EXPORT void pda_x3_fc_backward_duo(const void* d_dy, long d_dy_plane, const void* d_w, long d_w_plane,
                                       const void* d_y, void* d_dx, long d_dx_plane, int M, int Nout, int Kin,
                                       long lddy, long ldw, long ldy, long lddx, int act,
                                       const void* const* dy, const long* dy_plane, const void* const* x,
                                       const long* x_plane, void* const* gw, void* const* gb, int B,
                                       const int* wNout, const int* wKin, const long* wlddy, const long* wldx,
                                       const long* wldg, hipStream_t stream) {
    typedef SPLIT_A3(SrcDenseK) SA1;
    typedef SPLIT_B3(SrcDenseT) SB1;
    typedef SPLIT_A3(SrcDenseT) SA2;
    typedef SPLIT_B3(SrcDenseT) SB2;
    GemmArgs<SA1, SB1, EpiMaskBf16> a1;
    a1.g[0].a = {{(const bf16_t*)d_dy, lddy, M, Nout}, d_dy_plane, Nout};
    a1.g[0].b = {{(const bf16_t*)d_w, ldw, Kin, Nout}, d_w_plane, Nout};
    a1.g[0].e = {(bf16_t*)d_dx, (const bf16_t*)d_y, d_y ? act : ACT_NONE, lddx, ldy, d_dx_plane};
    a1.g[0].M = M; a1.g[0].N = Kin; a1.g[0].K = Nout;
    int3 g1 = plan_gemm<32, 64>(a1, 1, M, Kin, Nout, 0);
    GemmArgs<SA2, SB2, EpiWgrad> a2;
    int maxM = 0, maxN = 0;
    for (int i = 0; i < 2; ++i) {
        a2.g[i].a = {{(const bf16_t*)dy[i], wlddy[i], wNout[i], B}, dy_plane[i], B};
        a2.g[i].b = {{(const bf16_t*)x[i], wldx[i], wKin[i], B}, x_plane[i], B};
        a2.g[i].e = {(float*)gw[i], (float*)gb[i], wldg[i], 0, 0, 1.f};
        a2.g[i].M = wNout[i]; a2.g[i].N = wKin[i]; a2.g[i].K = B;
        maxM = wNout[i] > maxM ? wNout[i] : maxM;
        maxN = wKin[i] > maxN ? wKin[i] : maxN;
    }
    int3 g2 = plan_gemm<64, 64>(a2, 2, maxM, maxN, B, 0);
    const int xcd = g1.y == 1 && g2.y == 1;
    int blocks = xcd ? xcd_cols_blocks(g1, a1.tiles_n) + xcd_cols_blocks(g2, a2.tiles_n)
                     : g1.x * g1.y * g1.z + g2.x * g2.y * g2.z;
    hipLaunchKernelGGL((gemm_duo_kernel<32, 64, 1, 4, SA1, SB1, EpiMaskBf16, 64, 64, 2, 2, SA2, SB2, EpiWgrad>),
                       dim3(blocks), dim3(GEMM_THREADS), 0, stream, a1, g1, a2, g2, xcd);
}

// one or two conv weight gradients (same output-channel count) as problem groups of one launch, XCD-ordered
PDA_EXPORT void pda_x3_conv_wgrad(const void* const* dy, const long* dy_plane, const void* const* x,
                                  const long* x_plane, void* const* gw, void* const* gb, int n, int batch,
                                  const int* H, const int* W, const int* C, const int* KH, const int* KW,
                                  const int* S, int O, const long* slice_stride, const long* bslice_stride,
                                  int ksplit, hipStream_t stream) {
    GemmArgs<SPLIT_A3(SrcDenseT), SPLIT_B3(SrcIm2colT), EpiWgrad> a;
    int maxN = 0, maxK = 0;
    if (n > 2) n = 2;
    for (int i = 0; i < n; ++i) {
        ConvGeom g = make_geom(H[i], W[i], C[i], KH[i], KW[i], S[i]);
        int M = batch * g.OH * g.OW, K = C[i] * KH[i] * KW[i];
        check_fastdiv((long)M + 4 * GEMM_BK, (long)g.OH * g.OW, "conv wgrad rows");
        a.g[i].a = {{(const bf16_t*)dy[i], (long)O, O, M}, dy_plane[i], M};
        a.g[i].b = {{(const bf16_t*)x[i], g, K, M}, x_plane[i], M};
        a.g[i].e = {(float*)gw[i], (float*)gb[i], (long)K, slice_stride[i], bslice_stride[i], 1.f};
        a.g[i].M = O; a.g[i].N = K; a.g[i].K = M;
        if (K > maxN) maxN = K;
        if (M > maxK) maxK = M;
    }
    if (O <= 32) { launch_gemm<32, 64, 1, 4>(a, n, O, maxN, maxK, ksplit, stream); return; }
    int3 grid = plan_gemm<64, 64>(a, n, O, maxN, maxK, ksplit);
    int units = grid.y * grid.z;
    hipLaunchKernelGGL((gemm_xcd_kernel<64, 64, 2, 2, GEMM_BK, SPLIT_A3(SrcDenseT), SPLIT_B3(SrcIm2colT), EpiWgrad>),
                       dim3((units + 7) / 8 * 8 * grid.x), dim3(GEMM_THREADS), 0, stream, a, grid);
}

// conv1 weight gradient from uint8 frames: gw[O][(c,ky,kx)]; reduction 2 M = dy (hi, lo) against the exact frames
PDA_EXPORT void pda_x3_conv_wgrad_u8(const void* dy, long dy_plane, const void* frames, const void* idx, void* gw,
                                     void* gb, int batch, int H, int W, int C, int KH, int KW, int S, int O,
                                     float scale, long slice_stride, long bslice_stride, int ksplit,
                                     hipStream_t stream) {
    ConvGeom g = make_geom(H, W, C, KH, KW, S);
    int M = batch * g.OH * g.OW, K = C * KH * KW;
    check_fastdiv((long)M + 4 * GEMM_BK, (long)g.OH * g.OW, "conv rows");
    GemmArgs<SPLIT_HL2(SrcDenseT), SPLIT_X2(SrcFrameU8T), EpiWgrad> a;
    a.g[0].a = {{(const bf16_t*)dy, (long)O, O, M}, dy_plane, M};
    a.g[0].b = {{(const uint8_t*)frames, (const int*)idx, (long)C * H * W, g, K, M}, 0, M};
    a.g[0].e = {(float*)gw, (float*)gb, (long)K, slice_stride, bslice_stride, scale};
    a.g[0].M = O; a.g[0].N = K; a.g[0].K = M;
    if (O <= 32) launch_gemm<32, 64, 1, 4>(a, 1, O, K, M, ksplit, stream);
    else launch_gemm<64, 64, 2, 2>(a, 1, O, K, M, ksplit, stream);
}
