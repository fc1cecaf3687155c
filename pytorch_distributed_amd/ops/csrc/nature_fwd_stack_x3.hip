// ---- the same stack at fp32 precision (split-bf16 planes) ---------------------------------------------------
// Every activation is a (hi, lo) pair of bf16 planes, every weight too (the frames are exact in bf16); a product
// is w_hi x_hi + w_lo x_hi + w_hi x_lo into separate fp32 accumulation chains.  Same LDS-resident data flow as
// cnn_fwd_stack_kernel with both planes of a1 / a2 side by side, one image per workgroup (grid.x == batch).
// Register plan (one wave per SIMD, 512 registers per lane): the weights of a layer are resident only while the
// layer runs -- conv1 (128) and conv2 (128) are fetched at the start, conv3's hi plane (72) under conv2, its lo
// plane (72) under the a2 copy-out -- and a row block's activation fragments (hi + lo) are fetched in two
// half-reductions, double-buffered (128 / 144 registers).
#include "nature_cnn.h"

namespace {

#define X_A1LO (20 * F1_RP)     // LDS elements from a1's hi plane to its lo plane
#define X_A2LO (9 * F2_RP)

// relu(v * s + bias) -> (hi, lo) bf16 planes of the 4 values, each packed into 8 bytes
__device__ __forceinline__ void relu_split4(f32x4 v, float s, f32x4 bias, uint2& hi, uint2& lo) {
    const float x0 = fmaxf(v[0] * s + bias[0], 0.f), x1 = fmaxf(v[1] * s + bias[1], 0.f);
    const float x2 = fmaxf(v[2] * s + bias[2], 0.f), x3 = fmaxf(v[3] * s + bias[3], 0.f);
    hi = make_uint2(pack2bf(x0, x1), pack2bf(x2, x3));
    const float r0 = x0 - bf2f((bf16_t)(hi.x & 0xffffu)), r1 = x1 - bf2f((bf16_t)(hi.x >> 16));
    const float r2 = x2 - bf2f((bf16_t)(hi.y & 0xffffu)), r3 = x3 - bf2f((bf16_t)(hi.y >> 16));
    lo = make_uint2(pack2bf(r0, r1), pack2bf(r2, r3));
}

// MULTI (more images than CUs): gridDim.x workgroups per network walk the batch (image blockIdx.x, + gridDim.x, ...).
// The layers' weights are still resident only while a layer runs and come from the L2 per image, but conv1's stay staged
// in an LDS region of their own (2 x 32 x 264 elements behind the activations: 155 KB in all) and the NEXT image's frames
// are requested when conv3 starts -- an image no longer begins with a cold fetch of its frames and the staging of w1
// (7.4 k of the 32.8 k cycles of an image, profiles/r2_stack_kernel_phases.txt).
template <bool MULTI>
__global__ __launch_bounds__(256) void cnn_fwd_stack_x3_kernel(FwdStackArgs a) {
    extern __shared__ __attribute__((aligned(16))) bf16_t lds[];
    constexpr int C = 4, H = 84, W = 84, KS1 = 8, KS2 = 16, KS3 = 18;
    constexpr int OW1 = 20, P1 = 400, OW2 = 9, P2 = 81, OW3 = 7, P3 = 49;
    constexpr int NRB2 = (P2 + 15) / 16, NRB3 = (P3 + 15) / 16;
    constexpr long frame_bytes = (long)C * H * W;
    constexpr int PLANE = FWD_PLANE;
    const int grp = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g4 = lane >> 4, l15 = lane & 15;
    int img = blockIdx.x;
    bf16_t* patch = lds;
    // conv1's weights in LDS, [32][256 + 8] rows per plane: in the (still unused) a1 region, or (MULTI) behind everything
    constexpr int W1P = 264, W1S = MULTI ? 4 * PLANE + 2 * X_A1LO : C * PLANE;

    STACK_STAMP(a.prof, 0, 0)
    const long wp = a.w_plane;
    bf16x8 wf1h[KS1][2], wf1l[KS1][2], wf2h[KS2], wf2l[KS2], wf3h[KS3], wf3l[KS3];
    // conv1's weights go out first, as coalesced 16-byte chunks of the [32][256] tensor, and reach their fragment
    // registers through LDS (free until the frames arrive).  Loaded straight into fragments -- a lane's rows are 512 B
    // apart -- every load instruction touches 64 cache lines: 211 cycles per instruction, and with the other weight
    // loads more than the 63 a wave can keep outstanding (7.9 us from kernel start to the last load issued).
    u32x4 r1h[4], r1l[4];
    {
        const u32x4* sh = reinterpret_cast<const u32x4*>(a.w1[grp]);
        const u32x4* sl = reinterpret_cast<const u32x4*>(a.w1[grp] + wp);
#pragma unroll
        for (int i = 0; i < 4; ++i) { r1h[i] = sh[tid + i * 256]; r1l[i] = sl[tid + i * 256]; }
    }
    PatchRegs pr;
    {
        long f0 = a.idx[grp] ? (long)a.idx[grp][img] : (long)img;
        if (a.ref[grp]) f0 = (long)a.ref[grp][f0];
        patch_issue_at<7>(pr, a.frames[grp] + f0 * a.frame_stride, frame_bytes, tid);
    }
    long f_nxt = MULTI ? frame_index(a.idx[grp], a.ref[grp], img + (int)gridDim.x, a.batch) : 0;
    STACK_STAMP(a.prof, 0, 10)
    // conv2's weights are fetched under conv1's MFMAs, conv3's under conv2's: issued here they would share the L2 -> L1
    // path with the frames and conv1's weights, which the first MFMA waits for (190 KB per workgroup, 256 workgroups
    // at once: 11.6 k cycles from kernel start to the patch commit)
    const bf16_t* w2src = a.w2[grp] + (long)(wave * 16 + l15) * 512 + g4 * 8;
    STACK_STAMP(a.prof, 0, 12)
    const bf16_t* w3src = a.w3[grp] + (long)(wave * 16 + l15) * 576 + g4 * 8;
    const f32x4 b10 = *reinterpret_cast<const f32x4*>(a.b1[grp] + 4 * g4);
    const f32x4 b11 = *reinterpret_cast<const f32x4*>(a.b1[grp] + 16 + 4 * g4);
    const f32x4 b2v = *reinterpret_cast<const f32x4*>(a.b2[grp] + wave * 16 + 4 * g4);
    const f32x4 b3v = *reinterpret_cast<const f32x4*>(a.b3[grp] + wave * 16 + 4 * g4);
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    auto read_w1 = [&]() __attribute__((always_inline)) {      // fragments: 16 lanes x 16 B cover the 64 banks once
#pragma unroll
        for (int ks = 0; ks < KS1; ++ks)
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) {
                const bf16_t* q = lds + W1S + (nb * 16 + l15) * W1P + g4 * 64 + ks * 8;
                wf1h[ks][nb] = *reinterpret_cast<const bf16x8*>(q);
                wf1l[ks][nb] = *reinterpret_cast<const bf16x8*>(q + 32 * W1P);
            }
    };
    {   // w1 -> [32][256 + 8] rows in LDS -> fragments (MULTI: per image, inside the loop)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int t = tid + i * 256, off = W1S + (t >> 5) * W1P + (t & 31) * 8;
            *reinterpret_cast<u32x4*>(lds + off) = r1h[i];
            *reinterpret_cast<u32x4*>(lds + off + 32 * W1P) = r1l[i];
        }
        if constexpr (!MULTI) {
            __syncthreads();
            read_w1();
        }
    }
    STACK_STAMP(a.prof, 0, 13)
    int ld1[7], st1[7], co1[7], ld2[NRB2], st2[NRB2], co2[3], ld3[NRB3], st3[NRB3];
#pragma unroll
    for (int it = 0; it < 7; ++it) {
        int m = (wave + 4 * it) * 16 + l15;         // it = 6 exists for wave 0 only
        m = m < P1 ? m : 0;
        int oy = m / OW1, ox = m - oy * OW1;
        ld1[it] = g4 * PLANE + (4 * oy) * W + 4 * ox;
        st1[it] = C * PLANE + oy * F1_RP + ox * F1_CP + 4 * g4;
        int t = tid + it * 256, pix = (t < P1 * 4 ? t : 0) >> 2;
        int y = pix / OW1, x = pix - y * OW1;
        co1[it] = C * PLANE + y * F1_RP + x * F1_CP + (t & 3) * 8;
        asm volatile("" : "+v"(ld1[it]), "+v"(st1[it]), "+v"(co1[it]));
    }
#pragma unroll
    for (int rb = 0; rb < NRB2; ++rb) {
        int m = rb * 16 + l15;
        int mm = m < P2 ? m : 0;
        int oy = mm / OW2, ox = mm - oy * OW2;
        ld2[rb] = C * PLANE + (2 * oy) * F1_RP + (2 * ox) * F1_CP + g4 * 8;
        st2[rb] = oy * F2_RP + ox * F2_CP + wave * 16 + 4 * g4;
        asm volatile("" : "+v"(ld2[rb]), "+v"(st2[rb]));
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        int t = tid + i * 256, pix = (t < P2 * 8 ? t : 0) >> 3;
        int y = pix / OW2, x = pix - y * OW2;
        co2[i] = y * F2_RP + x * F2_CP + (t & 7) * 8;
        asm volatile("" : "+v"(co2[i]));
    }
#pragma unroll
    for (int rb = 0; rb < NRB3; ++rb) {
        int m = rb * 16 + l15;
        int mm = m < P3 ? m : 0;
        int oy = mm / OW3, ox = mm - oy * OW3;
        ld3[rb] = oy * F2_RP + ox * F2_CP + g4 * 8;
        st3[rb] = m * 64 + wave * 16 + 4 * g4;
        asm volatile("" : "+v"(ld3[rb]), "+v"(st3[rb]));
    }

    // (!MULTI: exactly one pass, resolved at compile time)
#pragma unroll 1
    for (bool first = true; MULTI ? img < a.batch : first; first = false, img += (int)gridDim.x) {
    STACK_STAMP(a.prof, 0, 1)
    if constexpr (MULTI) {
        __syncthreads();        // the previous image's conv3 is done with a2 (which lies over the patch)
        // the weight-fragment addresses are the same for every image: left visible they are all computed in front of
        // the image loop and held across it, a register pair each (nature_dgrad_stack_x3.hip: 141 spilled registers)
        asm volatile("" : "+v"(w2src), "+v"(w3src));
    }
    patch_commit<7, H * W / 16, PLANE - H * W>(pr, frame_bytes, patch, tid);
    __syncthreads();
    if constexpr (MULTI) read_w1();
#pragma unroll
    for (int ks = 0; ks < KS1; ++ks) {
        pin_agpr(wf1h[ks][0]); pin_agpr(wf1h[ks][1]); pin_agpr(wf1l[ks][0]); pin_agpr(wf1l[ks][1]);
    }
    STACK_STAMP(a.prof, 0, 2)
    // ---- conv1 (frames exact: two products per fragment pair) -> a1s hi / lo
    {
        auto load1 = [&](bf16x8 (&xf)[KS1], int it) {
            const bf16_t* base = lds + ld1[it];
#pragma unroll
            for (int ks = 0; ks < KS1; ++ks) {
                u32x2 lo = *(lds_v2)(base + ks * W), hi = *(lds_v2)(base + ks * W + 4);
                xf[ks] = __builtin_bit_cast(bf16x8, make_uint4(lo.x, lo.y, hi.x, hi.y));
            }
        };
        auto mul1 = [&](const bf16x8 (&xf)[KS1], f32x4& c0, f32x4& c1) {
            c0 = zero4; c1 = zero4;
#pragma unroll
            for (int ks = 0; ks < KS1; ++ks) {
                c0 = mfma_wx(wf1h[ks][0], xf[ks], c0);
                c1 = mfma_wx(wf1h[ks][1], xf[ks], c1);
                c0 = mfma_wx(wf1l[ks][0], xf[ks], c0);
                c1 = mfma_wx(wf1l[ks][1], xf[ks], c1);
            }
        };
        auto store1 = [&](const f32x4& c0, const f32x4& c1, int it) {
            bf16_t* dst = lds + st1[it];
            uint2 h, l;
            relu_split4(c0, a.scale, b10, h, l);
            *reinterpret_cast<uint2*>(dst) = h;
            *reinterpret_cast<uint2*>(dst + X_A1LO) = l;
            relu_split4(c1, a.scale, b11, h, l);
            *reinterpret_cast<uint2*>(dst + 16) = h;
            *reinterpret_cast<uint2*>(dst + 16 + X_A1LO) = l;
        };
        bf16x8 xa[KS1], xb[KS1];
        f32x4 ca0, ca1, cb0, cb1;
        load1(xa, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int it = 0; it < 6; ++it) {
            auto& cur = (it & 1) ? xb : xa;
            auto& nxt = (it & 1) ? xa : xb;
            if (it < 5) load1(nxt, it + 1);
            else if (wave == 0) load1(nxt, 6);
            if (it & 1) mul1(cur, cb0, cb1); else mul1(cur, ca0, ca1);
            if (it > 0) { if (it & 1) store1(ca0, ca1, it - 1); else store1(cb0, cb1, it - 1); }
            STACK_SCHED(4 * KS1, 1, 2, 3)
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int l = 6 * it; l < 6 * it + 6; ++l) {     // six of conv2's 32 weight-fragment loads per row block
                if (l < KS2) wf2h[l] = *reinterpret_cast<const bf16x8*>(w2src + l * 32);
                else if (l < 2 * KS2) wf2l[l - KS2] = *reinterpret_cast<const bf16x8*>(w2src + wp + (l - KS2) * 32);
            }
        }
        if (wave == 0) {
            mul1(xa, ca0, ca1);
            store1(cb0, cb1, 5);
            __builtin_amdgcn_sched_barrier(0);
            store1(ca0, ca1, 6);
        } else {
            store1(cb0, cb1, 5);
        }
    }
    STACK_STAMP(a.prof, 0, 3)
    __syncthreads();                    // a1 complete; the patch is dead from here on
#pragma unroll
    for (int ks = 0; ks < KS2; ++ks) { pin_agpr(wf2h[ks]); pin_agpr(wf2l[ks]); }
    STACK_STAMP(a.prof, 0, 5)
    // ---- conv2: wave w -> output channels [16 w, 16 w + 16); unit = half the reduction of a row block
    {
        auto load2 = [&](bf16x8 (&xh)[8], bf16x8 (&xl)[8], int rb, int half) {
            const bf16_t* base = lds + ld2[rb] + 2 * half * F1_RP;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const bf16_t* q = base + (j >> 2) * F1_RP + (j & 3) * F1_CP;
                xh[j] = *reinterpret_cast<const bf16x8*>(q);
                xl[j] = *reinterpret_cast<const bf16x8*>(q + X_A1LO);
            }
        };
        auto mul2 = [&](const bf16x8 (&xh)[8], const bf16x8 (&xl)[8], int half, f32x4 (&c)[3]) {
            if (half == 0) { c[0] = zero4; c[1] = zero4; c[2] = zero4; }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                c[0] = mfma_wx(wf2h[8 * half + j], xh[j], c[0]);
                c[1] = mfma_wx(wf2l[8 * half + j], xh[j], c[1]);
                c[2] = mfma_wx(wf2h[8 * half + j], xl[j], c[2]);
            }
        };
        auto store2 = [&](const f32x4 (&c)[3], int rb) {
            uint2 h, l;
            relu_split4(c[0] + (c[1] + c[2]), 1.f, b2v, h, l);
            if (rb * 16 + l15 < P2) {
                *reinterpret_cast<uint2*>(lds + st2[rb]) = h;
                *reinterpret_cast<uint2*>(lds + st2[rb] + X_A2LO) = l;
            }
        };
        bf16x8 xah[8], xal[8], xbh[8], xbl[8];
        f32x4 ca[3], cb[3];
        load2(xah, xal, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < 2 * NRB2; ++u) {
            const int rb = u >> 1, half = u & 1;
            if (u + 1 < 2 * NRB2) {
                if (u & 1) load2(xah, xal, (u + 1) >> 1, (u + 1) & 1); else load2(xbh, xbl, (u + 1) >> 1, (u + 1) & 1);
            }
            if (u & 1) { if (rb & 1) mul2(xbh, xbl, half, cb); else mul2(xbh, xbl, half, ca); }
            else { if (rb & 1) mul2(xah, xal, half, cb); else mul2(xah, xal, half, ca); }
            if (half == 0 && rb > 0) { if (rb & 1) store2(ca, rb - 1); else store2(cb, rb - 1); }
            STACK_SCHED(24, 2, 3, 3)
            __builtin_amdgcn_sched_barrier(0);
            // in the shadow of the MFMAs: conv3's weights (conv1's registers are free now) and the coalesced
            // copy-out of a1 (both planes; a1s is read-only in this phase)
#pragma unroll
            for (int l = 4 * u; l < 4 * u + 4; ++l) {       // four of the 36 fragment loads per unit (the queue stays
                if (l < KS3) wf3h[l] = *reinterpret_cast<const bf16x8*>(w3src + l * 32);         // shallow; the last
                else if (l < 2 * KS3) wf3l[l - KS3] = *reinterpret_cast<const bf16x8*>(w3src + wp + (l - KS3) * 32);
            }                                                                                    // three units: none)
            if (u >= 1 && u <= 7 && a.a1[grp]) {
                const int i = u - 1;
                if (i < 6 || tid < P1 * 4 - 1536) {
                    uint4* dst = reinterpret_cast<uint4*>(a.a1[grp] + (long)img * P1 * 32);
                    uint4* dstl = reinterpret_cast<uint4*>(a.a1[grp] + a.a1_plane + (long)img * P1 * 32);
                    dst[tid + i * 256] = *reinterpret_cast<const uint4*>(lds + co1[i]);
                    dstl[tid + i * 256] = *reinterpret_cast<const uint4*>(lds + co1[i] + X_A1LO);
                }
            }
        }
        if ((NRB2 - 1) & 1) store2(cb, NRB2 - 1); else store2(ca, NRB2 - 1);
    }
    STACK_STAMP(a.prof, 0, 6)
    __syncthreads();                    // a2 complete
#pragma unroll
    for (int ks = 0; ks < KS3; ++ks) { pin_agpr(wf3h[ks]); pin_agpr(wf3l[ks]); }
    if constexpr (MULTI) {      // the next image's frames fly under conv3 (past the end: frame_index clamps to image 0)
        patch_issue_at<7>(pr, a.frames[grp] + f_nxt * a.frame_stride, frame_bytes, tid);
        f_nxt = frame_index(a.idx[grp], a.ref[grp], img + 2 * (int)gridDim.x, a.batch);
    }
    STACK_STAMP(a.prof, 0, 8)
    // ---- conv3 -> global a3 hi / lo; unit = 9 of the 18 k-steps of a row block
    {
        bf16_t* out3 = a.a3[grp] + (long)img * P3 * 64;
        auto load3 = [&](bf16x8 (&xh)[9], bf16x8 (&xl)[9], int rb, int half) {
            const bf16_t* base = lds + ld3[rb];
#pragma unroll
            for (int j = 0; j < 9; ++j) {
                const int ks = 9 * half + j, tap = ks >> 1, ky = tap / 3, kx = tap - 3 * ky;
                const bf16_t* q = base + ky * F2_RP + kx * F2_CP + (ks & 1) * 32;
                xh[j] = *reinterpret_cast<const bf16x8*>(q);
                xl[j] = *reinterpret_cast<const bf16x8*>(q + X_A2LO);
            }
        };
        auto mul3 = [&](const bf16x8 (&xh)[9], const bf16x8 (&xl)[9], int half, f32x4 (&c)[3]) {
            if (half == 0) { c[0] = zero4; c[1] = zero4; c[2] = zero4; }
#pragma unroll
            for (int j = 0; j < 9; ++j) {
                c[0] = mfma_wx(wf3h[9 * half + j], xh[j], c[0]);
                c[1] = mfma_wx(wf3l[9 * half + j], xh[j], c[1]);
                c[2] = mfma_wx(wf3h[9 * half + j], xl[j], c[2]);
            }
        };
        auto store3 = [&](const f32x4 (&c)[3], int rb) {
            uint2 h, l;
            relu_split4(c[0] + (c[1] + c[2]), 1.f, b3v, h, l);
            if (rb * 16 + l15 < P3) {
                *reinterpret_cast<uint2*>(out3 + st3[rb]) = h;
                *reinterpret_cast<uint2*>(out3 + a.a3_plane + st3[rb]) = l;
            }
        };
        bf16x8 xah[9], xal[9], xbh[9], xbl[9];
        f32x4 ca[3], cb[3];
        load3(xah, xal, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < 2 * NRB3; ++u) {
            const int rb = u >> 1, half = u & 1;
            if (u + 1 < 2 * NRB3) {
                if (u & 1) load3(xah, xal, (u + 1) >> 1, (u + 1) & 1); else load3(xbh, xbl, (u + 1) >> 1, (u + 1) & 1);
            }
            if (u & 1) { if (rb & 1) mul3(xbh, xbl, half, cb); else mul3(xbh, xbl, half, ca); }
            else { if (rb & 1) mul3(xah, xal, half, cb); else mul3(xah, xal, half, ca); }
            if (half == 0 && rb > 0) { if (rb & 1) store3(ca, rb - 1); else store3(cb, rb - 1); }
            STACK_SCHED(27, 2, 3, 3)
            __builtin_amdgcn_sched_barrier(0);
            if (u < 3 && a.a2[grp]) {       // copy-out of a2 (both planes) under the MFMAs
                if (u < 2 || tid < P2 * 8 - 512) {
                    uint4* dst = reinterpret_cast<uint4*>(a.a2[grp] + (long)img * P2 * 64);
                    uint4* dstl = reinterpret_cast<uint4*>(a.a2[grp] + a.a2_plane + (long)img * P2 * 64);
                    dst[tid + u * 256] = *reinterpret_cast<const uint4*>(lds + co2[u]);
                    dstl[tid + u * 256] = *reinterpret_cast<const uint4*>(lds + co2[u] + X_A2LO);
                }
            }
        }
        if ((NRB3 - 1) & 1) store3(cb, NRB3 - 1); else store3(ca, NRB3 - 1);
    }
    STACK_STAMP(a.prof, 0, 9)
    }   // images of this workgroup
}

}  // namespace

// the same at fp32 precision: split weights (lo plane w_plane elements behind hi; one shadow buffer per network, so
// one offset for the three layers) and split activations.  One workgroup per image and network.
PDA_EXPORT void pda_cnn_fwd_stack_x3(const void* const* frames, const void* const* idx, const void* const* ref,
                                     long frame_stride, const void* const* w1,
                                     const void* const* b1, const void* const* w2, const void* const* b2,
                                     const void* const* w3, const void* const* b3, long w_plane, void* const* a1,
                                     long a1_plane, void* const* a2, long a2_plane, void* const* a3, long a3_plane,
                                     int ngroups, int batch, int H, int W, int C, float scale, hipStream_t stream) {
    FwdStackArgs a;
    for (int i = 0; i < ngroups; ++i) {
        a.frames[i] = (const uint8_t*)frames[i]; a.idx[i] = (const int*)idx[i];
        a.ref[i] = ref ? (const int*)ref[i] : nullptr;
        a.w1[i] = (const bf16_t*)w1[i]; a.b1[i] = (const float*)b1[i];
        a.w2[i] = (const bf16_t*)w2[i]; a.b2[i] = (const float*)b2[i];
        a.w3[i] = (const bf16_t*)w3[i]; a.b3[i] = (const float*)b3[i];
        a.a1[i] = (bf16_t*)a1[i]; a.a2[i] = (bf16_t*)a2[i]; a.a3[i] = (bf16_t*)a3[i];
    }
    a.batch = batch; a.scale = scale; a.prof = g_pda_stack_prof;
    a.frame_stride = frame_stride > 0 ? frame_stride : (long)C * H * W;
    a.w_plane = w_plane; a.a1_plane = a1_plane; a.a2_plane = a2_plane; a.a3_plane = a3_plane;
    size_t lds = ((size_t)4 * FWD_PLANE + 2 * (size_t)X_A1LO) * 2;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(cnn_fwd_stack_x3_kernel<false>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(cnn_fwd_stack_x3_kernel<true>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    const int per_group = 256 / (ngroups > 0 ? ngroups : 1);      // one workgroup per CU over all networks
    if (per_group >= 1 && batch >= 4 * per_group) {                 // >= 4 images per workgroup: walk the batch
        const size_t lds_multi = lds + (size_t)2 * 32 * 264 * 2;     // + conv1's weights, both planes
        static_assert(((size_t)4 * FWD_PLANE + 2 * (size_t)X_A1LO) * 2 + 2 * 32 * 264 * 2 <= 160 * 1024, "LDS budget");
        hipLaunchKernelGGL(cnn_fwd_stack_x3_kernel<true>, dim3(per_group, ngroups), dim3(256), lds_multi, stream, a);
        return;
    }
    hipLaunchKernelGGL(cnn_fwd_stack_x3_kernel<false>, dim3(batch, ngroups), dim3(256), lds, stream, a);
}
