// ---- whole forward conv stack per image: conv1 -> conv2 (4x4 s2, 64) -> conv3 (3x3 s1, 64) -------------
// At batch 128 the three conv forwards are three under-filled launches of latency-bound k-loops
// (11 + 12 + 12 us).  Here one workgroup carries an image through all three layers with every
// intermediate in LDS: the frame patch (bf16), a1 [400][32] and a2 [81][64] (row pitches padded to 5 and
// 9 sixteen-byte slots so that the strided pixel rows an MFMA A-fragment gathers fall on different
// banks), all weights as register-resident B-fragments (64 + 64 + 72 registers; wave w owns
// output-channel block w of conv2 / conv3), a3 and -- for the network that is trained -- a1 / a2
// written out with coalesced copies.  Nothing but the frames is read from HBM.
//
// Register plan (one wave per SIMD, so a wave owns the whole 512-entry unified file): the 200 weight
// registers are pinned to the accumulation half (pin_agpr) and feed the MFMA B operand from there.
// Left to itself the compiler keeps them in the 256 architectural VGPRs, runs out, and funnels every
// A-fragment through one register quad: each ds_read then waits out its full LDS latency in front of its
// MFMA (12.8 us per image in steady state, 8x the MFMA time).  With the VGPRs free, the A-fragments of a
// row block are fetched as a batch and the MFMAs run back to back.
// Geometry is the Nature one (4 x 84 x 84 frames), fixed at compile time: no integer divisions by
// run-time widths in the row-block loops.
#include "nature_cnn.h"

namespace {


// MULTI: workgroups loop over several images (next frame prefetched under the current one); otherwise one
// image per workgroup and no prefetch code at all.
template <bool MULTI>
__global__ __launch_bounds__(256) void cnn_fwd_stack_kernel(FwdStackArgs a) {
    extern __shared__ __attribute__((aligned(16))) bf16_t lds[];
    constexpr int C = 4, H = 84, W = 84, KS1 = 8, KS2 = 16, KS3 = 18;
    constexpr int OW1 = 20, P1 = 400, OW2 = 9, P2 = 81, OW3 = 7, P3 = 49;
    constexpr int NRB2 = (P2 + 15) / 16, NRB3 = (P3 + 15) / 16;
    constexpr long frame_bytes = (long)C * H * W;
    constexpr int PLANE = FWD_PLANE;        // channel-plane pitch of the patch
    const int grp = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g4 = lane >> 4, l15 = lane & 15;
    bf16_t* patch = lds;
    bf16_t* a1s = lds + C * PLANE;
    bf16_t* a2s = lds;                      // reuses the patch region once conv1 is done

    // conv1's weights first, as coalesced 16-byte chunks of the [32][256] tensor; they reach their fragment registers
    // through LDS below (a lane's fragment rows are 512 B apart in memory: loaded directly, every instruction touches
    // 64 cache lines -- 211 cycles per load instruction in the fp32-precision kernel's prologue)
    STACK_STAMP(a.prof, 0, 10)
    u32x4 r1[4];
    {
        const u32x4* sw = reinterpret_cast<const u32x4*>(a.w1[grp]);
#pragma unroll
        for (int i = 0; i < 4; ++i) r1[i] = sw[tid + i * 256];
    }
    PatchRegs pr;                           // frame loads next: the other weights arrive behind them
    const int G = gridDim.x;
    long f_nxt = MULTI ? frame_index(a.idx[grp], a.ref[grp], blockIdx.x + G, a.batch) : 0;
    if ((int)blockIdx.x < a.batch) {       // first image: the index is needed at once, as a scalar
        long f0 = a.idx[grp] ? (long)a.idx[grp][blockIdx.x] : (long)blockIdx.x;
        if (a.ref[grp]) f0 = (long)a.ref[grp][f0];
        patch_issue_at<7>(pr, a.frames[grp] + f0 * a.frame_stride, frame_bytes, tid);
    }
    // weight fragments (lane: channel n = l15 + 16 nb, 8 consecutive k).  conv1's k-steps are regrouped as
    // (ky = ks, c = g4, kx 0..7) -- any order works as long as both operands agree -- see conv1 below.
    bf16x8 wf1[KS1][2], wf2[KS2], wf3[KS3];
    // One image per workgroup: conv2's / conv3's weights are fetched under conv1's / conv2's MFMAs (issued here they
    // share the L2 -> L1 path with the frames and conv1's weights, which the first MFMA waits for: 12.8 k cycles from
    // kernel entry to the first image's commit, 5.5 k now).  MULTI: the weights serve every image of the workgroup
    // and are fetched once, here.
    const bf16_t* w2src = a.w2[grp] + (long)(wave * 16 + l15) * 512 + g4 * 8;
    const bf16_t* w3src = a.w3[grp] + (long)(wave * 16 + l15) * 576 + g4 * 8;
    if constexpr (MULTI) {
#pragma unroll
        for (int ks = 0; ks < KS2; ++ks) wf2[ks] = *reinterpret_cast<const bf16x8*>(w2src + ks * 32);
#pragma unroll
        for (int ks = 0; ks < KS3; ++ks) wf3[ks] = *reinterpret_cast<const bf16x8*>(w3src + ks * 32);
    }
    // biases of the 4 channels a lane owns in the swapped product
    const f32x4 b10 = *reinterpret_cast<const f32x4*>(a.b1[grp] + 4 * g4);
    const f32x4 b11 = *reinterpret_cast<const f32x4*>(a.b1[grp] + 16 + 4 * g4);
    const f32x4 b2v = *reinterpret_cast<const f32x4*>(a.b2[grp] + wave * 16 + 4 * g4);
    const f32x4 b3v = *reinterpret_cast<const f32x4*>(a.b3[grp] + wave * 16 + 4 * g4);
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    int it_no = 0;
    STACK_STAMP(a.prof, 0, 11)
    {   // w1 -> [32][256 + 8] rows in the (still unused) a1 region -> fragments: 16 lanes x 16 B cover the 64 banks once
        constexpr int W1S = C * PLANE, W1P = 264;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int t = tid + i * 256;
            *reinterpret_cast<u32x4*>(lds + W1S + (t >> 5) * W1P + (t & 31) * 8) = r1[i];
        }
        __syncthreads();
#pragma unroll
        for (int ks = 0; ks < KS1; ++ks)
#pragma unroll
            for (int nb = 0; nb < 2; ++nb)
                wf1[ks][nb] = *reinterpret_cast<const bf16x8*>(lds + W1S + (nb * 16 + l15) * W1P + g4 * 64 + ks * 8);
    }
    STACK_STAMP(a.prof, 0, 12)
    // Per-lane LDS offsets (bf16 elements) of every fragment gather, output store and copy-out chunk, computed
    // here -- in the shadow of the first frame / weight loads -- and made opaque so that the divisions are not
    // re-materialised inside the phases (a workgroup with a single image has no loop to hoist them out of).
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
        st3[rb] = m * 64 + wave * 16 + 4 * g4;        // global a3 offset within the image
        asm volatile("" : "+v"(ld3[rb]), "+v"(st3[rb]));
    }

    // One image.  Instantiated twice: a peeled first image, where the weight loads are still in flight and
    // are waited for as each layer first needs them, and the steady-state loop, whose waits then cover only
    // its own loads (one merged body made the compiler wait on the next image's frame loads in front of conv1).
    auto process = [&](int img) __attribute__((always_inline)) {
        STACK_STAMP(a.prof, it_no, 0)
        __syncthreads();                    // previous image fully consumed (a2s aliases the patch)
        patch_commit<7, H * W / 16, PLANE - H * W>(pr, frame_bytes, patch, tid);
        __syncthreads();
        STACK_STAMP(a.prof, it_no, 1)
        // unconditional (past the end: frame_index clamps to image 0): a conditional issue leaves the
        // compiler merging old and new registers with copies that wait for the loads on the spot
        if constexpr (MULTI) {
            patch_issue_at<7>(pr, a.frames[grp] + f_nxt * a.frame_stride, frame_bytes, tid);
            f_nxt = frame_index(a.idx[grp], a.ref[grp], img + 2 * G, a.batch);     // consumed one image from now
        }
#pragma unroll
        for (int ks = 0; ks < KS1; ++ks) { pin_agpr(wf1[ks][0]); pin_agpr(wf1[ks][1]); }

        // ---- conv1: 8x8 s4, 32 channels -> a1s (25 row blocks of 16 pixels, exactly; wave w takes w, w + 4, ..)
        // Pixel fragment of lane (pixel l15, g4): channel g4, kernel row ks, kx 0..7 = 16 bytes at an 8-byte
        // aligned address, as two ds_read_b64.  Pixel m sits at dword 2 m + 128 oy of its row set, so 16
        // consecutive pixels cover 32 distinct banks whatever rows they span, and with the plane pitch
        // == 32 (mod 64) dwords the other channel of a 32-lane group takes the other 32: conflict-free.
        // Software pipeline over row blocks (fully unrolled): the fragments of block i + 1 are fetched and
        // the outputs of block i - 1 are scaled / packed / stored in the shadow of block i's MFMAs.
        {
            auto load1 = [&](bf16x8 (&xf)[KS1], int it) {
                const bf16_t* base = lds + ld1[it];
#pragma unroll
                for (int ks = 0; ks < KS1; ++ks) {     // two ds_read_b64 (kept apart: see lds_v2)
                    u32x2 lo = *(lds_v2)(base + ks * W), hi = *(lds_v2)(base + ks * W + 4);
                    xf[ks] = __builtin_bit_cast(bf16x8, make_uint4(lo.x, lo.y, hi.x, hi.y));
                }
            };
            auto mul1 = [&](const bf16x8 (&xf)[KS1], f32x4& c0, f32x4& c1) {
                c0 = zero4; c1 = zero4;
#pragma unroll
                for (int ks = 0; ks < KS1; ++ks) {
                    c0 = mfma_wx(wf1[ks][0], xf[ks], c0);
                    c1 = mfma_wx(wf1[ks][1], xf[ks], c1);
                }
            };
            auto store1 = [&](const f32x4& c0, const f32x4& c1, int it) {
                bf16_t* dst = lds + st1[it];
                *reinterpret_cast<uint2*>(dst) = relu_pack4(c0, a.scale, b10);
                *reinterpret_cast<uint2*>(dst + 16) = relu_pack4(c1, a.scale, b11);
            };
            bf16x8 xa[KS1], xb[KS1];
            f32x4 ca0, ca1, cb0, cb1;
            load1(xa, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int it = 0; it < 6; ++it) {            // row blocks wave + 4 it <= 23: every wave has them
                auto& cur = (it & 1) ? xb : xa;
                auto& nxt = (it & 1) ? xa : xb;
                if (it < 5) load1(nxt, it + 1);
                else if (wave == 0) load1(nxt, 6);      // the 25th block is wave 0's
                if (it & 1) mul1(cur, cb0, cb1); else mul1(cur, ca0, ca1);
                if (it > 0) { if (it & 1) store1(ca0, ca1, it - 1); else store1(cb0, cb1, it - 1); }
                STACK_SCHED(2 * KS1, 1, 1, 2)
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (!MULTI) {
#pragma unroll
                    for (int l = 3 * it; l < 3 * it + 3 && l < KS2; ++l)
                        wf2[l] = *reinterpret_cast<const bf16x8*>(w2src + l * 32);
                }
            }
            if (wave == 0) {
                mul1(xa, ca0, ca1);                     // it = 6
                store1(cb0, cb1, 5);
                STACK_SCHED(2 * KS1, 0, 2, 4)
                __builtin_amdgcn_sched_barrier(0);
                store1(ca0, ca1, 6);
            } else {
                store1(cb0, cb1, 5);
            }
        }
        STACK_STAMP(a.prof, it_no, 2)
        __syncthreads();                    // a1 complete; the patch is dead from here on
        STACK_STAMP(a.prof, it_no, 3)
        if (a.a1[grp]) {                    // coalesced copy-out for the backward pass (4 x 16 B per pixel)
            uint4* dst = reinterpret_cast<uint4*>(a.a1[grp] + (long)img * P1 * 32);
            auto a1_chunk = [&](int i) { return *reinterpret_cast<const uint4*>(lds + co1[i]); };
            uint4 v0 = a1_chunk(0), v1 = a1_chunk(1), v2 = a1_chunk(2), v3 = a1_chunk(3), v4 = a1_chunk(4),
                  v5 = a1_chunk(5);
            dst[tid] = v0; dst[tid + 256] = v1; dst[tid + 512] = v2; dst[tid + 768] = v3;
            dst[tid + 1024] = v4; dst[tid + 1280] = v5;
            if (tid < P1 * 4 - 1536) dst[tid + 1536] = a1_chunk(6);
        }
        STACK_STAMP(a.prof, it_no, 4)
#pragma unroll
        for (int ks = 0; ks < KS2; ++ks) pin_agpr(wf2[ks]);
#pragma unroll
        for (int ks = 0; ks < KS3; ++ks) pin_agpr(wf3[ks]);
        STACK_STAMP(a.prof, it_no, 5)
        // ---- conv2: 4x4 s2, 32 -> 64; wave w computes output channels [16 w, 16 w + 16) -> a2s
        {
            auto load2 = [&](bf16x8 (&xf)[KS2], int rb) {
                const bf16_t* base = lds + ld2[rb];
#pragma unroll
                for (int ks = 0; ks < KS2; ++ks)        // one tap (ky, kx) = 32 channels per k-step
                    xf[ks] = *reinterpret_cast<const bf16x8*>(base + (ks >> 2) * F1_RP + (ks & 3) * F1_CP);
            };
            auto mul2 = [&](const bf16x8 (&xf)[KS2], f32x4& c0, f32x4& c1) {
                c0 = zero4; c1 = zero4;
#pragma unroll
                for (int ks = 0; ks < KS2; ks += 2) {   // two independent accumulation chains
                    c0 = mfma_wx(wf2[ks], xf[ks], c0);
                    c1 = mfma_wx(wf2[ks + 1], xf[ks + 1], c1);
                }
            };
            auto store2 = [&](const f32x4& c0, const f32x4& c1, int rb) {
                if (rb * 16 + l15 < P2) *reinterpret_cast<uint2*>(lds + st2[rb]) = relu_pack4(c0 + c1, 1.f, b2v);
            };
            bf16x8 xa[KS2], xb[KS2];
            f32x4 ca0, ca1, cb0, cb1;
            load2(xa, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int rb = 0; rb < NRB2; ++rb) {
                auto& cur = (rb & 1) ? xb : xa;
                auto& nxt = (rb & 1) ? xa : xb;
                if (rb + 1 < NRB2) load2(nxt, rb + 1);
                if (rb & 1) mul2(cur, cb0, cb1); else mul2(cur, ca0, ca1);
                if (rb > 0) { if (rb & 1) store2(ca0, ca1, rb - 1); else store2(cb0, cb1, rb - 1); }
                STACK_SCHED(KS2, 1, 1, 2)
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (!MULTI) {
#pragma unroll
                    for (int l = 4 * rb; l < 4 * rb + 4 && l < KS3; ++l)
                        wf3[l] = *reinterpret_cast<const bf16x8*>(w3src + l * 32);
                }
            }
            if ((NRB2 - 1) & 1) store2(cb0, cb1, NRB2 - 1); else store2(ca0, ca1, NRB2 - 1);
        }
        STACK_STAMP(a.prof, it_no, 6)
        __syncthreads();                    // a2 complete
        STACK_STAMP(a.prof, it_no, 7)
        if (a.a2[grp]) {
            uint4* dst = reinterpret_cast<uint4*>(a.a2[grp] + (long)img * P2 * 64);
            auto a2_chunk = [&](int i) { return *reinterpret_cast<const uint4*>(lds + co2[i]); };
            uint4 v0 = a2_chunk(0), v1 = a2_chunk(1);
            dst[tid] = v0; dst[tid + 256] = v1;
            if (tid < P2 * 8 - 512) dst[tid + 512] = a2_chunk(2);
        }
        STACK_STAMP(a.prof, it_no, 8)
        // ---- conv3: 3x3 s1, 64 -> 64 -> global a3 [(img, pix)][64]
        {
            bf16_t* out3 = a.a3[grp] + (long)img * P3 * 64;
            auto load3 = [&](bf16x8 (&xf)[KS3], int rb) {
                const bf16_t* base = lds + ld3[rb];
#pragma unroll
                for (int ks = 0; ks < KS3; ++ks) {      // tap = ks / 2, channel half = ks % 2
                    int tap = ks >> 1, ky = tap / 3, kx = tap - 3 * ky;
                    xf[ks] = *reinterpret_cast<const bf16x8*>(base + ky * F2_RP + kx * F2_CP + (ks & 1) * 32);
                }
            };
            auto mul3 = [&](const bf16x8 (&xf)[KS3], f32x4& c0, f32x4& c1) {
                c0 = zero4; c1 = zero4;
#pragma unroll
                for (int ks = 0; ks < KS3; ks += 2) {
                    c0 = mfma_wx(wf3[ks], xf[ks], c0);
                    c1 = mfma_wx(wf3[ks + 1], xf[ks + 1], c1);
                }
            };
            auto store3 = [&](const f32x4& c0, const f32x4& c1, int rb) {
                if (rb * 16 + l15 < P3) *reinterpret_cast<uint2*>(out3 + st3[rb]) = relu_pack4(c0 + c1, 1.f, b3v);
            };
            bf16x8 xa[KS3], xb[KS3];
            f32x4 ca0, ca1, cb0, cb1;
            load3(xa, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int rb = 0; rb < NRB3; ++rb) {
                auto& cur = (rb & 1) ? xb : xa;
                auto& nxt = (rb & 1) ? xa : xb;
                if (rb + 1 < NRB3) load3(nxt, rb + 1);
                if (rb & 1) mul3(cur, cb0, cb1); else mul3(cur, ca0, ca1);
                if (rb > 0) { if (rb & 1) store3(ca0, ca1, rb - 1); else store3(cb0, cb1, rb - 1); }
                STACK_SCHED(KS3, 1, 1, 2)
                __builtin_amdgcn_sched_barrier(0);
            }
            if ((NRB3 - 1) & 1) store3(cb0, cb1, NRB3 - 1); else store3(ca0, ca1, NRB3 - 1);
        }
        STACK_STAMP(a.prof, it_no, 9)
        ++it_no;
    };
    int img = blockIdx.x;
    if (img < a.batch) {
        process(img);
        if constexpr (MULTI)
            for (img += G; img < a.batch; img += G) process(img);
    }
}

}  // namespace

long long* g_pda_stack_prof = nullptr;
PDA_EXPORT void pda_cnn_stack_prof(void* buf) { g_pda_stack_prof = (long long*)buf; }

// conv1 -> conv2 -> conv3 forward of the Nature CNN in one launch (relu after each; see cnn_fwd_stack_kernel).
// Layer shapes are the Nature ones: 8x8 s4 x32, 4x4 s2 x64, 3x3 s1 x64 on C = 4 frame stacks.
PDA_EXPORT int pda_cnn_fwd_stack_supported(int H, int W, int C) { return (H == 84 && W == 84 && C == 4) ? 1 : 0; }

// ref / frame_stride: frame_index() in nature_cnn.h (ref == null or frame_stride == 0: rows of full stacks).
PDA_EXPORT void pda_cnn_fwd_stack(const void* const* frames, const void* const* idx, const void* const* ref,
                                  long frame_stride, const void* const* w1,
                                  const void* const* b1, const void* const* w2, const void* const* b2,
                                  const void* const* w3, const void* const* b3, void* const* a1, void* const* a2,
                                  void* const* a3, int ngroups, int batch, int H, int W, int C, float scale,
                                  hipStream_t stream) {
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
    size_t lds = ((size_t)4 * FWD_PLANE + (size_t)20 * F1_RP) * 2;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(cnn_fwd_stack_kernel<false>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(cnn_fwd_stack_kernel<true>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    int blocks = batch < 256 ? batch : 256;         // one workgroup per CU (89 KB LDS), looping over images
    if (blocks < batch)
        hipLaunchKernelGGL(cnn_fwd_stack_kernel<true>, dim3(blocks, ngroups), dim3(256), lds, stream, a);
    else
        hipLaunchKernelGGL(cnn_fwd_stack_kernel<false>, dim3(blocks, ngroups), dim3(256), lds, stream, a);
}
