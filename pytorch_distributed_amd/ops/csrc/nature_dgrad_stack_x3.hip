// ---- the data-gradient chain at fp32 precision (split-bf16 planes) ---------------------------------------
// da3 / da2 / da1 are (hi, lo) pairs of planes, the weights too; the forward activations only gate (relu'), so
// their hi planes suffice.  Same per-image LDS data flow as cnn_dgrad_stack_kernel with a second grid / row set
// for every lo plane (148 KB), one image per workgroup, products w_hi d_hi + w_lo d_hi + w_hi d_lo in three
// accumulation chains.  All four weight-fragment sets (272 registers) are staged through LDS at the start, under
// the input loads; conv3^T fetches its da3 fragments in thirds of the reduction to stay inside the register file.
#include "nature_cnn.h"

namespace {

#define DX_A1L (400 * D1_CP)                    // da1 lo rows (a1 hi / da1 hi rows sit at 0)
#define DX_A2H (2 * 400 * D1_CP)                // a2 hi -> da2 hi grid
#define DX_A2L (DX_A2H + 11 * D2_RP)            // da2 lo grid
#define DX_D3H (DX_A2L + 11 * D2_RP)            // da3 hi grid
#define DX_D3L (DX_D3H + 11 * D3_RP)            // da3 lo grid
#define DX_LDS_ELEMS (DX_D3L + 11 * D3_RP)

struct DgradStackX3Args {
    const bf16_t* da3; const bf16_t* a2; const bf16_t* a1;
    const bf16_t* w3t; const bf16_t* w2t;       // TRANSPOSED weight shadows [(tap, c)][o] (csrc/optim.h SegEntry::tr_off)
    bf16_t* da2; bf16_t* da1; int batch;
    long da3_plane, w_plane, da2_plane, da1_plane;
    long long* prof;    // optional phase stamps (pda_cnn_stack_prof)
    // optional tail (W1): conv1's weight gradient of the same image, from da1 (hi, lo) while it is still in LDS;
    // slab slice blockIdx.x
    const uint8_t* frames; const int* idx; float* gw1; float* gb1; long ws1, bs1; float scale1;
    const int* ref; long frame_stride;      // frame_index() in nature_cnn.h
};

// (hi, lo) planes of 4 gradient values, gated by relu'(act): 8 bytes each
__device__ __forceinline__ void gate_split4(uint2 act, f32x4 v, uint2& hi, uint2& lo) {
    const uint2 h = make_uint2(pack2bf(v[0], v[1]), pack2bf(v[2], v[3]));
    const float r0 = v[0] - bf2f((bf16_t)(h.x & 0xffffu)), r1 = v[1] - bf2f((bf16_t)(h.x >> 16));
    const float r2 = v[2] - bf2f((bf16_t)(h.y & 0xffffu)), r3 = v[3] - bf2f((bf16_t)(h.y >> 16));
    hi = make_uint2(relu_gate2(act.x, h.x), relu_gate2(act.y, h.y));
    lo = make_uint2(relu_gate2(act.x, pack2bf(r0, r1)), relu_gate2(act.y, pack2bf(r2, r3)));
}

// W1: the frame patch of the image (bf16, exact) is committed over the dead a2 / da3 grids once conv2^T is done, and
// the workgroup adds dW1 = da1^T im2col(frames) of its image with both planes of da1 read straight from LDS (as in
// cnn_dgrad_stack_kernel<., true>): one launch and one round trip of da1 through memory less per step.
// PAIR: two workgroups per image (small batches: 2 x batch workgroups for 256 CUs; batch 128: 10.25 k -> 10.5 k updates/s).
// Both run conv3^T, so a pair spends 57.8 k CU-cycles on an image where one workgroup that owns all four parity classes
// (wave = class, 7 row blocks) spends 33.8 k: large batches, where every CU has images queued anyway, take !PAIR
// (batch 4096: 403 -> 236 us expected from the phase stamps; the step went from 1.555 to 1.398 ms).
// (Measured and not kept for PAIR: workgroup h owning a1 rows [10 h, 10 h + 10) instead of a row parity, so that it needs
// only da2 rows 0-4 or 4-8 -- three conv3^T row blocks instead of six.  The phase went from 8.6 k to 7.3 k cycles, not to
// half: each wave then owns a parity class of its own and the workgroup pulls ALL of conv2's weights under conv3^T
// (131 KB instead of 65 KB; ~20 B/clk per CU), and the kernel as a whole stayed at 28.7 k cycles.  It is the 212 KB of
// split weights per workgroup, not the MFMAs, that bound this kernel at batch 128.)
// W1 && !PAIR: gridDim.x workgroups walk the batch (image blockIdx.x, + gridDim.x, ...) and keep ONE dW1 accumulator
// each across their images -- one slab slice per workgroup whatever the batch, so large batches get conv1's weight
// gradient from da1 in LDS too instead of from a launch of its own that reads da1 back from memory.
template <bool W1, bool PAIR = true>
__global__ __launch_bounds__(256) void cnn_dgrad_stack_x3_kernel(DgradStackX3Args a) {
    constexpr bool LOOP = W1 && !PAIR;
    extern __shared__ __attribute__((aligned(16))) bf16_t lds[];
    constexpr int P1 = 400, P2 = 81, P3 = 49, NRB3 = 6;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g4 = lane >> 4, l15 = lane & 15;
    // two workgroups per image: both run conv3^T (da2 is needed whole), workgroup h = blockIdx.x & 1 then owns the
    // a1 / da1 rows of parity h -- the parity classes (h, 0) and (h, 1) of conv2^T and those pixels' share of dW1
    int img = PAIR ? (int)blockIdx.x >> 1 : (int)blockIdx.x;
    const int hh = PAIR ? (int)blockIdx.x & 1 : 0;
    constexpr int FH = 84, FW = 84;
    constexpr long frame_bytes = 4L * FH * FW;
    f32x4 acc1[2][4], accb1[2];         // dW1 / db1 (W1): of this image, or (LOOP) of all of this workgroup's images
    if constexpr (W1) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            accb1[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < 4; ++j) acc1[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
    }
    bf16x8 wf3h[18], wf3l[18];      // conv3^T: this wave's 16 input channels; k-step = (tap, o half)
    // fragments [k0, k1) of conv3's weights (tap = ks / 2, o half = ks % 2), both planes, from the transposed shadow.
    // LOOP: an image's prologue is mostly the wait for these 147 KB per workgroup (a CU pulls ~20 B/clk through its L1:
    // 6.9 k of the 46.5 k cycles of an image), so the NEXT image's are requested under the conv1 weight-gradient tail
    auto issue_w3 = [&](const bf16_t* w3src, int k0, int k1) __attribute__((always_inline)) {
#pragma unroll
        for (int ks = 0; ks < 18; ++ks) {
            if (ks >= k0 && ks < k1) {
                const bf16_t* q = w3src + (ks >> 1) * 64 * 64 + (ks & 1) * 32;
                wf3h[ks] = *reinterpret_cast<const bf16x8*>(q);
                wf3l[ks] = *reinterpret_cast<const bf16x8*>(q + a.w_plane);
            }
        }
    };
    // this image's inputs (a1 / a2: hi planes only; da3: both): PAIR the 200 a1 pixels of row parity hh, else all 400.
    // LOOP: the next image's are requested under the tail as well
    constexpr int N1 = PAIR ? 4 : 7;     // 16-byte chunks of a1 per thread: 200 or 400 pixels x 4
    constexpr int N1_ROWS = PAIR ? 800 : 1600;
    u32x4 i1[N1], i2[3], i3h[2], i3l[2];
    auto issue_inputs = [&](int im, int tid_, const int (&g1_)[N1]) __attribute__((always_inline)) {
        const u32x4* s1 = reinterpret_cast<const u32x4*>(a.a1 + (long)im * 400 * 32);
        const u32x4* s2 = reinterpret_cast<const u32x4*>(a.a2 + (long)im * 81 * 64);
        const u32x4* s3 = reinterpret_cast<const u32x4*>(a.da3 + (long)im * 49 * 64);
        const u32x4* s3l = reinterpret_cast<const u32x4*>(a.da3 + a.da3_plane + (long)im * 49 * 64);
#pragma unroll
        for (int i = 0; i < N1; ++i) i1[i] = s1[g1_[i]];
#pragma unroll
        for (int i = 0; i < 3; ++i) { int t = tid_ + i * 256; i2[i] = s2[t < 648 ? t : 0]; }
#pragma unroll
        for (int i = 0; i < 2; ++i) { int t = tid_ + i * 256; i3h[i] = s3[t < 392 ? t : 0]; i3l[i] = s3l[t < 392 ? t : 0]; }
    };
    if constexpr (LOOP) {
        const int tid0 = (int)threadIdx.x, lane0 = tid0 & 63;
        issue_w3(a.w3t + (long)(wave * 16 + (lane0 & 15)) * 64 + (lane0 >> 4) * 8, 0, 18);
        int g0[N1];
#pragma unroll
        for (int i = 0; i < N1; ++i) { int t = tid0 + i * 256; g0[i] = t < N1_ROWS ? t : 0; }
        issue_inputs(img, tid0, g0);
    }
    // per-lane LDS offsets of the grids: the same for every image, computed by the first pass only (in1 / g1, which cost
    // no division when a workgroup owns whole images, per image: all of them held across the loop spill 5 registers)
    constexpr int NRB2W = PAIR ? 4 : 7;  // conv2^T row blocks per wave.  PAIR: wave = (px, part), part 0 -> blocks 0-3, 1 -> 4-6; else wave = (py, px), all 7
    int in1[N1], g1[N1], in2[3], in3[2];
    int ld3[NRB3], st3[NRB3], ld2[NRB2W], st2[NRB2W];
    // (!LOOP: exactly one pass, resolved at compile time)
#pragma unroll 1
    for (bool first = true; LOOP ? img < a.batch : first; first = false, img += (int)gridDim.x) {
    if constexpr (LOOP) __syncthreads();    // the previous image's tail is done with the patch / da1 rows
    // LOOP: everything below that depends only on the lane is the same for every image; left visible, the compiler
    // computes all of it in front of the image loop and holds it across (a 64-bit address per weight fragment, the
    // tail's offsets, ...: 141 spilled registers, 568 bytes of scratch per lane).  Opaque copies keep it in the body.
    int tid_v = (int)threadIdx.x;
    if constexpr (LOOP) asm volatile("" : "+v"(tid_v));
    const int tid = tid_v, lane = tid & 63, g4 = lane >> 4, l15 = lane & 15;
    long f0 = 0;
    if constexpr (W1) f0 = frame_index(a.idx, a.ref, img, a.batch);      // a per-lane load with a counted wait at its use

    STACK_STAMP(a.prof, 0, 0)
    // ---- conv3's weight fragments (both planes) and this image's inputs (a1 / a2: hi planes only) go out together;
    // conv2's fragments are fetched under conv3^T.  The fragments -- rows = input channel, k = output channel -- come
    // straight from the transposed shadows the optimizer keeps: 16 bytes per lane and fragment, no staging.  (Transposed
    // through LDS at the start of the launch, as the bf16 kernel does, the four tensors cost 8 barriers and 10 k cycles
    // before the first MFMA.)
    bf16x8 wf2h[8][2], wf2l[8][2];  // conv2^T: this wave's parity class; (tap a,b ; o half) x 2 channel blocks
    const int py = PAIR ? hh : wave >> 1, px = wave & 1, part = PAIR ? wave >> 1 : 0;
    const bf16_t* w3src = a.w3t + (long)(wave * 16 + l15) * 64 + g4 * 8;
    const bf16_t* w2src = a.w2t + (long)l15 * 64 + g4 * 8;
    if constexpr (LOOP) {
        // the fragment addresses below are the same for every image: left visible, they are all hoisted out of the image
        // loop and held across it (a register pair each: 141 spilled registers)
        asm volatile("" : "+v"(w3src), "+v"(w2src));
    }
    if constexpr (!LOOP) issue_w3(w3src, 0, 18);       // (LOOP: requested in front of the loop / under the previous tail)
    STACK_STAMP(a.prof, 0, 1)
#pragma unroll
    for (int i = 0; i < N1; ++i) {      // chunk t of this workgroup's pixels (PAIR: the 200 of row parity hh) x 4 x 16 bytes
        int t = tid + i * 256;
        t = t < N1_ROWS ? t : 0;
        const int pp = t >> 2, p = PAIR ? (2 * (pp / 20) + hh) * 20 + pp % 20 : pp;
        in1[i] = p * D1_CP + (t & 3) * 8;
        g1[i] = p * 4 + (t & 3);
        asm volatile("" : "+v"(in1[i]), "+v"(g1[i]));
    }
    if constexpr (!LOOP) issue_inputs(img, tid, g1);       // (LOOP: requested in front of the loop / under the previous tail)
    // fragment (t, nb) of conv2^T: t = (a * 2 + b) * 2 + o half, tap = (py + 2a) * 4 + px + 2b, channels 16 nb + l15
    auto w2_frag = [&](int t, int nb, long plane) {
        const int ta = t >> 2, tb = (t >> 1) & 1, half = t & 1;
        const int tap = (py + 2 * ta) * 4 + (px + 2 * tb);
        return *reinterpret_cast<const bf16x8*>(w2src + plane + (long)(tap * 32 + nb * 16) * 64 + half * 32);
    };
    STACK_STAMP(a.prof, 0, 2)
    // the four grids start out all zero; the image only writes their interiors
    for (int t = tid; t < (DX_LDS_ELEMS - DX_A2H) / 8; t += 256)
        *reinterpret_cast<uint4*>(lds + DX_A2H + t * 8) = make_uint4(0, 0, 0, 0);

    if (!LOOP || first) {
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        int t = tid + i * 256;
        t = t < P2 * 8 ? t : 0;
        int pix = t >> 3, y = pix / 9, x = pix - 9 * y;
        in2[i] = DX_A2H + (y + 1) * D2_RP + (x + 1) * D2_CP + (t & 7) * 8;
        asm volatile("" : "+v"(in2[i]));
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        int t = tid + i * 256;
        t = t < P3 * 8 ? t : 0;
        int pix = t >> 3, y = pix / 7, x = pix - 7 * y;
        in3[i] = DX_D3H + (y + 2) * D3_RP + (x + 2) * D3_CP + (t & 7) * 8;
        asm volatile("" : "+v"(in3[i]));
    }
#pragma unroll
    for (int rb = 0; rb < NRB3; ++rb) {
        int m = rb * 16 + l15;
        m = m < P2 ? m : 0;
        int iy = m / 9, ix = m - 9 * iy;
        ld3[rb] = DX_D3H + (iy + 2) * D3_RP + (ix + 2) * D3_CP + g4 * 8;
        st3[rb] = DX_A2H + (iy + 1) * D2_RP + (ix + 1) * D2_CP + wave * 16 + 4 * g4;
        asm volatile("" : "+v"(ld3[rb]), "+v"(st3[rb]));
    }
#pragma unroll
    for (int rb = 0; rb < NRB2W; ++rb) {
        int q = (part * NRB2W + rb) * 16 + l15;
        q = q < 100 ? q : 0;
        int iy2 = q / 10, ix2 = q - 10 * iy2;
        ld2[rb] = DX_A2H + (iy2 + 1) * D2_RP + (ix2 + 1) * D2_CP + g4 * 8;
        st2[rb] = ((2 * iy2 + py) * 20 + 2 * ix2 + px) * D1_CP + 4 * g4;
        asm volatile("" : "+v"(ld2[rb]), "+v"(st2[rb]));
    }
    }
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    __syncthreads();                    // grids zeroed
#pragma unroll
    for (int i = 0; i < N1 - 1; ++i) *reinterpret_cast<u32x4*>(lds + in1[i]) = i1[i];
    if (tid < N1_ROWS - (N1 - 1) * 256) *reinterpret_cast<u32x4*>(lds + in1[N1 - 1]) = i1[N1 - 1];
    *reinterpret_cast<u32x4*>(lds + in2[0]) = i2[0];
    *reinterpret_cast<u32x4*>(lds + in2[1]) = i2[1];
    if (tid < P2 * 8 - 512) *reinterpret_cast<u32x4*>(lds + in2[2]) = i2[2];
    *reinterpret_cast<u32x4*>(lds + in3[0]) = i3h[0];
    *reinterpret_cast<u32x4*>(lds + in3[0] + (DX_D3L - DX_D3H)) = i3l[0];
    if (tid < P3 * 8 - 256) {
        *reinterpret_cast<u32x4*>(lds + in3[1]) = i3h[1];
        *reinterpret_cast<u32x4*>(lds + in3[1] + (DX_D3L - DX_D3H)) = i3l[1];
    }
    __syncthreads();

    STACK_STAMP(a.prof, 0, 3)
#pragma unroll
    for (int ks = 0; ks < 18; ++ks) { pin_agpr(wf3h[ks]); pin_agpr(wf3l[ks]); }
    // ---- conv3^T: da2[(iy,ix)][c] = sum_{tap,o} w3[o][tap][c] da3[(iy-ky, ix-kx)][o], times relu'(a2); a unit is a
    // third (6 k-steps) of a row block's reduction
    {
        auto load = [&](bf16x8 (&xh)[6], bf16x8 (&xl)[6], int rb, int third) {
            const bf16_t* base = lds + ld3[rb];
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                const int ks = 6 * third + j, tap = ks >> 1, ky = tap / 3, kx = tap - 3 * ky;
                const bf16_t* q = base - ky * D3_RP - kx * D3_CP + (ks & 1) * 32;
                xh[j] = *reinterpret_cast<const bf16x8*>(q);
                xl[j] = *reinterpret_cast<const bf16x8*>(q + (DX_D3L - DX_D3H));
            }
        };
        auto mul = [&](const bf16x8 (&xh)[6], const bf16x8 (&xl)[6], int third, f32x4 (&c)[3]) {
            if (third == 0) { c[0] = zero4; c[1] = zero4; c[2] = zero4; }
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                c[0] = mfma_wx(wf3h[6 * third + j], xh[j], c[0]);
                c[1] = mfma_wx(wf3l[6 * third + j], xh[j], c[1]);
                c[2] = mfma_wx(wf3h[6 * third + j], xl[j], c[2]);
            }
        };
        auto store = [&](const f32x4 (&c)[3], int rb) {
            if (rb * 16 + l15 < P2) {
                uint2* cell = reinterpret_cast<uint2*>(lds + st3[rb]);      // this wave's channels only
                uint2 h, l;
                gate_split4(*cell, c[0] + (c[1] + c[2]), h, l);
                *cell = h;
                *reinterpret_cast<uint2*>(lds + st3[rb] + (DX_A2L - DX_A2H)) = l;
            }
        };
        bf16x8 xah[6], xal[6], xbh[6], xbl[6];
        f32x4 ca[3], cb[3];
        load(xah, xal, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < 3 * NRB3; ++u) {
            const int rb = u / 3, third = u - 3 * rb;
            if (u + 1 < 3 * NRB3) {
                if (u & 1) load(xah, xal, (u + 1) / 3, (u + 1) % 3); else load(xbh, xbl, (u + 1) / 3, (u + 1) % 3);
            }
            if (u & 1) { if (rb & 1) mul(xbh, xbl, third, cb); else mul(xbh, xbl, third, ca); }
            else { if (rb & 1) mul(xah, xal, third, cb); else mul(xah, xal, third, ca); }
            if (third == 0 && rb > 0) { if (rb & 1) store(ca, rb - 1); else store(cb, rb - 1); }
            STACK_SCHED(18, 2, 3, 3)
            __builtin_amdgcn_sched_barrier(0);
            if (u < 16) {       // two of conv2^T's 32 weight fragments per unit, under the MFMAs
                wf2h[u >> 1][u & 1] = w2_frag(u >> 1, u & 1, 0);
                wf2l[u >> 1][u & 1] = w2_frag(u >> 1, u & 1, a.w_plane);
            }
        }
        if ((NRB3 - 1) & 1) store(cb, NRB3 - 1); else store(ca, NRB3 - 1);
    }
    STACK_STAMP(a.prof, 0, 4)
    __syncthreads();                    // the a2 grids now hold da2 (hi, lo)
    {
        uint4* dst = reinterpret_cast<uint4*>(a.da2 + (long)img * P2 * 64);
        uint4* dstl = reinterpret_cast<uint4*>(a.da2 + a.da2_plane + (long)img * P2 * 64);
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            if (hh == 0 && (i < 2 || tid < P2 * 8 - 512)) {     // both workgroups of the pair hold the same da2
                dst[tid + i * 256] = *reinterpret_cast<const uint4*>(lds + in2[i]);
                dstl[tid + i * 256] = *reinterpret_cast<const uint4*>(lds + in2[i] + (DX_A2L - DX_A2H));
            }
        }
    }
#pragma unroll
    for (int t = 0; t < 8; ++t) { pin_agpr(wf2h[t][0]); pin_agpr(wf2h[t][1]); pin_agpr(wf2l[t][0]); pin_agpr(wf2l[t][1]); }
    PatchRegs pr;
    if constexpr (W1) patch_issue_at<7>(pr, a.frames + f0 * a.frame_stride, frame_bytes, tid);     // flies under conv2^T
    STACK_STAMP(a.prof, 0, 5)
    // ---- conv2^T for this wave's parity class: input pixel (2 iy2 + py, 2 ix2 + px), taps (py + 2a, px + 2b)
    {
        auto load = [&](bf16x8 (&xh)[8], bf16x8 (&xl)[8], int rb) {
            const bf16_t* base = lds + ld2[rb];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const bf16_t* q = base - (t >> 1) * D2_RP - (t & 1) * D2_CP;
                xh[2 * t] = *reinterpret_cast<const bf16x8*>(q);
                xh[2 * t + 1] = *reinterpret_cast<const bf16x8*>(q + 32);
                xl[2 * t] = *reinterpret_cast<const bf16x8*>(q + (DX_A2L - DX_A2H));
                xl[2 * t + 1] = *reinterpret_cast<const bf16x8*>(q + (DX_A2L - DX_A2H) + 32);
            }
        };
        auto mul = [&](const bf16x8 (&xh)[8], const bf16x8 (&xl)[8], f32x4 (&c0)[3], f32x4 (&c1)[3]) {
#pragma unroll
            for (int i = 0; i < 3; ++i) { c0[i] = zero4; c1[i] = zero4; }
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                c0[0] = mfma_wx(wf2h[t][0], xh[t], c0[0]);
                c1[0] = mfma_wx(wf2h[t][1], xh[t], c1[0]);
                c0[1] = mfma_wx(wf2l[t][0], xh[t], c0[1]);
                c1[1] = mfma_wx(wf2l[t][1], xh[t], c1[1]);
                c0[2] = mfma_wx(wf2h[t][0], xl[t], c0[2]);
                c1[2] = mfma_wx(wf2h[t][1], xl[t], c1[2]);
            }
        };
        auto store = [&](const f32x4 (&c0)[3], const f32x4 (&c1)[3], int rb) {
            if ((part * NRB2W + rb) * 16 + l15 < 100) {
                uint2* cell = reinterpret_cast<uint2*>(lds + st2[rb]);      // this class's pixels only
                uint2* celll = reinterpret_cast<uint2*>(lds + st2[rb] + DX_A1L);
                uint2 h, l;
                gate_split4(cell[0], c0[0] + (c0[1] + c0[2]), h, l);
                cell[0] = h; celll[0] = l;
                gate_split4(cell[4], c1[0] + (c1[1] + c1[2]), h, l);        // + 16 channels
                cell[4] = h; celll[4] = l;
            }
        };
        bf16x8 xah[8], xal[8], xbh[8], xbl[8];
        f32x4 ca0[3], ca1[3], cb0[3], cb1[3];
        load(xah, xal, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int rb = 0; rb < NRB2W; ++rb) {        // (part 1's fourth block is all padding: computed, never stored)
            if (rb + 1 < NRB2W) { if (rb & 1) load(xah, xal, rb + 1); else load(xbh, xbl, rb + 1); }
            if (rb & 1) mul(xbh, xbl, cb0, cb1); else mul(xah, xal, ca0, ca1);
            if (rb > 0) { if (rb & 1) store(ca0, ca1, rb - 1); else store(cb0, cb1, rb - 1); }
            STACK_SCHED(48, 1, 3, 2)
            __builtin_amdgcn_sched_barrier(0);
        }
        if ((NRB2W - 1) & 1) store(cb0, cb1, NRB2W - 1); else store(ca0, ca1, NRB2W - 1);
    }
    STACK_STAMP(a.prof, 0, 6)
    __syncthreads();                    // the a1 rows now hold da1 (hi), the rows behind them its lo plane
    {
        uint4* dst = reinterpret_cast<uint4*>(a.da1 + (long)img * P1 * 32);
        uint4* dstl = reinterpret_cast<uint4*>(a.da1 + a.da1_plane + (long)img * P1 * 32);
#pragma unroll
        for (int i = 0; i < N1; ++i) {          // this workgroup's rows
            if (i < N1 - 1 || tid < N1_ROWS - (N1 - 1) * 256) {
                dst[g1[i]] = *reinterpret_cast<const uint4*>(lds + in1[i]);
                dstl[g1[i]] = *reinterpret_cast<const uint4*>(lds + in1[i] + DX_A1L);
            }
        }
    }
    STACK_STAMP(a.prof, 0, 7)
    if constexpr (W1) {
        // ---- conv1 weight gradient: dW1[o][(c, ky, kx)] = sum_px da1[px][o] patch[c][4 oy + ky][4 ox + kx];
        // wave w = kernel channel w; this workgroup's 200 (PAIR) / 400 pixels in 7 / 13 chunks of 32 (pad pixels read
        // a zeroed strip behind the patch)
        constexpr int NPIX = PAIR ? 200 : 400, NCH = PAIR ? 6 : 12;      // NCH + 1 chunks
        bf16_t* patch = lds + DX_A2H;
        constexpr int ZOFF = DX_A2H + 4 * FH * FW;
        static_assert(ZOFF + 64 <= DX_LDS_ELEMS, "zero strip");
        patch_commit<7>(pr, frame_bytes, patch, tid);       // every wave is past the barrier that closed conv2^T
        if (tid < 8) *reinterpret_cast<uint4*>(lds + ZOFF + tid * 8) = make_uint4(0, 0, 0, 0);
        __syncthreads();
        const int tg = lane >> 4, tq = (lane >> 2) & 3, tp = lane & 3;
        // [32][256] tile <-> slab slice blockIdx.x (kernel layout == PyTorch's (c, ky, kx)); LOOP: the raw sums of the
        // workgroup's earlier images come back from it, the last image leaves them scaled
        int boff[4];
        bf16x8 ones;
#pragma unroll
        for (int j = 0; j < 4; ++j) boff[j] = wave * FH * FW + (2 * j + (tp >> 1)) * FW + 4 * (tp & 1);
#pragma unroll
        for (int i = 0; i < 8; ++i) ones[i] = (__bf16)1.0f;
        auto load = [&](bf16x8 (&fa)[4], bf16x8 (&fb)[4], int ch) {
            const int pp0 = ch * 32 + 8 * tg + tq, pp1 = pp0 + 4;
            const bool in0 = pp0 < NPIX, in1 = pp1 < NPIX;
            const int y0 = in0 ? (PAIR ? 2 * (pp0 / 20) + hh : pp0 / 20) : 0, x0 = in0 ? pp0 % 20 : 0;
            const int y1 = in1 ? (PAIR ? 2 * (pp1 / 20) + hh : pp1 / 20) : 0, x1 = in1 ? pp1 % 20 : 0;
            int a0 = (in0 ? (y0 * 20 + x0) * D1_CP : ZOFF) + 4 * tp, a1o = (in1 ? (y1 * 20 + x1) * D1_CP : ZOFF) + 4 * tp;
            int l0 = in0 ? DX_A1L : 0, l1 = in1 ? DX_A1L : 0;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                fa[i] = tr_frag(lds + a0 + i * 16, lds + a1o + i * 16);
                fa[2 + i] = tr_frag(lds + a0 + l0 + i * 16, lds + a1o + l1 + i * 16);
            }
            const bf16_t* b0 = patch + (4 * y0) * FW + 4 * x0;
            const bf16_t* b1 = patch + (4 * y1) * FW + 4 * x1;
#pragma unroll
            for (int j = 0; j < 4; ++j) fb[j] = tr_frag(b0 + boff[j], b1 + boff[j]);
        };
        auto mma = [&](const bf16x8 (&fa)[4], const bf16x8 (&fb)[4]) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    acc1[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb[j], acc1[i][j], 0, 0, 0);
                    acc1[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[2 + i], fb[j], acc1[i][j], 0, 0, 0);
                }
                if (wave == 0) {
                    accb1[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], ones, accb1[i], 0, 0, 0);
                    accb1[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[2 + i], ones, accb1[i], 0, 0, 0);
                }
            }
        };
        bf16x8 fa0[4], fb0[4], fa1[4], fb1[4];
        load(fa0, fb0, 0);
        if constexpr (LOOP) {
            // unrolled: the next image's conv3 weight fragments, three per pair of chunks, ride under the MFMAs
#pragma unroll
            for (int ch = 0; ch < NCH; ch += 2) {
                load(fa1, fb1, ch + 1);
                mma(fa0, fb0);
                issue_w3(w3src, 3 * (ch / 2), 3 * (ch / 2) + 3);
                load(fa0, fb0, ch + 2);
                mma(fa1, fb1);
            }
            static_assert(3 * (NCH / 2) == 18, "all 18 fragment pairs are requested by the tail");
            const int nxt = img + (int)gridDim.x;
            issue_inputs(nxt < a.batch ? nxt : img, tid, g1);     // (past the end: this image's again, never used)
        } else {
#pragma unroll 1
            for (int ch = 0; ch < NCH; ch += 2) {   // chunks ch, ch + 1; fragments one chunk ahead
                load(fa1, fb1, ch + 1);
                mma(fa0, fb0);
                load(fa0, fb0, ch + 2);
                mma(fa1, fb1);
            }
        }
        mma(fa0, fb0);                              // the last chunk
        STACK_STAMP(a.prof, 0, 8)
    }
    }   // images of this workgroup
    if constexpr (W1) {
        // [32][256] tile -> slab slice blockIdx.x (kernel layout == PyTorch's (c, ky, kx))
        const int lane = (int)threadIdx.x & 63, g4 = lane >> 4, l15 = lane & 15;
        float* dw = a.gw1 + (long)blockIdx.x * a.ws1;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            int o = i * 16 + g4 * 4;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                int k = wave * 64 + j * 16 + l15;
#pragma unroll
                for (int r = 0; r < 4; ++r) dw[(long)(o + r) * 256 + k] = acc1[i][j][r] * a.scale1;
            }
            if (wave == 0 && l15 == 0 && a.gb1) {
#pragma unroll
                for (int r = 0; r < 4; ++r) a.gb1[(long)blockIdx.x * a.bs1 + o + r] = accb1[i][r];
            }
        }
    }
}

}  // namespace

// conv1 weight-gradient slab slices the launch below writes when it is given the frames: one per workgroup -- two
// workgroups per image up to 256 images, 256 workgroups walking the batch above
PDA_EXPORT int pda_cnn_dgrad_stack_x3_slices(int batch) { return batch <= 256 ? 2 * batch : 256; }

// the chain at fp32 precision: da3 / da2 / da1 and the weights as (hi, lo) planes, a2 / a1 hi planes (relu masks)
// w3t / w2t: the TRANSPOSED weight shadows [(tap, c)][o].  frames != null: the launch also writes conv1's weight
// gradient, pda_cnn_dgrad_stack_x3_slices(batch) slab slices (ws1 / bs1 floats apart)
PDA_EXPORT void pda_cnn_dgrad_stack_x3(const void* da3, long da3_plane, const void* a2, const void* a1,
                                       const void* w3t, const void* w2t, long w_plane, void* da2, long da2_plane,
                                       void* da1, long da1_plane, int batch, const void* frames, const void* idx,
                                       void* gw1, void* gb1, long ws1, long bs1, float scale1, const void* ref,
                                       long frame_stride, hipStream_t stream) {
    DgradStackX3Args a = {(const bf16_t*)da3, (const bf16_t*)a2, (const bf16_t*)a1, (const bf16_t*)w3t,
                          (const bf16_t*)w2t, (bf16_t*)da2, (bf16_t*)da1, batch, da3_plane, w_plane, da2_plane, da1_plane,
                          g_pda_stack_prof, (const uint8_t*)frames, (const int*)idx, (float*)gw1, (float*)gb1, ws1, bs1,
                          scale1, (const int*)ref, frame_stride > 0 ? frame_stride : 4L * 84 * 84};
    static_assert(DX_LDS_ELEMS * 2 <= 160 * 1024, "LDS budget");
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(cnn_dgrad_stack_x3_kernel<false>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(cnn_dgrad_stack_x3_kernel<true>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(cnn_dgrad_stack_x3_kernel<false, false>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(cnn_dgrad_stack_x3_kernel<true, false>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    if (frames && batch > 256)  // pda_cnn_dgrad_stack_x3_slices(batch) workgroups walk the batch, one dW1 slab slice each
        hipLaunchKernelGGL((cnn_dgrad_stack_x3_kernel<true, false>), dim3(pda_cnn_dgrad_stack_x3_slices(batch)), dim3(256),
                           (size_t)DX_LDS_ELEMS * 2, stream, a);
    else if (frames)
        hipLaunchKernelGGL(cnn_dgrad_stack_x3_kernel<true>, dim3(2 * batch), dim3(256), (size_t)DX_LDS_ELEMS * 2, stream, a);
    else if (batch > 256)       // more images than CUs: one workgroup per image (no conv3^T computed twice)
        hipLaunchKernelGGL((cnn_dgrad_stack_x3_kernel<false, false>), dim3(batch), dim3(256), (size_t)DX_LDS_ELEMS * 2, stream, a);
    else
        hipLaunchKernelGGL(cnn_dgrad_stack_x3_kernel<false>, dim3(2 * batch), dim3(256), (size_t)DX_LDS_ELEMS * 2, stream, a);
}
