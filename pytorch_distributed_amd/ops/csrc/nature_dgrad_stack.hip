// ---- data-gradient chain per image: da3 -> (conv3^T, relu'(a2)) -> da2 -> (conv2^T, relu'(a1)) -> da1 -----
// Replaces four launches of the layer-wise backward (two "dcol" GEMMs writing [pixels][taps x channels]
// matrices and two col2im gathers reading them back: 34 us of the 141 us step at batch 128, 0.9 GB of
// scratch at batch 4096).  One workgroup carries an image: da3, the forward activations a2 / a1 (relu
// masks) and the intermediate da2 live in LDS; the gradients replace the activations in place and leave
// with coalesced copies.  Both layers are gather-form GEMMs over the taps (conv2: per input-pixel parity
// class, one class per wave).  The forward weight shadows w[o][(ky,kx,c)] are staged ONCE per workgroup
// through LDS as per-tap [o][c] tiles and picked up with the transposing ds_read_b64_tr_b16 as
// B-fragments (rows = c, k = o) that stay in registers for every image of the workgroup.
#include "nature_cnn.h"

namespace {

struct DgradStackArgs {
    const bf16_t* da3; const bf16_t* a2; const bf16_t* a1; const bf16_t* w3; const bf16_t* w2;
    bf16_t* da2; bf16_t* da1; int batch;
    // optional tail (W1): conv1's weight gradient of the same image, from da1 while it is still in LDS
    const uint8_t* frames; const int* idx; float* gw1; float* gb1; long ws1, bs1; float scale1;
    const int* ref; long frame_stride;      // frame_index() in nature_cnn.h
    long long* prof;    // optional phase stamps of workgroup 0 (pda_cnn_stack_prof)
};

// LDS plan (bf16 elements).  da3 and a2 / da2 live in zero-bordered 11 x 11 grids, so every tap of the two
// transposed convolutions is a constant offset from the lane's base (no per-tap bounds test); pixel and row
// pitches from tools/probes/lds_conflicts.py (4.6 - 4.7 LDS cycles per ds_read_b128, 4 = conflict-free).

struct DgradRegs { u32x4 a1[7], a2[3], d3[2]; };    // one image's inputs in flight: 1600 + 648 + 392 16-byte chunks

__device__ __forceinline__ void dgrad_issue(DgradRegs& r, const DgradStackArgs& a, int img, int tid) {
    const u32x4* s1 = reinterpret_cast<const u32x4*>(a.a1 + (long)img * 400 * 32);
    const u32x4* s2 = reinterpret_cast<const u32x4*>(a.a2 + (long)img * 81 * 64);
    const u32x4* s3 = reinterpret_cast<const u32x4*>(a.da3 + (long)img * 49 * 64);
#pragma unroll
    for (int i = 0; i < 7; ++i) { int t = tid + i * 256; r.a1[i] = s1[t < 1600 ? t : 0]; }
#pragma unroll
    for (int i = 0; i < 3; ++i) { int t = tid + i * 256; r.a2[i] = s2[t < 648 ? t : 0]; }
#pragma unroll
    for (int i = 0; i < 2; ++i) { int t = tid + i * 256; r.d3[i] = s3[t < 392 ? t : 0]; }
}


// MULTI: workgroups loop over images with the next image's inputs prefetched into registers.
// W1: the workgroup also accumulates conv1's weight gradient of its images (as conv1_wgrad_kernel does, wave w =
// kernel channel w, im2col fragments gathered from the bf16 frame patch in place) with da1 read straight from
// the LDS rows it was just written to: one launch, one round trip of da1 through memory and one dependent
// index -> frame load chain less per step.
template <bool MULTI, bool W1>
__global__ __launch_bounds__(256) void cnn_dgrad_stack_kernel(DgradStackArgs a) {
    extern __shared__ __attribute__((aligned(16))) bf16_t lds[];
    constexpr int P1 = 400, P2 = 81, P3 = 49, NRB3 = 6, NRB2 = 7;
    constexpr int FH = 84, FW = 84;
    constexpr long frame_bytes = 4L * FH * FW;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g4 = lane >> 4, l15 = lane & 15;
    const int G = gridDim.x;

    STACK_STAMP(a.prof, 0, 0)
    DgradRegs in;
    int img = blockIdx.x;
    PatchRegs pr;
    long f_nxt = 0;
    if constexpr (W1) {
        const int i0 = img < a.batch ? img : 0;
        long f0 = a.idx ? (long)a.idx[i0] : (long)i0;
        if (a.ref) f0 = (long)a.ref[f0];
        patch_issue_at<7>(pr, a.frames + f0 * a.frame_stride, frame_bytes, tid);
        if constexpr (MULTI) f_nxt = frame_index(a.idx, a.ref, img + G, a.batch);
    }
    dgrad_issue(in, a, img < a.batch ? img : 0, tid);       // flies under the weight staging

    // ---- weights: per-tap [o][c] tiles (T layout) -> fragments with rows = input channel, k = output channel
    bf16x8 wf3[18];         // conv3^T: this wave's 16 input channels; k-step = (tap, o half)
    bf16x8 wf2[8][2];       // conv2^T: this wave's parity class; (tap a,b ; o half) x 2 channel blocks
    {   // all 34 chunk loads of a thread are in flight together (a load -> store loop pays the L2 latency per trip)
        u32x4 r3[18], r2[16];
#pragma unroll
        for (int i = 0; i < 18; ++i) {
            int t = tid + i * 256, rg = t & 7, o = (t >> 3) & 63, tap = t >> 9;
            r3[i] = *reinterpret_cast<const u32x4*>(a.w3 + (long)o * 576 + tap * 64 + rg * 8);
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            int t = tid + i * 256, rg = t & 3, o = (t >> 2) & 63, tap = t >> 8;
            r2[i] = *reinterpret_cast<const u32x4*>(a.w2 + (long)o * 512 + tap * 32 + rg * 8);
        }
#pragma unroll
        for (int i = 0; i < 18; ++i) {
            int t = tid + i * 256, rg = t & 7, o = (t >> 3) & 63, tap = t >> 9;
            *reinterpret_cast<u32x4*>(lds + tap * 4096 + o * 64 + (((2 * rg) ^ swz_t<64>(o)) << 2)) = r3[i];
        }
        __syncthreads();
#pragma unroll
        for (int ks = 0; ks < 18; ++ks)
            wf3[ks] = read_frag<true, 64, 32>(lds + (ks >> 1) * 4096, wave * 16, ks & 1, lane);
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            int t = tid + i * 256, rg = t & 3, o = (t >> 2) & 63, tap = t >> 8;
            *reinterpret_cast<u32x4*>(lds + tap * 2048 + o * 32 + (((2 * rg) ^ swz_t<32>(o)) << 2)) = r2[i];
        }
    }
    __syncthreads();
    STACK_STAMP(a.prof, 0, 1)
    const int py = wave >> 1, px = wave & 1;
#pragma unroll
    for (int t = 0; t < 8; ++t) {       // t = (a * 2 + b) * 2 + half
        int ta = t >> 2, tb = (t >> 1) & 1, half = t & 1;
        int tap = (py + 2 * ta) * 4 + (px + 2 * tb);
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) wf2[t][nb] = read_frag<true, 32, 32>(lds + tap * 2048, nb * 16, half, lane);
    }
#pragma unroll
    for (int ks = 0; ks < 18; ++ks) pin_agpr(wf3[ks]);      // MFMA operands straight from the accumulation registers
#pragma unroll
    for (int t = 0; t < 8; ++t) { pin_agpr(wf2[t][0]); pin_agpr(wf2[t][1]); }
    __syncthreads();
    // the two grids start out all zero; images only ever write their interiors
    for (int t = tid; t < (D_LDS_ELEMS - D_A2OFF) / 8; t += 256)
        *reinterpret_cast<uint4*>(lds + D_A2OFF + t * 8) = make_uint4(0, 0, 0, 0);

    // ---- per-lane LDS offsets (elements), opaque so the divisions stay out of the phases
    int in1[7], in2[3], in3[2];             // where the input chunks land (and the gradients leave from)
    int ld3[NRB3], st3[NRB3], ld2[NRB2], st2[NRB2];
#pragma unroll
    for (int i = 0; i < 7; ++i) {
        int t = tid + i * 256;
        t = t < P1 * 4 ? t : 0;
        in1[i] = (t >> 2) * D1_CP + (t & 3) * 8;
        asm volatile("" : "+v"(in1[i]));
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        int t = tid + i * 256;
        t = t < P2 * 8 ? t : 0;
        int pix = t >> 3, y = pix / 9, x = pix - 9 * y;
        in2[i] = D_A2OFF + (y + 1) * D2_RP + (x + 1) * D2_CP + (t & 7) * 8;
        asm volatile("" : "+v"(in2[i]));
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        int t = tid + i * 256;
        t = t < P3 * 8 ? t : 0;
        int pix = t >> 3, y = pix / 7, x = pix - 7 * y;
        in3[i] = D_D3OFF + (y + 2) * D3_RP + (x + 2) * D3_CP + (t & 7) * 8;
        asm volatile("" : "+v"(in3[i]));
    }
#pragma unroll
    for (int rb = 0; rb < NRB3; ++rb) {
        int m = rb * 16 + l15;
        m = m < P2 ? m : 0;
        int iy = m / 9, ix = m - 9 * iy;
        ld3[rb] = D_D3OFF + (iy + 2) * D3_RP + (ix + 2) * D3_CP + g4 * 8;
        st3[rb] = D_A2OFF + (iy + 1) * D2_RP + (ix + 1) * D2_CP + wave * 16 + 4 * g4;
        asm volatile("" : "+v"(ld3[rb]), "+v"(st3[rb]));
    }
#pragma unroll
    for (int rb = 0; rb < NRB2; ++rb) {
        int q = rb * 16 + l15;
        q = q < 100 ? q : 0;
        int iy2 = q / 10, ix2 = q - 10 * iy2;
        ld2[rb] = D_A2OFF + (iy2 + 1) * D2_RP + (ix2 + 1) * D2_CP + g4 * 8;
        st2[rb] = ((2 * iy2 + py) * 20 + 2 * ix2 + px) * D1_CP + 4 * g4;
        asm volatile("" : "+v"(ld2[rb]), "+v"(st2[rb]));
    }
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    // conv1 weight-gradient state (W1): transposing-read roles, accumulators
    bf16_t* patch = lds + D_LDS_ELEMS;
    const int tg = lane >> 4, tq = (lane >> 2) & 3, tp = lane & 3;
    f32x4 acc1[2][4], accb1[2];
    int boff[4];
    bf16x8 ones;
    if constexpr (W1) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            accb1[i] = zero4;
#pragma unroll
            for (int j = 0; j < 4; ++j) acc1[i][j] = zero4;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) boff[j] = wave * FH * FW + (2 * j + (tp >> 1)) * FW + 4 * (tp & 1);
#pragma unroll
        for (int i = 0; i < 8; ++i) ones[i] = (__bf16)1.0f;
    }

    auto process = [&](int img) __attribute__((always_inline)) {
        __syncthreads();                    // previous image's gradients have left (and the grids are zeroed)
        if constexpr (W1) patch_commit<7>(pr, frame_bytes, patch, tid);
#pragma unroll
        for (int i = 0; i < 6; ++i) *reinterpret_cast<u32x4*>(lds + in1[i]) = in.a1[i];
        if (tid < P1 * 4 - 1536) *reinterpret_cast<u32x4*>(lds + in1[6]) = in.a1[6];
        *reinterpret_cast<u32x4*>(lds + in2[0]) = in.a2[0];
        *reinterpret_cast<u32x4*>(lds + in2[1]) = in.a2[1];
        if (tid < P2 * 8 - 512) *reinterpret_cast<u32x4*>(lds + in2[2]) = in.a2[2];
        *reinterpret_cast<u32x4*>(lds + in3[0]) = in.d3[0];
        if (tid < P3 * 8 - 256) *reinterpret_cast<u32x4*>(lds + in3[1]) = in.d3[1];
        __syncthreads();
        if constexpr (MULTI) dgrad_issue(in, a, img + G < a.batch ? img + G : 0, tid);   // unconditional: see cnn_fwd_stack_kernel
        if constexpr (MULTI && W1) {
            patch_issue_at<7>(pr, a.frames + f_nxt * a.frame_stride, frame_bytes, tid);
            f_nxt = frame_index(a.idx, a.ref, img + 2 * G, a.batch);
        }

        STACK_STAMP(a.prof, 0, 3)
        // ---- conv3^T: da2[(iy,ix)][c] = sum_{tap,o} w3[o][tap][c] da3[(iy-ky, ix-kx)][o], times relu'(a2).
        // Swapped product (weights as the A operand): a lane ends with 4 consecutive channels of one pixel.
        {
            auto load = [&](bf16x8 (&xf)[18], int rb) {
                const bf16_t* base = lds + ld3[rb];
#pragma unroll
                for (int tap = 0; tap < 9; ++tap) {
                    int ky = tap / 3, kx = tap - 3 * ky;
                    xf[2 * tap] = *reinterpret_cast<const bf16x8*>(base - ky * D3_RP - kx * D3_CP);
                    xf[2 * tap + 1] = *reinterpret_cast<const bf16x8*>(base - ky * D3_RP - kx * D3_CP + 32);
                }
            };
            auto mul = [&](const bf16x8 (&xf)[18], f32x4& c0, f32x4& c1) {
                c0 = zero4; c1 = zero4;
#pragma unroll
                for (int ks = 0; ks < 18; ks += 2) {
                    c0 = mfma_wx(wf3[ks], xf[ks], c0);
                    c1 = mfma_wx(wf3[ks + 1], xf[ks + 1], c1);
                }
            };
            auto store = [&](const f32x4& c0, const f32x4& c1, int rb) {
                if (rb * 16 + l15 < P2) {
                    uint2* cell = reinterpret_cast<uint2*>(lds + st3[rb]);      // this wave's channels only
                    *cell = gate_pack4(*cell, c0 + c1);
                }
            };
            bf16x8 xa[18], xb[18];
            f32x4 ca0, ca1, cb0, cb1;
            load(xa, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int rb = 0; rb < NRB3; ++rb) {
                auto& cur = (rb & 1) ? xb : xa;
                auto& nxt = (rb & 1) ? xa : xb;
                if (rb + 1 < NRB3) load(nxt, rb + 1);
                if (rb & 1) mul(cur, cb0, cb1); else mul(cur, ca0, ca1);
                if (rb > 0) { if (rb & 1) store(ca0, ca1, rb - 1); else store(cb0, cb1, rb - 1); }
                STACK_SCHED(18, 1, 1, 2)
                __builtin_amdgcn_sched_barrier(0);
            }
            if ((NRB3 - 1) & 1) store(cb0, cb1, NRB3 - 1); else store(ca0, ca1, NRB3 - 1);
        }
        STACK_STAMP(a.prof, 0, 4)
        __syncthreads();                    // the a2 grid now holds da2
        {
            uint4* dst = reinterpret_cast<uint4*>(a.da2 + (long)img * P2 * 64);
            uint4 v0 = *reinterpret_cast<const uint4*>(lds + in2[0]), v1 = *reinterpret_cast<const uint4*>(lds + in2[1]);
            dst[tid] = v0; dst[tid + 256] = v1;
            if (tid < P2 * 8 - 512) dst[tid + 512] = *reinterpret_cast<const uint4*>(lds + in2[2]);
        }
        STACK_STAMP(a.prof, 0, 5)
        // ---- conv2^T for this wave's parity class: input pixel (2 iy2 + py, 2 ix2 + px), taps (py + 2a, px + 2b)
        // read da2 at (iy2 - a, ix2 - b); both 16-channel blocks of a1 per fragment
        {
            auto load = [&](bf16x8 (&xf)[8], int rb) {
                const bf16_t* base = lds + ld2[rb];
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    xf[2 * t] = *reinterpret_cast<const bf16x8*>(base - (t >> 1) * D2_RP - (t & 1) * D2_CP);
                    xf[2 * t + 1] = *reinterpret_cast<const bf16x8*>(base - (t >> 1) * D2_RP - (t & 1) * D2_CP + 32);
                }
            };
            auto mul = [&](const bf16x8 (&xf)[8], f32x4& c0, f32x4& c1) {
                c0 = zero4; c1 = zero4;
#pragma unroll
                for (int t = 0; t < 8; ++t) {
                    c0 = mfma_wx(wf2[t][0], xf[t], c0);
                    c1 = mfma_wx(wf2[t][1], xf[t], c1);
                }
            };
            auto store = [&](const f32x4& c0, const f32x4& c1, int rb) {
                if (rb * 16 + l15 < 100) {
                    uint2* cell = reinterpret_cast<uint2*>(lds + st2[rb]);      // this class's pixels only
                    cell[0] = gate_pack4(cell[0], c0);
                    cell[4] = gate_pack4(cell[4], c1);                          // + 16 channels
                }
            };
            bf16x8 xa[8], xb[8];
            f32x4 ca0, ca1, cb0, cb1;
            load(xa, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int rb = 0; rb < NRB2; ++rb) {
                auto& cur = (rb & 1) ? xb : xa;
                auto& nxt = (rb & 1) ? xa : xb;
                if (rb + 1 < NRB2) load(nxt, rb + 1);
                if (rb & 1) mul(cur, cb0, cb1); else mul(cur, ca0, ca1);
                if (rb > 0) { if (rb & 1) store(ca0, ca1, rb - 1); else store(cb0, cb1, rb - 1); }
                STACK_SCHED(16, 1, 2, 4)
                __builtin_amdgcn_sched_barrier(0);
            }
            if ((NRB2 - 1) & 1) store(cb0, cb1, NRB2 - 1); else store(ca0, ca1, NRB2 - 1);
        }
        STACK_STAMP(a.prof, 0, 6)
        __syncthreads();                    // the a1 rows now hold da1
        {
            uint4* dst = reinterpret_cast<uint4*>(a.da1 + (long)img * P1 * 32);
            uint4 v[6];
#pragma unroll
            for (int i = 0; i < 6; ++i) v[i] = *reinterpret_cast<const uint4*>(lds + in1[i]);
#pragma unroll
            for (int i = 0; i < 6; ++i) dst[tid + i * 256] = v[i];
            if (tid < P1 * 4 - 1536) dst[tid + 1536] = *reinterpret_cast<const uint4*>(lds + in1[6]);
        }
        STACK_STAMP(a.prof, 0, 7)
        if constexpr (W1) {
            // ---- conv1 weight gradient: dW1[o][(c, ky, kx)] += sum_px da1[px][o] patch[c][4 oy + ky][4 ox + kx];
            // 13 chunks of 32 pixels; pixels >= 400 read the (always zero) first border row of the a2 grid
            auto load = [&](bf16x8 (&fa)[2], bf16x8 (&fb)[4], int ch) {
                int p0 = ch * 32 + 8 * tg + tq, p1 = p0 + 4;
                int a0 = (p0 < P1 ? p0 * D1_CP : D_A2OFF) + 4 * tp, a1o = (p1 < P1 ? p1 * D1_CP : D_A2OFF) + 4 * tp;
#pragma unroll
                for (int i = 0; i < 2; ++i) fa[i] = tr_frag(lds + a0 + i * 16, lds + a1o + i * 16);
                int q0 = p0 < P1 ? p0 : 0, q1 = p1 < P1 ? p1 : 0;
                const bf16_t* b0 = patch + (4 * (q0 / 20)) * FW + 4 * (q0 % 20);
                const bf16_t* b1 = patch + (4 * (q1 / 20)) * FW + 4 * (q1 % 20);
#pragma unroll
                for (int j = 0; j < 4; ++j) fb[j] = tr_frag(b0 + boff[j], b1 + boff[j]);
            };
            auto mma = [&](const bf16x8 (&fa)[2], const bf16x8 (&fb)[4]) {
#pragma unroll
                for (int i = 0; i < 2; ++i) {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc1[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb[j], acc1[i][j], 0, 0, 0);
                    if (wave == 0) accb1[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], ones, accb1[i], 0, 0, 0);
                }
            };
            bf16x8 fa0[2], fb0[4], fa1[2], fb1[4];
            load(fa0, fb0, 0);
#pragma unroll 1
            for (int ch = 0; ch < 12; ch += 2) {        // chunks ch, ch + 1; fragments one chunk ahead
                load(fa1, fb1, ch + 1);
                mma(fa0, fb0);
                load(fa0, fb0, ch + 2);
                mma(fa1, fb1);
            }
            mma(fa0, fb0);                              // chunk 12
        }
    };
    STACK_STAMP(a.prof, 0, 2)
    if (img < a.batch) {
        process(img);
        if constexpr (MULTI)
            for (img += G; img < a.batch; img += G) process(img);
    }
    STACK_STAMP(a.prof, 0, 8)
    if constexpr (W1) {
        // partial [32][256] tile -> slab slice blockIdx.x (kernel layout == PyTorch's (c, ky, kx))
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

// conv3^T -> relu'(a2) -> conv2^T -> relu'(a1): the data gradients da2 [(b, 9x9)][64] and da1 [(b, 20x20)][32]
// of the Nature CNN from da3 [(b, 7x7)][64] in one launch (cnn_dgrad_stack_kernel).  84x84 inputs only.
PDA_EXPORT int pda_cnn_dgrad_stack_supported(int H, int W, int C) { return (H == 84 && W == 84 && C == 4) ? 1 : 0; }

static void launch_dgrad_stack(DgradStackArgs& a, hipStream_t stream) {
    a.prof = g_pda_stack_prof;
    const bool w1 = a.frames != nullptr;
    size_t lds = ((size_t)D_LDS_ELEMS + (w1 ? 4 * 84 * 84 : 0)) * 2;   // >= the 9 x 4096-element weight staging tiles
    static_assert(D_LDS_ELEMS >= 9 * 4096, "weight staging must fit the working set's LDS");
    static bool attr_set = false;
    if (!attr_set) {
        const void* fns[4] = {reinterpret_cast<const void*>(cnn_dgrad_stack_kernel<false, false>),
                              reinterpret_cast<const void*>(cnn_dgrad_stack_kernel<true, false>),
                              reinterpret_cast<const void*>(cnn_dgrad_stack_kernel<false, true>),
                              reinterpret_cast<const void*>(cnn_dgrad_stack_kernel<true, true>)};
        for (const void* f : fns) (void)hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    int blocks = a.batch < 256 ? a.batch : 256;     // one workgroup per CU (a wave owns its SIMD's register file)
    const bool multi = blocks < a.batch;
    if (multi && w1) hipLaunchKernelGGL((cnn_dgrad_stack_kernel<true, true>), dim3(blocks), dim3(256), lds, stream, a);
    else if (multi) hipLaunchKernelGGL((cnn_dgrad_stack_kernel<true, false>), dim3(blocks), dim3(256), lds, stream, a);
    else if (w1) hipLaunchKernelGGL((cnn_dgrad_stack_kernel<false, true>), dim3(blocks), dim3(256), lds, stream, a);
    else hipLaunchKernelGGL((cnn_dgrad_stack_kernel<false, false>), dim3(blocks), dim3(256), lds, stream, a);
}

PDA_EXPORT void pda_cnn_dgrad_stack(const void* da3, const void* a2, const void* a1, const void* w3, const void* w2,
                                    void* da2, void* da1, int batch, hipStream_t stream) {
    DgradStackArgs a = {(const bf16_t*)da3, (const bf16_t*)a2, (const bf16_t*)a1, (const bf16_t*)w3,
                        (const bf16_t*)w2, (bf16_t*)da2, (bf16_t*)da1, batch, nullptr, nullptr, nullptr, nullptr, 0, 0, 1.f,
                        nullptr, 0, nullptr};
    launch_dgrad_stack(a, stream);
}

// same, plus conv1's weight gradient from uint8 frames (slab slice per workgroup: pda_cnn_dgrad_stack_slices(batch))
PDA_EXPORT int pda_cnn_dgrad_stack_slices(int batch) { return batch < 256 ? batch : 256; }
PDA_EXPORT void pda_cnn_dgrad_stack_w1(const void* da3, const void* a2, const void* a1, const void* w3, const void* w2,
                                       void* da2, void* da1, int batch, const void* frames, const void* idx,
                                       void* gw1, void* gb1, long ws1, long bs1, float scale1, const void* ref,
                                       long frame_stride, hipStream_t stream) {
    DgradStackArgs a = {(const bf16_t*)da3, (const bf16_t*)a2, (const bf16_t*)a1, (const bf16_t*)w3,
                        (const bf16_t*)w2, (bf16_t*)da2, (bf16_t*)da1, batch, (const uint8_t*)frames, (const int*)idx,
                        (float*)gw1, (float*)gb1, ws1, bs1, scale1, (const int*)ref,
                        frame_stride > 0 ? frame_stride : 4L * 84 * 84, nullptr};
    launch_dgrad_stack(a, stream);
}
