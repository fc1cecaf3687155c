// Nature-CNN conv1 (8x8 kernel, stride 4, 32 output channels, uint8 planar frames) -- forward and
// weight gradient specialised for gfx950.
//
// The generic implicit-GEMM loader (gemm.h, SrcFrameU8K/T) fetches every 8-byte kernel row of every
// output pixel straight from global memory: 8-byte accesses at 84-byte strides, each input byte
// fetched 4 times, u8->bf16 conversion repeated per use.  At batch 4096 that made conv1 the two most
// expensive launches of the learner step (365 + 470 us of 2.2 ms, profiles/r1_dqn_step_batch4096_rocprof.txt).
// Here one workgroup owns whole images:
//   * the 4x84x84 frame stack (28 KB, gathered through the replay index) is read ONCE with coalesced
//     16-byte loads, converted to bf16 once and kept in LDS (56 KB);
//   * forward: an MFMA A-fragment (pixel, 8 consecutive kx) is two ds_read_b64 of that patch; the
//     32x256 weights live in registers as B-fragments for the whole kernel (64 VGPRs);
//   * weight gradient: per 32 output pixels the im2col rows are expanded LDS->LDS into a [pixel][k]
//     tile and both operands come back through CDNA4's transposing ds_read_b64_tr_b16, accumulators
//     stay in registers across all images of the workgroup (split-K over images, one slab slice each).
#include "gemm.h"
#include "api.h"

namespace {

struct Conv1Geom { int C, H, W, OH, OW; };

// 16 uint8 -> 16 bf16 (exact), as two uint4
__device__ __forceinline__ void u8x16_to_bf16(uint4 v, uint4& lo, uint4& hi) {
    uint2 a = u8x4_to_bf16x4(v.x), b = u8x4_to_bf16x4(v.y), c = u8x4_to_bf16x4(v.z), d = u8x4_to_bf16x4(v.w);
    lo = make_uint4(a.x, a.y, b.x, b.y);
    hi = make_uint4(c.x, c.y, d.x, d.y);
}

// The frame stack of an image moves global -> registers (issue()) -> bf16 patch in LDS (commit()), same
// linear [c][y][x] order.  All loads of an image are issued back to back into a fixed register array
// (a runtime-trip-count loop that converts and stores after every load serialises 7 memory round trips:
// 11 us per image in the first version), and they are issued for image i+1 BEFORE image i is
// multiplied, so their latency hides under the MFMAs.
#define DY_IT 7         // dy rows of one image: up to 7 * 256 16-byte chunks = 448 output pixels x 32 channels
#define PATCH_IT 8      // 16-byte loads per thread: frames up to 8 * 256 * 16 = 32 KB
struct PatchRegs { uint4 v[PATCH_IT]; };

template <int IT = PATCH_IT>
__device__ __forceinline__ void patch_issue(PatchRegs& r, const uint8_t* frames, const int* idx, long frame_bytes,
                                            int img, int tid) {
    long f = idx ? (long)idx[img] : (long)img;
    const uint4* src = reinterpret_cast<const uint4*>(frames + f * frame_bytes);
    int n16 = (int)(frame_bytes >> 4);
#pragma unroll
    for (int i = 0; i < IT; ++i) {
        int t = tid + i * 256;
        r.v[i] = src[t < n16 ? t : 0];      // unconditional: lanes past the end re-read element 0
    }
}
// PLANE_CHUNKS / PLANE_PAD: 16-element chunks per channel plane and bf16 elements of padding after each plane
// (0 / 0 = contiguous image).
// The replay index of an image is itself a global load: fetched inside patch_issue it puts a full memory
// round trip (vmcnt(0)) in front of the frame loads of EVERY image.  Loops over images fetch the index one
// image further ahead (frame_index) and issue from the resolved pointer.
// The image number is made opaque so that the index stays a per-lane VGPR load with a counted wait at its
// use; recognised as wave-uniform it is moved to an SGPR on the spot, behind a vmcnt(0) that also drains the
// frame loads just issued.
__device__ __forceinline__ long frame_index(const int* idx, int img, int batch) {
    int iv = img < batch ? img : 0;
    asm volatile("" : "+v"(iv));
    return idx ? (long)idx[iv] : (long)iv;
}
template <int IT = PATCH_IT>
__device__ __forceinline__ void patch_issue_at(PatchRegs& r, const uint8_t* frame, long frame_bytes, int tid) {
    const uint4* src = reinterpret_cast<const uint4*>(frame);
    int n16 = (int)(frame_bytes >> 4);
#pragma unroll
    for (int i = 0; i < IT; ++i) {
        int t = tid + i * 256;
        r.v[i] = src[t < n16 ? t : 0];
    }
}
template <int IT = PATCH_IT, int PLANE_CHUNKS = 0, int PLANE_PAD = 0>
__device__ __forceinline__ void patch_commit(const PatchRegs& r, long frame_bytes, bf16_t* patch, int tid) {
    int n16 = (int)(frame_bytes >> 4);
#pragma unroll
    for (int i = 0; i < IT; ++i) {
        int t = tid + i * 256;
        if (t < n16) {
            uint4 lo, hi;
            u8x16_to_bf16(r.v[i], lo, hi);
            int pad = 0;
            if constexpr (PLANE_PAD != 0) pad = (t / PLANE_CHUNKS) * PLANE_PAD;
            uint4* dst = reinterpret_cast<uint4*>(patch + (long)t * 16 + pad);
            dst[0] = lo;
            dst[1] = hi;
        }
    }
}

__device__ __forceinline__ bf16x8 patch_frag(const bf16_t* p) {     // 8 consecutive bf16, 8-byte aligned
    uint2 a = *reinterpret_cast<const uint2*>(p);
    uint2 b = *reinterpret_cast<const uint2*>(p + 4);
    uint4 v = make_uint4(a.x, a.y, b.x, b.y);
    return __builtin_bit_cast(bf16x8, v);
}

// scheduling pattern of one row block: per MFMA one LDS read (next block's fragment) and a few VALU
// instructions (previous block's epilogue) ride in its shadow
#define STACK_SCHED(n_mfma, per_mfma_reads_num, per_mfma_reads_den, valu)                        \
    _Pragma("unroll") for (int q_ = 0; q_ < (n_mfma) / (per_mfma_reads_den); ++q_) {           \
        __builtin_amdgcn_sched_group_barrier(0x008, (per_mfma_reads_den), 0);                    \
        __builtin_amdgcn_sched_group_barrier(0x100, (per_mfma_reads_num), 0);                    \
        __builtin_amdgcn_sched_group_barrier(0x002, (valu), 0);                                  \
    }

struct Conv1FwdArgs {
    const uint8_t* frames[GEMM_MAX_GROUPS]; const int* idx[GEMM_MAX_GROUPS]; const bf16_t* w[GEMM_MAX_GROUPS];
    const float* bias[GEMM_MAX_GROUPS]; bf16_t* out[GEMM_MAX_GROUPS];
    int batch; Conv1Geom g; float scale; int act;
    int parts;      // workgroups per image (small batches: an image's 16-pixel row blocks are split)
    long w_plane, out_plane;    // X3: element offsets to the lo planes of the split weights / outputs
};

// grid (blocks, groups); each block loops over images blockIdx.x, += gridDim.x.
// X3 (fp32-precision learner): the frames are exact in bf16, the weights a (hi, lo) pair of planes -- two MFMAs per
// fragment into the same accumulator, 128 VGPRs of weight fragments -- and the output is stored as (hi, lo) planes.
template <int C, bool X3 = false>
__global__ __launch_bounds__(256) void conv1_fwd_kernel(Conv1FwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) bf16_t patch[];
    constexpr int KS = 2 * C;           // k-steps of 32: (c, ky half)
    constexpr int K = 64 * C;
    const int grp = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g4 = lane >> 4, l15 = lane & 15;
    const Conv1Geom g = a.g;
    const int P = g.OH * g.OW, nrb = (P + 15) >> 4;
    const long frame_bytes = (long)C * g.H * g.W;
    // weights as B fragments: lane (n = l15 + 16 nb, k = 32 ks + 8 g4 .. +7)
    bf16x8 wf[KS][2];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
            wf[ks][nb] = *reinterpret_cast<const bf16x8*>(a.w[grp] + (long)(nb * 16 + l15) * K + ks * 32 + g4 * 8);
    bf16x8 wl[X3 ? KS : 1][2];
    if constexpr (X3) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int nb = 0; nb < 2; ++nb)
                wl[ks][nb] = *reinterpret_cast<const bf16x8*>(a.w[grp] + a.w_plane + (long)(nb * 16 + l15) * K + ks * 32 +
                                                              g4 * 8);
    }
    const float b0 = a.bias[grp] ? a.bias[grp][l15] : 0.f, b1 = a.bias[grp] ? a.bias[grp][16 + l15] : 0.f;

    PatchRegs pr;
    const int part = blockIdx.x % a.parts, img0 = blockIdx.x / a.parts, img_step = gridDim.x / a.parts;
    const int rb_beg = part * nrb / a.parts, rb_end = (part + 1) * nrb / a.parts;
    long f_nxt = frame_index(a.idx[grp], img0 + img_step, a.batch);
    if (img0 < a.batch)
        patch_issue_at(pr, a.frames[grp] + frame_index(a.idx[grp], img0, a.batch) * frame_bytes, frame_bytes, tid);
    for (int img = img0; img < a.batch; img += img_step) {
        __syncthreads();                // everyone is done reading the previous patch
        patch_commit(pr, frame_bytes, patch, tid);
        __syncthreads();
        if (img + img_step < a.batch)           // next image's loads fly under this image's MFMAs
            patch_issue_at(pr, a.frames[grp] + f_nxt * frame_bytes, frame_bytes, tid);
        f_nxt = frame_index(a.idx[grp], img + 2 * img_step, a.batch);
        bf16_t* out = a.out[grp] + (long)img * P * 32;
        for (int rb = rb_beg + wave; rb < rb_end; rb += 4) {
            int m = rb * 16 + l15;
            int mm = m < P ? m : 0;
            int oy = mm / g.OW, ox = mm - oy * g.OW;
            const bf16_t* base = patch + (4 * oy) * g.W + 4 * ox;
            f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                int c = ks >> 1, ky = (ks & 1) * 4 + g4;
                bf16x8 af = patch_frag(base + (long)c * g.H * g.W + ky * g.W);
                acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, wf[ks][0], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, wf[ks][1], acc1, 0, 0, 0);
                if constexpr (X3) {
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, wl[ks][0], acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, wl[ks][1], acc1, 0, 0, 0);
                }
            }
            // D layout: col = lane & 15, row = (lane >> 4) * 4 + r
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                int pix = rb * 16 + g4 * 4 + r;
                if (pix < P) {
                    store_split(out + (long)pix * 32 + l15, X3 ? a.out_plane : 0, apply_act(acc0[r] * a.scale + b0, a.act));
                    store_split(out + (long)pix * 32 + 16 + l15, X3 ? a.out_plane : 0,
                                apply_act(acc1[r] * a.scale + b1, a.act));
                }
            }
        }
    }
}

struct Conv1WgradArgs {
    const uint8_t* frames; const int* idx; const bf16_t* dy;    // dy [(b, oy, ox)][32] bf16
    float* gw; float* gb; long slice_stride, bslice_stride;     // slab slice blockIdx.x
    int batch; Conv1Geom g; float scale;
    int parts;      // workgroups per image (small batches: the 32-pixel chunks of an image are split)
    long dy_plane;  // X3: element offset from dy's hi plane to its lo plane
};

// LDS: patch [C*H*W] | dyT [PP][32] (T layout, ROWS = 32) | X3: dyT of the lo plane.
// Both MFMA operands come through the transposing ds_read_b64_tr_b16, whose lanes each supply the address
// of an 8-byte piece (4 consecutive columns of one reduction row = pixel).  For dy that is the [pixel][32]
// tile; for the im2col operand the piece (pixel, c, ky, kx 0..3 or 4..7) already sits in the frame patch at
// patch[c][4 oy + ky][4 ox + kx], so the fragments are gathered from the patch in place: no [pixel][k]
// column tile, no LDS -> LDS expansion pass and no barriers inside an image (the first version expanded 32
// pixels at a time between two barriers: 13.9 us per image at batch 4096).  Wave w owns channel w of the
// kernel (k = 64 w .. 64 w + 63), n-block j = kernel rows 2 j, 2 j + 1.
__device__ __forceinline__ s16x4 lds_tr4(const bf16_t* p) {
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p));
}
// X3 (fp32-precision learner): dy is a (hi, lo) pair of planes, the frames are exact: two MFMAs per fragment pair.
template <int C, bool X3 = false>
__global__ __launch_bounds__(256) void conv1_wgrad_kernel(Conv1WgradArgs a) {
    extern __shared__ __attribute__((aligned(16))) bf16_t lds[];
    constexpr int K = 64 * C;           // 256
    constexpr int NA = X3 ? 4 : 2;      // A fragments per chunk: rows 0-15 / 16-31 of dy hi (and of dy lo)
    static_assert(C == 4, "one kernel channel per wave");
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const Conv1Geom g = a.g;
    const int P = g.OH * g.OW, PP = (P + 31) / 32 * 32, nchunk = PP / 32;
    const long frame_bytes = (long)C * g.H * g.W;
    bf16_t* patch = lds;
    bf16_t* dyt = patch + frame_bytes;
    bf16_t* dyl = dyt + PP * 32;        // (X3)

    f32x4 acc[2][4];
    f32x4 accb[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        accb[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    bf16x8 ones;
#pragma unroll
    for (int i = 0; i < 8; ++i) ones[i] = (__bf16)1.0f;

    // transposing-read roles of this lane: reduction rows (pixels) 8 tg + tq (+ 4), column piece tp
    const int tg = lane >> 4, tq = (lane >> 2) & 3, tp = lane & 3;
    int boff[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) boff[j] = wave * g.H * g.W + (2 * j + (tp >> 1)) * g.W + 4 * (tp & 1);
    auto pixbase = [&](int px) {
        px = px < P ? px : 0;           // pad pixels: their dy rows are zero, any finite operand will do
        int oy = px / g.OW, ox = px - oy * g.OW;
        return (4 * oy) * g.W + 4 * ox;
    };
    auto load = [&](bf16x8 (&fa)[NA], bf16x8 (&fb)[4], int ch) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            fa[i] = read_frag<true, 32, 32>(dyt + ch * 32 * 32, i * 16, 0, lane);
            if constexpr (X3) fa[2 + i] = read_frag<true, 32, 32>(dyl + ch * 32 * 32, i * 16, 0, lane);
        }
        const bf16_t* p0 = patch + pixbase(ch * 32 + 8 * tg + tq);
        const bf16_t* p1 = patch + pixbase(ch * 32 + 8 * tg + tq + 4);
        typedef __attribute__((ext_vector_type(8))) short s16x8;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            s16x4 lo = lds_tr4(p0 + boff[j]), hi = lds_tr4(p1 + boff[j]);
            s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            fb[j] = __builtin_bit_cast(bf16x8, v);
        }
    };
    auto mma = [&](const bf16x8 (&fa)[NA], const bf16x8 (&fb)[4]) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
                if constexpr (X3) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[2 + i], fb[j], acc[i][j], 0, 0, 0);
            }
            if (wave == 0) {
                accb[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], ones, accb[i], 0, 0, 0);
                if constexpr (X3) accb[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[2 + i], ones, accb[i], 0, 0, 0);
            }
        }
    };

    PatchRegs pr;
    uint4 dr[DY_IT], drl[X3 ? DY_IT : 1];
    auto dy_issue = [&](int img) {
        const uint4* dsrc = reinterpret_cast<const uint4*>(a.dy + (long)img * P * 32);
#pragma unroll
        for (int i = 0; i < DY_IT; ++i) {
            int t = tid + i * 256;
            dr[i] = dsrc[t < P * 4 ? t : 0];
        }
        if constexpr (X3) {
            const uint4* lsrc = reinterpret_cast<const uint4*>(a.dy + a.dy_plane + (long)img * P * 32);
#pragma unroll
            for (int i = 0; i < DY_IT; ++i) {
                int t = tid + i * 256;
                drl[i] = lsrc[t < P * 4 ? t : 0];
            }
        }
    };
    const int part = blockIdx.x % a.parts, img0 = blockIdx.x / a.parts, img_step = gridDim.x / a.parts;
    const int ch_beg = part * nchunk / a.parts, ch_end = (part + 1) * nchunk / a.parts;
    long f_nxt = frame_index(a.idx, img0 + img_step, a.batch);
    if (img0 < a.batch) {
        patch_issue_at(pr, a.frames + frame_index(a.idx, img0, a.batch) * frame_bytes, frame_bytes, tid);
        dy_issue(img0);
    }
    for (int img = img0; img < a.batch; img += img_step) {
        __syncthreads();                // every wave is done with the previous image's patch / dy tile
        patch_commit(pr, frame_bytes, patch, tid);
        // dy rows of this image -> T layout [px][32]: 4 x 16-byte chunks per pixel row, chunk pair j = 2 rg
#pragma unroll
        for (int i = 0; i < DY_IT; ++i) {
            int t = tid + i * 256;
            if (t < PP * 4) {
                int px = t >> 2, rg = t & 3;
                uint4 v = px < P ? dr[i] : make_uint4(0, 0, 0, 0);
                *reinterpret_cast<uint4*>(dyt + px * 32 + (((2 * rg) ^ swz_t<32>(px)) << 2)) = v;
                if constexpr (X3) {
                    uint4 vl = px < P ? drl[i] : make_uint4(0, 0, 0, 0);
                    *reinterpret_cast<uint4*>(dyl + px * 32 + (((2 * rg) ^ swz_t<32>(px)) << 2)) = vl;
                }
            }
        }
        __syncthreads();
        if (img + img_step < a.batch) {         // next image's loads fly under this image's MFMAs
            patch_issue_at(pr, a.frames + f_nxt * frame_bytes, frame_bytes, tid);
            dy_issue(img + img_step);
        }
        f_nxt = frame_index(a.idx, img + 2 * img_step, a.batch);
        // chunks of 32 pixels, fragments of chunk i + 1 in flight under the MFMAs of chunk i (two register
        // sets, the loop advances two chunks per trip; the look-ahead past the end re-reads the last chunk)
        bf16x8 fa0[NA], fb0[4], fa1[NA], fb1[4];
        int ch = ch_beg;
        const int last = ch_end - 1;
        load(fa0, fb0, ch);
        while (true) {
            load(fa1, fb1, ch + 1 < last ? ch + 1 : last);
            mma(fa0, fb0);
            STACK_SCHED(8, 2, 1, 3)
            __builtin_amdgcn_sched_barrier(0);
            if (++ch >= ch_end) break;
            load(fa0, fb0, ch + 1 < last ? ch + 1 : last);
            mma(fa1, fb1);
            STACK_SCHED(8, 2, 1, 3)
            __builtin_amdgcn_sched_barrier(0);
            if (++ch >= ch_end) break;
        }
    }
    // this block's partial [32][K] tile -> slab slice blockIdx.x (kernel layout == PyTorch's (c, ky, kx))
    float* dw = a.gw + (long)blockIdx.x * a.slice_stride;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        int o = i * 16 + (lane >> 4) * 4;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            int k = wave * 64 + j * 16 + (lane & 15);
#pragma unroll
            for (int r = 0; r < 4; ++r) dw[(long)(o + r) * K + k] = acc[i][j][r] * a.scale;
        }
        if (wave == 0 && (lane & 15) == 0 && a.gb) {
#pragma unroll
            for (int r = 0; r < 4; ++r) a.gb[(long)blockIdx.x * a.bslice_stride + o + r] = accb[i][r];
        }
    }
}

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
struct FwdStackArgs {
    const uint8_t* frames[GEMM_MAX_GROUPS]; const int* idx[GEMM_MAX_GROUPS];
    const bf16_t* w1[GEMM_MAX_GROUPS]; const float* b1[GEMM_MAX_GROUPS];
    const bf16_t* w2[GEMM_MAX_GROUPS]; const float* b2[GEMM_MAX_GROUPS];
    const bf16_t* w3[GEMM_MAX_GROUPS]; const float* b3[GEMM_MAX_GROUPS];
    bf16_t* a1[GEMM_MAX_GROUPS]; bf16_t* a2[GEMM_MAX_GROUPS]; bf16_t* a3[GEMM_MAX_GROUPS];   // a1 / a2 may be null
    int batch; float scale;
    long long* prof;    // optional phase clock stamps of workgroup (0, 0), thread 0 (tools/probes/stack_kernels_bench.py)
    long w_plane, a1_plane, a2_plane, a3_plane;     // fp32-precision kernel: element offsets hi plane -> lo plane
};

#define A1_PITCH 40     // bf16 elements per a1 pixel row in LDS (32 channels + 8 pad = 5 slots)
#define A2_PITCH 72     // 64 channels + 8 pad = 9 slots
#define FWD_PLANE 7104  // bf16 elements per channel plane of the forward kernel's frame patch: 84 * 84 + 48, == 32 (mod 64) dwords

typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
// volatile LDS pointer: keeps two adjacent 8-byte reads as two ds_read_b64 (2 LDS cycles each, 64-bank rules);
// merged into one ds_read2_b64 they cost 8 cycles under 32-bank rules
typedef const volatile __attribute__((address_space(3))) u32x2* lds_v2;
__device__ __forceinline__ void pin_agpr(bf16x8& v) { asm volatile("" : "+a"(v)); }
// phase stamp: 16 slots per image, first 4 images of the stamped workgroup
#define STACK_STAMP(prof, it, slot)                                                          \
    if ((prof) && (it) < 4 && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0)        \
        (prof)[(it) * 16 + (slot)] = (long long)__builtin_readcyclecounter();

// LDS pitches of the forward kernel's activations (bf16 elements), chosen with tools/probes/lds_conflicts.py
// so that the A-fragment gathers of conv2 (stride-2 pixels of a1) and conv3 (pixels of a2) are served in
// 4.7 and 5.0 LDS cycles per ds_read_b128 (4 = conflict-free; the dense pitches 40 / 72 gave 8 and 10).
#define F1_CP 40        // a1: per pixel (32 channels + 8 pad)
#define F1_RP 808       // a1: per row of 20 pixels (+ 8 pad)
#define F2_CP 80        // a2: per pixel (64 channels + 16 pad)
#define F2_RP 816       // a2: per row of 9 pixels (+ 96 pad)

// one k-step of the swapped product: D[channel][pixel] += W[channel][k] * X[k][pixel].  With the weights as
// the A operand a lane ends up with 4 CONSECUTIVE channels (4 g4 .. + 3) of ONE pixel (l15), which leave as
// one packed 8-byte store instead of four 2-byte ones.
__device__ __forceinline__ f32x4 mfma_wx(bf16x8 w, bf16x8 x, f32x4 acc) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(w, x, acc, 0, 0, 0);
}
__device__ __forceinline__ uint2 relu_pack4(f32x4 v, float s, f32x4 bias) {
    return make_uint2(pack2bf(fmaxf(v[0] * s + bias[0], 0.f), fmaxf(v[1] * s + bias[1], 0.f)),
                      pack2bf(fmaxf(v[2] * s + bias[2], 0.f), fmaxf(v[3] * s + bias[3], 0.f)));
}

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
    long f_nxt = MULTI ? frame_index(a.idx[grp], blockIdx.x + G, a.batch) : 0;
    if ((int)blockIdx.x < a.batch) {       // first image: the index is needed at once, as a scalar
        const long f0 = a.idx[grp] ? (long)a.idx[grp][blockIdx.x] : (long)blockIdx.x;
        patch_issue_at<7>(pr, a.frames[grp] + f0 * frame_bytes, frame_bytes, tid);
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
            patch_issue_at<7>(pr, a.frames[grp] + f_nxt * frame_bytes, frame_bytes, tid);
            f_nxt = frame_index(a.idx[grp], img + 2 * G, a.batch);     // consumed one image from now
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

// ---- the same stack at fp32 precision (split-bf16 planes) ---------------------------------------------------
// Every activation is a (hi, lo) pair of bf16 planes, every weight too (the frames are exact in bf16); a product
// is w_hi x_hi + w_lo x_hi + w_hi x_lo into separate fp32 accumulation chains.  Same LDS-resident data flow as
// cnn_fwd_stack_kernel with both planes of a1 / a2 side by side, one image per workgroup (grid.x == batch).
// Register plan (one wave per SIMD, 512 registers per lane): the weights of a layer are resident only while the
// layer runs -- conv1 (128) and conv2 (128) are fetched at the start, conv3's hi plane (72) under conv2, its lo
// plane (72) under the a2 copy-out -- and a row block's activation fragments (hi + lo) are fetched in two
// half-reductions, double-buffered (128 / 144 registers).
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

__global__ __launch_bounds__(256) void cnn_fwd_stack_x3_kernel(FwdStackArgs a) {
    extern __shared__ __attribute__((aligned(16))) bf16_t lds[];
    constexpr int C = 4, H = 84, W = 84, KS1 = 8, KS2 = 16, KS3 = 18;
    constexpr int OW1 = 20, P1 = 400, OW2 = 9, P2 = 81, OW3 = 7, P3 = 49;
    constexpr int NRB2 = (P2 + 15) / 16, NRB3 = (P3 + 15) / 16;
    constexpr long frame_bytes = (long)C * H * W;
    constexpr int PLANE = FWD_PLANE;
    const int grp = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g4 = lane >> 4, l15 = lane & 15;
    const int img = blockIdx.x;
    bf16_t* patch = lds;

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
        const long f0 = a.idx[grp] ? (long)a.idx[grp][img] : (long)img;
        patch_issue_at<7>(pr, a.frames[grp] + f0 * frame_bytes, frame_bytes, tid);
    }
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
    {   // w1 -> [32][256 + 8] rows in the (still unused) a1 region -> fragments: 16 lanes x 16 B cover the 64 banks once
        constexpr int W1S = C * PLANE, W1P = 264;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int t = tid + i * 256, off = W1S + (t >> 5) * W1P + (t & 31) * 8;
            *reinterpret_cast<u32x4*>(lds + off) = r1h[i];
            *reinterpret_cast<u32x4*>(lds + off + 32 * W1P) = r1l[i];
        }
        __syncthreads();
#pragma unroll
        for (int ks = 0; ks < KS1; ++ks)
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) {
                const bf16_t* q = lds + W1S + (nb * 16 + l15) * W1P + g4 * 64 + ks * 8;
                wf1h[ks][nb] = *reinterpret_cast<const bf16x8*>(q);
                wf1l[ks][nb] = *reinterpret_cast<const bf16x8*>(q + 32 * W1P);
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

    STACK_STAMP(a.prof, 0, 1)
    patch_commit<7, H * W / 16, PLANE - H * W>(pr, frame_bytes, patch, tid);
    __syncthreads();
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
}

// ---- data-gradient chain per image: da3 -> (conv3^T, relu'(a2)) -> da2 -> (conv2^T, relu'(a1)) -> da1 -----
// Replaces four launches of the layer-wise backward (two "dcol" GEMMs writing [pixels][taps x channels]
// matrices and two col2im gathers reading them back: 34 us of the 141 us step at batch 128, 0.9 GB of
// scratch at batch 4096).  One workgroup carries an image: da3, the forward activations a2 / a1 (relu
// masks) and the intermediate da2 live in LDS; the gradients replace the activations in place and leave
// with coalesced copies.  Both layers are gather-form GEMMs over the taps (conv2: per input-pixel parity
// class, one class per wave).  The forward weight shadows w[o][(ky,kx,c)] are staged ONCE per workgroup
// through LDS as per-tap [o][c] tiles and picked up with the transposing ds_read_b64_tr_b16 as
// B-fragments (rows = c, k = o) that stay in registers for every image of the workgroup.
struct DgradStackArgs {
    const bf16_t* da3; const bf16_t* a2; const bf16_t* a1; const bf16_t* w3; const bf16_t* w2;
    bf16_t* da2; bf16_t* da1; int batch;
    // optional tail (W1): conv1's weight gradient of the same image, from da1 while it is still in LDS
    const uint8_t* frames; const int* idx; float* gw1; float* gb1; long ws1, bs1; float scale1;
    long long* prof;    // optional phase stamps of workgroup 0 (pda_cnn_stack_prof)
};

// LDS plan (bf16 elements).  da3 and a2 / da2 live in zero-bordered 11 x 11 grids, so every tap of the two
// transposed convolutions is a constant offset from the lane's base (no per-tap bounds test); pixel and row
// pitches from tools/probes/lds_conflicts.py (4.6 - 4.7 LDS cycles per ds_read_b128, 4 = conflict-free).
#define D1_CP 40        // a1 / da1: dense [400][32 + 8]
#define D2_CP 80        // a2 / da2 grid, 1-pixel border
#define D2_RP 928
#define D3_CP 80        // da3 grid, 2-pixel border
#define D3_RP 976
#define D_A2OFF (400 * D1_CP)
#define D_D3OFF (D_A2OFF + 11 * D2_RP)
#define D_LDS_ELEMS (D_D3OFF + 11 * D3_RP)

typedef short s16x2 __attribute__((ext_vector_type(2)));
// keep the halves of val whose activation (bf16 bits) is > 0: relu'
__device__ __forceinline__ uint32_t relu_gate2(uint32_t act, uint32_t val) {
    s16x2 pos = __builtin_bit_cast(s16x2, act) > (s16x2){0, 0};
    return val & __builtin_bit_cast(uint32_t, pos);
}
__device__ __forceinline__ uint2 gate_pack4(uint2 act, f32x4 v) {
    return make_uint2(relu_gate2(act.x, pack2bf(v[0], v[1])), relu_gate2(act.y, pack2bf(v[2], v[3])));
}

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

__device__ __forceinline__ bf16x8 tr_frag(const bf16_t* p0, const bf16_t* p1) {
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    s16x4 lo = lds_tr4(p0), hi = lds_tr4(p1);
    s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v);
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
        const long f0 = a.idx ? (long)a.idx[i0] : (long)i0;
        patch_issue_at<7>(pr, a.frames + f0 * frame_bytes, frame_bytes, tid);
        if constexpr (MULTI) f_nxt = frame_index(a.idx, img + G, a.batch);
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
            patch_issue_at<7>(pr, a.frames + f_nxt * frame_bytes, frame_bytes, tid);
            f_nxt = frame_index(a.idx, img + 2 * G, a.batch);
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

// ---- the data-gradient chain at fp32 precision (split-bf16 planes) ---------------------------------------
// da3 / da2 / da1 are (hi, lo) pairs of planes, the weights too; the forward activations only gate (relu'), so
// their hi planes suffice.  Same per-image LDS data flow as cnn_dgrad_stack_kernel with a second grid / row set
// for every lo plane (148 KB), one image per workgroup, products w_hi d_hi + w_lo d_hi + w_hi d_lo in three
// accumulation chains.  All four weight-fragment sets (272 registers) are staged through LDS at the start, under
// the input loads; conv3^T fetches its da3 fragments in thirds of the reduction to stay inside the register file.
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
template <bool W1>
__global__ __launch_bounds__(256) void cnn_dgrad_stack_x3_kernel(DgradStackX3Args a) {
    extern __shared__ __attribute__((aligned(16))) bf16_t lds[];
    constexpr int P1 = 400, P2 = 81, P3 = 49, NRB3 = 6, NRB2 = 7;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g4 = lane >> 4, l15 = lane & 15;
    // two workgroups per image: both run conv3^T (da2 is needed whole), workgroup h = blockIdx.x & 1 then owns the
    // a1 / da1 rows of parity h -- the parity classes (h, 0) and (h, 1) of conv2^T and those pixels' share of dW1
    const int img = blockIdx.x >> 1, hh = blockIdx.x & 1;
    constexpr int FH = 84, FW = 84;
    constexpr long frame_bytes = 4L * FH * FW;
    long f0 = 0;
    if constexpr (W1) f0 = frame_index(a.idx, img, a.batch);      // a per-lane load with a counted wait at its use

    STACK_STAMP(a.prof, 0, 0)
    // ---- conv3's weight fragments (both planes) and this image's inputs (a1 / a2: hi planes only) go out together;
    // conv2's fragments are fetched under conv3^T.  The fragments -- rows = input channel, k = output channel -- come
    // straight from the transposed shadows the optimizer keeps: 16 bytes per lane and fragment, no staging.  (Transposed
    // through LDS at the start of the launch, as the bf16 kernel does, the four tensors cost 8 barriers and 10 k cycles
    // before the first MFMA.)
    constexpr int NRB2W = 4;             // conv2^T row blocks per wave: wave = (px, part), part 0 -> blocks 0-3, 1 -> 4-6
    u32x4 i1[4], i2[3], i3h[2], i3l[2];
    int in1[4], g1[4], in2[3], in3[2];
    int ld3[NRB3], st3[NRB3], ld2[NRB2W], st2[NRB2W];
    bf16x8 wf3h[18], wf3l[18];      // conv3^T: this wave's 16 input channels; k-step = (tap, o half)
    bf16x8 wf2h[8][2], wf2l[8][2];  // conv2^T: this wave's parity class; (tap a,b ; o half) x 2 channel blocks
    const int py = hh, px = wave & 1, part = wave >> 1;
    const bf16_t* w3src = a.w3t + (long)(wave * 16 + l15) * 64 + g4 * 8;
    const bf16_t* w2src = a.w2t + (long)l15 * 64 + g4 * 8;
#pragma unroll
    for (int ks = 0; ks < 18; ++ks) {       // tap = ks / 2, o half = ks % 2
        const bf16_t* q = w3src + (ks >> 1) * 64 * 64 + (ks & 1) * 32;
        wf3h[ks] = *reinterpret_cast<const bf16x8*>(q);
        wf3l[ks] = *reinterpret_cast<const bf16x8*>(q + a.w_plane);
    }
    STACK_STAMP(a.prof, 0, 1)
#pragma unroll
    for (int i = 0; i < 4; ++i) {       // chunk t of this workgroup's 200 pixels (rows of parity py) x 4 x 16 bytes
        int t = tid + i * 256;
        t = t < 800 ? t : 0;
        const int pp = t >> 2, p = (2 * (pp / 20) + py) * 20 + pp % 20;
        in1[i] = p * D1_CP + (t & 3) * 8;
        g1[i] = p * 4 + (t & 3);
        asm volatile("" : "+v"(in1[i]), "+v"(g1[i]));
    }
    {
        const u32x4* s1 = reinterpret_cast<const u32x4*>(a.a1 + (long)img * 400 * 32);
        const u32x4* s2 = reinterpret_cast<const u32x4*>(a.a2 + (long)img * 81 * 64);
        const u32x4* s3 = reinterpret_cast<const u32x4*>(a.da3 + (long)img * 49 * 64);
        const u32x4* s3l = reinterpret_cast<const u32x4*>(a.da3 + a.da3_plane + (long)img * 49 * 64);
#pragma unroll
        for (int i = 0; i < 4; ++i) i1[i] = s1[g1[i]];
#pragma unroll
        for (int i = 0; i < 3; ++i) { int t = tid + i * 256; i2[i] = s2[t < 648 ? t : 0]; }
#pragma unroll
        for (int i = 0; i < 2; ++i) { int t = tid + i * 256; i3h[i] = s3[t < 392 ? t : 0]; i3l[i] = s3l[t < 392 ? t : 0]; }
    }
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
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    __syncthreads();                    // grids zeroed
#pragma unroll
    for (int i = 0; i < 3; ++i) *reinterpret_cast<u32x4*>(lds + in1[i]) = i1[i];
    if (tid < 800 - 768) *reinterpret_cast<u32x4*>(lds + in1[3]) = i1[3];
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
    if constexpr (W1) patch_issue_at<7>(pr, a.frames + f0 * frame_bytes, frame_bytes, tid);     // flies under conv2^T
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
        for (int i = 0; i < 4; ++i) {           // this workgroup's rows
            if (i < 3 || tid < 800 - 768) {
                dst[g1[i]] = *reinterpret_cast<const uint4*>(lds + in1[i]);
                dstl[g1[i]] = *reinterpret_cast<const uint4*>(lds + in1[i] + DX_A1L);
            }
        }
    }
    STACK_STAMP(a.prof, 0, 7)
    if constexpr (W1) {
        // ---- conv1 weight gradient: dW1[o][(c, ky, kx)] = sum_px da1[px][o] patch[c][4 oy + ky][4 ox + kx];
        // wave w = kernel channel w; this workgroup's 200 pixels in 7 chunks of 32 (pad pixels read a zeroed strip
        // behind the patch)
        bf16_t* patch = lds + DX_A2H;
        constexpr int ZOFF = DX_A2H + 4 * FH * FW;
        static_assert(ZOFF + 64 <= DX_LDS_ELEMS, "zero strip");
        patch_commit<7>(pr, frame_bytes, patch, tid);       // every wave is past the barrier that closed conv2^T
        if (tid < 8) *reinterpret_cast<uint4*>(lds + ZOFF + tid * 8) = make_uint4(0, 0, 0, 0);
        __syncthreads();
        const int tg = lane >> 4, tq = (lane >> 2) & 3, tp = lane & 3;
        f32x4 acc1[2][4], accb1[2];
        int boff[4];
        bf16x8 ones;
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
        auto load = [&](bf16x8 (&fa)[4], bf16x8 (&fb)[4], int ch) {
            const int pp0 = ch * 32 + 8 * tg + tq, pp1 = pp0 + 4;
            const bool in0 = pp0 < 200, in1 = pp1 < 200;
            const int y0 = in0 ? 2 * (pp0 / 20) + py : 0, x0 = in0 ? pp0 % 20 : 0;
            const int y1 = in1 ? 2 * (pp1 / 20) + py : 0, x1 = in1 ? pp1 % 20 : 0;
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
#pragma unroll 1
        for (int ch = 0; ch < 6; ch += 2) {         // chunks ch, ch + 1; fragments one chunk ahead
            load(fa1, fb1, ch + 1);
            mma(fa0, fb0);
            load(fa0, fb0, ch + 2);
            mma(fa1, fb1);
        }
        mma(fa0, fb0);                              // chunk 6
        // [32][256] tile -> slab slice blockIdx.x (kernel layout == PyTorch's (c, ky, kx))
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
        STACK_STAMP(a.prof, 0, 8)
    }
}

// ---- conv2 + conv3 weight gradients per image (large batches) -------------------------------------------
// dW2[o][(tap,c)] = sum_pix da2[pix][o] a1[(2 oy + ky, 2 ox + kx)][c],  dW3 likewise from da3 and a2.
// As implicit GEMMs over all images these two are the most expensive launches of a batch-4096 step (2 x 133
// us): the im2col operand is re-gathered from global memory for every k-tile.  Here a workgroup loops over its
// images with a1 / a2 / da2 / da3 of the current image in LDS and BOTH gradients accumulating in registers
// (64 x 512 + 64 x 576 fp32 = 272 registers per lane over the 4 waves: the reason a wave owns its SIMD's whole
// register file), reduction over pixels, both MFMA operands through the transposing ds_read_b64_tr_b16 whose
// lanes point straight at the (strided / shifted) activation pixels, exactly like conv1_wgrad_kernel.  Each
// workgroup leaves one split-K slab slice per layer; the optimizer launch sums them.
// Wave w owns kernel row w of conv2 (n-blocks 8 w .. 8 w + 7) and n-blocks 9 w .. 9 w + 8 of conv3.
#define WG_A1 0                         // [400][40]
#define WG_A2 (400 * 40)                // [81][72]
#define WG_D2 (WG_A2 + 81 * 72)         // [96][72]  da2, rows >= 81 zero
#define WG_D3 (WG_D2 + 96 * 72)         // [64][72]  da3, rows >= 49 zero
#define WG_LDS_ELEMS (WG_D3 + 64 * 72)

struct WgradStackArgs {
    const bf16_t* a1; const bf16_t* a2; const bf16_t* da2; const bf16_t* da3;
    float* gw2; float* gb2; float* gw3; float* gb3;     // slab bases, slice blockIdx.x
    long ws2, bs2, ws3, bs3;
    int batch;
};
struct WgradRegs { u32x4 a1[7], a2[3], d2[3], d3[2]; };

__device__ __forceinline__ void wgrad_issue(WgradRegs& r, const WgradStackArgs& a, int img, int tid) {
    const u32x4* s1 = reinterpret_cast<const u32x4*>(a.a1 + (long)img * 400 * 32);
    const u32x4* s2 = reinterpret_cast<const u32x4*>(a.a2 + (long)img * 81 * 64);
    const u32x4* s3 = reinterpret_cast<const u32x4*>(a.da2 + (long)img * 81 * 64);
    const u32x4* s4 = reinterpret_cast<const u32x4*>(a.da3 + (long)img * 49 * 64);
#pragma unroll
    for (int i = 0; i < 7; ++i) { int t = tid + i * 256; r.a1[i] = s1[t < 1600 ? t : 0]; }
#pragma unroll
    for (int i = 0; i < 3; ++i) { int t = tid + i * 256; r.a2[i] = s2[t < 648 ? t : 0]; r.d2[i] = s3[t < 648 ? t : 0]; }
#pragma unroll
    for (int i = 0; i < 2; ++i) { int t = tid + i * 256; r.d3[i] = s4[t < 392 ? t : 0]; }
}


__global__ __launch_bounds__(256) void cnn_wgrad_stack_kernel(WgradStackArgs a) {
    extern __shared__ __attribute__((aligned(16))) bf16_t lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tg = lane >> 4, tq = (lane >> 2) & 3, tp = lane & 3;
    const int G = gridDim.x;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

    WgradRegs in;
    int img = blockIdx.x;
    wgrad_issue(in, a, img < a.batch ? img : 0, tid);

    f32x4 acc2[4][8], acc3[4][9];
    float bsum2 = 0.f, bsum3 = 0.f;         // bias gradients: lane = channel, this wave's share of the pixels
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int j = 0; j < 8; ++j) acc2[i][j] = zero4;
#pragma unroll
        for (int j = 0; j < 9; ++j) acc3[i][j] = zero4;
    }
    for (int t = tid; t < (15 * 72) / 8; t += 256)      // pad rows of the two dy tiles stay zero
        *reinterpret_cast<uint4*>(lds + WG_D2 + 81 * 72 + t * 8) = make_uint4(0, 0, 0, 0);
    for (int t = tid; t < (15 * 72) / 8; t += 256)
        *reinterpret_cast<uint4*>(lds + WG_D3 + 49 * 72 + t * 8) = make_uint4(0, 0, 0, 0);

    // landing offsets of the input chunks: chunk t of a [pixels][4 or 8 chunks] tile; + 256 threads = + 64 / 32 pixels
    const int in1 = (tid >> 2) * 40 + (tid & 3) * 8, in2 = (tid >> 3) * 72 + (tid & 7) * 8;
    // transposing-read bases of this lane: reduction rows (output pixels) 8 tg + tq (+ 4) of each k-step
    const int dlo = (8 * tg + tq) * 72 + 4 * tp;        // dy tiles: + 32 * 72 per k-step, + 4 * 72 for the upper half
    int b2lo[3], b2hi[3], b3lo[2], b3hi[2];
#pragma unroll
    for (int ks = 0; ks < 3; ++ks) {
        int p0 = ks * 32 + 8 * tg + tq, p1 = p0 + 4;
        int q0 = p0 < 81 ? p0 : 0, q1 = p1 < 81 ? p1 : 0;      // pad pixels: their dy rows are zero
        b2lo[ks] = WG_A1 + ((2 * (q0 / 9) + wave) * 20 + 2 * (q0 % 9)) * 40 + 4 * tp;
        b2hi[ks] = WG_A1 + ((2 * (q1 / 9) + wave) * 20 + 2 * (q1 % 9)) * 40 + 4 * tp;
    }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        int p0 = ks * 32 + 8 * tg + tq, p1 = p0 + 4;
        int q0 = p0 < 49 ? p0 : 0, q1 = p1 < 49 ? p1 : 0;
        b3lo[ks] = WG_A2 + ((q0 / 7) * 9 + q0 % 7) * 72 + 4 * tp;
        b3hi[ks] = WG_A2 + ((q1 / 7) * 9 + q1 % 7) * 72 + 4 * tp;
    }

    for (; img < a.batch; img += G) {
        __syncthreads();                    // every wave is done with the previous image's tiles
#pragma unroll
        for (int i = 0; i < 6; ++i) *reinterpret_cast<u32x4*>(lds + WG_A1 + in1 + i * (64 * 40)) = in.a1[i];
        if (tid < 1600 - 1536) *reinterpret_cast<u32x4*>(lds + WG_A1 + in1 + 6 * (64 * 40)) = in.a1[6];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            *reinterpret_cast<u32x4*>(lds + WG_A2 + in2 + i * (32 * 72)) = in.a2[i];
            *reinterpret_cast<u32x4*>(lds + WG_D2 + in2 + i * (32 * 72)) = in.d2[i];
        }
        if (tid < 648 - 512) {
            *reinterpret_cast<u32x4*>(lds + WG_A2 + in2 + 2 * (32 * 72)) = in.a2[2];
            *reinterpret_cast<u32x4*>(lds + WG_D2 + in2 + 2 * (32 * 72)) = in.d2[2];
        }
        *reinterpret_cast<u32x4*>(lds + WG_D3 + in2) = in.d3[0];
        if (tid < 392 - 256) *reinterpret_cast<u32x4*>(lds + WG_D3 + in2 + 32 * 72) = in.d3[1];
        __syncthreads();
        wgrad_issue(in, a, img + G < a.batch ? img + G : 0, tid);      // unconditional: see cnn_fwd_stack_kernel

        // ---- conv2: 3 k-steps of 32 output pixels; wave w = kernel row w, n-block j = (kx = j / 2, 16 channels j % 2)
#pragma unroll
        for (int ks = 0; ks < 3; ++ks) {
            bf16x8 fa[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) fa[i] = tr_frag(lds + WG_D2 + dlo + ks * (32 * 72) + i * 16, lds + WG_D2 + dlo + ks * (32 * 72) + 4 * 72 + i * 16);
#pragma unroll
            for (int jh = 0; jh < 8; jh += 4) {         // four n-blocks at a time: bounded fragment registers
                bf16x8 fb[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    int off = ((jh + j) >> 1) * 40 + ((jh + j) & 1) * 16;
                    fb[j] = tr_frag(lds + b2lo[ks] + off, lds + b2hi[ks] + off);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc2[i][jh + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb[j], acc2[i][jh + j], 0, 0, 0);
            }
        }
        // ---- conv3: 2 k-steps; n-block jj = 9 w + j = (tap = jj / 4, 16 channels jj % 4)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 fa[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) fa[i] = tr_frag(lds + WG_D3 + dlo + ks * (32 * 72) + i * 16, lds + WG_D3 + dlo + ks * (32 * 72) + 4 * 72 + i * 16);
#pragma unroll
            for (int jh = 0; jh < 9; jh += 3) {
                bf16x8 fb[3];
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    int jj = wave * 9 + jh + j, tap = jj >> 2, cb = jj & 3;
                    int off = ((tap / 3) * 9 + tap % 3) * 72 + cb * 16;
                    fb[j] = tr_frag(lds + b3lo[ks] + off, lds + b3hi[ks] + off);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 3; ++j)
                        acc3[i][jh + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb[j], acc3[i][jh + j], 0, 0, 0);
            }
        }
        // bias gradients: plain column sums of the two dy tiles, pixels dealt over the 4 waves (lane = channel)
        {
            float s2 = 0.f, s3 = 0.f;
#pragma unroll
            for (int i = 0; i < 21; ++i) {
                int px = wave + 4 * i;          // rows 81..83 of the da2 tile are zero pads
                s2 += bf2f(lds[WG_D2 + px * 72 + lane]);
            }
#pragma unroll
            for (int i = 0; i < 13; ++i) {
                int px = wave + 4 * i;          // rows 49..51 likewise
                s3 += bf2f(lds[WG_D3 + px * 72 + lane]);
            }
            bsum2 += s2;
            bsum3 += s3;
        }
    }
    // this workgroup's partial sums -> slab slice blockIdx.x (D layout: col = lane & 15, row = (lane >> 4) * 4 + r)
    const int l15 = lane & 15, g4 = lane >> 4;
    float* w2 = a.gw2 + (long)blockIdx.x * a.ws2;
    float* w3 = a.gw3 + (long)blockIdx.x * a.ws3;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int o = i * 16 + 4 * g4;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            int k = (wave * 8 + j) * 16 + l15;
#pragma unroll
            for (int r = 0; r < 4; ++r) w2[(long)(o + r) * 512 + k] = acc2[i][j][r];
        }
#pragma unroll
        for (int j = 0; j < 9; ++j) {
            int k = (wave * 9 + j) * 16 + l15;
#pragma unroll
            for (int r = 0; r < 4; ++r) w3[(long)(o + r) * 576 + k] = acc3[i][j][r];
        }
    }
    __syncthreads();                        // the tiles are dead: reuse LDS to add the four waves' bias sums
    float* red = reinterpret_cast<float*>(lds);
    red[wave * 128 + lane] = bsum2;
    red[wave * 128 + 64 + lane] = bsum3;
    __syncthreads();
    if (tid < 128) {
        float v = (red[tid] + red[128 + tid]) + (red[256 + tid] + red[384 + tid]);
        if (tid < 64) a.gb2[(long)blockIdx.x * a.bs2 + tid] = v;
        else a.gb3[(long)blockIdx.x * a.bs3 + tid - 64] = v;
    }
}

}  // namespace

// Does the specialised path cover this conv?  (Nature conv1 on 4-frame stacks)
PDA_EXPORT int pda_conv1_supported(int H, int W, int C, int KH, int KW, int S, int O) {
    return (C == 4 && KH == 8 && KW == 8 && S == 4 && O == 32 && W % 4 == 0 && ((long)C * H * W) % 16 == 0 &&
            (long)C * H * W <= PATCH_IT * 256 * 16 && ((H - 8) / 4 + 1) * ((W - 8) / 4 + 1) <= DY_IT * 64) ? 1 : 0;
}

PDA_EXPORT void pda_conv1_fwd_u8(const void* const* frames, const void* const* idx, const void* const* w,
                                 const void* const* bias, void* const* out, int ngroups, int batch, int H, int W,
                                 int C, float scale, int act, hipStream_t stream) {
    Conv1FwdArgs a;
    for (int i = 0; i < ngroups; ++i) {
        a.frames[i] = (const uint8_t*)frames[i]; a.idx[i] = (const int*)idx[i]; a.w[i] = (const bf16_t*)w[i];
        a.bias[i] = (const float*)bias[i]; a.out[i] = (bf16_t*)out[i];
    }
    a.batch = batch; a.g = {C, H, W, (H - 8) / 4 + 1, (W - 8) / 4 + 1}; a.scale = scale; a.act = act;
    a.parts = batch * ngroups <= 128 ? 4 : (batch * ngroups <= 256 ? 2 : 1);
    a.w_plane = 0; a.out_plane = 0;
    int blocks = batch * a.parts < 512 ? batch * a.parts : 512;     // 2 resident per CU (56 KB LDS each)
    size_t lds = (size_t)C * H * W * 2;
    hipLaunchKernelGGL(conv1_fwd_kernel<4>, dim3(blocks, ngroups), dim3(256), lds, stream, a);
}

// the same launch for the fp32-precision learner: split weights (lo plane w_plane elements behind hi), split outputs
PDA_EXPORT void pda_conv1_fwd_u8_x3(const void* const* frames, const void* const* idx, const void* const* w,
                                    long w_plane, const void* const* bias, void* const* out, long out_plane,
                                    int ngroups, int batch, int H, int W, int C, float scale, int act,
                                    hipStream_t stream) {
    Conv1FwdArgs a;
    for (int i = 0; i < ngroups; ++i) {
        a.frames[i] = (const uint8_t*)frames[i]; a.idx[i] = (const int*)idx[i]; a.w[i] = (const bf16_t*)w[i];
        a.bias[i] = (const float*)bias[i]; a.out[i] = (bf16_t*)out[i];
    }
    a.batch = batch; a.g = {C, H, W, (H - 8) / 4 + 1, (W - 8) / 4 + 1}; a.scale = scale; a.act = act;
    a.parts = batch * ngroups <= 128 ? 4 : (batch * ngroups <= 256 ? 2 : 1);
    a.w_plane = w_plane; a.out_plane = out_plane;
    int blocks = batch * a.parts < 512 ? batch * a.parts : 512;
    size_t lds = (size_t)C * H * W * 2;
    hipLaunchKernelGGL((conv1_fwd_kernel<4, true>), dim3(blocks, ngroups), dim3(256), lds, stream, a);
}

// Slices the weight-gradient launch writes for this batch (one per workgroup).
static int conv1_wgrad_parts(int batch) { return batch <= 64 ? 4 : (batch <= 128 ? 2 : 1); }
PDA_EXPORT int pda_conv1_wgrad_slices(int batch) {
    int blocks = batch * conv1_wgrad_parts(batch);
    return blocks < 256 ? blocks : 256;
}

PDA_EXPORT void pda_conv1_wgrad_u8(const void* dy, const void* frames, const void* idx, void* gw, void* gb,
                                   int batch, int H, int W, int C, float scale, long slice_stride,
                                   long bslice_stride, hipStream_t stream) {
    Conv1WgradArgs a;
    a.frames = (const uint8_t*)frames; a.idx = (const int*)idx; a.dy = (const bf16_t*)dy;
    a.gw = (float*)gw; a.gb = (float*)gb; a.slice_stride = slice_stride; a.bslice_stride = bslice_stride;
    a.batch = batch; a.g = {C, H, W, (H - 8) / 4 + 1, (W - 8) / 4 + 1}; a.scale = scale;
    a.parts = conv1_wgrad_parts(batch);
    a.dy_plane = 0;
    int P = a.g.OH * a.g.OW, PP = (P + 31) / 32 * 32;
    size_t lds = ((size_t)C * H * W + (size_t)PP * 32) * 2;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv1_wgrad_kernel<4>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    hipLaunchKernelGGL(conv1_wgrad_kernel<4>, dim3(pda_conv1_wgrad_slices(batch)), dim3(256), lds, stream, a);
}

// the same launch for the fp32-precision learner: dy as (hi, lo) planes
PDA_EXPORT void pda_conv1_wgrad_u8_x3(const void* dy, long dy_plane, const void* frames, const void* idx, void* gw,
                                      void* gb, int batch, int H, int W, int C, float scale, long slice_stride,
                                      long bslice_stride, hipStream_t stream) {
    Conv1WgradArgs a;
    a.frames = (const uint8_t*)frames; a.idx = (const int*)idx; a.dy = (const bf16_t*)dy;
    a.gw = (float*)gw; a.gb = (float*)gb; a.slice_stride = slice_stride; a.bslice_stride = bslice_stride;
    a.batch = batch; a.g = {C, H, W, (H - 8) / 4 + 1, (W - 8) / 4 + 1}; a.scale = scale;
    a.parts = conv1_wgrad_parts(batch);
    a.dy_plane = dy_plane;
    int P = a.g.OH * a.g.OW, PP = (P + 31) / 32 * 32;
    size_t lds = ((size_t)C * H * W + 2 * (size_t)PP * 32) * 2;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv1_wgrad_kernel<4, true>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    hipLaunchKernelGGL((conv1_wgrad_kernel<4, true>), dim3(pda_conv1_wgrad_slices(batch)), dim3(256), lds, stream, a);
}

// Phase-stamp buffer (device, >= 64 int64) for the next stack-kernel launches; null switches stamping off.
static long long* g_stack_prof = nullptr;
PDA_EXPORT void pda_cnn_stack_prof(void* buf) { g_stack_prof = (long long*)buf; }

// conv1 -> conv2 -> conv3 forward of the Nature CNN in one launch (relu after each; see cnn_fwd_stack_kernel).
// Layer shapes are the Nature ones: 8x8 s4 x32, 4x4 s2 x64, 3x3 s1 x64 on C = 4 frame stacks.
PDA_EXPORT int pda_cnn_fwd_stack_supported(int H, int W, int C) { return (H == 84 && W == 84 && C == 4) ? 1 : 0; }

PDA_EXPORT void pda_cnn_fwd_stack(const void* const* frames, const void* const* idx, const void* const* w1,
                                  const void* const* b1, const void* const* w2, const void* const* b2,
                                  const void* const* w3, const void* const* b3, void* const* a1, void* const* a2,
                                  void* const* a3, int ngroups, int batch, int H, int W, int C, float scale,
                                  hipStream_t stream) {
    FwdStackArgs a;
    for (int i = 0; i < ngroups; ++i) {
        a.frames[i] = (const uint8_t*)frames[i]; a.idx[i] = (const int*)idx[i];
        a.w1[i] = (const bf16_t*)w1[i]; a.b1[i] = (const float*)b1[i];
        a.w2[i] = (const bf16_t*)w2[i]; a.b2[i] = (const float*)b2[i];
        a.w3[i] = (const bf16_t*)w3[i]; a.b3[i] = (const float*)b3[i];
        a.a1[i] = (bf16_t*)a1[i]; a.a2[i] = (bf16_t*)a2[i]; a.a3[i] = (bf16_t*)a3[i];
    }
    a.batch = batch; a.scale = scale; a.prof = g_stack_prof;
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

// the same at fp32 precision: split weights (lo plane w_plane elements behind hi; one shadow buffer per network, so
// one offset for the three layers) and split activations.  One workgroup per image and network.
PDA_EXPORT void pda_cnn_fwd_stack_x3(const void* const* frames, const void* const* idx, const void* const* w1,
                                     const void* const* b1, const void* const* w2, const void* const* b2,
                                     const void* const* w3, const void* const* b3, long w_plane, void* const* a1,
                                     long a1_plane, void* const* a2, long a2_plane, void* const* a3, long a3_plane,
                                     int ngroups, int batch, int H, int W, int C, float scale, hipStream_t stream) {
    FwdStackArgs a;
    for (int i = 0; i < ngroups; ++i) {
        a.frames[i] = (const uint8_t*)frames[i]; a.idx[i] = (const int*)idx[i];
        a.w1[i] = (const bf16_t*)w1[i]; a.b1[i] = (const float*)b1[i];
        a.w2[i] = (const bf16_t*)w2[i]; a.b2[i] = (const float*)b2[i];
        a.w3[i] = (const bf16_t*)w3[i]; a.b3[i] = (const float*)b3[i];
        a.a1[i] = (bf16_t*)a1[i]; a.a2[i] = (bf16_t*)a2[i]; a.a3[i] = (bf16_t*)a3[i];
    }
    a.batch = batch; a.scale = scale; a.prof = g_stack_prof;
    a.w_plane = w_plane; a.a1_plane = a1_plane; a.a2_plane = a2_plane; a.a3_plane = a3_plane;
    size_t lds = ((size_t)4 * FWD_PLANE + 2 * (size_t)X_A1LO) * 2;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(cnn_fwd_stack_x3_kernel),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    hipLaunchKernelGGL(cnn_fwd_stack_x3_kernel, dim3(batch, ngroups), dim3(256), lds, stream, a);
}

// conv3^T -> relu'(a2) -> conv2^T -> relu'(a1): the data gradients da2 [(b, 9x9)][64] and da1 [(b, 20x20)][32]
// of the Nature CNN from da3 [(b, 7x7)][64] in one launch (cnn_dgrad_stack_kernel).  84x84 inputs only.
PDA_EXPORT int pda_cnn_dgrad_stack_supported(int H, int W, int C) { return (H == 84 && W == 84 && C == 4) ? 1 : 0; }

static void launch_dgrad_stack(DgradStackArgs& a, hipStream_t stream) {
    a.prof = g_stack_prof;
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
                        nullptr};
    launch_dgrad_stack(a, stream);
}

// the chain at fp32 precision: da3 / da2 / da1 and the weights as (hi, lo) planes, a2 / a1 hi planes (relu masks)
// w3t / w2t: the TRANSPOSED weight shadows [(tap, c)][o].  frames != null: the launch also writes conv1's weight
// gradient, TWO slab slices (ws1 / bs1 floats apart) per image (one per workgroup)
PDA_EXPORT void pda_cnn_dgrad_stack_x3(const void* da3, long da3_plane, const void* a2, const void* a1,
                                       const void* w3t, const void* w2t, long w_plane, void* da2, long da2_plane,
                                       void* da1, long da1_plane, int batch, const void* frames, const void* idx,
                                       void* gw1, void* gb1, long ws1, long bs1, float scale1, hipStream_t stream) {
    DgradStackX3Args a = {(const bf16_t*)da3, (const bf16_t*)a2, (const bf16_t*)a1, (const bf16_t*)w3t,
                          (const bf16_t*)w2t, (bf16_t*)da2, (bf16_t*)da1, batch, da3_plane, w_plane, da2_plane, da1_plane,
                          g_stack_prof, (const uint8_t*)frames, (const int*)idx, (float*)gw1, (float*)gb1, ws1, bs1,
                          scale1};
    static_assert(DX_LDS_ELEMS * 2 <= 160 * 1024, "LDS budget");
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(cnn_dgrad_stack_x3_kernel<false>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(cnn_dgrad_stack_x3_kernel<true>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    if (frames)
        hipLaunchKernelGGL(cnn_dgrad_stack_x3_kernel<true>, dim3(2 * batch), dim3(256), (size_t)DX_LDS_ELEMS * 2, stream, a);
    else
        hipLaunchKernelGGL(cnn_dgrad_stack_x3_kernel<false>, dim3(2 * batch), dim3(256), (size_t)DX_LDS_ELEMS * 2, stream, a);
}

// same, plus conv1's weight gradient from uint8 frames (slab slice per workgroup: pda_cnn_dgrad_stack_slices(batch))
PDA_EXPORT int pda_cnn_dgrad_stack_slices(int batch) { return batch < 256 ? batch : 256; }
PDA_EXPORT void pda_cnn_dgrad_stack_w1(const void* da3, const void* a2, const void* a1, const void* w3, const void* w2,
                                       void* da2, void* da1, int batch, const void* frames, const void* idx,
                                       void* gw1, void* gb1, long ws1, long bs1, float scale1, hipStream_t stream) {
    DgradStackArgs a = {(const bf16_t*)da3, (const bf16_t*)a2, (const bf16_t*)a1, (const bf16_t*)w3,
                        (const bf16_t*)w2, (bf16_t*)da2, (bf16_t*)da1, batch, (const uint8_t*)frames, (const int*)idx,
                        (float*)gw1, (float*)gb1, ws1, bs1, scale1, nullptr};
    launch_dgrad_stack(a, stream);
}

// conv2 + conv3 weight gradients of the Nature CNN, one slab slice per workgroup and layer (cnn_wgrad_stack_kernel).
// pda_cnn_wgrad_stack_slices(batch) is the number of slices (= workgroups) the launch writes.
PDA_EXPORT int pda_cnn_wgrad_stack_slices(int batch) { return batch < 256 ? batch : 256; }

PDA_EXPORT void pda_cnn_wgrad_stack(const void* a1, const void* a2, const void* da2, const void* da3, void* gw2,
                                    void* gb2, void* gw3, void* gb3, long ws2, long bs2, long ws3, long bs3,
                                    int batch, hipStream_t stream) {
    WgradStackArgs a = {(const bf16_t*)a1, (const bf16_t*)a2, (const bf16_t*)da2, (const bf16_t*)da3,
                        (float*)gw2, (float*)gb2, (float*)gw3, (float*)gb3, ws2, bs2, ws3, bs3, batch};
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(cnn_wgrad_stack_kernel),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    hipLaunchKernelGGL(cnn_wgrad_stack_kernel, dim3(pda_cnn_wgrad_stack_slices(batch)), dim3(256),
                       (size_t)WG_LDS_ELEMS * 2, stream, a);
}
