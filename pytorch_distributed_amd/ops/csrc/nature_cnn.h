// Shared pieces of the Nature-CNN kernel family (4 x 84 x 84 uint8 frame stacks; conv 8x8 s4 x32, 4x4 s2 x64, 3x3 s1 x64):
// frame-patch loaders, LDS pitches, fragment helpers and the argument blocks that more than one file uses.
//   nature_conv1.hip            whole-image conv1 forward and weight gradient (bf16 and fp32-class)
//   nature_fwd_stack.hip        conv1 -> conv2 -> conv3 forward per image, bf16
//   nature_fwd_stack_x3.hip     the same on split-bf16 planes (fp32-class)
//   nature_dgrad_stack.hip      da3 -> da2 -> da1 per image (+ conv1 weight gradient tail), bf16
//   nature_dgrad_stack_x3.hip   the same on split-bf16 planes
//   nature_wgrad_stack.hip      conv2 + conv3 weight gradients per image (large batches), bf16
// A CNN of another geometry adds files of its own beside these; nothing here is selected by `if constexpr` on
// a geometry.
#pragma once
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
// ref != null (frame-deduplicated replay, parallel/csrc/frame_runs.h): idx[img] is a transition row and ref[row] the
// first plane of its frame window in the plane ring; the caller multiplies by the plane stride instead of the stack
// size.  (One more dependent load, also kept per-lane.)
__device__ __forceinline__ long frame_index(const int* idx, const int* ref, int img, int batch) {
    int iv = img < batch ? img : 0;
    asm volatile("" : "+v"(iv));
    if (!idx) return (long)iv;
    int row = idx[iv];
    if (ref) {
        asm volatile("" : "+v"(row));
        row = ref[row];
    }
    return (long)row;
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

__device__ __forceinline__ s16x4 lds_tr4(const bf16_t* p) {
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p));
}

struct FwdStackArgs {
    const uint8_t* frames[GEMM_MAX_GROUPS]; const int* idx[GEMM_MAX_GROUPS];
    const int* ref[GEMM_MAX_GROUPS]; long frame_stride;    // see frame_index(): bytes between the stacks idx / ref count in
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

__device__ __forceinline__ bf16x8 tr_frag(const bf16_t* p0, const bf16_t* p1) {
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    s16x4 lo = lds_tr4(p0), hi = lds_tr4(p1);
    s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v);
}

}  // namespace

// Phase-stamp buffer (device, >= 64 int64) for the next stack-kernel launches; null switches stamping off
// (nature_fwd_stack.hip: pda_cnn_stack_prof).
extern long long* g_pda_stack_prof;
