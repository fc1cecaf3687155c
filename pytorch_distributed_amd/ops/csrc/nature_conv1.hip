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
#include "nature_cnn.h"

namespace {

struct Conv1FwdArgs {
    const uint8_t* frames[GEMM_MAX_GROUPS]; const int* idx[GEMM_MAX_GROUPS]; const bf16_t* w[GEMM_MAX_GROUPS];
    const float* bias[GEMM_MAX_GROUPS]; bf16_t* out[GEMM_MAX_GROUPS];
    int batch; Conv1Geom g; float scale; int act;
    int parts;      // workgroups per image (small batches: an image's 16-pixel row blocks are split)
    long w_plane, out_plane;    // X3: element offsets to the lo planes of the split weights / outputs
    const int* ref[GEMM_MAX_GROUPS]; long frame_stride;     // frame_index() in nature_cnn.h
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
    long f_nxt = frame_index(a.idx[grp], a.ref[grp], img0 + img_step, a.batch);
    if (img0 < a.batch)
        patch_issue_at(pr, a.frames[grp] + frame_index(a.idx[grp], a.ref[grp], img0, a.batch) * a.frame_stride,
                       frame_bytes, tid);
    for (int img = img0; img < a.batch; img += img_step) {
        __syncthreads();                // everyone is done reading the previous patch
        patch_commit(pr, frame_bytes, patch, tid);
        __syncthreads();
        if (img + img_step < a.batch)           // next image's loads fly under this image's MFMAs
            patch_issue_at(pr, a.frames[grp] + f_nxt * a.frame_stride, frame_bytes, tid);
        f_nxt = frame_index(a.idx[grp], a.ref[grp], img + 2 * img_step, a.batch);
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
    const int* ref; long frame_stride;      // frame_index() in nature_cnn.h
};

// LDS: patch [C*H*W] | dyT [PP][32] (T layout, ROWS = 32) | X3: dyT of the lo plane.
// Both MFMA operands come through the transposing ds_read_b64_tr_b16, whose lanes each supply the address
// of an 8-byte piece (4 consecutive columns of one reduction row = pixel).  For dy that is the [pixel][32]
// tile; for the im2col operand the piece (pixel, c, ky, kx 0..3 or 4..7) already sits in the frame patch at
// patch[c][4 oy + ky][4 ox + kx], so the fragments are gathered from the patch in place: no [pixel][k]
// column tile, no LDS -> LDS expansion pass and no barriers inside an image (the first version expanded 32
// pixels at a time between two barriers: 13.9 us per image at batch 4096).  Wave w owns channel w of the
// kernel (k = 64 w .. 64 w + 63), n-block j = kernel rows 2 j, 2 j + 1.
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
    long f_nxt = frame_index(a.idx, a.ref, img0 + img_step, a.batch);
    if (img0 < a.batch) {
        patch_issue_at(pr, a.frames + frame_index(a.idx, a.ref, img0, a.batch) * a.frame_stride, frame_bytes, tid);
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
            patch_issue_at(pr, a.frames + f_nxt * a.frame_stride, frame_bytes, tid);
            dy_issue(img + img_step);
        }
        f_nxt = frame_index(a.idx, a.ref, img + 2 * img_step, a.batch);
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

}  // namespace

// Does the specialised path cover this conv?  (Nature conv1 on 4-frame stacks)
PDA_EXPORT int pda_conv1_supported(int H, int W, int C, int KH, int KW, int S, int O) {
    return (C == 4 && KH == 8 && KW == 8 && S == 4 && O == 32 && W % 4 == 0 && ((long)C * H * W) % 16 == 0 &&
            (long)C * H * W <= PATCH_IT * 256 * 16 && ((H - 8) / 4 + 1) * ((W - 8) / 4 + 1) <= DY_IT * 64) ? 1 : 0;
}

// ref / frame_stride: frame_index() in nature_cnn.h (ref == null or frame_stride == 0: rows of full stacks)
PDA_EXPORT void pda_conv1_fwd_u8_w(const void* const* frames, const void* const* idx, const void* const* ref,
                                   long frame_stride, const void* const* w, const void* const* bias, void* const* out,
                                   int ngroups, int batch, int H, int W, int C, float scale, int act,
                                   hipStream_t stream) {
    Conv1FwdArgs a;
    for (int i = 0; i < ngroups; ++i) {
        a.frames[i] = (const uint8_t*)frames[i]; a.idx[i] = (const int*)idx[i]; a.w[i] = (const bf16_t*)w[i];
        a.bias[i] = (const float*)bias[i]; a.out[i] = (bf16_t*)out[i];
        a.ref[i] = ref ? (const int*)ref[i] : nullptr;
    }
    a.frame_stride = frame_stride > 0 ? frame_stride : (long)C * H * W;
    a.batch = batch; a.g = {C, H, W, (H - 8) / 4 + 1, (W - 8) / 4 + 1}; a.scale = scale; a.act = act;
    a.parts = batch * ngroups <= 128 ? 4 : (batch * ngroups <= 256 ? 2 : 1);
    a.w_plane = 0; a.out_plane = 0;
    int blocks = batch * a.parts < 512 ? batch * a.parts : 512;     // 2 resident per CU (56 KB LDS each)
    size_t lds = (size_t)C * H * W * 2;
    hipLaunchKernelGGL(conv1_fwd_kernel<4>, dim3(blocks, ngroups), dim3(256), lds, stream, a);
}
PDA_EXPORT void pda_conv1_fwd_u8(const void* const* frames, const void* const* idx, const void* const* w,
                                 const void* const* bias, void* const* out, int ngroups, int batch, int H, int W,
                                 int C, float scale, int act, hipStream_t stream) {
    pda_conv1_fwd_u8_w(frames, idx, nullptr, 0, w, bias, out, ngroups, batch, H, W, C, scale, act, stream);
}

// the same launch for the fp32-precision learner: split weights (lo plane w_plane elements behind hi), split outputs
PDA_EXPORT void pda_conv1_fwd_u8_x3_w(const void* const* frames, const void* const* idx, const void* const* ref,
                                      long frame_stride, const void* const* w, long w_plane, const void* const* bias,
                                      void* const* out, long out_plane, int ngroups, int batch, int H, int W, int C,
                                      float scale, int act, hipStream_t stream) {
    Conv1FwdArgs a;
    for (int i = 0; i < ngroups; ++i) {
        a.frames[i] = (const uint8_t*)frames[i]; a.idx[i] = (const int*)idx[i]; a.w[i] = (const bf16_t*)w[i];
        a.bias[i] = (const float*)bias[i]; a.out[i] = (bf16_t*)out[i];
        a.ref[i] = ref ? (const int*)ref[i] : nullptr;
    }
    a.frame_stride = frame_stride > 0 ? frame_stride : (long)C * H * W;
    a.batch = batch; a.g = {C, H, W, (H - 8) / 4 + 1, (W - 8) / 4 + 1}; a.scale = scale; a.act = act;
    a.parts = batch * ngroups <= 128 ? 4 : (batch * ngroups <= 256 ? 2 : 1);
    a.w_plane = w_plane; a.out_plane = out_plane;
    int blocks = batch * a.parts < 512 ? batch * a.parts : 512;
    size_t lds = (size_t)C * H * W * 2;
    hipLaunchKernelGGL((conv1_fwd_kernel<4, true>), dim3(blocks, ngroups), dim3(256), lds, stream, a);
}
PDA_EXPORT void pda_conv1_fwd_u8_x3(const void* const* frames, const void* const* idx, const void* const* w,
                                    long w_plane, const void* const* bias, void* const* out, long out_plane,
                                    int ngroups, int batch, int H, int W, int C, float scale, int act,
                                    hipStream_t stream) {
    pda_conv1_fwd_u8_x3_w(frames, idx, nullptr, 0, w, w_plane, bias, out, out_plane, ngroups, batch, H, W, C, scale, act,
                          stream);
}

// Slices the weight-gradient launch writes for this batch (one per workgroup).
static int conv1_wgrad_parts(int batch) { return batch <= 64 ? 4 : (batch <= 128 ? 2 : 1); }
PDA_EXPORT int pda_conv1_wgrad_slices(int batch) {
    int blocks = batch * conv1_wgrad_parts(batch);
    return blocks < 256 ? blocks : 256;
}

PDA_EXPORT void pda_conv1_wgrad_u8(const void* dy, const void* frames, const void* idx, void* gw, void* gb,
                                   int batch, int H, int W, int C, float scale, long slice_stride,
                                   long bslice_stride, const void* ref, long frame_stride, hipStream_t stream) {
    Conv1WgradArgs a;
    a.ref = (const int*)ref; a.frame_stride = frame_stride > 0 ? frame_stride : (long)C * H * W;
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
                                      long bslice_stride, const void* ref, long frame_stride, hipStream_t stream) {
    Conv1WgradArgs a;
    a.ref = (const int*)ref; a.frame_stride = frame_stride > 0 ? frame_stride : (long)C * H * W;
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
