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
#include "nature_cnn.h"

namespace {

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
