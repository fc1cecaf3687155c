// fp32 MLP learner kernels: the reference-precision path of the MLP agents (Distributed DDPG, MLP-DQN bodies).
//
// The reference's DDPG learner is plain fp32 PyTorch (/root/reference/core/single_processes/ddpg_learner.py:66-95,
// core/models/ddpg_mlp_model.py:16-72).  The bf16 engine (mlp_fused.hip) rounds every operand to 8 significand bits
// (gradients within ~6 % of fp32); this file runs the SAME step on gfx950's fp32-input matrix instruction,
// v_mfma_f32_16x16x4_f32: products and sums are exact fp32 FMA chains, operands are the fp32 master weights themselves
// (no shadow copies, no transposed copies), activations and gradients stay fp32 in HBM / L2.  The problem is ~0.1 GFLOP per
// step, so the instruction's rate (1/16 of bf16 MFMA) is irrelevant; what matters is that every launch fills the chip:
// a launch is a LIST of small GEMMs (forward layers of the online and target networks, data gradients, all six weight
// gradients), cut into 16x16 output tiles, one workgroup of four waves per tile, the reduction split over the waves.
//
//   C[i][j] = sum_r A(i, r) B(r, j)         A(i, r) = a[i a_si + r a_sr],   B(r, j) = b[r b_sr + j b_sj]
//
//   forward      i = batch row, j = output unit, r = input unit:  A = x (rows), B = W[j][r] (rows)      -> act(C + bias)
//   data grad    i = batch row, j = input unit,  r = output unit: A = dy (rows), B = W[r][j] (columns)  -> C * act'(y_in)
//   weight grad  i = output unit, j = input unit, r = batch row:  A = dy (columns), B = x (columns)     -> dW, db = sum_r A
//
// Lane l of a wave holds row / column l & 15 and reduction elements 16 c + 4 (l >> 4) + e, e = 0..3, of chunk c: with a unit
// reduction stride that is ONE 16-byte load per operand and chunk, and element e of both operands goes into MFMA e of the
// chunk (a sum does not care in which order its terms arrive, as long as both operands use the same order).
#include "api.h"

namespace {

constexpr int F32_MAX_JOBS = 8;
constexpr int KIND_FWD = 0, KIND_DGRAD = 1, KIND_WGRAD = 2;

struct F32Job {
    const float* a; long a_si, a_sr;
    const float* b; long b_sr, b_sj;
    int I, J, R;                        // output rows, output columns, reduction length
    float* c; long c_ld; int c_col0;    // C[i c_ld + c_col0 + j]
    const float* bias;                  // forward: per output column (may be null)
    float* z; long z_ld;                // forward: optional copy of the pre-activation (C + bias)
    const float* yin; long yin_ld;      // data gradient: the activation OUTPUT at (i, j), for act'
    int act, kind;
    float* gb;                          // weight gradient: bias gradient sum_r A(i, r) (may be null)
    // data gradient extra (DDPG: the actor's pre-activation penalty enters behind tanh'): rows i < x_rows, columns
    // j >= x_col0 get + x_coef * gate[0] * x_src[i x_ld + j - x_col0]; the same columns of those rows are also copied to
    // x_out[i x_out_ld + j - x_col0] (dL/dz of the actor's output layer, contiguous for the next launch)
    const float* x_src; long x_ld; int x_col0, x_rows; float x_coef; const float* x_gate;
    float* x_out; long x_out_ld;
    int tiles_j, tile0, a_vec, b_vec;   // host-filled: j tiles, first workgroup of the job, 16-byte paths
};
struct F32Jobs { F32Job j[F32_MAX_JOBS]; int n; };

__device__ __forceinline__ float act_fwd(float v, int act) {
    if (act == ACT_RELU) return fmaxf(v, 0.f);
    if (act == ACT_TANH) { const float e = __expf(2.f * v); return 1.f - 2.f * __builtin_amdgcn_rcpf(e + 1.f); }
    return v;
}
__device__ __forceinline__ float act_bwd(float y, int act) {
    if (act == ACT_RELU) return y > 0.f ? 1.f : 0.f;
    if (act == ACT_TANH) return 1.f - y * y;
    return 1.f;
}

// 4 reduction elements r0 .. r0+3 of one operand row / column `line` (clamped, then masked: never a branch around a load)
__device__ __forceinline__ void load4(const float* base, long line_off, long sr, int r0, int R, bool line_ok, bool vec,
                                      float (&v)[4]) {
    if (vec) {              // sr == 1, 16-byte aligned lines, R padded to 4 in memory
        const int rc = min(r0, max(R - 1, 0) & ~3);
        const float4 q = *reinterpret_cast<const float4*>(base + line_off + rc);
        v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = base[line_off + (long)min(r0 + e, R - 1) * sr];
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = (line_ok && r0 + e < R) ? v[e] : 0.f;
}

__global__ __launch_bounds__(256) void mlp_f32_jobs_kernel(F32Jobs jobs) {
    __shared__ float red[4][4][64];
    __shared__ float bsum[4][16];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int ji = 0;
#pragma unroll
    for (int k = 1; k < F32_MAX_JOBS; ++k)
        if (k < jobs.n && (int)blockIdx.x >= jobs.j[k].tile0) ji = k;
    const F32Job& J = jobs.j[ji];
    const int t = (int)blockIdx.x - J.tile0;
    const int ti = t / J.tiles_j, tj = t - ti * J.tiles_j;
    const int i0 = ti * 16, j0 = tj * 16;
    const int ln = lane & 15, kg = lane >> 4;
    const int ia = i0 + ln, jb = j0 + ln;
    const bool ia_ok = ia < J.I, jb_ok = jb < J.J;
    const long a_off = (long)min(ia, J.I - 1) * J.a_si, b_off = (long)min(jb, J.J - 1) * J.b_sj;
    const int nchunks = (J.R + 15) >> 4;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    float asum = 0.f;
#pragma unroll 2
    for (int c = wave; c < nchunks; c += 4) {
        const int r0 = c * 16 + kg * 4;
        float av[4], bv[4];
        load4(J.a, a_off, J.a_sr, r0, J.R, ia_ok, J.a_vec != 0, av);
        load4(J.b, b_off, J.b_sr, r0, J.R, jb_ok, J.b_vec != 0, bv);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[e], bv[e], acc, 0, 0, 0);
            asum += av[e];
        }
    }
    // the four waves' partial tiles meet in LDS; wave w then finishes accumulator register w: row 4 kg + w, column ln
#pragma unroll
    for (int q = 0; q < 4; ++q) red[wave][q][lane] = acc[q];
    if (J.kind == KIND_WGRAD && J.gb != nullptr && tj == 0) {
        asum += __shfl_xor(asum, 16, 64);
        asum += __shfl_xor(asum, 32, 64);
        if (lane < 16) bsum[wave][lane] = asum;
    }
    __syncthreads();
    const float v0 = (red[0][wave][lane] + red[1][wave][lane]) + (red[2][wave][lane] + red[3][wave][lane]);
    const int i = i0 + kg * 4 + wave, j = j0 + ln;
    if (J.kind == KIND_WGRAD && J.gb != nullptr && tj == 0 && threadIdx.x < 16 && i0 + (int)threadIdx.x < J.I)
        J.gb[i0 + threadIdx.x] = (bsum[0][threadIdx.x] + bsum[1][threadIdx.x]) + (bsum[2][threadIdx.x] + bsum[3][threadIdx.x]);
    if (i >= J.I || j >= J.J) return;
    float v = v0;
    if (J.kind == KIND_FWD) {
        if (J.bias) v += J.bias[j];
        if (J.z) J.z[(long)i * J.z_ld + j] = v;
        v = act_fwd(v, J.act);
    } else if (J.kind == KIND_DGRAD) {
        if (J.yin) v *= act_bwd(J.yin[(long)i * J.yin_ld + j], J.act);
        if (J.x_out != nullptr && j >= J.x_col0 && i < J.x_rows) {
            if (J.x_src) v += J.x_coef * J.x_gate[0] * J.x_src[(long)i * J.x_ld + (j - J.x_col0)];
            J.x_out[(long)i * J.x_out_ld + (j - J.x_col0)] = v;
        }
    }
    J.c[(long)i * J.c_ld + J.c_col0 + j] = v;
}

// ---- replay rows -> fp32 operands (gather through the sampled indices) ---------------------------------------------------
struct F32StageArgs {
    const float* s0; const float* s1; long ld_s; const float* tails; int tail_ld; int off_action;
    const int* idx; int B, D, A;
    float* x0; long ld_x0;          // [2B][ld]: rows b and B + b = s0[idx[b]]
    float* x1; long ld_x1;          // [B][ld]:  s_n[idx[b]]
    float* cat; long ld_cat; int c1;    // rows B + b, columns c1 .. c1 + A: the replay action
    float scale;
};
__global__ __launch_bounds__(256) void mlp_f32_stage_kernel(F32StageArgs p) {
    const int total = p.B * (2 * p.D + p.A);
    for (int t = blockIdx.x * 256 + threadIdx.x; t < total; t += gridDim.x * 256) {
        const int per = 2 * p.D + p.A;
        const int b = t / per, d = t - b * per;
        const long row = p.idx ? (long)p.idx[b] : (long)b;
        if (d < p.D) {
            const float v = p.s0[row * p.ld_s + d] * p.scale;
            p.x0[(long)b * p.ld_x0 + d] = v;
            p.x0[(long)(p.B + b) * p.ld_x0 + d] = v;
        } else if (d < 2 * p.D) {
            p.x1[(long)b * p.ld_x1 + (d - p.D)] = p.s1[row * p.ld_s + (d - p.D)] * p.scale;
        } else if (p.cat) {
            p.cat[(long)(p.B + b) * p.ld_cat + p.c1 + (d - 2 * p.D)] = p.tails[row * p.tail_ld + p.off_action + (d - 2 * p.D)];
        }
    }
}

// ---- DDPG critic head in fp32: the three Q values, n-step target, both losses, dL/d(pre-activation of critic.1.0) -------
struct DdpgHeadF32Args {
    const float* h2; long ldh;                              // [3B][ldh]: Q(s0, pi(s0)) | Q(s0, a) | Q'(s_n, pi'(s_n)) hidden
    const float* w3_on; const float* b3_on; const float* w3_tg; const float* b3_tg;
    const float* tails; int tail_ld, off_reward, off_gamma, off_term;
    const int* idx; const float* isw;
    float* dh2;                                             // [2B][ldh]
    float* dq;                                              // [2B]: dL/dQ of the two online passes (row 0..B: 0 unless leak)
    const float* za; int A; float actor_l2;                 // the online actor's pre-tanh output [B][A] (penalty term)
    float* gate;                                            // gate[0] = 1 while the actor loss is on, else 0
    float* q_out; float* td_out; float* loss_sum; float* leaf_sum; float* leaf_min; float prio_eps, prio_alpha;
    int B, H, leak, actor_warmup;
    unsigned long long* rng_counter; int* opt_step;
};
__global__ __launch_bounds__(256) void ddpg_head_f32_kernel(DdpgHeadF32Args p) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int b = blockIdx.x * 4 + wave;
    // critic warm-up: no actor-loss gradient yet.  (Workgroup 0 bumps the counter in this launch: at the one boundary
    // step some samples may see either value, which is harmless -- same as the bf16 head, mlp_kernels.hip.)
    const bool actor_on = !(p.actor_warmup > 0 && p.opt_step && *p.opt_step < p.actor_warmup);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        p.gate[0] = actor_on ? 1.f : 0.f;
        if (p.rng_counter) *p.rng_counter += 1;
        if (p.opt_step) *p.opt_step += 1;
    }
    if (b >= p.B) return;
    const float* hp = p.h2 + (long)b * p.ldh;
    const float* hc = p.h2 + (long)(p.B + b) * p.ldh;
    const float* ht = p.h2 + (long)(2 * p.B + b) * p.ldh;
    float sp = 0.f, sc = 0.f, st = 0.f;
    for (int j = lane; j < p.H; j += 64) {
        const float wo = p.w3_on[j];
        sp += wo * hp[j];
        sc += wo * hc[j];
        st += p.w3_tg[j] * ht[j];
    }
    const float q_pi = wave_sum(sp) + p.b3_on[0], q_c = wave_sum(sc) + p.b3_on[0], q_t = wave_sum(st) + p.b3_tg[0];
    const int ridx = p.idx ? p.idx[b] : b;
    const float* tl = p.tails + (long)ridx * p.tail_ld;
    const float y = tl[p.off_reward] + tl[p.off_gamma] * q_t * (1.f - tl[p.off_term]);
    const float td = q_c - y;
    const float w = p.isw ? p.isw[b] : 1.f;
    const float invB = 1.f / (float)p.B;
    const float dq_c = 2.f * w * td * invB;
    const float dq_pi = actor_on ? -invB : 0.f;
    float* dp = p.dh2 + (long)b * p.ldh;
    float* dc = p.dh2 + (long)(p.B + b) * p.ldh;
    for (int j = lane; j < p.H; j += 64) {
        const float wo = p.w3_on[j];
        const float a = hp[j], c = hc[j];
        dp[j] = dq_pi * wo * (1.f - a * a);
        dc[j] = dq_c * wo * (1.f - c * c);
    }
    if (lane == 0) {
        p.dq[b] = p.leak ? dq_pi : 0.f;
        p.dq[p.B + b] = dq_c;
        if (p.q_out) { p.q_out[b] = q_pi; p.q_out[p.B + b] = q_c; p.q_out[2 * p.B + b] = q_t; }
        if (p.td_out) p.td_out[b] = td;
        float pen = 0.f;
        if (p.actor_l2 > 0.f)
            for (int a = 0; a < p.A; ++a) pen += p.za[(long)b * p.A + a] * p.za[(long)b * p.A + a];
        atomicAdd(p.loss_sum, (-q_pi + p.actor_l2 * pen) * invB);
        atomicAdd(p.loss_sum + 1, w * td * td * invB);
        if (p.leaf_sum) {
            const float leaf = __powf(fabsf(td) + p.prio_eps, p.prio_alpha);
            p.leaf_sum[ridx] = leaf;
            p.leaf_min[ridx] = leaf;
        }
    }
}

}  // namespace

// jobs: F32Job[n] as laid out above (tiles_j / tile0 / a_vec / b_vec are filled here)
PDA_EXPORT int pda_mlp_f32_jobs(const void* jobs_in, int n, hipStream_t stream) {
    if (n <= 0 || n > F32_MAX_JOBS) return -1;
    F32Jobs jobs;
    const F32Job* in = static_cast<const F32Job*>(jobs_in);
    int tiles = 0;
    for (int k = 0; k < n; ++k) {
        F32Job j = in[k];
        if (j.I <= 0 || j.J <= 0 || j.R <= 0) return -2;
        j.tiles_j = (j.J + 15) / 16;
        j.tile0 = tiles;
        tiles += ((j.I + 15) / 16) * j.tiles_j;
        // 16-byte path: unit reduction stride, every line 16-byte aligned, and the bytes up to the next multiple of 4
        // behind the last element readable (activation buffers are padded; weight rows qualify when R % 4 == 0)
        j.a_vec = (j.a_sr == 1 && j.a_si % 4 == 0 && ((uintptr_t)j.a % 16) == 0 && (j.kind == KIND_FWD || j.kind == KIND_DGRAD)) ? 1 : 0;
        if (j.kind == KIND_FWD && j.a_si < ((j.R + 3) & ~3)) j.a_vec = 0;
        if (j.kind == KIND_DGRAD && j.a_si < ((j.R + 3) & ~3)) j.a_vec = 0;
        j.b_vec = (j.b_sr == 1 && j.b_sj % 4 == 0 && j.R % 4 == 0 && ((uintptr_t)j.b % 16) == 0) ? 1 : 0;
        jobs.j[k] = j;
    }
    jobs.n = n;
    hipLaunchKernelGGL(mlp_f32_jobs_kernel, dim3(tiles), dim3(256), 0, stream, jobs);
    return (int)hipGetLastError();
}
PDA_EXPORT int pda_mlp_f32_job_bytes() { return (int)sizeof(F32Job); }

PDA_EXPORT void pda_mlp_f32_stage(const F32StageArgs* args, hipStream_t stream) {
    F32StageArgs p = *args;
    const int total = p.B * (2 * p.D + p.A);
    int blocks = (total + 255) / 256;
    if (blocks > 256) blocks = 256;
    hipLaunchKernelGGL(mlp_f32_stage_kernel, dim3(blocks), dim3(256), 0, stream, p);
}

PDA_EXPORT void pda_ddpg_head_f32(const DdpgHeadF32Args* args, hipStream_t stream) {
    DdpgHeadF32Args p = *args;
    hipLaunchKernelGGL(ddpg_head_f32_kernel, dim3((p.B + 3) / 4), dim3(256), 0, stream, p);
}
