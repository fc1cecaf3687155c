// Kernels specific to the MLP agents (DDPG actor/critic, MLP DQN): replay-row
// staging into padded bf16 GEMM operands and the fused DDPG critic head
// (critic output layer + n-step TD target + both losses + the gradients that
// start the two backward passes).  The dense layers themselves run on the MFMA
// GEMM family in gemm.hip.
#include "api.h"

// ----------------------------------------------------------------------------
// staging: up to 6 jobs in one launch.  Job j copies `ncols` fp32 columns
// (starting at src_col) of the sampled replay rows into a bf16 matrix:
//   dst[(dst_row0 + b) * ld_dst + d] = bf16(src[row(b) * ld_src + src_col + d] * scale)
// row(b) = idx[b] (HBM replay, sampled indices) or b (host batch already staged).
// Pad columns of dst are never written (they stay zero from allocation).
// ----------------------------------------------------------------------------
// (StageJob / StageJobs live in api.h)

__global__ __launch_bounds__(256) void mlp_stage_kernel(StageJobs jobs, int njobs, const int* __restrict__ idx, int B) {
    const StageJob J = jobs.j[blockIdx.y];
    long total = (long)B * J.ncols;
    for (long t = (long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long)gridDim.x * 256) {
        int b = (int)(t / J.ncols), d = (int)(t - (long)b * J.ncols);
        long r = idx ? (long)idx[b] : (long)b;
        J.dst[(long)(J.dst_row0 + b) * J.ld_dst + d] = f2bf(J.src[r * J.ld_src + J.src_col + d] * J.scale);
    }
}

PDA_EXPORT void pda_mlp_stage(const StageJobs* jobs, int njobs, const void* idx, int B, int max_cols,
                              hipStream_t stream) {
    if (njobs <= 0 || B <= 0) return;
    long total = (long)B * max_cols;
    int bx = (int)((total + 255) / 256);
    if (bx > 256) bx = 256;
    hipLaunchKernelGGL(mlp_stage_kernel, dim3(bx, njobs), dim3(256), 0, stream, *jobs, njobs, (const int*)idx, B);
}

// ----------------------------------------------------------------------------
// DDPG critic head.  One wave64 per sample b; lanes stride over the H hidden units.
// h2 holds the last hidden layer of three critic passes, [3B][ldh] bf16:
//   rows [0,B)    online  Q(s0, pi(s0))        -> actor loss  -mean(q_pi)
//   rows [B,2B)   online  Q(s0, a_replay)      -> critic loss  mean(w (q_c - y)^2)
//   rows [2B,3B)  target  Q'(s_n, pi'(s_n))    -> y = r + gamma^n q_t (1 - done)
// Outputs: dh2 [2B][ldh] bf16 = dL/d(pre-activation of that layer) for the two online
// passes (tanh' folded in), dq [2B][dq_ld] bf16 (column 0) for the output layer's
// weight gradient, |td| leaves for the sum/min trees, the two loss sums.
// With leak == 0 the actor-loss rows of dq are zero, so critic parameters only see the
// critic loss (the reference lets the actor loss leak into them, ddpg_learner.py:66-86).
// actor_warmup > 0 zeroes the actor-loss gradient for the first updates: the tanh policy is not pushed into
// saturation by the gradient of a critic that has not seen any data yet (agent_params.actor_warmup_steps).
// ----------------------------------------------------------------------------
struct DdpgHeadArgs {
    const bf16_t* h2; long ldh;
    const float* w3_on; const float* b3_on; const float* w3_tg; const float* b3_tg;
    const float* tails; int tail_ld; int off_reward, off_gamma, off_term;
    const int* idx; const float* isw;
    bf16_t* dh2; bf16_t* dq; int dq_ld;
    float* q_out; float* td_out; float* loss_sum;          // loss_sum[0] = actor, [1] = critic
    float* leaf_sum; float* leaf_min; float prio_eps, prio_alpha;
    int B, H, leak;
    int actor_warmup;       // the actor loss is switched off while the update count is below this (critic warm-up)
    unsigned long long* rng_counter; int* opt_step;
};

__global__ __launch_bounds__(256) void ddpg_head_kernel(DdpgHeadArgs p) {
    int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    int b = blockIdx.x * 4 + wave;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        if (p.rng_counter) *p.rng_counter += 1;
        if (p.opt_step) *p.opt_step += 1;
    }
    if (b >= p.B) return;
    const bf16_t* hp = p.h2 + (long)b * p.ldh;
    const bf16_t* hc = p.h2 + (long)(p.B + b) * p.ldh;
    const bf16_t* ht = p.h2 + (long)(2 * p.B + b) * p.ldh;
    float sp = 0.f, sc = 0.f, st = 0.f;
    for (int j = lane; j < p.H; j += 64) {
        float wo = p.w3_on[j];
        sp += wo * bf2f(hp[j]);
        sc += wo * bf2f(hc[j]);
        st += p.w3_tg[j] * bf2f(ht[j]);
    }
    float q_pi = wave_sum(sp) + p.b3_on[0];
    float q_c = wave_sum(sc) + p.b3_on[0];
    float q_t = wave_sum(st) + p.b3_tg[0];

    int ridx = p.idx ? p.idx[b] : b;
    const float* tl = p.tails + (long)ridx * p.tail_ld;
    float y = tl[p.off_reward] + tl[p.off_gamma] * q_t * (1.f - tl[p.off_term]);
    float td = q_c - y;
    float w = p.isw ? p.isw[b] : 1.f;
    float invB = 1.f / (float)p.B;
    float dq_c = 2.f * w * td * invB;
    // critic warm-up: no actor-loss gradient yet.  (The counter is bumped by workgroup 0 of this launch: at the
    // one boundary step some samples may see either value, which is harmless.)
    const bool actor_on = !(p.actor_warmup > 0 && p.opt_step && *p.opt_step < p.actor_warmup);
    float dq_pi = actor_on ? -invB : 0.f;

    bf16_t* dp = p.dh2 + (long)b * p.ldh;
    bf16_t* dc = p.dh2 + (long)(p.B + b) * p.ldh;
    for (int j = lane; j < p.H; j += 64) {
        float wo = p.w3_on[j];
        float a = bf2f(hp[j]), c = bf2f(hc[j]);
        dp[j] = f2bf(dq_pi * wo * (1.f - a * a));
        dc[j] = f2bf(dq_c * wo * (1.f - c * c));
    }
    if (lane < p.dq_ld) {
        p.dq[(long)b * p.dq_ld + lane] = f2bf((lane == 0 && p.leak) ? dq_pi : 0.f);
        p.dq[(long)(p.B + b) * p.dq_ld + lane] = f2bf(lane == 0 ? dq_c : 0.f);
    }
    if (lane == 0) {
        if (p.q_out) { p.q_out[b] = q_pi; p.q_out[p.B + b] = q_c; p.q_out[2 * p.B + b] = q_t; }
        if (p.td_out) p.td_out[b] = td;
        atomicAdd(p.loss_sum, -q_pi * invB);
        atomicAdd(p.loss_sum + 1, w * td * td * invB);
        if (p.leaf_sum) {
            float leaf = __powf(fabsf(td) + p.prio_eps, p.prio_alpha);
            p.leaf_sum[ridx] = leaf;
            p.leaf_min[ridx] = leaf;
        }
    }
}

PDA_EXPORT void pda_ddpg_head(const DdpgHeadArgs* args, hipStream_t stream) {
    DdpgHeadArgs p = *args;
    hipLaunchKernelGGL(ddpg_head_kernel, dim3((p.B + 3) / 4), dim3(256), 0, stream, p);
}

// ----------------------------------------------------------------------------
// deterministic-policy epilogue for the batched actor server: out[b][0..A) =
// float(action bf16) (the actor's last layer already applied tanh).
// ----------------------------------------------------------------------------
__global__ void bf16_rows_to_f32_kernel(const bf16_t* in, long ld_in, float* out, long ld_out, int rows, int cols) {
    long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long)rows * cols) return;
    int r = (int)(t / cols), c = (int)(t - (long)r * cols);
    out[(long)r * ld_out + c] = bf2f(in[(long)r * ld_in + c]);
}
PDA_EXPORT void pda_bf16_rows_to_f32(const void* in, long ld_in, void* out, long ld_out, int rows, int cols,
                                     hipStream_t stream) {
    long total = (long)rows * cols;
    if (total <= 0) return;
    hipLaunchKernelGGL(bf16_rows_to_f32_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream,
                       (const bf16_t*)in, ld_in, (float*)out, ld_out, rows, cols);
}
