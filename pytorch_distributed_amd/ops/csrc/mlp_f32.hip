// fp32 MLP learner kernels: the reference-precision path of the MLP agents (Distributed DDPG, MLP-DQN bodies).
//
// The reference's DDPG learner is plain fp32 PyTorch (/root/reference/core/single_processes/ddpg_learner.py:66-95,
// core/models/ddpg_mlp_model.py:16-72).  The bf16 engine (mlp_fused.hip) rounds every operand to 8 significand bits
// (gradients within ~6 % of fp32); this file runs the SAME step on gfx950's fp32-input matrix instruction,
// v_mfma_f32_16x16x4_f32: products and sums are exact fp32 FMA chains, operands are the fp32 master weights themselves
// (no shadow copies, no transposed copies), activations and gradients stay fp32 in HBM / L2.  The problem is ~0.1 GFLOP per
// step, so the instruction's rate (1/16 of bf16 MFMA) is irrelevant; a step is a chain of DEPENDENT launches and each
// costs 3-5 us of latency whatever it computes, so the design minimises (a) the number of launches -- seven per step --
// and (b) the memory round trips and instruction bytes inside each:
//
//   a launch is a LIST of small GEMMs of one kind (forward layers of the online and target networks; a data gradient;
//   all six weight gradients), cut into 16x16 output tiles, one workgroup per tile, the reduction split over its waves:
//
//   C[i][j] = sum_r A(i, r) B(r, j)         A(i, r) = a[i a_si + r a_sr],   B(r, j) = b[r b_sr + j b_sj]
//
//   forward      i = batch row, j = output unit, r = input unit:  A = x (rows), B = W[j][r] (rows)      -> act(C + bias)
//   data grad    i = batch row, j = input unit,  r = output unit: A = dy (rows), B = W[r][j] (columns)  -> C * act'(y_in)
//   weight grad  i = output unit, j = input unit, r = batch row:  A = dy (columns), B = x (columns)     -> dW, db = sum_r A
//
//   the narrow layers (actor output, critic Q: 1-17 units) never get a launch of their own: the layer before leaves
//   per-tile partial dot products in its epilogue, the layer after finishes them in its prologue (F32Job: narrow / tail /
//   producer), and the critic head (TD target, both losses, dL/dQ, PER leaves) runs at the top of the first
//   data-gradient launch.  The replay gather, the step / rng counters ride in the first / the weight-gradient launch.
//
// Lane l of a wave holds row / column l & 15 and reduction elements 16 c + 4 (l >> 4) + e, e = 0..3, of chunk c, and
// element e of both operands goes into MFMA e of the chunk (a sum does not care in which order its terms arrive, as long
// as both operands use the same order).  Measurements behind the layout: profiles/r3_ddpg_fp32_step.txt.
#include "api.h"

namespace {

constexpr int F32_MAX_JOBS = 6;
constexpr int KIND_FWD = 0, KIND_DGRAD = 1, KIND_WGRAD = 2;

// One small GEMM of a launch.  Besides the plain operands a job can carry the pieces that let a whole DDPG step run in
// SEVEN launches (each dependent launch costs ~4.5 us on this chip, the arithmetic almost nothing):
//   gather     A row i is replay row a_idx[i % a_mod]: the first layers read the replay directly
//   narrow     forward epilogue: the NEXT layer is narrow (the actor's output layer, the critic's Q layer: 1-17 units over
//              300 inputs), so each output tile also leaves its share of that layer's dot products,
//              npart[tile_j][i][k] = sum_{j in tile} y[i][j] nar_w[k][j]; the consumer adds the tiles in a fixed order
//   tail       forward loader: A columns >= t_col0 of rows < t_rows are act(bias + sum of such partials) -- torch.cat's
//              second half, the online / target action, computed where it is consumed (tile column 0 stores it back)
//   producer   data-gradient loader: A(i, r) = (sum_{k < p_n} P(i, k) p_q[k][r]) act'(p_y[i][r]) -- the data gradient of a
//              narrow layer, formed in registers; P(i, .) comes from memory or, p_head, from the critic head below
struct F32Job {
    const float* a; long a_si, a_sr;
    const float* b; long b_sr, b_sj;
    int I, J, R;                        // output rows, output columns, reduction length
    float* c; long c_ld; int c_col0;    // C[i c_ld + c_col0 + j]
    float c_scale;                      // forward: act(c_scale C + bias) (the 1 / norm_val of a raw input layer);
                                        // weight gradient: dW = c_scale * A^T B (the same factor, on the input side)
    const float* bias;                  // forward: per output column (may be null)
    const float* yin; long yin_ld;      // data gradient: the activation OUTPUT at (i, j), for act'
    int act, kind;
    float* gb;                          // weight gradient: bias gradient sum_r A(i, r) (may be null)
    // data gradient extra (DDPG: the actor's pre-activation penalty enters behind tanh'): rows i < x_rows, columns
    // j >= x_col0 get + x_coef * [actor loss on] * x_src[i x_ld + j - x_col0]; the same columns of those rows are also
    // copied to x_out[i x_out_ld + j - x_col0] (dL/dz of the actor's output layer, contiguous for the next launch)
    const float* x_src; long x_ld; int x_col0, x_rows; float x_coef; float* x_out; long x_out_ld;
    const int* a_idx; int a_mod;
    const float* nar_w; long nar_ld; int nar_n; float* npart;
    const float* t_part; const float* t_bias; float* t_z; int t_col0, t_rows, t_tiles, t_I, t_n;
    const float* p_p; const float* p_q; long p_qld; const float* p_y; long p_yld; float* p_out; long p_outld;
    int p_n, p_act, p_head;
    int tiles_j, tiles_i, tile0;        // host-filled: tile counts, first workgroup of the job
};

// replay rows -> the fp32 operands that later launches read as columns (weight gradients) or rows (the replay action
// inside the critic's concat rows); rides in the first launch as extra workgroups
struct F32StageArgs {
    const float* s0; long ld_s; const float* tails; int tail_ld; int off_action;
    const int* idx; int B, D, A;
    float* x0; long ld_x0;          // [2B][ld]: rows b and B + b = s0[idx[b]]
    float* cat; long ld_cat; int c1;    // rows B + b, columns c1 .. c1 + A: the replay action
    float* hd; int off_reward, off_gamma, off_term;     // [B][4]: reward, discount, terminal flag of the sampled rows
};

// DDPG critic head: Q values from the narrow-layer partials, n-step target, both losses, dL/dQ, PER leaves
struct F32Head {
    const float* qp_on; const float* qp_tg; int q_tiles;    // [q_tiles][2B], [q_tiles][B]
    const float* b3_on; const float* b3_tg;
    const float* hd;                                        // [B][4] (F32StageArgs::hd)
    const int* idx; const float* isw;
    float* dq;                                              // [2B]: dL/dQ of the two online passes (row 0..B: 0 unless leak)
    const float* za; int A; float actor_l2;                 // the online actor's pre-tanh output [B][A] (penalty term)
    float* q_out; float* td_out; float* loss_sum; float* leaf_sum; float* leaf_min; float prio_eps, prio_alpha;
    int B, leak, actor_warmup;
    const int* opt_step;
};

struct F32Launch {
    F32Job j[F32_MAX_JOBS];
    int n, tiles, stage_blocks, pad_;
    F32StageArgs stage;
    F32Head head;
    unsigned long long* bump_counter; int* bump_step;      // workgroup 0 bumps these (the launch after their last reader)
};

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

// 4 reduction elements r0 .. r0+3 of one operand row / column, for each of the CB chunks of a pass: the REQUEST only
// (clamped addresses, no branch around a load, nothing that waits on the data) -- every request of a wave is in flight
// before the first use; mask4 at the use.  Plain 4-byte loads: a 16-byte variant doubled the code of every launch for
// nothing measurable (these launches are latency, not issue, bound).
template <int CB, int NW>
__device__ __forceinline__ void request(const float* base, long line_off, long sr, int c0, int kg, int R,
                                        float (&v)[CB][4]) {
#pragma unroll
    for (int u = 0; u < CB; ++u)
#pragma unroll
        for (int e = 0; e < 4; ++e)
            v[u][e] = base[line_off + (long)min((c0 + NW * u) * 16 + kg * 4 + e, R - 1) * sr];
}
__device__ __forceinline__ void mask4(float (&v)[4], int r0, int R, bool line_ok) {
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = (line_ok && r0 + e < R) ? v[e] : 0.f;
}

__device__ __forceinline__ bool actor_on(const F32Head& h) {
    // critic warm-up: no actor-loss gradient yet (the step counter is bumped by a LATER launch of the step)
    return !(h.actor_warmup > 0 && h.opt_step && *h.opt_step < h.actor_warmup);
}

__device__ __forceinline__ float group16_sum(float s) {
    s += __shfl_xor(s, 1, 64);
    s += __shfl_xor(s, 2, 64);
    s += __shfl_xor(s, 4, 64);
    s += __shfl_xor(s, 8, 64);
    return s;
}

// rows i0 .. i0+15 of the [policy pass | critic pass] batch: dL/dQ into dqs[16]; `writer` workgroups (one per row tile)
// also leave the losses, Q values, TD errors and PER leaves.  16 threads per row: the tile partials of the Q layer arrive
// in ONE memory round trip (a thread walking all 19 of them is 19 dependent trips, longer than the launch it replaces).
__device__ __forceinline__ void head_rows(const F32Head& h, int i0, bool writer, float* dqs) {
    if (threadIdx.x < 256) {
        const int row = threadIdx.x >> 4, slot = threadIdx.x & 15;
        const int i = i0 + row;
        const int ic = min(i, 2 * h.B - 1);
        const int b = max(ic - h.B, 0);
        // every load of the head up front (one round trip): the sample's reward / discount / terminal flag come from
        // the compact copy that the step's first launch left (no index -> replay row chain here)
        const float rew = h.hd[b * 4], gam = h.hd[b * 4 + 1], term = h.hd[b * 4 + 2];
        const float w = h.isw ? h.isw[b] : 1.f;
        const float b3o = h.b3_on[0], b3t = h.b3_tg[0];
        const bool on = actor_on(h);
        const int k1 = min(slot + 16, h.q_tiles - 1);
        float q = h.qp_on[(long)slot * 2 * h.B + ic], q1 = h.qp_on[(long)k1 * 2 * h.B + ic];
        float qt = h.qp_tg[(long)slot * h.B + b], qt1 = h.qp_tg[(long)k1 * h.B + b];
        if (slot >= h.q_tiles) q = qt = 0.f;
        if (slot + 16 < h.q_tiles) { q += q1; qt += qt1; }
        for (int k = slot + 32; k < h.q_tiles; k += 16) {
            q += h.qp_on[(long)k * 2 * h.B + ic];
            qt += h.qp_tg[(long)k * h.B + b];
        }
        q = group16_sum(q) + b3o;
        qt = group16_sum(qt) + b3t;
        if (slot == 0) {
            const float invB = 1.f / (float)h.B;
            const float td = q - (rew + gam * qt * (1.f - term));
            const bool policy = i < h.B;
            float dq = policy ? (on ? -invB : 0.f) : 2.f * w * td * invB;
            if (i >= 2 * h.B) dq = 0.f;
            dqs[row] = dq;
            if (writer && i < 2 * h.B) {
                if (policy) {
                    float pen = 0.f;
                    if (h.actor_l2 > 0.f)
                        for (int a = 0; a < h.A; ++a) pen += h.za[(long)i * h.A + a] * h.za[(long)i * h.A + a];
                    atomicAdd(h.loss_sum, (-q + h.actor_l2 * pen) * invB);
                    h.dq[i] = h.leak ? dq : 0.f;
                    if (h.q_out) h.q_out[i] = q;
                } else {
                    h.dq[i] = dq;
                    if (h.q_out) { h.q_out[i] = q; h.q_out[2 * h.B + b] = qt; }
                    if (h.td_out) h.td_out[b] = td;
                    atomicAdd(h.loss_sum + 1, w * td * td * invB);
                    if (h.leaf_sum) {
                        const int ridx = h.idx ? h.idx[b] : b;
                        const float leaf = __powf(fabsf(td) + h.prio_eps, h.prio_alpha);
                        h.leaf_sum[ridx] = leaf;
                        h.leaf_min[ridx] = leaf;
                    }
                }
            }
        }
    }
    __syncthreads();
}

// the narrow layer finished for rows i0 .. i0+15: zt[row][k] = bias[k] + sum over tile columns of its partials
constexpr int ZT_LD = 33;
__device__ __forceinline__ void tail_rows(const F32Job& J, int i0, float* zt) {
    if (threadIdx.x < 256) {
        const int row = threadIdx.x >> 4, slot = threadIdx.x & 15;
        const int ic = min(i0 + row, J.t_rows - 1);
        const int k1 = min(slot + 16, J.t_tiles - 1);
        const bool has0 = slot < J.t_tiles, has1 = slot + 16 < J.t_tiles;
        for (int a0 = 0; a0 < J.t_n; a0 += 4) {         // four outputs' loads in flight together
            float p0[4], p1[4], bs[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int a = min(a0 + u, J.t_n - 1);
                p0[u] = J.t_part[((long)min(slot, J.t_tiles - 1) * J.t_I + ic) * J.t_n + a];
                p1[u] = J.t_part[((long)k1 * J.t_I + ic) * J.t_n + a];
                bs[u] = J.t_bias[a];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int a = a0 + u;
                float sdot = (has0 ? p0[u] : 0.f) + (has1 ? p1[u] : 0.f);
                if (a < J.t_n)
                    for (int q = slot + 32; q < J.t_tiles; q += 16) sdot += J.t_part[((long)q * J.t_I + ic) * J.t_n + a];
                sdot = group16_sum(sdot);
                if (slot == 0 && a < J.t_n) zt[row * ZT_LD + a] = sdot + bs[u];
            }
        }
    }
    __syncthreads();
}

__device__ __forceinline__ void stage_block(const F32StageArgs& p, int bid, int nblocks) {
    const int per = p.D + p.A + 3;
    const int total = p.B * per;
    for (int t = bid * (int)blockDim.x + threadIdx.x; t < total; t += nblocks * (int)blockDim.x) {
        const int b = t / per, d = t - b * per;
        const long row = p.idx ? (long)p.idx[b] : (long)b;
        if (d < p.D) {
            const float v = p.s0[row * p.ld_s + d];
            p.x0[(long)b * p.ld_x0 + d] = v;
            p.x0[(long)(p.B + b) * p.ld_x0 + d] = v;
        } else if (d < p.D + p.A) {
            p.cat[(long)(p.B + b) * p.ld_cat + p.c1 + (d - p.D)] = p.tails[row * p.tail_ld + p.off_action + (d - p.D)];
        } else {
            const int f = d - p.D - p.A;
            p.hd[b * 4 + f] = p.tails[row * p.tail_ld + (f == 0 ? p.off_reward : f == 1 ? p.off_gamma : p.off_term)];
        }
    }
}

// Which tile a workgroup runs.  (Tile order is plain row-major: dealing tile columns to XCDs so that each weight slice is
// fetched into one L2 only was measured -- no gain in isolation, a loss inside the step: these launches are bound by
// latency, not by L2 fill traffic.)
__device__ __forceinline__ bool tile_of(const F32Job& J, int t, int& ti, int& tj) {
    ti = t / J.tiles_j;
    tj = t - ti * J.tiles_j;
    return ti < J.tiles_i;
}
__device__ __forceinline__ int job_of(const F32Launch& L) {
    int ji = 0;
#pragma unroll
    for (int k = 1; k < F32_MAX_JOBS; ++k)
        if (k < L.n && (int)blockIdx.x >= L.j[k].tile0) ji = k;
    return ji;
}

// ---- forward / data gradient: operands whose reduction runs ALONG memory rows ----------------------------------------------
// A(i, r) = a[i a_si + r] (and, forward, B(r, j) = w[j b_sj + r]): lane l & 15 of an MFMA operand is a different row, so
// loading operands straight into registers makes every load instruction touch 16 rows -- the launch time then grows with
// the reduction length alone (3.3 us at R = 64, 7.4 us at R = 300, whatever the tile count; tools/probes/f32_job_probe.py).
// Here the NW waves first copy the tile's 16 operand rows into LDS with row-contiguous loads (wave w: rows w, w + NW, ..;
// lanes along the row), then read their MFMA operands from LDS as 16-byte vectors (row pitch = 4 mod 64 words: the 16
// lanes of a quarter-wave hit 64 different banks).  A producer (F32Job) forms its A rows in the same pass, a lane per
// element, and the first tile column stores them with coalesced writes.
// NW waves x CB chunks of 16 cover the launch's longest reduction in one pass (R <= min(NW CB 16, 448)).
template <int KIND, int NW, int CB, bool SPECIAL>
__global__ __launch_bounds__(NW * 64) void mlp_f32_rows_kernel(F32Launch L) {
    constexpr int RCAP = NW * CB * 16 > 448 ? 448 : NW * CB * 16;
    constexpr int PITCH = RCAP + 4;
    constexpr int SLOTS = (RCAP + 63) / 64;
    constexpr int RW = 16 / NW;                 // rows a wave copies (NW = 4: 4, NW = 16: 1)
    constexpr bool B_ROWS = KIND == KIND_FWD;   // forward: the weights are row operands too
    constexpr int STAGE_F = 16 * PITCH * (B_ROWS ? 2 : 1), RED_F = NW * 256;
    __shared__ __attribute__((aligned(16))) float smem[STAGE_F > RED_F ? STAGE_F : RED_F];
    __shared__ float dqs[16];
    __shared__ float zt[16 * ZT_LD];
    float* lds_a = smem;
    float* lds_b = smem + 16 * PITCH;
    if (KIND == KIND_FWD && (int)blockIdx.x >= L.tiles) {
        stage_block(L.stage, (int)blockIdx.x - L.tiles, L.stage_blocks);
        return;
    }
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const F32Job& J = L.j[job_of(L)];
    int ti, tj;
    if (!tile_of(J, (int)blockIdx.x - J.tile0, ti, tj)) return;     // (whole workgroup, before any barrier)
    const int i0 = ti * 16, j0 = tj * 16;
    const int ln = lane & 15, kg = lane >> 4;
    const bool writer = tj == 0;
    const int nchunks = (J.R + 15) >> 4;
    const bool prod = KIND == KIND_DGRAD && SPECIAL && J.p_q != nullptr;
    const bool tail = KIND == KIND_FWD && SPECIAL && J.t_part != nullptr && i0 < J.t_rows;
    // epilogue operands, requested now so that their latency hides behind the reduction
    const int ie = i0 + kg * 4 + wave, je = j0 + ln;
    const bool oke = wave < 4 && ie < J.I && je < J.J;
    float e_bias = 0.f, e_nar = 0.f, e_yin = 0.f;
    if (KIND == KIND_FWD) {
        if (J.bias != nullptr && wave < 4) e_bias = J.bias[min(je, J.J - 1)];
        if (J.nar_w != nullptr && oke) e_nar = J.nar_w[je];
    }
    if (KIND == KIND_DGRAD && J.yin != nullptr && oke) e_yin = J.yin[(long)ie * J.yin_ld + je];
    // ---- requests: this wave's operand rows (a lane per row element), and, data gradient, its B chunks
    float ra[RW][SLOTS], rb[RW][SLOTS], rq[SLOTS], bv[CB][4];
    float pk0[RW];
#pragma unroll
    for (int rw = 0; rw < RW; ++rw) {
        const int row = wave * RW + rw;
        const int gi = min(i0 + row, J.I - 1);
        pk0[rw] = 0.f;
        if (prod) {
            if (!J.p_head) pk0[rw] = J.p_p[(long)gi * J.p_n];
#pragma unroll
            for (int sl = 0; sl < SLOTS; ++sl) ra[rw][sl] = J.p_y[(long)gi * J.p_yld + min(sl * 64 + lane, J.R - 1)];
        } else {
            long arow = gi;
            if (KIND == KIND_FWD) {
                if (J.a_mod > 0) arow = gi % J.a_mod;
                if (J.a_idx) arow = J.a_idx[arow];
            }
#pragma unroll
            for (int sl = 0; sl < SLOTS; ++sl) ra[rw][sl] = J.a[arow * J.a_si + min(sl * 64 + lane, J.R - 1)];
        }
        if (B_ROWS) {
            const long brow = min(j0 + row, J.J - 1);
#pragma unroll
            for (int sl = 0; sl < SLOTS; ++sl) rb[rw][sl] = J.b[brow * J.b_sj + min(sl * 64 + lane, J.R - 1)];
        }
    }
    if (prod) {
#pragma unroll
        for (int sl = 0; sl < SLOTS; ++sl) rq[sl] = J.p_q[min(sl * 64 + lane, J.R - 1)];
    }
    if (!B_ROWS) request<CB, NW>(J.b, (long)min(j0 + ln, J.J - 1) * J.b_sj, J.b_sr, wave, kg, J.R, bv);
    // ---- prologues (they wait on their own loads: behind the requests above)
    if (SPECIAL) {
        if (KIND == KIND_DGRAD && prod && J.p_head) {
            head_rows(L.head, i0, writer, dqs);
#pragma unroll
            for (int rw = 0; rw < RW; ++rw) pk0[rw] = dqs[wave * RW + rw];
        }
        if (KIND == KIND_FWD && tail) tail_rows(J, i0, zt);
    }
    // ---- operand rows -> LDS (zero beyond the matrix: the chunks need no masks)
#pragma unroll
    for (int rw = 0; rw < RW; ++rw) {
        const int row = wave * RW + rw;
        const int gi = i0 + row;
        const bool row_ok = gi < J.I;
#pragma unroll
        for (int sl = 0; sl < SLOTS; ++sl) {
            const int k = sl * 64 + lane;
            float v = ra[rw][sl];
            if (KIND == KIND_DGRAD && SPECIAL && prod) {
                float sacc = pk0[rw] * rq[sl];
                for (int kk = 1; kk < J.p_n; ++kk)
                    sacc += J.p_p[(long)min(gi, J.I - 1) * J.p_n + kk] * J.p_q[(long)kk * J.p_qld + min(k, J.R - 1)];
                v = sacc * act_bwd(v, J.p_act);
                if (writer && row_ok && k < J.R) J.p_out[(long)gi * J.p_outld + k] = v;
            }
            if (KIND == KIND_FWD && SPECIAL && tail && gi < J.t_rows && k >= J.t_col0 && k < J.t_col0 + J.t_n) {
                const float zv = zt[row * ZT_LD + (k - J.t_col0)];
                v = act_fwd(zv, ACT_TANH);
                if (writer) {           // the narrow layer's output (and pre-activation) for the weight gradients
                    if (J.t_z) J.t_z[(long)gi * J.t_n + (k - J.t_col0)] = zv;
                    const_cast<float*>(J.a)[(long)gi * J.a_si + k] = v;
                }
            }
            if (k < RCAP) {
                lds_a[row * PITCH + k] = (row_ok && k < J.R) ? v : 0.f;
                if (B_ROWS) lds_b[row * PITCH + k] = (j0 + row < J.J && k < J.R) ? rb[rw][sl] : 0.f;
            }
        }
    }
    __syncthreads();
    // ---- the reduction: chunk c = wave + NW u
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < CB; ++u) {
        const int c = wave + NW * u;
        if (c < nchunks && c * 16 < RCAP) {         // wave-uniform
            const int r0 = c * 16 + kg * 4;
            const float4 a4 = *reinterpret_cast<const float4*>(lds_a + ln * PITCH + r0);
            float b4[4];
            if (B_ROWS) {
                const float4 t4 = *reinterpret_cast<const float4*>(lds_b + ln * PITCH + r0);
                b4[0] = t4.x; b4[1] = t4.y; b4[2] = t4.z; b4[3] = t4.w;
            } else {
                mask4(bv[u], r0, J.R, j0 + ln < J.J);
#pragma unroll
                for (int e = 0; e < 4; ++e) b4[e] = bv[u][e];
            }
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.x, b4[0], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.y, b4[1], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.z, b4[2], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.w, b4[3], acc, 0, 0, 0);
        }
    }
    __syncthreads();            // the partial tiles reuse the operand rows' LDS
    float (*red)[4][64] = reinterpret_cast<float (*)[4][64]>(smem);
#pragma unroll
    for (int q = 0; q < 4; ++q) red[wave][q][lane] = acc[q];
    __syncthreads();
    if (wave >= 4) return;
    float v = 0.f;      // wave w < 4 finishes accumulator register w: row 4 kg + w, column ln
#pragma unroll
    for (int w = 0; w < NW; w += 4)
        v += (red[w][wave][lane] + red[w + 1][wave][lane]) + (red[w + 2][wave][lane] + red[w + 3][wave][lane]);
    const int i = ie, j = je;
    const bool ok = oke;
    if (KIND == KIND_FWD) {
        v = act_fwd(v * J.c_scale + e_bias, J.act);
        if (J.nar_w != nullptr) {
            for (int k = 0; k < J.nar_n; ++k) {
                float sdot = ok ? v * (k == 0 ? e_nar : J.nar_w[(long)k * J.nar_ld + j]) : 0.f;
                sdot = group16_sum(sdot);
                if (ln == 0 && i < J.I) J.npart[((long)tj * J.I + i) * J.nar_n + k] = sdot;
            }
        }
    } else if (ok) {
        if (J.yin) v *= act_bwd(e_yin, J.act);
        if (J.x_out != nullptr && j >= J.x_col0 && i < J.x_rows) {
            if (J.x_src) v += J.x_coef * (actor_on(L.head) ? 1.f : 0.f) * J.x_src[(long)i * J.x_ld + (j - J.x_col0)];
            J.x_out[(long)i * J.x_out_ld + (j - J.x_col0)] = v;
        }
    }
    if (ok) J.c[(long)i * J.c_ld + J.c_col0 + j] = v;
}

// ---- weight gradient: the reduction runs over batch rows, both operands are read along their unit-stride side ------------
template <int NW, int CB>
__global__ __launch_bounds__(NW * 64) void mlp_f32_wgrad_kernel(F32Launch L) {
    __shared__ float red[NW][4][64];
    __shared__ float bsum[NW][16];
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        if (L.bump_counter) *L.bump_counter += 1;
        if (L.bump_step) *L.bump_step += 1;
    }
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const F32Job& J = L.j[job_of(L)];
    int ti, tj;
    if (!tile_of(J, (int)blockIdx.x - J.tile0, ti, tj)) return;
    const int i0 = ti * 16, j0 = tj * 16;
    const int ln = lane & 15, kg = lane >> 4;
    const int ia = i0 + ln, jb = j0 + ln;
    const bool ia_ok = ia < J.I, jb_ok = jb < J.J;
    const long a_off = (long)min(ia, J.I - 1) * J.a_si, b_off = (long)min(jb, J.J - 1) * J.b_sj;
    const int nchunks = (J.R + 15) >> 4;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    float asum = 0.f;
    int c0 = wave;
    do {
        float av[CB][4], bv[CB][4];
        request<CB, NW>(J.a, a_off, J.a_sr, c0, kg, J.R, av);
        request<CB, NW>(J.b, b_off, J.b_sr, c0, kg, J.R, bv);
#pragma unroll
        for (int u = 0; u < CB; ++u) {
            const int r0 = (c0 + NW * u) * 16 + kg * 4;
            mask4(av[u], r0, J.R, ia_ok);
            mask4(bv[u], r0, J.R, jb_ok);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u][e], bv[u][e], acc, 0, 0, 0);
                asum += av[u][e];
            }
        }
        c0 += NW * CB;
    } while (c0 < nchunks);
#pragma unroll
    for (int q = 0; q < 4; ++q) red[wave][q][lane] = acc[q];
    if (J.gb != nullptr && tj == 0) {
        asum += __shfl_xor(asum, 16, 64);
        asum += __shfl_xor(asum, 32, 64);
        if (lane < 16) bsum[wave][lane] = asum;
    }
    __syncthreads();
    if (wave >= 4) return;
    float v = 0.f;
#pragma unroll
    for (int w = 0; w < NW; w += 4)
        v += (red[w][wave][lane] + red[w + 1][wave][lane]) + (red[w + 2][wave][lane] + red[w + 3][wave][lane]);
    const int i = i0 + kg * 4 + wave, j = j0 + ln;
    if (J.gb != nullptr && tj == 0 && threadIdx.x < 16 && i0 + (int)threadIdx.x < J.I) {
        float sb = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) sb += bsum[w][threadIdx.x];
        J.gb[i0 + threadIdx.x] = sb;
    }
    if (i < J.I && j < J.J) J.c[(long)i * J.c_ld + J.c_col0 + j] = v * J.c_scale;
}

}  // namespace

// launch: F32Launch as laid out above (tiles / tiles_j / tile0 / the 16-byte flags are filled here)
PDA_EXPORT int pda_mlp_f32_launch(const void* launch_in, hipStream_t stream) {
    F32Launch L = *static_cast<const F32Launch*>(launch_in);
    if (L.n <= 0 || L.n > F32_MAX_JOBS) return -1;
    int tiles = 0, maxchunks = 1;
    const int kind = L.j[0].kind;
    bool special = false;
    for (int k = 0; k < L.n; ++k) {
        F32Job& j = L.j[k];
        if (j.I <= 0 || j.J <= 0 || j.R <= 0 || j.kind != kind || kind < KIND_FWD || kind > KIND_WGRAD) return -2;
        if (j.c_scale == 0.f) j.c_scale = 1.f;
        maxchunks = max(maxchunks, (j.R + 15) / 16);
        special = special || j.p_q != nullptr || j.t_part != nullptr;
        if ((j.p_q != nullptr && kind != KIND_DGRAD) || (j.t_part != nullptr && kind != KIND_FWD) ||
            ((j.a_idx != nullptr || j.a_mod > 0 || j.nar_w != nullptr) && kind != KIND_FWD)) return -8;
        j.tiles_j = (j.J + 15) / 16;
        j.tiles_i = (j.I + 15) / 16;
        j.tile0 = tiles;
        tiles += j.tiles_i * j.tiles_j;
        if (j.p_q != nullptr) {
            if (j.p_y == nullptr || j.p_out == nullptr || (!j.p_head && j.p_p == nullptr)) return -3;
        } else if (j.a == nullptr) {
            return -4;
        }
        if (kind != KIND_WGRAD && ((j.p_q == nullptr && j.a_sr != 1) || (kind == KIND_FWD && j.b_sr != 1))) return -11;
        if (j.t_part != nullptr && (j.a_sr != 1 || j.a_idx != nullptr || j.a_mod > 0 || j.t_rows > j.I || j.t_rows <= 0 ||
                                    j.t_n <= 0 || j.t_n >= ZT_LD)) return -5;
    }
    L.tiles = tiles;
    if (L.stage_blocks < 0 || L.stage_blocks > 256 || (L.stage_blocks > 0 && kind != KIND_FWD)) return -6;
    if ((L.bump_counter || L.bump_step) && kind != KIND_WGRAD) return -7;
    // (waves per workgroup, chunks per wave): the longest reduction of the launch in one pass
    const int shape = maxchunks <= 4 ? 0 : maxchunks <= 8 ? 1 : maxchunks <= 16 ? 2 : 3;
    if (kind != KIND_WGRAD && maxchunks > 28) return -10;       // row operands live in LDS: R <= 448
    const dim3 grid(tiles + L.stage_blocks);
#define PDA_F32_ROWS(K, W, C, S) hipLaunchKernelGGL((mlp_f32_rows_kernel<K, W, C, S>), grid, dim3(W * 64), 0, stream, L)
#define PDA_F32_SHAPE(K, S)                             \
    do {                                                \
        if (shape == 0) PDA_F32_ROWS(K, 4, 1, S);       \
        else if (shape == 1) PDA_F32_ROWS(K, 4, 2, S);  \
        else if (shape == 2) PDA_F32_ROWS(K, 16, 1, S); \
        else PDA_F32_ROWS(K, 16, 2, S);                 \
    } while (0)
    if (kind == KIND_FWD) { if (special) PDA_F32_SHAPE(KIND_FWD, true); else PDA_F32_SHAPE(KIND_FWD, false); }
    else if (kind == KIND_DGRAD) { if (special) PDA_F32_SHAPE(KIND_DGRAD, true); else PDA_F32_SHAPE(KIND_DGRAD, false); }
    else if (shape == 0) hipLaunchKernelGGL((mlp_f32_wgrad_kernel<4, 1>), grid, dim3(256), 0, stream, L);
    else if (shape == 1) hipLaunchKernelGGL((mlp_f32_wgrad_kernel<4, 2>), grid, dim3(256), 0, stream, L);
    else if (shape == 2) hipLaunchKernelGGL((mlp_f32_wgrad_kernel<16, 1>), grid, dim3(1024), 0, stream, L);
    else hipLaunchKernelGGL((mlp_f32_wgrad_kernel<16, 2>), grid, dim3(1024), 0, stream, L);
#undef PDA_F32_SHAPE
#undef PDA_F32_ROWS
    return (int)hipGetLastError();
}
PDA_EXPORT int pda_mlp_f32_launch_bytes() { return (int)sizeof(F32Launch); }
PDA_EXPORT int pda_mlp_f32_job_bytes() { return (int)sizeof(F32Job); }
