// RL-specific fused kernels: DQN head (fc2 + n-step TD target + loss + dQ +
// priorities), critic TD for DDPG, fused Adam / Polyak over flat buffers with
// bf16 shadow-weight emission, epsilon-greedy actor epilogue.
#include "api.h"
#include "optim.h"

// ----------------------------------------------------------------------------
// DQN head.  One workgroup (4 waves) per sample; the 512-wide hidden row is 2 values per thread.
//   h      = relu(h1pre + b4)                    (fc1 split-K sums arrive as fp32)
//   q      = W5 h + b5                           (fp32 master weights, A <= 32 actions)
//   y      = r + gamma^n * max_a q_tgt(s_n)[a] * (1 - done)      (double: a* from online(s_n))
//   loss  += w (q[a] - y)^2 / B,   dq[a] = 2 w (q[a]-y) / B
//   dh1    = dq[a] * W5[a,:] * (h > 0)           -> bf16, feeds fc1 wgrad/dgrad
//   leaf   = (|q[a]-y| + eps)^alpha              -> written straight into the sum/min tree leaves
// The fp32 split-K buffers are zeroed after reading so the next step can
// accumulate into them again without a memset node.
// ----------------------------------------------------------------------------
struct DqnHeadArgs {
    float* h1pre_on0; float* h1pre_tg1; float* h1pre_on1;   // [B][HID]; on1 may be null (no double-DQN)
    const float* b4_on; const float* b4_tg;
    const float* w5_on; const float* b5_on; const float* w5_tg; const float* b5_tg;
    const float* tails; int tail_ld; int off_action, off_reward, off_gamma, off_term;
    const int* idx; const float* isw;                        // sampled replay rows, importance weights (may be null)
    bf16_t* h1_on0; bf16_t* dh1; bf16_t* dq; int dq_ld;      // outputs
    float* q_out; float* td_out; float* loss_sum;           // loss_sum: [B] per-sample accumulators
    float* leaf_sum; float* leaf_min; float prio_eps, prio_alpha;
    int B, HID, A;
    unsigned long long* rng_counter; int* opt_step;     // bumped here (nobody reads them in this launch)
    // fp32-precision learner: element offsets to the lo planes of the split-bf16 outputs (0: plain bf16 outputs)
    long h1_plane, dh1_plane, dq_plane;
    // fc1 split-K sums delivered as slabs (pda_linear_fwd_slab): h1pre_*[s * h1_slice_stride + ...], s < h1_slices,
    // added here in slice order; 0 = one atomically accumulated buffer that this kernel zeroes after reading
    int h1_slices; long h1_slice_stride;
    // fp32 learners (ops/mlp_f32_engine.py): the same three outputs as plain fp32 rows (null: not written); the bf16
    // pointers above may then be null
    float* h1_f32; float* dh1_f32; float* dq_f32;
};

// One WORKGROUP (4 waves) per sample: thread t owns hidden columns t PER .. t PER + PER - 1 (PER = HID / 256).
// The kernel is a latency chain, not a throughput problem (128 samples, ~50 KB each way), so everything that does not
// depend on other data is requested in the first instructions -- the index first, so that the tail record can follow
// as soon as it lands, then every split-K slab slice, both bias rows and the first 8 rows of both W5 -- and the wave
// waits once.  (One wave per sample with the loads issued where they were used was 5-6 dependent round trips: 8.6 us.)
template <int PER>
__device__ __forceinline__ void ldv(const float* src, float (&v)[PER]) {
    if constexpr (PER == 2) {
        const float2 t = *reinterpret_cast<const float2*>(src);
        v[0] = t.x; v[1] = t.y;
    } else {
#pragma unroll
        for (int j = 0; j < PER; ++j) v[j] = src[j];
    }
}

template <int HID>
__global__ __launch_bounds__(256) void dqn_head_kernel(DqnHeadArgs p) {
    constexpr int PER = HID / 256;
    __shared__ float part[4][3][32];        // [wave][online(s0) | target(s_n) | online(s_n)][action]
    __shared__ float qfin[3][32];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int b = blockIdx.x;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        if (p.rng_counter) *p.rng_counter += 1;     // the sampler of the NEXT step draws a fresh stream
        if (p.opt_step) *p.opt_step += 1;           // Adam of THIS step sees t = old + 1
    }
    const int col = tid * PER;
    const long row = (long)b * HID + col;
    const bool dbl = p.h1pre_on1 != nullptr;
    // ---- requests
    const int ridx = p.idx ? p.idx[b] : b;
    const float w = p.isw ? p.isw[b] : 1.f;
    float b4o[PER], b4t[PER];
    ldv<PER>(p.b4_on + col, b4o);
    ldv<PER>(p.b4_tg + col, b4t);
    float a0[PER], at[PER], ad[PER];
#pragma unroll
    for (int j = 0; j < PER; ++j) { a0[j] = 0.f; at[j] = 0.f; ad[j] = 0.f; }
    const int nsl = max(p.h1_slices, 1);
    const long sstride = p.h1_slices > 0 ? p.h1_slice_stride : 0;
    float v0[8][PER], vt[8][PER], vd[8][PER];
#pragma unroll
    for (int u = 0; u < 8; ++u) {           // the first 8 slices (fc1 forward at batch 128 makes 8)
        const long o = (long)min(u, nsl - 1) * sstride + row;
        ldv<PER>(p.h1pre_on0 + o, v0[u]);
        ldv<PER>(p.h1pre_tg1 + o, vt[u]);
        if (dbl) ldv<PER>(p.h1pre_on1 + o, vd[u]);
    }
    float w_on8[8][PER], w_tg8[8][PER];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const long o = (long)min(u, p.A - 1) * HID + col;
        ldv<PER>(p.w5_on + o, w_on8[u]);
        ldv<PER>(p.w5_tg + o, w_tg8[u]);
    }
    const float b5o = tid < p.A ? p.b5_on[tid] : 0.f, b5t = tid < p.A ? p.b5_tg[tid] : 0.f;
    const float* tl = p.tails + (long)ridx * p.tail_ld;     // (waits for the index only: it was requested first)
    const float t_act = tl[p.off_action], t_rew = tl[p.off_reward], t_gam = tl[p.off_gamma], t_term = tl[p.off_term];
    // ---- hidden rows: slices added in slice order (bit-reproducible)
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        if (u < nsl) {
#pragma unroll
            for (int j = 0; j < PER; ++j) { a0[j] += v0[u][j]; at[j] += vt[u][j]; if (dbl) ad[j] += vd[u][j]; }
        }
    }
    for (int sl = 8; sl < nsl; ++sl) {
        const long o = (long)sl * sstride + row;
        float x0[PER], xt[PER], xd[PER];
        ldv<PER>(p.h1pre_on0 + o, x0);
        ldv<PER>(p.h1pre_tg1 + o, xt);
        if (dbl) ldv<PER>(p.h1pre_on1 + o, xd);
#pragma unroll
        for (int j = 0; j < PER; ++j) { a0[j] += x0[j]; at[j] += xt[j]; if (dbl) ad[j] += xd[j]; }
    }
    if (p.h1_slices <= 0) {     // one atomically accumulated buffer: zero it for the next step
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            p.h1pre_on0[row + j] = 0.f;
            p.h1pre_tg1[row + j] = 0.f;
            if (dbl) p.h1pre_on1[row + j] = 0.f;
        }
    }
    float h0[PER], ht[PER], hd[PER];
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        h0[j] = fmaxf(a0[j] + b4o[j], 0.f);
        ht[j] = fmaxf(at[j] + b4t[j], 0.f);
        hd[j] = dbl ? fmaxf(ad[j] + b4o[j], 0.f) : 0.f;
    }
    // ---- fc2: per-wave partial dot products, 8 actions at a time (their reductions overlap), combined through LDS
    for (int g0 = 0; g0 < p.A; g0 += 8) {
        float s0[8], st[8], sd[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            float won[PER], wtg[PER];
            if (g0 == 0) {
#pragma unroll
                for (int j = 0; j < PER; ++j) { won[j] = w_on8[u][j]; wtg[j] = w_tg8[u][j]; }
            } else {
                const long o = (long)min(g0 + u, p.A - 1) * HID + col;
                ldv<PER>(p.w5_on + o, won);
                ldv<PER>(p.w5_tg + o, wtg);
            }
            s0[u] = 0.f; st[u] = 0.f; sd[u] = 0.f;
#pragma unroll
            for (int j = 0; j < PER; ++j) {
                s0[u] += h0[j] * won[j];
                st[u] += ht[j] * wtg[j];
                sd[u] += hd[j] * won[j];
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                s0[u] += __shfl_xor(s0[u], o, 64);
                st[u] += __shfl_xor(st[u], o, 64);
                if (dbl) sd[u] += __shfl_xor(sd[u], o, 64);
            }
        }
        if (lane == 0) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (g0 + u < p.A) { part[wave][0][g0 + u] = s0[u]; part[wave][1][g0 + u] = st[u]; part[wave][2][g0 + u] = sd[u]; }
            }
        }
    }
    __syncthreads();
    if (tid < p.A) {
        qfin[0][tid] = ((part[0][0][tid] + part[1][0][tid]) + (part[2][0][tid] + part[3][0][tid])) + b5o;
        qfin[1][tid] = ((part[0][1][tid] + part[1][1][tid]) + (part[2][1][tid] + part[3][1][tid])) + b5t;
        qfin[2][tid] = ((part[0][2][tid] + part[1][2][tid]) + (part[2][2][tid] + part[3][2][tid])) + b5o;
    }
    __syncthreads();
    float best_t = -INFINITY, best_d = -INFINITY;
    int arg_d = 0;
    for (int a = 0; a < p.A; ++a) {
        best_t = fmaxf(best_t, qfin[1][a]);
        if (qfin[2][a] > best_d) { best_d = qfin[2][a]; arg_d = a; }
    }
    const float next_q = dbl ? qfin[1][arg_d] : best_t;
    const int act = min(max((int)t_act, 0), p.A - 1);
    float wa[PER];
#pragma unroll
    for (int j = 0; j < PER; ++j) wa[j] = 0.f;
    if (act < 8) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (u == act) {
#pragma unroll
                for (int j = 0; j < PER; ++j) wa[j] = w_on8[u][j];
            }
    } else {
        ldv<PER>(p.w5_on + (long)act * HID + col, wa);
    }
    const float y = t_rew + t_gam * next_q * (1.f - t_term);
    const float td = qfin[0][act] - y;
    const float dqa = 2.f * w * td / (float)p.B;
    float dv[PER];
#pragma unroll
    for (int j = 0; j < PER; ++j) dv[j] = h0[j] > 0.f ? dqa * wa[j] : 0.f;
    if (p.h1_f32 != nullptr) {
#pragma unroll
        for (int j = 0; j < PER; ++j) { p.h1_f32[row + j] = h0[j]; p.dh1_f32[row + j] = dv[j]; }
        if (tid < p.dq_ld) p.dq_f32[(long)b * p.dq_ld + tid] = tid == act ? dqa : 0.f;
    }
    if (p.h1_on0 == nullptr) {
        // (fp32 outputs only)
    } else if constexpr (PER == 2) {
        const uint32_t hw = pack2bf(h0[0], h0[1]), dw = pack2bf(dv[0], dv[1]);
        *reinterpret_cast<uint32_t*>(p.h1_on0 + row) = hw;
        *reinterpret_cast<uint32_t*>(p.dh1 + row) = dw;
        if (p.h1_plane) {       // residuals of the two rows: x - bf16(x), itself rounded to bf16
            *reinterpret_cast<uint32_t*>(p.h1_on0 + p.h1_plane + row) =
                pack2bf(h0[0] - bf2f((bf16_t)(hw & 0xffffu)), h0[1] - bf2f((bf16_t)(hw >> 16)));
            *reinterpret_cast<uint32_t*>(p.dh1 + p.dh1_plane + row) =
                pack2bf(dv[0] - bf2f((bf16_t)(dw & 0xffffu)), dv[1] - bf2f((bf16_t)(dw >> 16)));
        }
    } else {
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            const bf16_t hh = f2bf(h0[j]), dh = f2bf(dv[j]);
            p.h1_on0[row + j] = hh;
            p.dh1[row + j] = dh;
            if (p.h1_plane) {
                p.h1_on0[p.h1_plane + row + j] = f2bf(h0[j] - bf2f(hh));
                p.dh1[p.dh1_plane + row + j] = f2bf(dv[j] - bf2f(dh));
            }
        }
    }
    if (tid < p.dq_ld && p.dq != nullptr) {
        const float dq = tid == act ? dqa : 0.f;
        const bf16_t dh = f2bf(dq);
        p.dq[(long)b * p.dq_ld + tid] = dh;
        if (p.dq_plane) p.dq[p.dq_plane + (long)b * p.dq_ld + tid] = f2bf(dq - bf2f(dh));
    }
    if (tid < p.A && p.q_out) p.q_out[(long)b * p.A + tid] = qfin[0][tid];
    if (tid == 0) {
        if (p.td_out) p.td_out[b] = td;
        // per-sample loss accumulator (summed on the host side when the logger asks): 128 fp32
        // atomics on one address serialise memory-side and were half of this kernel's time
        p.loss_sum[b] += w * td * td / (float)p.B;
        if (p.leaf_sum) {
            const float leaf = __powf(fabsf(td) + p.prio_eps, p.prio_alpha);
            p.leaf_sum[ridx] = leaf;
            p.leaf_min[ridx] = leaf;
        }
    }
}

PDA_EXPORT void pda_dqn_head(const DqnHeadArgs* args, hipStream_t stream) {
    DqnHeadArgs p = *args;
    if (p.A > 32 || p.dq_ld > 256) return;      // (lane-per-action layout; the python side checks the same)
    dim3 grid(p.B), block(256);
    if (p.HID == 512) hipLaunchKernelGGL(dqn_head_kernel<512>, grid, block, 0, stream, p);
    else if (p.HID == 256) hipLaunchKernelGGL(dqn_head_kernel<256>, grid, block, 0, stream, p);
}

// forward-only head for the actor server: q = W5 relu(h1pre + b4) + b5, then epsilon-greedy
// (ActHeadArgs lives in api.h: the native inference loop fills it too)

template <int HID>
__global__ __launch_bounds__(256) void act_head_kernel(ActHeadArgs p) {
    constexpr int PER = HID / 64;
    int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    int b = blockIdx.x * (blockDim.x >> 6) + wave;   // 1 or 4 waves per workgroup, see the launcher
    if (b >= p.B) return;
    long row = (long)b * HID + lane * PER;
    float h[PER];
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        h[j] = fmaxf(p.h1pre[row + j] + p.b4[lane * PER + j], 0.f);
        p.h1pre[row + j] = 0.f;
    }
    float best = -INFINITY, myq = 0.f;
    int arg = 0;
    for (int a = 0; a < p.A; ++a) {
        float s = 0.f;
        const float* w = p.w5 + (long)a * HID + lane * PER;
#pragma unroll
        for (int j = 0; j < PER; ++j) s += h[j] * w[j];
        s = wave_sum(s) + p.b5[a];
        if (lane == a) myq = s;
        if (s > best) { best = s; arg = a; }
    }
    if (lane < p.A && p.q_out) p.q_out[(long)b * p.A + lane] = myq;
    unsigned long long ctr = p.counter ? *p.counter : 0ull;
    uint32_t r0 = hash_u32(p.seed ^ (ctr * 0x9e3779b97f4a7c15ULL + (unsigned long long)b * 2 + 1));
    uint32_t r1 = hash_u32(p.seed ^ (ctr * 0x9e3779b97f4a7c15ULL + (unsigned long long)b * 2 + 2));
    int action = arg;
    float e = p.eps ? p.eps[b] : 0.f;
    if (e > 0.f && u01(r0) < e) action = (int)(r1 % (uint32_t)p.A);
    float q_action = __shfl(myq, action, 64);
    if (lane == 0) {
        float* o = p.out + (long)b * p.out_ld;
        o[0] = (float)action;
        o[1] = q_action;
        o[2] = best;
    }
}

PDA_EXPORT void pda_act_head(const ActHeadArgs* args, hipStream_t stream) {
    ActHeadArgs p = *args;
    const int wpb = p.B <= 1024 ? 1 : 4;
    dim3 grid((p.B + wpb - 1) / wpb), block(64 * wpb);
    if (p.HID == 512) hipLaunchKernelGGL(act_head_kernel<512>, grid, block, 0, stream, p);
    else if (p.HID == 256) hipLaunchKernelGGL(act_head_kernel<256>, grid, block, 0, stream, p);
}

// ----------------------------------------------------------------------------
// generic n-step TD for value heads given as vectors (DDPG critic, MLP DQN):
//   y = r + gamma^n * next_q * (1 - done);  td = q - y;  dq = 2 w td / B
// ----------------------------------------------------------------------------
__global__ void td_vector_kernel(const float* q, const float* next_q, const float* tails, int tail_ld,
                                 int off_reward, int off_gamma, int off_term, const int* idx, const float* isw,
                                 float* dq, float* td_out, float* loss_sum, float* leaf_sum, float* leaf_min,
                                 float prio_eps, float prio_alpha, int B, unsigned long long* rng_counter,
                                 int* opt_step) {
    int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b == 0) {
        if (rng_counter) *rng_counter += 1;
        if (opt_step) *opt_step += 1;
    }
    float contrib = 0.f;
    if (b < B) {
        int ridx = idx ? idx[b] : b;
        const float* tl = tails + (long)ridx * tail_ld;
        float y = tl[off_reward] + tl[off_gamma] * next_q[b] * (1.f - tl[off_term]);
        float td = q[b] - y;
        float w = isw ? isw[b] : 1.f;
        dq[b] = 2.f * w * td / (float)B;
        if (td_out) td_out[b] = td;
        contrib = w * td * td / (float)B;
        if (leaf_sum) {
            float leaf = __powf(fabsf(td) + prio_eps, prio_alpha);
            leaf_sum[ridx] = leaf;
            leaf_min[ridx] = leaf;
        }
    }
    contrib = wave_sum(contrib);
    if ((threadIdx.x & 63) == 0 && contrib != 0.f) atomicAdd(loss_sum, contrib);
}

PDA_EXPORT void pda_td_vector(const void* q, const void* next_q, const void* tails, int tail_ld, int off_reward,
                              int off_gamma, int off_term, const void* idx, const void* isw, void* dq, void* td_out,
                              void* loss_sum, void* leaf_sum, void* leaf_min, float prio_eps, float prio_alpha,
                              int B, void* rng_counter, void* opt_step, hipStream_t stream) {
    hipLaunchKernelGGL(td_vector_kernel, dim3((B + 255) / 256), dim3(256), 0, stream, (const float*)q,
                       (const float*)next_q, (const float*)tails, tail_ld, off_reward, off_gamma, off_term,
                       (const int*)idx, (const float*)isw, (float*)dq, (float*)td_out, (float*)loss_sum,
                       (float*)leaf_sum, (float*)leaf_min, prio_eps, prio_alpha, B,
                       (unsigned long long*)rng_counter, (int*)opt_step);
}

// flat-buffer optimizer (optim.h): segment table, chunk rules, adam_block
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v,
                                                   bf16_t* __restrict__ shadow, const SegEntry* __restrict__ segs,
                                                   const int2* __restrict__ chunks, int nchunks,
                                                   const float* __restrict__ hyper,
                                                   const int* __restrict__ step, AdamExtras x) {
    __shared__ __attribute__((aligned(16))) float row[OPT_ROW_LDS];
    adam_block(p, g, m, v, shadow, segs, chunks, nchunks, hyper, step, x, (int)blockIdx.x, row);
}

__global__ void tick_kernel(int* step, unsigned long long* counter) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        if (step) *step += 1;
        if (counter) *counter += 1;
    }
}
PDA_EXPORT void pda_tick(void* step, void* counter, hipStream_t stream) {
    hipLaunchKernelGGL(tick_kernel, dim3(1), dim3(64), 0, stream, (int*)step, (unsigned long long*)counter);
}

PDA_EXPORT void pda_adam(void* p, void* g, void* m, void* v, void* shadow, const void* segs, const void* chunks,
                         int nchunks, const void* hyper, void* step, hipStream_t stream) {
    AdamExtras x = {};
    hipLaunchKernelGGL(adam_kernel, dim3(nchunks), dim3(256), 0, stream, (float*)p, (float*)g, (float*)m, (float*)v,
                       (bf16_t*)shadow, (const SegEntry*)segs, (const int2*)chunks, nchunks, (const float*)hyper,
                       (const int*)step, x);
}
// same, with the folded-in slab reduction and/or next-step sampler (AdamExtras above)
PDA_EXPORT void pda_adam_x(void* p, void* g, void* m, void* v, void* shadow, const void* segs, const void* chunks,
                           int nchunks, const void* hyper, void* step, const AdamExtras* extras, hipStream_t stream) {
    AdamExtras x = *extras;
    int extra_blocks = adam_sampler_blocks(x) + adam_repair_blocks(x) + adam_per_sampler_blocks(x);
    hipLaunchKernelGGL(adam_kernel, dim3(nchunks + extra_blocks), dim3(256), 0, stream, (float*)p, (float*)g,
                       (float*)m, (float*)v, (bf16_t*)shadow, (const SegEntry*)segs, (const int2*)chunks, nchunks,
                       (const float*)hyper, (const int*)step, x);
}

// tgt = (1 - tau) tgt + tau src (tau >= 1: hard copy) on fp32 masters, with the target's shadow weights
__global__ __launch_bounds__(256) void polyak_kernel(float* tgt, const float* src,
                                                     bf16_t* __restrict__ shadow, const SegEntry* __restrict__ segs,
                                                     const int2* __restrict__ chunks, float tau, long shadow_plane) {
    int2 ch = chunks[blockIdx.x];
    SegEntry s = segs[ch.x];
    int end = min(ch.y + chunk_len(s), s.count);
    for (int e = ch.y + threadIdx.x; e < end; e += 256) {
        long si;
        long i = master_index(s, e, &si);
        float t = tgt[i] + tau * (src[i] - tgt[i]);
        if (tau >= 1.f) t = src[i];
        tgt[i] = t;
        if (shadow && si >= 0) { put_shadow(shadow, si, t, shadow_plane); put_tr_elem(shadow, s, e, t, shadow_plane); }
    }
}

PDA_EXPORT void pda_polyak(void* tgt, const void* src, void* shadow, const void* segs, const void* chunks,
                           int nchunks, float tau, long shadow_plane, hipStream_t stream) {
    hipLaunchKernelGGL(polyak_kernel, dim3(nchunks), dim3(256), 0, stream, (float*)tgt, (const float*)src,
                       (bf16_t*)shadow, (const SegEntry*)segs, (const int2*)chunks, tau, shadow_plane);
}

// plain multi-tensor Polyak over a flat fp32 buffer (no shadow): used by utils.helpers
__global__ void polyak_flat_kernel(float* tgt, const float* src, long n, float tau) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    long stride = (long)gridDim.x * blockDim.x;
    for (; i < n; i += stride) tgt[i] += tau * (src[i] - tgt[i]);
}
PDA_EXPORT void pda_polyak_flat(void* tgt, const void* src, long n, float tau, hipStream_t stream) {
    long blocks = (n + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(polyak_flat_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, (float*)tgt,
                       (const float*)src, n, tau);
}

// elementwise helpers for the MLP engines ---------------------------------------------------
// out_bf16[i] = bf16(in_f32[i]) over a [rows][cols] block with pitches (pads untouched)
__global__ void f32_to_bf16_kernel(const float* in, long ld_in, bf16_t* out, long ld_out, int rows, int cols) {
    long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long)rows * cols) return;
    int r = (int)(t / cols), c = (int)(t - (long)r * cols);
    out[(long)r * ld_out + c] = f2bf(in[(long)r * ld_in + c]);
}
PDA_EXPORT void pda_f32_to_bf16(const void* in, long ld_in, void* out, long ld_out, int rows, int cols,
                                hipStream_t stream) {
    long total = (long)rows * cols;
    hipLaunchKernelGGL(f32_to_bf16_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream,
                       (const float*)in, ld_in, (bf16_t*)out, ld_out, rows, cols);
}

// gather float state rows (replay, fp32 [M][D]) through idx into a padded bf16 matrix [B][ld]
__global__ void gather_rows_bf16_kernel(const float* src, long ld_src, const int* idx, bf16_t* out, long ld_out,
                                        int B, int D) {
    long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long)B * D) return;
    int b = (int)(t / D), d = (int)(t - (long)b * D);
    int r = idx ? idx[b] : b;
    out[(long)b * ld_out + d] = f2bf(src[(long)r * ld_src + d]);
}
PDA_EXPORT void pda_gather_rows_bf16(const void* src, long ld_src, const void* idx, void* out, long ld_out, int B,
                                     int D, hipStream_t stream) {
    long total = (long)B * D;
    hipLaunchKernelGGL(gather_rows_bf16_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream,
                       (const float*)src, ld_src, (const int*)idx, (bf16_t*)out, ld_out, B, D);
}
