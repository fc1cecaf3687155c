// RL-specific fused kernels: DQN head (fc2 + n-step TD target + loss + dQ +
// priorities), critic TD for DDPG, fused Adam / Polyak over flat buffers with
// bf16 shadow-weight emission, epsilon-greedy actor epilogue.
#include "api.h"

// ----------------------------------------------------------------------------
// DQN head.  One wave64 per sample; the 512-wide hidden row is 8 values per lane.
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
};

template <int HID>
__global__ __launch_bounds__(256) void dqn_head_kernel(DqnHeadArgs p) {
    constexpr int PER = HID / 64;       // values per lane (8 for 512)
    const int wpb = blockDim.x >> 6;    // waves per workgroup: 1 for small batches (one sample per CU), else 4
    int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    int b = blockIdx.x * wpb + wave;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        if (p.rng_counter) *p.rng_counter += 1;     // the sampler of the NEXT step draws a fresh stream
        if (p.opt_step) *p.opt_step += 1;           // Adam of THIS step sees t = old + 1
    }
    if (b >= p.B) return;
    long row = (long)b * HID + lane * PER;

    // The replay row (index -> tail record -> action -> that action's W5 row) is a chain of three dependent
    // global loads.  It is started here, ahead of everything else and ahead of the zeroing stores below
    // (which the compiler will not move loads across), so that it runs under the hidden-row loads and the
    // dot products instead of after them.
    const int ridx = p.idx ? p.idx[b] : b;
    const float* tl = p.tails + (long)ridx * p.tail_ld;
    const float t_act = tl[p.off_action], t_rew = tl[p.off_reward], t_gam = tl[p.off_gamma], t_term = tl[p.off_term];
    const int act = min(max((int)t_act, 0), p.A - 1);
    const float w = p.isw ? p.isw[b] : 1.f;
    float wa[PER];
    {
        const float* w_act = p.w5_on + (long)act * HID + lane * PER;
#pragma unroll
        for (int j = 0; j < PER; ++j) wa[j] = w_act[j];
    }

    float h0[PER], ht[PER], hd[PER];
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        h0[j] = fmaxf(p.h1pre_on0[row + j] + p.b4_on[lane * PER + j], 0.f);
        ht[j] = fmaxf(p.h1pre_tg1[row + j] + p.b4_tg[lane * PER + j], 0.f);
        hd[j] = p.h1pre_on1 ? fmaxf(p.h1pre_on1[row + j] + p.b4_on[lane * PER + j], 0.f) : 0.f;
        p.h1pre_on0[row + j] = 0.f;
        p.h1pre_tg1[row + j] = 0.f;
        if (p.h1pre_on1) p.h1pre_on1[row + j] = 0.f;
    }

    // lane a keeps q(.)[a]: no runtime-indexed arrays (they would live in scratch).  Actions go in
    // groups of 8 with the 24 dot products formed first and reduced together, so the shuffles of
    // independent reductions overlap (one wave per sample = nothing else hides their latency).
    float myq0 = 0.f, myqt = 0.f, best_t = -INFINITY, best_d = -INFINITY;
    int arg_d = 0;
    for (int a0 = 0; a0 < p.A; a0 += 8) {
        float s0[8], st[8], sd[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            s0[u] = 0.f; st[u] = 0.f; sd[u] = 0.f;
            if (a0 + u < p.A) {
                const float* w_on = p.w5_on + (long)(a0 + u) * HID + lane * PER;
                const float* w_tg = p.w5_tg + (long)(a0 + u) * HID + lane * PER;
#pragma unroll
                for (int j = 0; j < PER; ++j) {
                    s0[u] += h0[j] * w_on[j];
                    st[u] += ht[j] * w_tg[j];
                    sd[u] += hd[j] * w_on[j];
                }
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                s0[u] += __shfl_xor(s0[u], o, 64);
                st[u] += __shfl_xor(st[u], o, 64);
                sd[u] += __shfl_xor(sd[u], o, 64);
            }
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            int a = a0 + u;
            if (a < p.A) {
                float q0 = s0[u] + p.b5_on[a], qt = st[u] + p.b5_tg[a], qd = sd[u] + p.b5_on[a];
                if (lane == a) { myq0 = q0; myqt = qt; }
                best_t = fmaxf(best_t, qt);
                if (qd > best_d) { best_d = qd; arg_d = a; }
            }
        }
    }
    float next_q = p.h1pre_on1 ? __shfl(myqt, arg_d, 64) : best_t;

    float y = t_rew + t_gam * next_q * (1.f - t_term);
    float td = __shfl(myq0, act, 64) - y;
    float dqa = 2.f * w * td / (float)p.B;

    uint32_t hw[PER / 2], dw[PER / 2];
#pragma unroll
    for (int j = 0; j < PER; j += 2) {
        hw[j / 2] = pack2bf(h0[j], h0[j + 1]);
        dw[j / 2] = pack2bf(h0[j] > 0.f ? dqa * wa[j] : 0.f, h0[j + 1] > 0.f ? dqa * wa[j + 1] : 0.f);
    }
    uint32_t* ho = reinterpret_cast<uint32_t*>(p.h1_on0 + row);
    uint32_t* dout = reinterpret_cast<uint32_t*>(p.dh1 + row);
#pragma unroll
    for (int j = 0; j < PER / 2; ++j) { ho[j] = hw[j]; dout[j] = dw[j]; }

    if (lane < p.dq_ld) p.dq[(long)b * p.dq_ld + lane] = f2bf(lane == act ? dqa : 0.f);
    if (lane < p.A && p.q_out) p.q_out[(long)b * p.A + lane] = myq0;
    if (lane == 0) {
        if (p.td_out) p.td_out[b] = td;
        // per-sample loss accumulator (summed on the host side when the logger asks): 128 fp32
        // atomics on one address serialise memory-side and were half of this kernel's time
        p.loss_sum[b] += w * td * td / (float)p.B;
        if (p.leaf_sum) {
            float leaf = __powf(fabsf(td) + p.prio_eps, p.prio_alpha);
            p.leaf_sum[ridx] = leaf;
            p.leaf_min[ridx] = leaf;
        }
    }
}

PDA_EXPORT void pda_dqn_head(const DqnHeadArgs* args, hipStream_t stream) {
    DqnHeadArgs p = *args;
    const int wpb = p.B <= 1024 ? 1 : 4;
    dim3 grid((p.B + wpb - 1) / wpb), block(64 * wpb);
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

// ----------------------------------------------------------------------------
// flat-buffer optimizer.  One launch updates every parameter, clips the gradient
// by value and emits the bf16 "shadow" weights in the layout the MFMA kernels
// read.  Weights that have a shadow keep their gradient and Adam moments in that
// same kernel layout ([row][p*C + c]: what the wgrad GEMMs produce with coalesced
// stores); only the fp32 master stays in PyTorch's layout ([row][c*P + p]).  Threads
// walk kernel order, so g/m/v/shadow accesses are coalesced and the strided master
// accesses stay inside one row (<= 12.5 KB), which the vector L1 absorbs.
// Segment table entry (8 ints): start, count, cols, C, P, shadow_off, shadow_ld, flags(1 = has shadow)
// hyper (device floats): lr, beta1, beta2, eps, weight_decay, clip, grad_scale
// step: number of updates including this one (bumped earlier in the step, see dqn_head_kernel / pda_tick)
// ----------------------------------------------------------------------------
struct SegEntry { int start, count, cols, C, P, shadow_off, shadow_ld, flags; };

// kernel-order element j of segment s -> index of its fp32 master
__device__ __forceinline__ long master_index(const SegEntry& s, int j, long* shadow_idx) {
    if (!(s.flags & 1)) { *shadow_idx = -1; return (long)s.start + j; }
    int row = j / s.cols, q = j - row * s.cols;
    int pp = q / s.C, c = q - pp * s.C;
    *shadow_idx = (long)s.shadow_off + (long)row * s.shadow_ld + q;
    return (long)s.start + (long)row * s.cols + c * s.P + pp;
}

// Optional work folded into the optimizer launch (each saves a ~4-8 us kernel of the 20-kernel step):
//   * split-K gradient slabs are summed here instead of by wgrad_reduce_kernel (single-rank steps;
//     a data-parallel step needs the reduced gradient before the all-reduce and keeps the reduce launch)
//   * the uniform replay indices of the NEXT step are drawn by extra blocks of this launch: every
//     consumer of the current indices has finished by now, and the rng counter was bumped by the head
struct AdamExtras {
    const float* slab; const int4* slabinfo;        // per segment (slab_off, nslices, slice_stride, -)
    int* next_idx; float* next_w; int next_B; const int* size; unsigned long long seed;
    const unsigned long long* counter;
};

#define OPT_CHUNK 2048      // elements per block (flat chunks)
#define OPT_ROW_LDS 4352    // floats of LDS for one weight row in the transposition path (row pitch padded to odd)

// A chunk is (segment, first element).  Segments whose master layout is a per-row transpose of the
// kernel layout (C > 1: conv2/conv3/fc1 weights) are chunked one ROW per block when the row fits in
// LDS; everything else in flat OPT_CHUNK pieces.  (ops/optim.py:SegmentTable builds the list.)
__device__ __forceinline__ bool row_mode(const SegEntry& s) {
    return (s.flags & 1) && s.C > 1 && (s.P | 1) * s.C <= OPT_ROW_LDS;
}
// flags bit 1: "fine" flat chunks of 64 elements (segments whose gradient is summed from many split-K slabs
// inside the optimizer): a lane per element, the slices dealt over the 4 waves of the workgroup
__device__ __forceinline__ int chunk_len(const SegEntry& s) {
    return row_mode(s) ? s.cols : ((s.flags & 2) ? 64 : OPT_CHUNK);
}

// gradient element e (kernel order within segment seg): the flat buffer, or the sum of the split-K slabs
__device__ __forceinline__ float grad_at(const AdamExtras& x, int seg, int e, float* g, long i) {
    if (x.slab == nullptr) return g[i];
    int4 si = x.slabinfo[seg];
    if (si.y <= 0) return g[i];
    const float* q = x.slab + si.x + e;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    int sl = 0;
    for (; sl + 4 <= si.y; sl += 4) {
        a0 += q[(long)sl * si.z];
        a1 += q[(long)(sl + 1) * si.z];
        a2 += q[(long)(sl + 2) * si.z];
        a3 += q[(long)(sl + 3) * si.z];
    }
    for (; sl < si.y; ++sl) a0 += q[(long)sl * si.z];
    float gi = (a0 + a1) + (a2 + a3);
    g[i] = gi;          // keep the flat gradient readable (tests, debugging)
    return gi;
}

__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v,
                                                   bf16_t* __restrict__ shadow, const SegEntry* __restrict__ segs,
                                                   const int2* __restrict__ chunks, int nchunks,
                                                   const float* __restrict__ hyper,
                                                   const int* __restrict__ step, AdamExtras x) {
    __shared__ float row[OPT_ROW_LDS];
    if ((int)blockIdx.x >= nchunks) {       // sampler blocks (only launched when x.next_idx is set)
        int i = ((int)blockIdx.x - nchunks) * 256 + threadIdx.x;
        if (i < x.next_B) {
            unsigned long long ctr = *x.counter;
            int n = max(*x.size, 1);
            uint32_t r = hash_u32(x.seed ^ (ctr * 0x9e3779b97f4a7c15ULL + (unsigned long long)i + 1));
            x.next_idx[i] = (int)(((unsigned long long)r * (unsigned long long)n) >> 32);
            if (x.next_w) x.next_w[i] = 1.f;
        }
        return;
    }
    int2 ch = chunks[blockIdx.x];           // (segment, first element within the segment)
    SegEntry s = segs[ch.x];
    float lr = hyper[0], b1 = hyper[1], b2 = hyper[2], eps = hyper[3], wd = hyper[4], clip = hyper[5],
          gscale = hyper[6];
    float t = (float)max(*step, 1);
    float bc1 = 1.f - __powf(b1, t), bc2 = 1.f - __powf(b2, t);
    float step_size = lr / bc1, inv_sqrt_bc2 = rsqrtf(bc2);
    if (row_mode(s)) {
        // One weight row.  The fp32 master keeps PyTorch's [c][p] order, gradients / moments / bf16 shadow
        // the kernels' [p][c] order.  Walking kernel order against the master directly costs one 64-byte
        // sector per 4-byte access (24 us for the 1.6 M fc1 weights); here the master row crosses LDS:
        // coalesced in, transposed access in LDS (the pitch P, padded to odd, spreads the banks),
        // coalesced out.
        int r = ch.y / s.cols;
        float* prow = p + (long)s.start + (long)r * s.cols;
        int P = s.P, C = s.C, pitch = P | 1;
        const long gbase0 = (long)s.start + (long)r * s.cols;
        const long sbase0 = (long)s.shadow_off + (long)r * s.shadow_ld;
        const bool summed = x.slab != nullptr && x.slabinfo[ch.x].y > 0;
        if (!summed && ((C | s.cols | (int)gbase0 | (int)sbase0) & 3) == 0) {
            // 16-byte path (fc1: 512 rows of 3136 = 95 % of the parameters).  Every load of a phase is issued
            // before the first use: the scalar loops below make ~13 dependent round trips per row, which kept
            // this launch at 3.4 TB/s.
            constexpr int IT = (OPT_ROW_LDS + 1023) / 1024;
            float4 a4[IT], m4[IT], v4[IT];
#pragma unroll
            for (int it = 0; it < IT; ++it) {
                int i = (threadIdx.x + it * 256) * 4;
                if (i < s.cols) a4[it] = *reinterpret_cast<const float4*>(prow + i);
            }
#pragma unroll
            for (int it = 0; it < IT; ++it) {
                int i = (threadIdx.x + it * 256) * 4;
                if (i < s.cols) {
                    const float e4[4] = {a4[it].x, a4[it].y, a4[it].z, a4[it].w};
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        int c = (i + k) / P, pp = (i + k) - c * P;
                        row[c * pitch + pp] = e4[k];
                    }
                }
            }
#pragma unroll
            for (int it = 0; it < IT; ++it) {       // gradient / moments: kernel order, coalesced
                int q = (threadIdx.x + it * 256) * 4;
                if (q < s.cols) {
                    a4[it] = *reinterpret_cast<const float4*>(g + gbase0 + q);
                    m4[it] = *reinterpret_cast<const float4*>(m + gbase0 + q);
                    v4[it] = *reinterpret_cast<const float4*>(v + gbase0 + q);
                }
            }
            __syncthreads();
#pragma unroll
            for (int it = 0; it < IT; ++it) {
                int q = (threadIdx.x + it * 256) * 4;
                if (q < s.cols) {
                    int pp = q / C, c = q - pp * C;         // C % 4 == 0: the four share pp
                    float gq[4] = {a4[it].x, a4[it].y, a4[it].z, a4[it].w};
                    float mq[4] = {m4[it].x, m4[it].y, m4[it].z, m4[it].w};
                    float vq[4] = {v4[it].x, v4[it].y, v4[it].z, v4[it].w};
                    float pq[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        float gi = fminf(fmaxf(gq[k] * gscale, -clip), clip);
                        float pi = row[(c + k) * pitch + pp];
                        gi += wd * pi;
                        mq[k] = b1 * mq[k] + (1.f - b1) * gi;
                        vq[k] = b2 * vq[k] + (1.f - b2) * gi * gi;
                        pi -= step_size * mq[k] / (sqrtf(vq[k]) * inv_sqrt_bc2 + eps);
                        row[(c + k) * pitch + pp] = pi;
                        pq[k] = pi;
                    }
                    *reinterpret_cast<float4*>(m + gbase0 + q) = make_float4(mq[0], mq[1], mq[2], mq[3]);
                    *reinterpret_cast<float4*>(v + gbase0 + q) = make_float4(vq[0], vq[1], vq[2], vq[3]);
                    *reinterpret_cast<uint2*>(shadow + sbase0 + q) = make_uint2(pack2bf(pq[0], pq[1]), pack2bf(pq[2], pq[3]));
                }
            }
            __syncthreads();
#pragma unroll
            for (int it = 0; it < IT; ++it) {
                int i = (threadIdx.x + it * 256) * 4;
                if (i < s.cols) {
                    float e4[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        int c = (i + k) / P, pp = (i + k) - c * P;
                        e4[k] = row[c * pitch + pp];
                    }
                    *reinterpret_cast<float4*>(prow + i) = make_float4(e4[0], e4[1], e4[2], e4[3]);
                }
            }
            return;
        }
        for (int i = threadIdx.x; i < s.cols; i += 256) {
            int c = i / P, pp = i - c * P;
            row[c * pitch + pp] = prow[i];
        }
        __syncthreads();
        long gbase = (long)s.start + (long)r * s.cols;
        long sbase = (long)s.shadow_off + (long)r * s.shadow_ld;
        for (int q = threadIdx.x; q < s.cols; q += 256) {   // q = pp * C + c (kernel order)
            int pp = q / C, c = q - pp * C;
            long i = gbase + q;
            float gi = grad_at(x, ch.x, r * s.cols + q, g, i) * gscale;
            gi = fminf(fmaxf(gi, -clip), clip);
            float pi = row[c * pitch + pp];
            gi += wd * pi;
            float mi = b1 * m[i] + (1.f - b1) * gi;
            float vi = b2 * v[i] + (1.f - b2) * gi * gi;
            pi -= step_size * mi / (sqrtf(vi) * inv_sqrt_bc2 + eps);
            row[c * pitch + pp] = pi; m[i] = mi; v[i] = vi;
            shadow[sbase + q] = f2bf(pi);
        }
        __syncthreads();
        for (int i = threadIdx.x; i < s.cols; i += 256) {
            int c = i / P, pp = i - c * P;
            prow[i] = row[c * pitch + pp];
        }
        return;
    }
    int end = min(ch.y + chunk_len(s), s.count);
    if ((s.flags & 2) && x.slab != nullptr && x.slabinfo[ch.x].y > 0) {
        // Slab-summed segment: 64 elements per workgroup, wave w adds slices w, w + 4, .. with 8 loads in
        // flight per lane, the four partial sums meet in LDS.  (One thread per element walking all the
        // slices -- 256 of them for conv1 at batch 128 -- was a chain of ~64 dependent memory round trips
        // on 32 workgroups: it set the duration of the whole launch.)
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        const int e = ch.y + lane;
        const bool ok = e < end;
        const int4 sinfo = x.slabinfo[ch.x];
        const float* q = x.slab + sinfo.x + (ok ? e : ch.y);
        float acc[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) acc[u] = 0.f;
        int sl = wave;
        for (; sl + 28 < sinfo.y; sl += 32) {
#pragma unroll
            for (int u = 0; u < 8; ++u) acc[u] += q[(long)(sl + 4 * u) * sinfo.z];
        }
        for (; sl < sinfo.y; sl += 4) acc[0] += q[(long)sl * sinfo.z];
        row[wave * 64 + lane] = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
        __syncthreads();
        if (wave == 0 && ok) {
            long si;
            long ip = master_index(s, e, &si);
            long i = (long)s.start + e;
            float gi = (row[lane] + row[64 + lane]) + (row[128 + lane] + row[192 + lane]);
            g[i] = gi;          // keep the flat gradient readable (tests, debugging)
            gi = fminf(fmaxf(gi * gscale, -clip), clip);
            float pi = p[ip];
            gi += wd * pi;
            float mi = b1 * m[i] + (1.f - b1) * gi;
            float vi = b2 * v[i] + (1.f - b2) * gi * gi;
            pi -= step_size * mi / (sqrtf(vi) * inv_sqrt_bc2 + eps);
            p[ip] = pi; m[i] = mi; v[i] = vi;
            if (si >= 0) shadow[si] = f2bf(pi);
        }
        return;
    }
    for (int e = ch.y + threadIdx.x; e < end; e += 256) {
        long si;
        long ip = master_index(s, e, &si);
        long i = (long)s.start + e;
        float gi = grad_at(x, ch.x, e, g, i) * gscale;
        gi = fminf(fmaxf(gi, -clip), clip);
        float pi = p[ip];
        gi += wd * pi;
        float mi = b1 * m[i] + (1.f - b1) * gi;
        float vi = b2 * v[i] + (1.f - b2) * gi * gi;
        pi -= step_size * mi / (sqrtf(vi) * inv_sqrt_bc2 + eps);
        p[ip] = pi; m[i] = mi; v[i] = vi;
        if (si >= 0) shadow[si] = f2bf(pi);
    }
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
    int extra_blocks = x.next_idx ? (x.next_B + 255) / 256 : 0;
    hipLaunchKernelGGL(adam_kernel, dim3(nchunks + extra_blocks), dim3(256), 0, stream, (float*)p, (float*)g,
                       (float*)m, (float*)v, (bf16_t*)shadow, (const SegEntry*)segs, (const int2*)chunks, nchunks,
                       (const float*)hyper, (const int*)step, x);
}

// tgt = (1 - tau) tgt + tau src (tau >= 1: hard copy) on fp32 masters, with the target's shadow weights
__global__ __launch_bounds__(256) void polyak_kernel(float* tgt, const float* src,
                                                     bf16_t* __restrict__ shadow, const SegEntry* __restrict__ segs,
                                                     const int2* __restrict__ chunks, float tau) {
    int2 ch = chunks[blockIdx.x];
    SegEntry s = segs[ch.x];
    int end = min(ch.y + chunk_len(s), s.count);
    for (int e = ch.y + threadIdx.x; e < end; e += 256) {
        long si;
        long i = master_index(s, e, &si);
        float t = tgt[i] + tau * (src[i] - tgt[i]);
        if (tau >= 1.f) t = src[i];
        tgt[i] = t;
        if (shadow && si >= 0) shadow[si] = f2bf(t);
    }
}

PDA_EXPORT void pda_polyak(void* tgt, const void* src, void* shadow, const void* segs, const void* chunks,
                           int nchunks, float tau, hipStream_t stream) {
    hipLaunchKernelGGL(polyak_kernel, dim3(nchunks), dim3(256), 0, stream, (float*)tgt, (const float*)src,
                       (bf16_t*)shadow, (const SegEntry*)segs, (const int2*)chunks, tau);
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
