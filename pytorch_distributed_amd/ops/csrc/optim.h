// Flat-buffer optimizer: segment table, chunking rules and the Adam workgroup body (shared by
// rl_kernels.hip, which owns the plain optimizer launches, and gemm.hip, which co-launches optimizer
// workgroups with the conv weight-gradient GEMMs).
#pragma once
#include "common.h"
#include "segtree_dev.h"

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
// tr_ld > 0: the segment also keeps a TRANSPOSED copy of its shadow, element (row, q) at shadow[tr_off + q * tr_ld + row]
// (+ the lo plane): the conv data-gradient kernels read w[o][(tap, c)] as fragments with the output channel o as the
// reduction index, i.e. 8 consecutive o of one (tap, c) -- one 16-byte load from this copy instead of a transposition
// of the whole tensor through LDS at the start of every launch.  Written wherever the shadow is (every optimizer
// path and the Polyak / refresh kernel), so it cannot go stale.
struct SegEntry { int start, count, cols, C, P, shadow_off, shadow_ld, flags, tr_off, tr_ld; };

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
    long shadow_plane;      // fp32-precision engines: the shadow's lo plane (bf16(w - bf16(w))) lies this many
                            // elements after the hi plane; 0 = plain bf16 shadow
    // MLP engines (flat chunks only, C == 1 segments):
    const int2* tinfo;      // per segment (offset, pitch) of a TRANSPOSED bf16 copy [col][row] in `shadow` that the fused
                            // backward reads as its B operand (mlp_fused.hip); pitch 0 = none
    float* tgt; bf16_t* tgt_shadow; float tau;      // target network updated in the same pass: tgt += tau (p - tgt),
                                                    // with its shadow (same segment table); tgt == nullptr = off
    // prioritized replay: the first repair level of the sum / min tree (replay_kernels.hip) rides in this launch --
    // the leaves of the rep_n sampled rows were written by the loss kernel earlier in the step; one wave per row
    // recomputes that leaf's parent from its 64 children (idempotent on duplicates).  rep_sum == nullptr = off
    float* rep_sum; float* rep_min; const int* rep_idx; int rep_n; int rep_parent_off;
    // trees of three levels: the last repair block to finish also recomputes the rep_top_n nodes of the top level
    // (at rep_top_off) from the repaired level, so such a tree is complete when this launch ends.  rep_count: a
    // device int, zero between launches.  rep_top_n == 0 = off
    // (the level offsets and the depth come from ps_offs / ps_depth below); rep_bar: barrier counter of the repair
    // blocks (trees of four levels), zero between launches
    int* rep_count; int rep_top_off; int rep_top_n; int* rep_bar;
    // ... and the NEXT step's prioritized draw rides behind it (trees of up to three levels, complete once the last
    // repair block has finished): ps_B sampler waves, four per block, wait for that block's flag *ps_done (bounded
    // spin), then descend the tree rep_sum / rep_min exactly as segtree_sample_kernel does (uniforms from seed and
    // *counter).  They overwrite rep_idx's buffer -- every repair block has read it by then.  The last sampler block
    // to finish clears the flag for the next launch.  ps_idx == nullptr = off
    int* ps_idx; float* ps_w; int ps_B; int ps_depth; int ps_leaves; float ps_beta; TreeOffsets ps_offs;
    int* ps_done; int* ps_count;
};
__host__ __device__ inline int adam_sampler_blocks(const AdamExtras& x) { return x.next_idx ? (x.next_B + 255) / 256 : 0; }
__host__ __device__ inline int adam_repair_blocks(const AdamExtras& x) { return x.rep_sum ? (x.rep_n + 3) / 4 : 0; }
__host__ __device__ inline int adam_per_sampler_blocks(const AdamExtras& x) { return x.ps_idx ? (x.ps_B + 3) / 4 : 0; }

// the extra copies of a freshly updated weight (see AdamExtras): e = element within segment `seg`, ip = master index
__device__ __forceinline__ void put_extras(const AdamExtras& x, bf16_t* shadow, const SegEntry& s, int seg, int e,
                                           long ip, long si, float w) {
    if (x.tinfo) {
        int2 ti = x.tinfo[seg];
        if (ti.y > 0) {
            int r = e / s.cols, q = e - r * s.cols;
            shadow[(long)ti.x + (long)q * ti.y + r] = f2bf(w);
        }
    }
    if (x.tgt) {
        float t = x.tgt[ip];
        t += x.tau * (w - t);
        x.tgt[ip] = t;
        if (si >= 0 && x.tgt_shadow) x.tgt_shadow[si] = f2bf(t);
    }
}

__device__ __forceinline__ void put_shadow(bf16_t* shadow, long si, float w, long plane) {
    bf16_t h = f2bf(w);
    shadow[si] = h;
    if (plane) shadow[si + plane] = f2bf(w - bf2f(h));
}
// the transposed copy of shadow element (row, q) of segment s (see SegEntry)
__device__ __forceinline__ void put_tr(bf16_t* shadow, const SegEntry& s, int row, int q, float w, long plane) {
    if (s.tr_ld > 0) put_shadow(shadow, (long)s.tr_off + (long)q * s.tr_ld + row, w, plane);
}
__device__ __forceinline__ void put_tr_elem(bf16_t* shadow, const SegEntry& s, int e, float w, long plane) {
    if (s.tr_ld > 0) {
        int row = e / s.cols;
        put_shadow(shadow, (long)s.tr_off + (long)(e - row * s.cols) * s.tr_ld + row, w, plane);
    }
}

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
// flags bit 2: "medium" flat chunks of 256 elements, one per thread: a single memory round trip per workgroup (small
// models, where the optimizer launch is latency- and not bandwidth-bound; 2048-element chunks are 8 dependent trips)
__device__ __forceinline__ int chunk_len(const SegEntry& s) {
    return row_mode(s) ? s.cols : ((s.flags & 2) ? 64 : (s.flags & 4) ? 256 : OPT_CHUNK);
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

// One workgroup (256 threads) of the optimizer launch: chunk `bid` of the table, or a sampler block past
// it.  `row` = OPT_ROW_LDS floats of LDS.
__device__ __forceinline__ void adam_block(float* __restrict__ p, float* __restrict__ g,
                                           float* __restrict__ m, float* __restrict__ v,
                                           bf16_t* __restrict__ shadow, const SegEntry* __restrict__ segs,
                                           const int2* __restrict__ chunks, int nchunks,
                                           const float* __restrict__ hyper,
                                           const int* __restrict__ step, const AdamExtras& x, int bid, float* row) {
    // Tree work of a prioritized step.  Blocks of this launch hand values to each other, so everything they exchange goes
    // through agent-scope atomics (performed at the coherence point) and NO fence is used: an agent-scope fence writes the
    // L2 back, which at this moment is full of the optimizer's own dirty lines (measured: the launch grew by the whole
    // sampling kernel it was meant to absorb).  A store that must be complete before a counter / flag update is an
    // atomic exchange whose returned value the thread waits for.
    if (bid >= nchunks + adam_sampler_blocks(x) + adam_repair_blocks(x)) {      // prioritized sampler blocks
        if (threadIdx.x == 0) {
            unsigned spins = 0;
            while (__hip_atomic_load(x.ps_done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0 && ++spins < (1u << 22))
                __builtin_amdgcn_s_sleep(2);
        }
        __syncthreads();
        const int wave = (bid - nchunks - adam_sampler_blocks(x) - adam_repair_blocks(x)) * 4 + (threadIdx.x >> 6);
        const int lane = threadIdx.x & 63;
        if (wave < x.ps_B) {
            const float u = u01(hash_u32(x.seed ^ (*x.counter * 0x9e3779b97f4a7c15ULL + (unsigned long long)wave + 1)));
            segtree_sample_wave<true>(x.rep_sum, x.rep_min, x.ps_offs, x.ps_depth, x.ps_leaves, u, wave, lane, x.ps_B,
                                      x.ps_beta, x.ps_idx, x.ps_w, 0, nullptr, nullptr);
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            const int prev = atomicAdd(x.ps_count, 1);
            if (prev + 1 == adam_per_sampler_blocks(x)) {       // the last sampler block: ready for the next launch
                __hip_atomic_store(x.ps_count, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(x.ps_done, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        return;
    }
    if (bid >= nchunks + adam_sampler_blocks(x)) {      // tree-repair blocks (only launched when x.rep_sum is set)
        const int wave = (bid - nchunks - adam_sampler_blocks(x)) * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
        if (!x.rep_count) {         // level 1 only (deeper trees: the caller repairs the rest, tree.lazy_top)
            if (wave < x.rep_n) {
                const int node = x.rep_idx[wave] >> 6;
                const long child = (long)node * 64 + lane;      // level 0 starts at offset 0
                const float s = wave_sum(x.rep_sum[child]);
                const float q = wave_min(x.rep_min[child]);
                if (lane == 0) {
                    x.rep_sum[x.rep_parent_off + node] = s;
                    x.rep_min[x.rep_parent_off + node] = q;
                }
            }
            return;
        }
        // The whole tree (2 .. 4 levels).  Levels 1 .. L are repaired per sampled row, a wave per row and level, with a
        // barrier of the repair blocks between two levels (they are few and all resident); the last block to finish
        // recomputes every node of the top level and releases the sampler blocks.
        int* flag = reinterpret_cast<int*>(row);
        const int L = x.ps_depth <= 2 ? 1 : x.ps_depth - 2;
        for (int lvl = 1; lvl <= L; ++lvl) {
            if (wave < x.rep_n) {
                const int node = x.rep_idx[wave] >> (6 * lvl);
                const long child = (long)x.ps_offs.off[lvl - 1] + (long)node * 64 + lane;
                float cs, cm;
                if (lvl == 1) { cs = x.rep_sum[child]; cm = x.rep_min[child]; }       // leaves: an earlier launch's
                else {
                    cs = __hip_atomic_load(x.rep_sum + child, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    cm = __hip_atomic_load(x.rep_min + child, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                const float s = wave_sum(cs), q = wave_min(cm);
                if (lane == 0) {
                    const float r0 = atomicExch(x.rep_sum + x.ps_offs.off[lvl] + node, s);
                    const float r1 = atomicExch(x.rep_min + x.ps_offs.off[lvl] + node, q);
                    asm volatile("" :: "v"(r0), "v"(r1));       // both exchanges have returned: the node is in place
                }
            }
            if (lvl < L) {          // barrier of the repair blocks (x.rep_bar: reset by the last block of the launch)
                __syncthreads();
                if (threadIdx.x == 0) {
                    atomicAdd(x.rep_bar, 1);
                    unsigned spins = 0;
                    while (__hip_atomic_load(x.rep_bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) <
                               (lvl * adam_repair_blocks(x)) && ++spins < (1u << 22))
                        __builtin_amdgcn_s_sleep(2);
                }
                __syncthreads();
            }
        }
        __syncthreads();                                        // this block's nodes are stored
        if (threadIdx.x == 0) {
            const int prev = atomicAdd(x.rep_count, 1);
            *flag = prev + 1 == adam_repair_blocks(x) ? 1 : 0;
            if (*flag) {
                __hip_atomic_store(x.rep_count, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(x.rep_bar, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        __syncthreads();
        if (*flag) {                                            // every repair block has published its nodes
            // top level: wave w takes nodes w, w + 4, ...; four rows in flight per wave (13 nodes for 50 k leaves: one
            // memory round trip -- one row at a time put four of them on the step's critical path)
            const long below = (long)x.ps_offs.off[x.ps_depth - 2];
            for (int j0 = threadIdx.x >> 6; j0 < x.rep_top_n; j0 += 16) {
                float cs[4], cm[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int j = min(j0 + 4 * t, x.rep_top_n - 1);
                    cs[t] = __hip_atomic_load(x.rep_sum + below + (long)j * 64 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    cm[t] = __hip_atomic_load(x.rep_min + below + (long)j * 64 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                float r = 0.f;
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const float s = wave_sum(cs[t]), q = wave_min(cm[t]);
                    const int j = j0 + 4 * t;
                    if (lane == 0 && j < x.rep_top_n) {
                        r += atomicExch(x.rep_sum + x.rep_top_off + j, s);
                        r += atomicExch(x.rep_min + x.rep_top_off + j, q);
                    }
                }
                asm volatile("" :: "v"(r));                     // every exchange has returned
            }
            if (x.ps_done) {                                    // the tree is complete: release the sampler blocks
                __syncthreads();
                if (threadIdx.x == 0) __hip_atomic_store(x.ps_done, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        return;
    }
    if (bid >= nchunks) {       // sampler blocks (only launched when x.next_idx is set)
        int i = (bid - nchunks) * 256 + threadIdx.x;
        if (i < x.next_B) {
            unsigned long long ctr = *x.counter;
            uint32_t r = hash_u32(x.seed ^ (ctr * 0x9e3779b97f4a7c15ULL + (unsigned long long)i + 1));
            x.next_idx[i] = window_pick(x.size, r);
            if (x.next_w) x.next_w[i] = 1.f;
        }
        return;
    }
    int2 ch = chunks[bid];           // (segment, first element within the segment)
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
            float4 a4[IT], g4[IT], m4[IT], v4[IT];
#pragma unroll
            for (int it = 0; it < IT; ++it) {
                int i = (threadIdx.x + it * 256) * 4;
                if (i < s.cols) a4[it] = *reinterpret_cast<const float4*>(prow + i);
            }
#pragma unroll
            for (int it = 0; it < IT; ++it) {       // gradient / moments: kernel order, coalesced; in flight together
                int q = (threadIdx.x + it * 256) * 4;   // with the master row (one memory round trip, not two)
                if (q < s.cols) {
                    g4[it] = *reinterpret_cast<const float4*>(g + gbase0 + q);
                    m4[it] = *reinterpret_cast<const float4*>(m + gbase0 + q);
                    v4[it] = *reinterpret_cast<const float4*>(v + gbase0 + q);
                }
            }
            // P odd (fc1: 49, conv3: 9): pitch == P, the LDS image of the row IS the master row, so it goes in and
            // out as 16-byte vectors.  (Scalar stores of the four components are stride-4 across lanes: 16 banks
            // for 64 lanes, the 64 % bank-conflict rate of this kernel in profiles/r2_b128_bf16_counters.txt.)
            const bool flat_row = pitch == P;
#pragma unroll
            for (int it = 0; it < IT; ++it) {
                int i = (threadIdx.x + it * 256) * 4;
                if (i < s.cols) {
                    if (flat_row) {
                        *reinterpret_cast<float4*>(row + i) = a4[it];
                    } else {
                        const float e4[4] = {a4[it].x, a4[it].y, a4[it].z, a4[it].w};
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            int c = (i + k) / P, pp = (i + k) - c * P;
                            row[c * pitch + pp] = e4[k];
                        }
                    }
                }
            }
            __syncthreads();
#pragma unroll
            for (int it = 0; it < IT; ++it) {
                int q = (threadIdx.x + it * 256) * 4;
                if (q < s.cols) {
                    int pp = q / C, c = q - pp * C;         // C % 4 == 0: the four share pp
                    float gq[4] = {g4[it].x, g4[it].y, g4[it].z, g4[it].w};
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
                    const uint2 hi2 = make_uint2(pack2bf(pq[0], pq[1]), pack2bf(pq[2], pq[3]));
                    *reinterpret_cast<uint2*>(shadow + sbase0 + q) = hi2;
                    if (x.shadow_plane) {
                        const float r0 = pq[0] - bf2f((bf16_t)(hi2.x & 0xffffu)), r1 = pq[1] - bf2f((bf16_t)(hi2.x >> 16));
                        const float r2 = pq[2] - bf2f((bf16_t)(hi2.y & 0xffffu)), r3 = pq[3] - bf2f((bf16_t)(hi2.y >> 16));
                        *reinterpret_cast<uint2*>(shadow + x.shadow_plane + sbase0 + q) =
                            make_uint2(pack2bf(r0, r1), pack2bf(r2, r3));
                    }
                    if (s.tr_ld > 0) {
#pragma unroll
                        for (int k = 0; k < 4; ++k) put_tr(shadow, s, r, q + k, pq[k], x.shadow_plane);
                    }
                }
            }
            __syncthreads();
#pragma unroll
            for (int it = 0; it < IT; ++it) {
                int i = (threadIdx.x + it * 256) * 4;
                if (i < s.cols) {
                    if (flat_row) {
                        *reinterpret_cast<float4*>(prow + i) = *reinterpret_cast<const float4*>(row + i);
                    } else {
                        float e4[4];
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            int c = (i + k) / P, pp = (i + k) - c * P;
                            e4[k] = row[c * pitch + pp];
                        }
                        *reinterpret_cast<float4*>(prow + i) = make_float4(e4[0], e4[1], e4[2], e4[3]);
                    }
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
            put_shadow(shadow, sbase + q, pi, x.shadow_plane);
            put_tr(shadow, s, r, q, pi, x.shadow_plane);
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
            if (si >= 0) { put_shadow(shadow, si, pi, x.shadow_plane); put_tr_elem(shadow, s, e, pi, x.shadow_plane); }
            put_extras(x, shadow, s, ch.x, e, ip, si, pi);
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
        if (si >= 0) { put_shadow(shadow, si, pi, x.shadow_plane); put_tr_elem(shadow, s, e, pi, x.shadow_plane); }
        put_extras(x, shadow, s, ch.x, e, ip, si, pi);
    }
}
