// Device side of the radix-64 sum / min tree (ops/segtree.py): the stratified descent of ONE wave, shared by the
// sampling kernel (replay_kernels.hip) and the sampler blocks of the optimizer launch (optim.h).
#pragma once
#include "common.h"

#define RADIX 64
struct TreeOffsets { int off[8]; };

// Sample `wave` of B (stratum [wave, wave + 1) / B of the total mass, position u in it) -> out_idx / out_w[wave].
// ntop > 0: the top level (ntop <= 64 nodes) is NOT read but re-derived from the level below -- ntop coalesced rows of
// 64, eight in flight at a time -- and written back by wave 0 (trees whose repair stops below the top level:
// DeviceSumMinTree.lazy_top).
// COH: inner nodes (every level above the leaves) are read with agent-scope atomic loads -- for the optimizer launch's
// sampler blocks, which read nodes that repair blocks of the SAME launch have just written with agent-scope atomics
// (no fences anywhere: an agent-scope fence writes the whole L2 back, which is full of the optimizer's dirty lines).
template <bool COH>
__device__ __forceinline__ float seg_ld(const float* p) {
    if constexpr (COH) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else return *p;
}
template <bool COH = false>
__device__ __forceinline__ void segtree_sample_wave(const float* sum, const float* mn, const TreeOffsets& offs, int depth,
                                                    int num_leaves, float u, int wave, int lane, int B, float beta,
                                                    int* out_idx, float* out_w, int ntop, float* sum_w, float* mn_w) {
    int top = offs.off[depth - 1];
    float kid, kmin;
    if (ntop > 0 && depth >= 2) {
        // The top level (ntop <= 64 nodes) is NOT read: every wave re-derives it from the level below -- ntop
        // coalesced rows of 64, eight in flight at a time -- so a priority update has to repair the tree only up to
        // that level (the engines fold the first repair level into their optimizer launch: a prioritized step then
        // has no repair launches of its own on trees of <= 262 k leaves, one on larger ones).  Wave 0 writes the
        // level back for host-side readers.
        const long below = offs.off[depth - 2];
        kid = 0.f; kmin = INFINITY;
        for (int j0 = 0; j0 < ntop; j0 += 8) {
            float c[8], m[8];
#pragma unroll
            for (int t8 = 0; t8 < 8; ++t8) {
                const int j = min(j0 + t8, ntop - 1);
                c[t8] = sum[below + (long)j * RADIX + lane];
                m[t8] = mn[below + (long)j * RADIX + lane];
            }
#pragma unroll
            for (int t8 = 0; t8 < 8; ++t8) {
                const float s = wave_sum(c[t8]), q = wave_min(m[t8]);
                if (lane == j0 + t8 && j0 + t8 < ntop) { kid = s; kmin = q; }
            }
        }
        if (wave == 0 && sum_w) { sum_w[top + lane] = kid; mn_w[top + lane] = kmin; }
    } else {
        kid = depth >= 2 ? seg_ld<COH>(sum + top + lane) : sum[top + lane];
        kmin = depth >= 2 ? seg_ld<COH>(mn + top + lane) : mn[top + lane];
    }
    float total = wave_sum(kid);
    float pmin = wave_min(kmin);
    float v = ((float)wave + u) / (float)B * total;
    int node = 0;
    for (int lvl = depth - 1; lvl >= 0; --lvl) {
        if (lvl != depth - 1) {
            const float* q = sum + (long)offs.off[lvl] + (long)node * RADIX + lane;
            kid = lvl >= 1 ? seg_ld<COH>(q) : *q;
        }
        float incl = wave_scan_incl(kid, lane);
        bool hit = (incl >= v) && (kid > 0.f);
        unsigned long long mask = __ballot(hit);
        int pick;
        if (mask) {
            pick = __ffsll((long long)mask) - 1;
        } else {                                    // rounding pushed v past the end: last positive child
            unsigned long long pos = __ballot(kid > 0.f);
            pick = pos ? 63 - __clzll((long long)pos) : 0;
        }
        float excl = __shfl(incl - kid, pick, 64);
        v -= excl;
        node = node * RADIX + pick;
    }
    node = min(node, num_leaves - 1);
    if (lane == 0) {
        float p = sum[node];
        out_idx[wave] = node;
        // (min tree of an empty / partly filled tree holds +inf in its unused leaves: a tree with no positive
        // leaf at all gives weight 1 instead of inf^beta)
        const float ratio = (pmin < 3.0e38f && p > 0.f) ? pmin / p : 1.f;
        out_w[wave] = __powf(ratio, beta);
    }
}
