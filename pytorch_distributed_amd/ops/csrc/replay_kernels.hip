// Prioritized-replay segment tree (radix 64 = one child per wavefront lane) and
// HBM replay data movement.
#include "common.h"
#include "segtree_dev.h"
#include "../../parallel/csrc/frame_runs.h"

// ---- leaves: tree[idx[i]] = value[i] for both trees ---------------------------------
__global__ void segtree_set_leaves_kernel(float* sum, float* mn, const int* idx, const float* val, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        int leaf = idx[i];
        float v = val[i];
        sum[leaf] = v;
        mn[leaf] = v;
    }
}
PDA_EXPORT void pda_segtree_set_leaves(void* sum, void* mn, const void* idx, const void* val, int n,
                                       hipStream_t stream) {
    hipLaunchKernelGGL(segtree_set_leaves_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, (float*)sum,
                       (float*)mn, (const int*)idx, (const float*)val, n);
}

// ---- repair one level: for each touched leaf, one wave recomputes its level-`lvl`
// ancestor from that ancestor's 64 children (idempotent on duplicates, no atomics) ----
__global__ __launch_bounds__(256) void segtree_repair_kernel(float* sum, float* mn, const int* idx, int n,
                                                             int child_off, int parent_off, int shift) {
    int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    if (wave >= n) return;
    int node = idx[wave] >> shift;              // ancestor index within its level (shift = 6 * lvl)
    long child = (long)child_off + (long)node * RADIX + lane;
    float s = wave_sum(sum[child]);
    float m = wave_min(mn[child]);
    if (lane == 0) {
        sum[parent_off + node] = s;
        mn[parent_off + node] = m;
    }
}
PDA_EXPORT void pda_segtree_repair_level(void* sum, void* mn, const void* idx, int n, int child_off,
                                         int parent_off, int lvl, hipStream_t stream) {
    long threads = (long)n * 64;
    hipLaunchKernelGGL(segtree_repair_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, stream,
                       (float*)sum, (float*)mn, (const int*)idx, n, child_off, parent_off, 6 * lvl);
}

// ---- stratified prefix-sum sampling: one wave per sample (segtree_dev.h) ---------------------------------
__global__ __launch_bounds__(256) void segtree_sample_kernel(const float* __restrict__ sum,
                                                             const float* __restrict__ mn, TreeOffsets offs,
                                                             int depth, int num_leaves,
                                                             const float* __restrict__ uniforms, int B, float beta,
                                                             int* __restrict__ out_idx, float* __restrict__ out_w,
                                                             unsigned long long seed,
                                                             const unsigned long long* __restrict__ counter,
                                                             int ntop, float* __restrict__ sum_w,
                                                             float* __restrict__ mn_w) {
    int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    if (wave >= B) return;
    // uniforms == nullptr: the stratum's uniform is drawn here (the same counter hash as uniform_f32_kernel: one launch
    // less per prioritized step)
    const float u = uniforms ? uniforms[wave]
                             : u01(hash_u32(seed ^ (*counter * 0x9e3779b97f4a7c15ULL + (unsigned long long)wave + 1)));
    segtree_sample_wave(sum, mn, offs, depth, num_leaves, u, wave, lane, B, beta, out_idx, out_w, ntop, sum_w, mn_w);
}
// ntop: nodes of the top level (re-derived from the level below by every wave and written back); 0 = read it as stored
PDA_EXPORT void pda_segtree_sample(void* sum, void* mn, const int* offs, int depth, int num_leaves,
                                   const void* uniforms, int B, float beta, void* out_idx, void* out_w,
                                   int ntop, hipStream_t stream) {
    TreeOffsets o;
    for (int i = 0; i < 8; ++i) o.off[i] = offs[i];
    long threads = (long)B * 64;
    hipLaunchKernelGGL(segtree_sample_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, stream,
                       (const float*)sum, (const float*)mn, o, depth, num_leaves, (const float*)uniforms, B, beta,
                       (int*)out_idx, (float*)out_w, 0ull, (const unsigned long long*)nullptr, ntop, (float*)sum,
                       (float*)mn);
}
// the same draw with the uniforms generated in the kernel from (seed, *counter): see uniform_f32_kernel
PDA_EXPORT void pda_segtree_sample_rng(void* sum, void* mn, const int* offs, int depth, int num_leaves,
                                       int B, float beta, void* out_idx, void* out_w, unsigned long long seed,
                                       const void* counter, int ntop, hipStream_t stream) {
    TreeOffsets o;
    for (int i = 0; i < 8; ++i) o.off[i] = offs[i];
    long threads = (long)B * 64;
    hipLaunchKernelGGL(segtree_sample_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, stream,
                       (const float*)sum, (const float*)mn, o, depth, num_leaves, (const float*)nullptr, B, beta,
                       (int*)out_idx, (float*)out_w, seed, (const unsigned long long*)counter, ntop, (float*)sum,
                       (float*)mn);
}

// uniform index draw on device (counter-based, graph-replayable: the counter lives in memory).
// size: int32[4] = {rows to draw from, first row, ring length (0: no wrap), -} (common.h window_pick)
__global__ void uniform_idx_kernel(int* out_idx, float* out_w, int B, const int* size, unsigned long long seed,
                                   unsigned long long* counter) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long ctr = *counter;
    if (i < B) {
        uint32_t r = hash_u32(seed ^ (ctr * 0x9e3779b97f4a7c15ULL + (unsigned long long)i + 1));
        out_idx[i] = window_pick(size, r);
        if (out_w) out_w[i] = 1.f;
    }
}
// (the counter is bumped by a later kernel of the step -- dqn_head_kernel / td_vector_kernel /
// pda_tick -- never by a kernel that reads it)
PDA_EXPORT void pda_uniform_idx(void* out_idx, void* out_w, int B, const void* size, unsigned long long seed,
                                void* counter, hipStream_t stream) {
    hipLaunchKernelGGL(uniform_idx_kernel, dim3((B + 255) / 256), dim3(256), 0, stream, (int*)out_idx,
                       (float*)out_w, B, (const int*)size, seed, (unsigned long long*)counter);
}
// uniforms[i] in [0,1) for the stratified sampler, same counter scheme
__global__ void uniform_f32_kernel(float* out, int B, unsigned long long seed, unsigned long long* counter) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long ctr = *counter;
    if (i < B) out[i] = u01(hash_u32(seed ^ (ctr * 0x9e3779b97f4a7c15ULL + (unsigned long long)i + 1)));
}
PDA_EXPORT void pda_uniform_f32(void* out, int B, unsigned long long seed, void* counter, hipStream_t stream) {
    hipLaunchKernelGGL(uniform_f32_kernel, dim3((B + 255) / 256), dim3(256), 0, stream, (float*)out, B, seed,
                       (unsigned long long*)counter);
}

// ---- replay: unpack packed transition records into the ring buffers -------------------------
// One workgroup per record: 16-byte vector copies of both states, then the float tail.
// pos == nullptr: record r goes to ring slot (pos_start + r) mod ring_size (uniform replay: no index tensor, no
// launches to build one); otherwise to pos[r] (prioritized replay keeps the tensor for the tree update).
__global__ __launch_bounds__(256) void replay_unpack_kernel(const uint8_t* __restrict__ rec, long rec_stride,
                                                            const int* __restrict__ pos, long pos_start, long ring_size,
                                                            uint8_t* __restrict__ s0, uint8_t* __restrict__ s1,
                                                            int state_bytes, int state_stride,
                                                            float* __restrict__ tails, int tail_ld, int tail_floats,
                                                            float* __restrict__ leaf_sum, float* __restrict__ leaf_min,
                                                            int prio_at, float prio_eps, float prio_alpha) {
    int r = blockIdx.x;
    const uint8_t* src = rec + (long)r * rec_stride;
    const long slot = pos ? (long)pos[r] : (pos_start + r) % ring_size;
    long dst_row = slot * state_bytes;
    int vec = state_bytes / 16;
    const uint4* a = reinterpret_cast<const uint4*>(src);
    const uint4* b = reinterpret_cast<const uint4*>(src + state_stride);
    uint4* d0 = reinterpret_cast<uint4*>(s0 + dst_row);
    uint4* d1 = reinterpret_cast<uint4*>(s1 + dst_row);
    if (state_bytes % 16 == 0 && dst_row % 16 == 0) {
        for (int i = threadIdx.x; i < vec; i += blockDim.x) { d0[i] = a[i]; d1[i] = b[i]; }
    } else {
        for (int i = threadIdx.x; i < state_bytes; i += blockDim.x) {
            s0[dst_row + i] = src[i];
            s1[dst_row + i] = src[state_stride + i];
        }
    }
    const float* tl = reinterpret_cast<const float*>(src + 2 * (long)state_stride);
    if (threadIdx.x < tail_floats) tails[slot * tail_ld + threadIdx.x] = tl[threadIdx.x];
    if (threadIdx.x == 128 && leaf_sum) {           // prioritized replay: the row's leaf (|p| + eps)^alpha in both trees
        const float v = powf(fabsf(tl[prio_at]) + prio_eps, prio_alpha);
        leaf_sum[slot] = v;
        leaf_min[slot] = v;
    }
}
PDA_EXPORT void pda_replay_unpack(const void* rec, long rec_stride, int n, const void* pos, long pos_start,
                                  long ring_size, void* s0, void* s1, int state_bytes, int state_stride, void* tails,
                                  int tail_ld, int tail_floats, void* leaf_sum, void* leaf_min, int prio_at,
                                  float prio_eps, float prio_alpha, hipStream_t stream) {
    hipLaunchKernelGGL(replay_unpack_kernel, dim3(n), dim3(256), 0, stream, (const uint8_t*)rec, rec_stride,
                       (const int*)pos, pos_start, ring_size > 0 ? ring_size : 1, (uint8_t*)s0, (uint8_t*)s1,
                       state_bytes, state_stride, (float*)tails, tail_ld, tail_floats, (float*)leaf_sum,
                       (float*)leaf_min, prio_at, prio_eps, prio_alpha);
}

// ---- replay: gather B rows of row_bytes (multiple of 16) ------------------------------------------
__global__ __launch_bounds__(256) void replay_gather_kernel(const uint8_t* __restrict__ src,
                                                            const int* __restrict__ idx, int row_bytes,
                                                            uint8_t* __restrict__ out) {
    int r = blockIdx.x;
    const uint4* a = reinterpret_cast<const uint4*>(src + (long)idx[r] * row_bytes);
    uint4* d = reinterpret_cast<uint4*>(out + (long)r * row_bytes);
    for (int i = threadIdx.x; i < row_bytes / 16; i += blockDim.x) d[i] = a[i];
}
PDA_EXPORT void pda_replay_gather(const void* src, const void* idx, int B, int row_bytes, void* out,
                                  hipStream_t stream) {
    hipLaunchKernelGGL(replay_gather_kernel, dim3(B), dim3(256), 0, stream, (const uint8_t*)src, (const int*)idx,
                       row_bytes, (uint8_t*)out);
}

// ---- frame-deduplicated replay: unpack run records (parallel/csrc/frame_runs.h) ----------------------------
// Workgroup (j, r): plane j and transition j of run r.  The planes of a run land contiguously at plane_base[r] of the
// plane ring; a transition's two window offsets become plane indices of that ring.  Workgroup (0, 0) also publishes the
// sampling window {rows, first row, ring length} that is valid once this chunk is in (the host computed it; written
// here so that it changes in stream order with the data, without launches of its own).
#define PDA_MAX_RUNS 16
struct RunPlacement { int plane_base[PDA_MAX_RUNS]; int trans_pos[PDA_MAX_RUNS]; };

__global__ __launch_bounds__(256) void replay_unpack_runs_kernel(const uint8_t* __restrict__ rec, long rec_stride,
                                                                 RunPlacement pl, int trans_ring,
                                                                 uint8_t* __restrict__ planes, int plane_stride,
                                                                 int planes_off, int trans_stride, int tail_floats,
                                                                 float* __restrict__ tails, int tail_ld,
                                                                 int* __restrict__ ref0, int* __restrict__ refN,
                                                                 int* __restrict__ window, int w_rows, int w_first,
                                                                 int w_ring, float* __restrict__ leaf_sum,
                                                                 float* __restrict__ leaf_min, int prio_at,
                                                                 float prio_eps, float prio_alpha) {
    const int j = blockIdx.x, r = blockIdx.y;
    const uint8_t* src = rec + (long)r * rec_stride;
    const PdaRunHeader* h = reinterpret_cast<const PdaRunHeader*>(src);
    const int np = h->num_planes, nt = h->num_transitions;
    const int base = pl.plane_base[r];
    if (j < np) {
        const uint4* a = reinterpret_cast<const uint4*>(src + planes_off + (long)j * plane_stride);
        uint4* d = reinterpret_cast<uint4*>(planes + (long)(base + j) * plane_stride);
        for (int i = threadIdx.x; i < plane_stride / 16; i += blockDim.x) d[i] = a[i];
    }
    if (j < nt) {
        const uint8_t* e = src + sizeof(PdaRunHeader) + (long)j * trans_stride;
        int slot = pl.trans_pos[r] + j;
        if (slot >= trans_ring) slot -= trans_ring;
        const float* tl = reinterpret_cast<const float*>(e);
        if ((int)threadIdx.x < tail_floats) tails[(long)slot * tail_ld + threadIdx.x] = tl[threadIdx.x];
        if (threadIdx.x == 64) {
            const int* offs = reinterpret_cast<const int*>(e + 4 * tail_floats);
            ref0[slot] = base + offs[0];
            refN[slot] = base + offs[1];
        }
        if (threadIdx.x == 128 && leaf_sum) {       // prioritized replay: the row's leaf (|p| + eps)^alpha in both trees
            const float v = powf(fabsf(tl[prio_at]) + prio_eps, prio_alpha);
            leaf_sum[slot] = v;
            leaf_min[slot] = v;
        }
    }
    if (j == 0 && r == 0 && threadIdx.x == 192 && window) {
        window[0] = w_rows; window[1] = w_first; window[2] = w_ring;
    }
}
PDA_EXPORT int pda_replay_unpack_runs(const void* rec, long rec_stride, int n_runs, const int* plane_base,
                                      const int* trans_pos, int trans_ring, void* planes, int plane_stride,
                                      int plane_cap, int trans_cap, int tail_floats, void* tails, int tail_ld,
                                      void* ref0, void* refN, void* window, int w_rows, int w_first, int w_ring,
                                      void* leaf_sum, void* leaf_min, int prio_at, float prio_eps, float prio_alpha,
                                      hipStream_t stream) {
    if (n_runs < 1 || n_runs > PDA_MAX_RUNS || plane_stride % 16 != 0 || tail_floats > 64) return -1;
    RunPlacement pl;
    for (int i = 0; i < PDA_MAX_RUNS; ++i) {
        pl.plane_base[i] = i < n_runs ? plane_base[i] : 0;
        pl.trans_pos[i] = i < n_runs ? trans_pos[i] : 0;
    }
    const int gx = plane_cap > trans_cap ? plane_cap : trans_cap;
    hipLaunchKernelGGL(replay_unpack_runs_kernel, dim3(gx, n_runs), dim3(256), 0, stream, (const uint8_t*)rec,
                       rec_stride, pl, trans_ring, (uint8_t*)planes, plane_stride,
                       (int)pda_run_planes_offset((uint32_t)tail_floats, (uint32_t)trans_cap),
                       (int)pda_run_trans_stride((uint32_t)tail_floats), tail_floats, (float*)tails, tail_ld, (int*)ref0,
                       (int*)refN, (int*)window, w_rows, w_first, w_ring, (float*)leaf_sum, (float*)leaf_min, prio_at,
                       prio_eps, prio_alpha);
    return 0;
}

// ---- tree repair for a contiguous (mod ring) run of `count` leaves starting at `start`: levels 1 .. stop - 1 in ONE
// launch of one workgroup (16 waves; a wave per node, levels separated by workgroup barriers).  What replay ingestion
// needs after it has written the leaves of a chunk of new rows: a few hundred consecutive leaves touch ~10 nodes of
// level 1 and one or two of every level above, so there is nothing to spread over the GPU -- and nothing to launch
// level by level (set_leaves + 3 repair launches + the index arithmetic around them: ~12 launches per chunk before).
__global__ __launch_bounds__(1024) void segtree_repair_range_kernel(float* __restrict__ sum, float* __restrict__ mn,
                                                                    TreeOffsets offs, int stop, int start, int count,
                                                                    int ring) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    // the run as up to two plain ranges [a0, b0) and [0, b1)
    const int a0 = start, b0 = min(start + count, ring), b1 = start + count > ring ? start + count - ring : 0;
    for (int lvl = 1; lvl < stop; ++lvl) {
        const int sh = 6 * lvl;
        const int n0lo = a0 >> sh, n0hi = (b0 - 1) >> sh;                   // nodes of this level above [a0, b0)
        const int c0 = b0 > a0 ? n0hi - n0lo + 1 : 0;
        const int c1 = b1 > 0 ? ((b1 - 1) >> sh) + 1 : 0;                   // ... and above [0, b1)
        for (int i = wave; i < c0 + c1; i += 16) {
            const int node = i < c0 ? n0lo + i : i - c0;
            const long child = (long)offs.off[lvl - 1] + (long)node * RADIX + lane;
            const float s = wave_sum(sum[child]);
            const float q = wave_min(mn[child]);
            if (lane == 0) {
                sum[offs.off[lvl] + node] = s;
                mn[offs.off[lvl] + node] = q;
            }
        }
        __syncthreads();        // (plain global stores of this workgroup are visible to it after the barrier)
    }
}
PDA_EXPORT void pda_segtree_repair_range(void* sum, void* mn, const int* offs, int stop, int start, int count,
                                         int ring, hipStream_t stream) {
    if (count <= 0 || stop <= 1) return;
    TreeOffsets o;
    for (int i = 0; i < 8; ++i) o.off[i] = offs[i];
    hipLaunchKernelGGL(segtree_repair_range_kernel, dim3(1), dim3(1024), 0, stream, (float*)sum, (float*)mn, o, stop,
                       start, count, ring);
}
