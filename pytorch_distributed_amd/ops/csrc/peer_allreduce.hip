// Latency-optimised all-reduce of a flat fp32 gradient over direct peer stores (xGMI), one kernel per call.
//
// The reference's extra learners are Hogwild writers on one shared CUDA model; its A3C-style gradient hand-off
// (utils/helpers.py:28-34 ensure_global_grads, commented out at core/single_processes/dqn_learner.py:85-86) never
// ran.  Here learners are synchronous data-parallel ranks, one process per MI355X, and what they exchange every
// ~80 us of compute is one 6.75 MB gradient.  A ring all-reduce moves that over ONE outgoing link per GPU in
// 2(N-1) dependent hops; MI355X nodes are fully connected (7 links x ~153 GB/s per GPU), so this kernel uses all
// links at once and is two hops deep:
//
//   phase 1  scatter   rank r stores slice p of its gradient into rank p's staging area, slot r  (p != r; N-1
//                      links busy at once), then raises flag1[r][chunk] in rank p's memory
//   phase 2  reduce    rank r sums its own slice over the N slots in rank order (every rank reduces ONE slice, so
//                      the result is bit-identical everywhere) and stores the sum into every rank's result area,
//                      then raises flag2[r][chunk][sub]
//   phase 3  gather    every rank copies the N reduced slices from its result area into its gradient buffer
//
// All hand-off memory (staging, result, flags) is one uncached (fine-grained, MTYPE_UC) allocation per rank,
// exported with hipIpcGetMemHandle and mapped by every peer: uncached accesses bypass the per-XCD L2s on both
// sides, so a consumer never holds a stale line and a producer needs no write-back -- only a system-scope fence
// between its data stores and its flag store.  Flags carry a monotonically increasing epoch that the kernel itself
// reads and bumps in device memory, so the launch takes no per-call argument and replays inside a hipGraph.
// Work is dealt as (peer p, chunk j) to N x CH workgroups which all stay resident (<= 256 of 256 threads), every
// spin is bounded (a timeout raises an error word instead of hanging the GPU), and no workgroup ever waits on
// another workgroup of its own GPU, only on flags stored by peers.
// Optional bf16 wire format: slices cross the links as bf16 (converted while being stored: no conversion launches)
// and are summed in fp32; the reducing rank rounds its sum through bf16 as well, so all ranks still end up with
// identical gradients.
#include "common.h"
#include <cstring>

#define PAR_MAX_RANKS 8
#define PAR_THREADS 256
#define PAR_MAX_CHUNKS 64           // chunks per slice (flag array sizing)
#define PAR_SPIN_LIMIT (1u << 22)   // polls (~64 cycles of sleep each) before a wait gives up: a few seconds

struct PeerRegionHeader {           // at the start of every rank's region
    unsigned flag1[PAR_MAX_RANKS][PAR_MAX_CHUNKS];                    // [source rank][chunk]
    unsigned flag2[PAR_MAX_RANKS][PAR_MAX_CHUNKS][PAR_MAX_RANKS];     // [source rank][chunk][sub-chunk]
    unsigned epoch;                 // calls completed by this rank
    unsigned done;                  // workgroups of the current call that have finished
    unsigned error;                 // != 0: a wait timed out (code = phase << 8 | awaited rank)
    unsigned pad[13];
};

struct PeerArArgs {
    char* region[PAR_MAX_RANKS];    // every rank's region as mapped in THIS process (own rank: the local pointer)
    float* local;                   // this rank's gradient (ordinary device memory)
    long n;                         // elements
    long slice;                     // elements per rank slice (multiple of 8)
    long chunk;                     // elements per chunk (multiple of 8 * world)
    int rank, world, chunks;        // chunks per slice (<= PAR_MAX_CHUNKS)
    int wire_bf16;
};

namespace {

__host__ __device__ inline long par_stage_off() { return (long)sizeof(PeerRegionHeader); }
// staging slot s (what rank s sent us) and the result area follow the header; both in wire elements
__host__ __device__ inline long par_region_bytes(long slice, int world, int wire_bytes) {
    return par_stage_off() + (long)world * slice * wire_bytes          // staging [world][slice]
           + (long)world * slice * wire_bytes;                         // result  [world][slice]
}

__device__ __forceinline__ unsigned ld_sys(const unsigned* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ void st_sys(unsigned* p, unsigned v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// thread 0 waits until *flag has reached `epoch` (flags only grow; the comparison is wrap-safe), everybody follows
__device__ __forceinline__ bool wait_flag(const unsigned* flag, unsigned epoch, unsigned* err, unsigned code) {
    __shared__ int ok_s;
    if (threadIdx.x == 0) {
        int ok = 1;
        unsigned spins = 0;
        while ((int)(ld_sys(flag) - epoch) < 0) {
            __builtin_amdgcn_s_sleep(8);
            if (++spins > PAR_SPIN_LIMIT) { ok = 0; atomicExch(err, code); break; }
        }
        __atomic_thread_fence(__ATOMIC_ACQUIRE);       // system scope: later loads see the data behind the flag
        ok_s = ok;
    }
    __syncthreads();
    const bool ok = ok_s != 0;
    __syncthreads();        // ok_s may be rewritten by the next wait
    return ok;
}

// every thread's stores are done and visible system-wide, then thread 0 raises the flag
__device__ __forceinline__ void raise_flag(unsigned* flag, unsigned epoch) {
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) st_sys(flag, epoch);
}

// 4 gradient elements from ordinary device memory (16-byte aligned base); elements past n read as zero
__device__ __forceinline__ void load4(const float* base, long e, long n, float* f) {
    if (e + 3 < n) {
        const float4 v = *reinterpret_cast<const float4*>(base + e);
        f[0] = v.x; f[1] = v.y; f[2] = v.z; f[3] = v.w;
    } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) f[k] = (e + k < n) ? base[e + k] : 0.f;
    }
}

template <bool BF16> struct Wire;
template <> struct Wire<false> {
    typedef float4 V;                       // 4 elements per access
    static constexpr int N = 4;
    __device__ static V pack(const float* f) { return make_float4(f[0], f[1], f[2], f[3]); }
    __device__ static void unpack(V v, float* f) { f[0] = v.x; f[1] = v.y; f[2] = v.z; f[3] = v.w; }
};
template <> struct Wire<true> {
    typedef uint2 V;                        // 4 bf16
    static constexpr int N = 4;
    __device__ static V pack(const float* f) { return make_uint2(pack2bf(f[0], f[1]), pack2bf(f[2], f[3])); }
    __device__ static void unpack(V v, float* f) {
        f[0] = bf2f((bf16_t)(v.x & 0xffffu)); f[1] = bf2f((bf16_t)(v.x >> 16));
        f[2] = bf2f((bf16_t)(v.y & 0xffffu)); f[3] = bf2f((bf16_t)(v.y >> 16));
    }
};

template <bool BF16>
__global__ __launch_bounds__(PAR_THREADS) void peer_allreduce_kernel(PeerArArgs a) {
    typedef Wire<BF16> W;
    typedef typename W::V V;
    const int p = blockIdx.x / a.chunks, j = blockIdx.x % a.chunks;       // (peer, chunk of that peer's slice)
    const int r = a.rank, world = a.world;
    PeerRegionHeader* me = reinterpret_cast<PeerRegionHeader*>(a.region[r]);
    const unsigned epoch = ld_sys(&me->epoch) + 1u;         // every workgroup reads it before any can finish the call
    const long slice = a.slice, chunk = a.chunk;
    const long c0 = (long)j * chunk, c1 = min(c0 + chunk, slice);          // chunk range within a slice
    auto stage_of = [&](int owner, int slot) {      // staging slot `slot` in rank `owner`'s region
        return reinterpret_cast<V*>(a.region[owner] + par_stage_off()) + ((long)slot * slice) / W::N;
    };
    auto result_of = [&](int owner, int sl) {       // reduced slice `sl` in rank `owner`'s result area
        return reinterpret_cast<V*>(a.region[owner] + par_stage_off()) + ((long)(world + sl) * slice) / W::N;
    };
    bool ok = true;

    // ---- phase 1: my copy of peer p's slice -> p's staging slot r ----------------------------------------------
    if (p != r) {
        V* dst = stage_of(p, r);
        for (long e = c0 + (long)threadIdx.x * W::N; e < c1; e += (long)PAR_THREADS * W::N) {
            float f[4];
            load4(a.local, (long)p * slice + e, a.n, f);
            dst[e / W::N] = W::pack(f);
        }
        raise_flag(&reinterpret_cast<PeerRegionHeader*>(a.region[p])->flag1[r][j], epoch);
    }

    // ---- phase 2: reduce sub-chunk p of chunk j of MY slice, store it everywhere -----------------------------
    {
        const long sub = (chunk / world + W::N - 1) / W::N * W::N;
        const long s0 = min(c0 + (long)p * sub, c1), s1 = (p == world - 1) ? c1 : min(s0 + sub, c1);
        for (int s = 0; s < world && ok; ++s)
            if (s != r) ok = wait_flag(&me->flag1[s][j], epoch, &me->error, (1u << 8) | (unsigned)s);
        if (ok) {
            for (long e = s0 + (long)threadIdx.x * W::N; e < s1; e += (long)PAR_THREADS * W::N) {
                float acc[4] = {0.f, 0.f, 0.f, 0.f};
                for (int s = 0; s < world; ++s) {           // rank order: the same sum on every run
                    float f[4];
                    if (s == r) {
                        load4(a.local, (long)r * slice + e, a.n, f);
                        if (BF16) { V t = W::pack(f); W::unpack(t, f); }       // what the peers would have received
                    } else {
                        W::unpack(stage_of(r, s)[e / W::N], f);
                    }
#pragma unroll
                    for (int k = 0; k < 4; ++k) acc[k] += f[k];
                }
                const V v = W::pack(acc);
                for (int q = 0; q < world; ++q) result_of(q, r)[e / W::N] = v;
            }
        }
        __threadfence_system();
        __syncthreads();
        if (threadIdx.x == 0 && ok)
            for (int q = 0; q < world; ++q)
                st_sys(&reinterpret_cast<PeerRegionHeader*>(a.region[q])->flag2[r][j][p], epoch);
    }

    // ---- phase 3: reduced chunk j of slice p -> my gradient buffer ---------------------------------------------
    for (int sb = 0; sb < world && ok; ++sb)
        ok = wait_flag(&me->flag2[p][j][sb], epoch, &me->error, (2u << 8) | (unsigned)p);
    if (ok) {
        const V* src = result_of(r, p);
        float* dst = a.local + (long)p * slice;
        for (long e = c0 + (long)threadIdx.x * W::N; e < c1; e += (long)PAR_THREADS * W::N) {
            float f[4];
            W::unpack(src[e / W::N], f);
            if ((long)p * slice + e + 3 < a.n) {
                *reinterpret_cast<float4*>(dst + e) = make_float4(f[0], f[1], f[2], f[3]);
            } else {
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if ((long)p * slice + e + k < a.n) dst[e + k] = f[k];
            }
        }
    }

    // ---- the last workgroup to finish closes the call: epoch += 1 ------------------------------------------------
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();
        const unsigned prev = atomicAdd(&me->done, 1u);
        if (prev + 1u == gridDim.x) {
            me->done = 0u;
            st_sys(&me->epoch, epoch);
        }
    }
}

}  // namespace

// ---- host side ---------------------------------------------------------------------------------------------------
// geometry for n elements over `world` ranks: slice / chunk lengths and chunks per slice
PDA_EXPORT void pda_peer_plan(long n, int world, long* slice, long* chunk, int* chunks) {
    long sl = (n + world - 1) / world;
    sl = (sl + 7) / 8 * 8;
    int target = 224 / world;                       // ~224 resident workgroups in all
    if (target > PAR_MAX_CHUNKS) target = PAR_MAX_CHUNKS;
    if (target < 1) target = 1;
    const long gran = 8L * world;                   // sub-chunks stay 16-byte aligned
    long ch = (sl + target - 1) / target;
    ch = (ch + gran - 1) / gran * gran;
    if (ch < gran) ch = gran;
    int nch = (int)((sl + ch - 1) / ch);
    *slice = sl; *chunk = ch; *chunks = nch;
}

PDA_EXPORT long pda_peer_region_bytes(long n, int world, int wire_bf16) {
    long slice, chunk; int chunks;
    pda_peer_plan(n, world, &slice, &chunk, &chunks);
    return par_region_bytes(slice, world, wire_bf16 ? 2 : 4);
}

// uncached device memory + its IPC handle (64 bytes).  Returns 0 on success, a hipError_t otherwise.
PDA_EXPORT int pda_peer_alloc(long bytes, void** ptr, void* handle64) {
    hipError_t e = hipExtMallocWithFlags(ptr, (size_t)bytes, hipDeviceMallocUncached);
    if (e != hipSuccess) return (int)e;
    e = hipMemset(*ptr, 0, (size_t)bytes);
    if (e != hipSuccess) return (int)e;
    e = hipDeviceSynchronize();
    if (e != hipSuccess) return (int)e;
    hipIpcMemHandle_t h;
    e = hipIpcGetMemHandle(&h, *ptr);
    if (e != hipSuccess) return (int)e;
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "IPC handle size");
    std::memcpy(handle64, &h, 64);
    return 0;
}

PDA_EXPORT int pda_peer_open(const void* handle64, void** ptr) {
    hipIpcMemHandle_t h;
    std::memcpy(&h, handle64, 64);
    return (int)hipIpcOpenMemHandle(ptr, h, hipIpcMemLazyEnablePeerAccess);
}

PDA_EXPORT int pda_peer_close(void* ptr) { return (int)hipIpcCloseMemHandle(ptr); }
PDA_EXPORT int pda_peer_free(void* ptr) { return (int)hipFree(ptr); }

// error word of the local region (0 = no wait has timed out); synchronises the device
PDA_EXPORT int pda_peer_error(void* region) {
    unsigned v = 0;
    if (hipMemcpy(&v, (char*)region + offsetof(PeerRegionHeader, error), sizeof(v), hipMemcpyDeviceToHost) != hipSuccess)
        return -1;
    return (int)v;
}

PDA_EXPORT int pda_peer_allreduce(void* const* regions, void* local, long n, int rank, int world, int wire_bf16,
                                  hipStream_t stream) {
    if (world < 2 || world > PAR_MAX_RANKS) return -1;
    PeerArArgs a;
    for (int i = 0; i < PAR_MAX_RANKS; ++i) a.region[i] = i < world ? (char*)regions[i] : nullptr;
    a.local = (float*)local; a.n = n; a.rank = rank; a.world = world; a.wire_bf16 = wire_bf16;
    pda_peer_plan(n, world, &a.slice, &a.chunk, &a.chunks);
    dim3 grid(world * a.chunks), block(PAR_THREADS);
    if (wire_bf16) hipLaunchKernelGGL(peer_allreduce_kernel<true>, grid, block, 0, stream, a);
    else hipLaunchKernelGGL(peer_allreduce_kernel<false>, grid, block, 0, stream, a);
    return (int)hipGetLastError();
}
