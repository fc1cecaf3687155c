// Latency-optimised all-reduce of a flat fp32 gradient over direct peer stores (xGMI), one kernel per call.
//
// The reference's extra learners are Hogwild writers on one shared CUDA model; its A3C-style gradient hand-off
// (utils/helpers.py:28-34 ensure_global_grads, commented out at core/single_processes/dqn_learner.py:85-86) never
// ran.  Here learners are synchronous data-parallel ranks, one process per MI355X, and what they exchange every
// ~80 us of compute is one 6.75 MB gradient.  A ring all-reduce moves that over ONE outgoing link per GPU in
// 2(N-1) dependent hops; MI355X nodes are fully connected (7 links per GPU), so this kernel uses all links at once
// and is two hops deep:
//
//   phase 1  scatter   rank r stores slice p of its gradient into rank p's staging area, slot r (p != r: N-1 links
//                      busy at once)
//   phase 2  reduce    rank r sums its own slice over the N slots IN RANK ORDER (every rank reduces one slice, so the
//                      result is bit-identical everywhere), writes it into its gradient buffer and stores it into
//                      every peer's result area
//   phase 3  gather    every rank copies the N-1 reduced peer slices from its result area into its gradient buffer
//
// Hand-off protocol: the data IS the flag.  Everything that crosses processes travels as naturally aligned 8-byte
// granules {payload, tag = epoch of this call}, written by ONE store instruction each (two granules per 16-byte
// write-through store) and polled by the consumer until the tag matches -- no separate flag word whose store could
// be observed before the data it covers, no fences, no barriers.  (A first version used flag words behind
// release / acquire fences, exactly as within one process; across processes it delivered, only under load, a few
// 128-byte lines per call whose data arrived after their flag -- found by the multi-process test, which checks every
// element.)  A granule of an earlier call carries an older tag and reads as "not yet"; epochs of different regions
// never meet (pda_peer_alloc: epoch_base).  The epoch is read and bumped by the kernel itself in device memory, so
// the launch takes no per-call argument and replays inside a hipGraph.
// All hand-off memory is one uncached allocation per rank, exported with hipIpcGetMemHandle and mapped by every
// peer; it is only ever touched by system-scope accesses (sc0 sc1: write-through stores, cache-missing loads).
// Work is dealt as (peer p, chunk j) to N x CH workgroups that all stay resident (<= 224 of 256 threads), every spin
// is bounded (a timeout raises an error word instead of hanging the GPU), and a wait only ever depends on PEERS'
// stores, never on another workgroup of the same GPU.
// Optional bf16 wire format: two bf16 per granule (converted while being stored: no conversion launches), summed in
// fp32; the reducing rank rounds its own result through bf16 as well, so all ranks still hold identical gradients.
// Packed fp32 wire format (wire 2): the plain fp32 wire spends half of every granule on the tag (8 bytes per element);
// here a granule carries 48 payload bits and a 16-bit tag, so three fp32 values travel in two granules
//   {p0, p1.lo16 | tag16 << 16}  {p2, p1.hi16 | tag16 << 16}
// = 5.33 bytes per element, still exact fp32, still one 8-byte store unit per tag.  tag16 = 0x8000 | (epoch & 0x7fff):
// never zero (a fresh region is all zeros), and since every granule of a region is rewritten by every call the only
// other tag a reader can meet is the previous call's.
#include "common.h"
#include <cstring>
#include <cstdlib>

#define PAR_MAX_RANKS 8
#define PAR_THREADS 256
#define PAR_MAX_CHUNKS 64           // chunks per slice
#define PAR_SPIN_LIMIT (1u << 21)   // polls (~0.5 us each) before a wait gives up: about a second

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

struct PeerRegionHeader {           // at the start of every rank's region
    unsigned epoch;                 // calls completed by this rank (+ the region's epoch base)
    unsigned done;                  // workgroups of the current call that have finished
    unsigned error;                 // != 0: a wait timed out (code = phase << 8 | awaited rank)
    unsigned fault;                 // test hook: != 0 withholds this rank's phase-1 stores (its peers' waits run out)
    unsigned spin_limit;            // polls before a wait gives up; 0 = PAR_SPIN_LIMIT (tests shorten it)
    unsigned pad[11];
};

struct PeerArArgs {
    char* region[PAR_MAX_RANKS];    // every rank's region as mapped in THIS process (own rank: the local pointer)
    float* local;                   // this rank's gradient (ordinary device memory, 16-byte aligned)
    long n;                         // elements
    long slice;                     // elements per rank slice (multiple of 8)
    long chunk;                     // elements per chunk (multiple of 8 * world)
    long region_bytes;
    int rank, world, chunks;        // chunks per slice (<= PAR_MAX_CHUNKS)
};

namespace {

// bytes of one slot (a slice's granules): 8 bytes per fp32 element, 8 bytes per two bf16 elements, 16 bytes per three
// packed fp32 elements (slice is then a multiple of 24)
__host__ __device__ inline long par_slot_bytes(long slice, int wire) {
    return wire == 1 ? slice * 4 : (wire == 2 ? slice / 3 * 16 : slice * 8);
}
__host__ __device__ inline long par_region_bytes(long slice, int world, int wire) {
    return (long)sizeof(PeerRegionHeader) + 2L * world * par_slot_bytes(slice, wire);    // staging + result slots
}

__device__ __forceinline__ unsigned ld_sys(const unsigned* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ void st_sys(unsigned* p, unsigned v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
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
__device__ __forceinline__ void store4(float* base, long e, long n, const float* f) {
    if (e + 3 < n) {
        *reinterpret_cast<float4*>(base + e) = make_float4(f[0], f[1], f[2], f[3]);
    } else {
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (e + k < n) base[e + k] = f[k];
    }
}

// One 16-byte system-scope access = two granules {payload, tag, payload, tag}.  aux 17 = sc0 | sc1.
__device__ __forceinline__ void st_pair(__amdgpu_buffer_rsrc_t rsrc, unsigned off, unsigned p0, unsigned p1,
                                        unsigned tag) {
    u32x4 v = {p0, tag, p1, tag};
    __builtin_amdgcn_raw_buffer_store_b128(v, rsrc, off, 0, 17);
}
__device__ __forceinline__ u32x4 ld_pair(__amdgpu_buffer_rsrc_t rsrc, unsigned off) {
    return __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, 17);
}

// EL consecutive elements from ordinary device memory / back: EL = 4 (16-byte aligned) or 6 (8-byte aligned)
template <int EL>
__device__ __forceinline__ void load_el(const float* base, long e, long n, float* f) {
    if constexpr (EL == 4) {
        load4(base, e, n, f);
    } else {
        if (e + EL - 1 < n) {
#pragma unroll
            for (int q = 0; q < EL; q += 2) {
                const float2 v = *reinterpret_cast<const float2*>(base + e + q);
                f[q] = v.x; f[q + 1] = v.y;
            }
        } else {
#pragma unroll
            for (int k = 0; k < EL; ++k) f[k] = (e + k < n) ? base[e + k] : 0.f;
        }
    }
}
template <int EL>
__device__ __forceinline__ void store_el(float* base, long e, long n, const float* f) {
    if constexpr (EL == 4) {
        store4(base, e, n, f);
    } else {
        if (e + EL - 1 < n) {
#pragma unroll
            for (int q = 0; q < EL; q += 2) *reinterpret_cast<float2*>(base + e + q) = make_float2(f[q], f[q + 1]);
        } else {
#pragma unroll
            for (int k = 0; k < EL; ++k)
                if (e + k < n) base[e + k] = f[k];
        }
    }
}

// the wire format: one work item = EL consecutive elements <-> NP granule pairs (16-byte accesses) = BYTES bytes
template <int WIRE> struct Wire;
template <> struct Wire<0> {            // one fp32 per granule: 4 elements = 2 pairs = 32 bytes
    static constexpr int EL = 4, NP = 2, BYTES = 32;
    __device__ static unsigned tag_of(unsigned epoch) { return epoch; }
    __device__ static bool ok(u32x4 v, unsigned tag) { return v.y == tag && v.w == tag; }
    __device__ static void store(__amdgpu_buffer_rsrc_t r, unsigned off, const float* f, unsigned tag) {
        st_pair(r, off, __float_as_uint(f[0]), __float_as_uint(f[1]), tag);
        st_pair(r, off + 16, __float_as_uint(f[2]), __float_as_uint(f[3]), tag);
    }
    __device__ static void unpack(const u32x4* v, float* f) {
        f[0] = __uint_as_float(v[0].x); f[1] = __uint_as_float(v[0].z);
        f[2] = __uint_as_float(v[1].x); f[3] = __uint_as_float(v[1].z);
    }
    __device__ static void round_trip(float*) {}
};
template <> struct Wire<1> {            // two bf16 per granule: 4 elements = 1 pair = 16 bytes
    static constexpr int EL = 4, NP = 1, BYTES = 16;
    __device__ static unsigned tag_of(unsigned epoch) { return epoch; }
    __device__ static bool ok(u32x4 v, unsigned tag) { return v.y == tag && v.w == tag; }
    __device__ static void store(__amdgpu_buffer_rsrc_t r, unsigned off, const float* f, unsigned tag) {
        st_pair(r, off, pack2bf(f[0], f[1]), pack2bf(f[2], f[3]), tag);
    }
    __device__ static void unpack(const u32x4* v, float* f) {
        f[0] = bf2f((bf16_t)(v[0].x & 0xffffu)); f[1] = bf2f((bf16_t)(v[0].x >> 16));
        f[2] = bf2f((bf16_t)(v[0].z & 0xffffu)); f[3] = bf2f((bf16_t)(v[0].z >> 16));
    }
    __device__ static void round_trip(float* f) {       // what a peer receives of these four values
#pragma unroll
        for (int k = 0; k < 4; ++k) f[k] = bf2f(f2bf(f[k]));
    }
};
template <> struct Wire<2> {            // three fp32 per granule pair (48 payload + 16 tag bits per granule): 6 elements = 32 bytes
    static constexpr int EL = 6, NP = 2, BYTES = 32;  // (the same registers per work item as the plain wire: all workgroups stay resident)
    __device__ static unsigned tag_of(unsigned epoch) { return 0x8000u | (epoch & 0x7fffu); }
    __device__ static bool ok(u32x4 v, unsigned tag) { return (v.y >> 16) == tag && (v.w >> 16) == tag; }
    __device__ static void store(__amdgpu_buffer_rsrc_t r, unsigned off, const float* f, unsigned tag) {
#pragma unroll
        for (int h = 0; h < NP; ++h) {
            const unsigned p0 = __float_as_uint(f[3 * h]), p1 = __float_as_uint(f[3 * h + 1]), p2 = __float_as_uint(f[3 * h + 2]);
            u32x4 v = {p0, (p1 & 0xffffu) | (tag << 16), p2, (p1 >> 16) | (tag << 16)};
            __builtin_amdgcn_raw_buffer_store_b128(v, r, off + 16 * h, 0, 17);
        }
    }
    __device__ static void unpack(const u32x4* v, float* f) {
#pragma unroll
        for (int h = 0; h < NP; ++h) {
            f[3 * h] = __uint_as_float(v[h].x);
            f[3 * h + 1] = __uint_as_float((v[h].y & 0xffffu) | (v[h].w << 16));
            f[3 * h + 2] = __uint_as_float(v[h].z);
        }
    }
    __device__ static void round_trip(float*) {}
};

template <int WIRE>
__global__ __launch_bounds__(PAR_THREADS) void peer_allreduce_kernel(PeerArArgs a) {
    typedef Wire<WIRE> W;
    constexpr int EL = W::EL;
    const int p = blockIdx.x / a.chunks, j = blockIdx.x % a.chunks;       // (peer, chunk of that peer's slice)
    const int r = a.rank, world = a.world;
    PeerRegionHeader* me = reinterpret_cast<PeerRegionHeader*>(a.region[r]);
    const unsigned epoch = ld_sys(&me->epoch) + 1u;         // every workgroup reads it before any can finish the call
    const unsigned tag = W::tag_of(epoch);
    // (both hooks live in device memory, not in the launch arguments: a captured launch replays them as they are NOW)
    const bool withhold = ld_sys(&me->fault) != 0u;
    const unsigned lim = ld_sys(&me->spin_limit);
    const unsigned spin_limit = lim ? lim : PAR_SPIN_LIMIT;
    const long slice = a.slice, chunk = a.chunk;
    const long c0 = (long)j * chunk, c1 = min(c0 + chunk, slice);          // chunk range within a slice
    const unsigned slot = (unsigned)par_slot_bytes(slice, WIRE);
    const unsigned hdr = (unsigned)sizeof(PeerRegionHeader);
    // region of rank `owner`: staging slot s at hdr + s * slot, result slot s at hdr + (world + s) * slot
    auto rsrc_of = [&](int owner) {
        return __builtin_amdgcn_make_buffer_rsrc(a.region[owner], 0, (int)a.region_bytes, 0x00020000);
    };
    const __amdgpu_buffer_rsrc_t mine = rsrc_of(r);

    // ---- phase 1: my copy of peer p's slice -> p's staging slot r ----------------------------------------------
    if (p != r && !withhold) {
        const __amdgpu_buffer_rsrc_t dst = rsrc_of(p);
        const unsigned base = hdr + (unsigned)r * slot;
        for (long e = c0 + (long)threadIdx.x * EL; e < c1; e += (long)PAR_THREADS * EL) {
            float f[EL];
            load_el<EL>(a.local, (long)p * slice + e, a.n, f);
            W::store(dst, base + (unsigned)(e / EL) * W::BYTES, f, tag);
        }
    }

    // ---- phase 2: reduce sub-chunk p of chunk j of MY slice; into my gradient and every peer's result slot r ----
    {
        const long sub = chunk / world;                     // multiple of 8 / 24 (pda_peer_plan3)
        const long s0 = min(c0 + (long)p * sub, c1), s1 = (p == world - 1) ? c1 : min(s0 + sub, c1);
        for (long e = s0 + (long)threadIdx.x * EL; e < s1; e += (long)PAR_THREADS * EL) {
            const unsigned goff = (unsigned)(e / EL) * W::BYTES;
            // all peers' granules in flight together, then re-poll whatever has not arrived
            u32x4 v[PAR_MAX_RANKS][W::NP];
#pragma unroll
            for (int s = 0; s < PAR_MAX_RANKS; ++s)
                if (s < world && s != r) {
#pragma unroll
                    for (int h = 0; h < W::NP; ++h) v[s][h] = ld_pair(mine, hdr + (unsigned)s * slot + goff + 16 * h);
                }
            for (unsigned spins = 0;;) {
                bool all_ok = true;
#pragma unroll
                for (int s = 0; s < PAR_MAX_RANKS; ++s)
                    if (s < world && s != r) {
#pragma unroll
                        for (int h = 0; h < W::NP; ++h)
                            if (!W::ok(v[s][h], tag)) {
                                all_ok = false;
                                v[s][h] = ld_pair(mine, hdr + (unsigned)s * slot + goff + 16 * h);
                            }
                    }
                if (all_ok) break;
                __builtin_amdgcn_s_sleep(4);
                if (++spins > spin_limit) { atomicExch(&me->error, (1u << 8) | 0xffu); break; }
                // a wait of this call has already run out somewhere on this GPU: the result is void anyway, drain fast
                if ((spins & 63u) == 0u && ld_sys(&me->error) != 0u) break;
            }
            float acc[EL];
#pragma unroll
            for (int k = 0; k < EL; ++k) acc[k] = 0.f;
#pragma unroll
            for (int s = 0; s < PAR_MAX_RANKS; ++s)             // rank order: the same sum on every run and rank
                if (s < world) {
                    float f[EL];
                    if (s == r) {
                        load_el<EL>(a.local, (long)r * slice + e, a.n, f);
                        W::round_trip(f);
                    } else {
                        W::unpack(v[s], f);
                    }
#pragma unroll
                    for (int k = 0; k < EL; ++k) acc[k] += f[k];
                }
            W::round_trip(acc);                                 // every rank ends up with the same bits
            store_el<EL>(a.local, (long)r * slice + e, a.n, acc);
#pragma unroll
            for (int q = 0; q < PAR_MAX_RANKS; ++q)
                if (q < world && q != r)
                    W::store(rsrc_of(q), hdr + (unsigned)(world + r) * slot + goff, acc, tag);
        }
    }

    // ---- phase 3: reduced chunk j of peer p's slice -> my gradient buffer ----------------------------------------
    if (p != r) {
        const unsigned base = hdr + (unsigned)(world + p) * slot;
        for (long e = c0 + (long)threadIdx.x * EL; e < c1; e += (long)PAR_THREADS * EL) {
            const unsigned goff = base + (unsigned)(e / EL) * W::BYTES;
            u32x4 v[W::NP];
#pragma unroll
            for (int h = 0; h < W::NP; ++h) v[h] = ld_pair(mine, goff + 16 * h);
            for (unsigned spins = 0;;) {
                bool all_ok = true;
#pragma unroll
                for (int h = 0; h < W::NP; ++h)
                    if (!W::ok(v[h], tag)) { all_ok = false; v[h] = ld_pair(mine, goff + 16 * h); }
                if (all_ok) break;
                __builtin_amdgcn_s_sleep(4);
                if (++spins > spin_limit) { atomicExch(&me->error, (2u << 8) | (unsigned)p); break; }
                if ((spins & 63u) == 0u && ld_sys(&me->error) != 0u) break;
            }
            float f[EL];
            W::unpack(v, f);
            store_el<EL>(a.local, (long)p * slice + e, a.n, f);
        }
    }

    // ---- the last workgroup to finish closes the call: epoch += 1 ------------------------------------------------
    // (No fence here: the gradient written above is read by LATER kernels of this stream, which the end of this one
    // orders; an agent-scope fence would write back the L2 -- full of the concurrent backward kernels' dirty lines --
    // once per workgroup.  The epoch bump below only has to follow every workgroup's READ of the epoch, and a workgroup
    // counts itself done after that read.)
    __syncthreads();
    if (threadIdx.x == 0) {
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
// budget: workgroups of the launch as a whole (0 = 224, all resident on a GPU of its own; several ranks SHARING one
// card -- rehearsals on a one-GPU box -- must fit all their launches on it together and ask for fewer).  Every rank
// must use the same budget: the chunk geometry addresses the peers' regions.
PDA_EXPORT void pda_peer_plan3(long n, int world, int budget, int wire, long* slice, long* chunk, int* chunks) {
    const long unit = wire == 2 ? 24 : 8;           // whole 16-byte accesses of every wire format (packed fp32: 6-element items)
    long sl = (n + world - 1) / world;
    sl = (sl + unit - 1) / unit * unit;
    if (budget <= 0) budget = 224;
    int target = budget / world;                    // ~224 resident workgroups in all
    if (target > PAR_MAX_CHUNKS) target = PAR_MAX_CHUNKS;
    if (target < 1) target = 1;
    const long gran = unit * world;                 // sub-chunks stay whole work items
    long ch = (sl + target - 1) / target;
    ch = (ch + gran - 1) / gran * gran;
    if (ch < gran) ch = gran;
    int nch = (int)((sl + ch - 1) / ch);
    *slice = sl; *chunk = ch; *chunks = nch;
}
PDA_EXPORT void pda_peer_plan2(long n, int world, int budget, long* slice, long* chunk, int* chunks) {
    pda_peer_plan3(n, world, budget, 0, slice, chunk, chunks);
}

PDA_EXPORT void pda_peer_plan(long n, int world, long* slice, long* chunk, int* chunks) {
    pda_peer_plan3(n, world, 0, 0, slice, chunk, chunks);
}

// wire: 0 = fp32 (8 bytes per element), 1 = bf16 (4), 2 = packed fp32 (5.33)
PDA_EXPORT long pda_peer_region_bytes(long n, int world, int wire) {
    long slice, chunk; int chunks;
    pda_peer_plan3(n, world, 0, wire, &slice, &chunk, &chunks);
    return par_region_bytes(slice, world, wire);
}

// Uncached device memory, zero-filled, + its IPC handle (64 bytes).  Returns 0 on success, a hipError_t otherwise.
// epoch_base: the region's first call uses tag epoch_base + 1.  Every region a process creates gets a base far above
// the epochs of all regions before it (the caller counts regions; all ranks create them in the same order), so a
// granule left over from an EARLIER region at the same address can never look valid to this one.
PDA_EXPORT int pda_peer_alloc(long bytes, unsigned epoch_base, void** ptr, void* handle64) {
    hipError_t e = hipExtMallocWithFlags(ptr, (size_t)bytes, hipDeviceMallocUncached);
    if (e != hipSuccess) return (int)e;
    e = hipMemset(*ptr, 0, (size_t)bytes);
    if (e != hipSuccess) return (int)e;
    e = hipMemcpy((char*)*ptr + offsetof(PeerRegionHeader, epoch), &epoch_base, sizeof(unsigned), hipMemcpyHostToDevice);
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

// clears the error word (after the caller has dealt with a timed-out wait); synchronises the device
PDA_EXPORT int pda_peer_clear_error(void* region) {
    return (int)hipMemset((char*)region + offsetof(PeerRegionHeader, error), 0, sizeof(unsigned));
}

// test hooks in the local region's header (synchronise the device): withhold this rank's phase-1 stores; polls before a
// wait gives up (0 = default, about a second)
PDA_EXPORT int pda_peer_set_fault(void* region, unsigned withhold, unsigned spin_limit) {
    unsigned v[2] = {withhold, spin_limit};
    return (int)hipMemcpy((char*)region + offsetof(PeerRegionHeader, fault), v, sizeof(v), hipMemcpyHostToDevice);
}

PDA_EXPORT int pda_peer_allreduce(void* const* regions, void* local, long n, int rank, int world, int wire,
                                  int budget, hipStream_t stream) {
    if (world < 2 || world > PAR_MAX_RANKS || wire < 0 || wire > 2) return -1;
    PeerArArgs a;
    for (int i = 0; i < PAR_MAX_RANKS; ++i) a.region[i] = i < world ? (char*)regions[i] : nullptr;
    a.local = (float*)local; a.n = n; a.rank = rank; a.world = world;
    pda_peer_plan3(n, world, budget, wire, &a.slice, &a.chunk, &a.chunks);
    a.region_bytes = par_region_bytes(a.slice, world, wire);
    if (a.region_bytes >= (1L << 31)) return -2;            // buffer descriptors address 2 GB
    dim3 grid(world * a.chunks), block(PAR_THREADS);
    if (wire == 1) hipLaunchKernelGGL(peer_allreduce_kernel<1>, grid, block, 0, stream, a);
    else if (wire == 2) hipLaunchKernelGGL(peer_allreduce_kernel<2>, grid, block, 0, stream, a);
    else hipLaunchKernelGGL(peer_allreduce_kernel<0>, grid, block, 0, stream, a);
    return (int)hipGetLastError();
}
