// Lock-free bounded MPMC ring over a caller-provided shared-memory region.
//
// Many CPU actor processes push fixed-size transition records; learner ranks
// pop them in batches straight into a pinned staging buffer that is then DMA'd
// to HBM-resident replay.  Replaces the reference's single mp.Lock around every
// feed and sample (core/memories/shared_memory.py:69-75), which serialises all
// actors behind the learner's gather.
//
// Algorithm: per-slot sequence numbers (bounded MPMC queue).  A producer claims
// position `pos` by CAS on `head` when slot.seq == pos, writes the payload, then
// publishes with slot.seq = pos + 1 (release).  A consumer claims `pos` by CAS on
// `tail` when slot.seq == pos + 1, reads, then frees with slot.seq = pos + N.
// No syscalls, no locks; a stalled producer only delays its own slot.
//
// A producer that DIES between its claim and its publish (fault injection, OOM kill, the supervisor's terminate())
// would leave the slot unpublished for good and the consumer stuck behind it.  So a claim also records the
// claimant's pid in the slot; a consumer that finds the slot at `tail` claimed but unpublished for longer than
// Header::stuck_ms skips it when that process no longer exists (kill(pid, 0) == ESRCH), or when no pid was recorded
// yet (death between the two stores).  A live, merely slow producer is never skipped.
//
// The region is plain memory (a torch shared-memory tensor on the Python side),
// so the same code works across spawn'ed processes.  C ABI for ctypes.
#include <atomic>
#include <cerrno>
#include <csignal>
#include <cstdint>
#include <cstring>
#include <chrono>
#include <new>
#include <thread>
#include <vector>

#include "mailbox.h"
#include "frame_runs.h"

namespace {

constexpr uint64_t kMagic = 0x50444152494e4731ull;  // "PDARING1"
constexpr size_t kLine = 64;

struct alignas(kLine) Header {
    uint64_t magic;
    uint32_t num_slots;      // power of two
    uint32_t payload_bytes;  // bytes copied per record
    uint32_t slot_stride;    // bytes between slots (header + padded payload)
    uint32_t stuck_ms;       // how long the consumer lets a claimed, unpublished slot block the tail (see above)
    alignas(kLine) std::atomic<uint64_t> head;
    alignas(kLine) std::atomic<uint64_t> tail;
    alignas(kLine) std::atomic<uint64_t> pushed;
    alignas(kLine) std::atomic<uint64_t> popped;
    alignas(kLine) std::atomic<uint64_t> full_events;
    alignas(kLine) std::atomic<uint64_t> skipped;       // slots abandoned by dead producers
    std::atomic<uint64_t> stuck_pos;                    // consumer side: the tail position that was found blocked ...
    std::atomic<uint64_t> stuck_since_ms;               // ... and when (steady clock, ms)
};

struct alignas(kLine) Slot {
    std::atomic<uint64_t> seq;
    std::atomic<uint32_t> owner;        // pid of the producer between claim and publish, else 0
};

inline uint64_t now_ms() {
    return (uint64_t)std::chrono::duration_cast<std::chrono::milliseconds>(
               std::chrono::steady_clock::now().time_since_epoch()).count();
}

static_assert(std::atomic<uint64_t>::is_always_lock_free, "need lock-free 64-bit atomics");

inline size_t round_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
inline Header* hdr(void* base) { return reinterpret_cast<Header*>(base); }
inline char* slots_base(void* base) { return reinterpret_cast<char*>(base) + sizeof(Header); }
inline Slot* slot_at(void* base, uint64_t pos) {
    Header* h = hdr(base);
    return reinterpret_cast<Slot*>(slots_base(base) + (pos & (h->num_slots - 1)) * (size_t)h->slot_stride);
}
inline char* payload_of(Slot* s) { return reinterpret_cast<char*>(s) + sizeof(Slot); }

// The slot at tail position `pos` is claimed (head > pos) but not published.  True once that has lasted stuck_ms
// AND its claimant is gone.
inline bool abandoned(Header* h, Slot* s, uint64_t pos) {
    const uint64_t t = now_ms();
    if (h->stuck_pos.load(std::memory_order_relaxed) != pos) {
        h->stuck_pos.store(pos, std::memory_order_relaxed);
        h->stuck_since_ms.store(t, std::memory_order_relaxed);
        return false;
    }
    if (t - h->stuck_since_ms.load(std::memory_order_relaxed) < h->stuck_ms) return false;
    const uint32_t pid = s->owner.load(std::memory_order_relaxed);
    if (pid == 0) return true;                              // died between the claim and the owner store
    return kill((pid_t)pid, 0) != 0 && errno == ESRCH;      // the claimant no longer exists
}

}  // namespace

extern "C" {

// Bytes the region must have for `num_slots` records of `payload_bytes`.
uint64_t pda_ring_bytes(uint32_t num_slots, uint32_t payload_bytes) {
    return sizeof(Header) + (uint64_t)num_slots * (sizeof(Slot) + round_up(payload_bytes, kLine));
}

// Initialise (once, before any producer/consumer starts).  Returns 0 on success.
int pda_ring_init(void* base, uint64_t region_bytes, uint32_t num_slots, uint32_t payload_bytes) {
    if (num_slots == 0 || (num_slots & (num_slots - 1)) != 0) return -1;
    if (reinterpret_cast<uintptr_t>(base) % kLine != 0) return -2;
    if (region_bytes < pda_ring_bytes(num_slots, payload_bytes)) return -3;
    Header* h = new (base) Header();
    h->num_slots = num_slots;
    h->payload_bytes = payload_bytes;
    h->slot_stride = (uint32_t)(sizeof(Slot) + round_up(payload_bytes, kLine));
    h->head.store(0, std::memory_order_relaxed);
    h->tail.store(0, std::memory_order_relaxed);
    h->pushed.store(0, std::memory_order_relaxed);
    h->popped.store(0, std::memory_order_relaxed);
    h->full_events.store(0, std::memory_order_relaxed);
    h->skipped.store(0, std::memory_order_relaxed);
    h->stuck_pos.store(~0ull, std::memory_order_relaxed);
    h->stuck_since_ms.store(0, std::memory_order_relaxed);
    h->stuck_ms = 2000;
    for (uint32_t i = 0; i < num_slots; ++i) {
        Slot* s = new (slots_base(base) + (size_t)i * h->slot_stride) Slot();
        s->seq.store(i, std::memory_order_relaxed);
        s->owner.store(0, std::memory_order_relaxed);
    }
    std::atomic_thread_fence(std::memory_order_seq_cst);
    h->magic = kMagic;
    return 0;
}

int pda_ring_valid(void* base) { return hdr(base)->magic == kMagic ? 1 : 0; }

// Push one record assembled from up to 4 source segments (scatter-gather, so the
// caller never builds an intermediate copy).  Returns 1 on success, 0 if full.
int pda_ring_push(void* base, const void* seg0, uint32_t n0, const void* seg1, uint32_t n1,
                  const void* seg2, uint32_t n2, const void* seg3, uint32_t n3) {
    Header* h = hdr(base);
    if ((uint64_t)n0 + n1 + n2 + n3 > h->payload_bytes) return -1;
    uint64_t pos = h->head.load(std::memory_order_relaxed);
    Slot* s;
    for (;;) {
        s = slot_at(base, pos);
        uint64_t seq = s->seq.load(std::memory_order_acquire);
        int64_t dif = (int64_t)(seq - pos);
        if (dif == 0) {
            if (h->head.compare_exchange_weak(pos, pos + 1, std::memory_order_relaxed)) break;
        } else if (dif < 0) {
            h->full_events.fetch_add(1, std::memory_order_relaxed);
            return 0;
        } else {
            pos = h->head.load(std::memory_order_relaxed);
        }
    }
    s->owner.store((uint32_t)getpid(), std::memory_order_relaxed);
    char* dst = payload_of(s);
    if (n0) { std::memcpy(dst, seg0, n0); dst += n0; }
    if (n1) { std::memcpy(dst, seg1, n1); dst += n1; }
    if (n2) { std::memcpy(dst, seg2, n2); dst += n2; }
    if (n3) { std::memcpy(dst, seg3, n3); }
    s->owner.store(0, std::memory_order_relaxed);
    s->seq.store(pos + 1, std::memory_order_release);
    h->pushed.fetch_add(1, std::memory_order_relaxed);
    return 1;
}

// Pop up to max_records into dst (records packed dst_stride bytes apart).
// Returns the number popped (0 when empty).
int pda_ring_pop_many(void* base, void* dst, uint32_t dst_stride, uint32_t max_records) {
    Header* h = hdr(base);
    uint32_t got = 0;
    char* out = reinterpret_cast<char*>(dst);
    while (got < max_records) {
        uint64_t pos = h->tail.load(std::memory_order_relaxed);
        Slot* s;
        bool empty = false;
        for (;;) {
            s = slot_at(base, pos);
            uint64_t seq = s->seq.load(std::memory_order_acquire);
            int64_t dif = (int64_t)(seq - (pos + 1));
            if (dif == 0) {
                if (h->tail.compare_exchange_weak(pos, pos + 1, std::memory_order_relaxed)) break;
            } else if (dif < 0) {
                // not published.  Empty ring (nobody claimed `pos`), a producer still copying, or one that died?
                if (h->head.load(std::memory_order_relaxed) > pos && abandoned(h, s, pos)) {
                    if (h->tail.compare_exchange_strong(pos, pos + 1, std::memory_order_relaxed)) {
                        s->owner.store(0, std::memory_order_relaxed);
                        s->seq.store(pos + h->num_slots, std::memory_order_release);
                        h->skipped.fetch_add(1, std::memory_order_relaxed);
                    }
                    pos = h->tail.load(std::memory_order_relaxed);
                    continue;
                }
                empty = true;
                break;
            } else {
                pos = h->tail.load(std::memory_order_relaxed);
            }
        }
        if (empty) break;
        std::memcpy(out + (size_t)got * dst_stride, payload_of(s), h->payload_bytes);
        s->seq.store(pos + h->num_slots, std::memory_order_release);
        ++got;
    }
    if (got) h->popped.fetch_add(got, std::memory_order_relaxed);
    return (int)got;
}

uint64_t pda_ring_pushed(void* base) { return hdr(base)->pushed.load(std::memory_order_relaxed); }
uint64_t pda_ring_popped(void* base) { return hdr(base)->popped.load(std::memory_order_relaxed); }
uint64_t pda_ring_full_events(void* base) { return hdr(base)->full_events.load(std::memory_order_relaxed); }
uint64_t pda_ring_skipped(void* base) { return hdr(base)->skipped.load(std::memory_order_relaxed); }
void pda_ring_set_stuck_ms(void* base, uint32_t ms) { hdr(base)->stuck_ms = ms; }
// Test hook: claim the next slot as this process and never publish it (what a producer killed mid-copy leaves behind).
// Returns the claimed position, or -1 if the ring is full.
int64_t pda_ring_claim_and_abandon(void* base) {
    Header* h = hdr(base);
    uint64_t pos = h->head.load(std::memory_order_relaxed);
    for (;;) {
        Slot* s = slot_at(base, pos);
        int64_t dif = (int64_t)(s->seq.load(std::memory_order_acquire) - pos);
        if (dif == 0) {
            if (h->head.compare_exchange_weak(pos, pos + 1, std::memory_order_relaxed)) {
                s->owner.store((uint32_t)getpid(), std::memory_order_relaxed);
                return (int64_t)pos;
            }
        } else if (dif < 0) {
            return -1;
        } else {
            pos = h->head.load(std::memory_order_relaxed);
        }
    }
}
uint64_t pda_ring_backlog(void* base) {
    Header* h = hdr(base);
    uint64_t hd = h->head.load(std::memory_order_relaxed), tl = h->tail.load(std::memory_order_relaxed);
    return hd >= tl ? hd - tl : 0;
}

// ---- shared 64-bit counters / seqlocked mailboxes (actor <-> inference server) ----
uint64_t pda_atomic_load(void* p) {
    return reinterpret_cast<std::atomic<uint64_t>*>(p)->load(std::memory_order_acquire);
}
void pda_atomic_store(void* p, uint64_t v) {
    reinterpret_cast<std::atomic<uint64_t>*>(p)->store(v, std::memory_order_release);
}
uint64_t pda_atomic_add(void* p, uint64_t v) {
    return reinterpret_cast<std::atomic<uint64_t>*>(p)->fetch_add(v, std::memory_order_acq_rel);
}
// Spin (with pause) until *p >= target or `spins` iterations elapse.  Returns the
// last value seen; lets a Python caller wait without a 50 us sleep granularity.
uint64_t pda_atomic_wait_ge(void* p, uint64_t target, uint64_t spins) {
    auto* a = reinterpret_cast<std::atomic<uint64_t>*>(p);
    uint64_t v = a->load(std::memory_order_acquire);
    for (uint64_t i = 0; i < spins && v < target; ++i) {
#if defined(__x86_64__)
        __builtin_ia32_pause();
#endif
        v = a->load(std::memory_order_acquire);
    }
    return v;
}

// ---- ingestion pump ----------------------------------------------------------------------
// A host thread owned by the learner process that moves records ring -> pinned staging
// buffers, so the learner's own thread never spends time in memcpy (at 90k transitions/s of
// Atari frames that is ~5 GB/s) and only enqueues the H2D copy + unpack kernel of a buffer
// that is already full.  Buffers cycle FREE -> (pump fills) -> READY -> (learner enqueues the
// DMA) -> BUSY -> (learner saw the copy's event complete) -> FREE, strictly in order.
}  // extern "C"

namespace {

struct IngestPump {
    void* ring;
    int nbuf;
    uint32_t chunk, stride;
    std::vector<char*> bufs;
    std::vector<std::atomic<int>> state;      // 0 free, 1 ready, 2 busy (DMA in flight)
    std::vector<int> count;
    std::atomic<bool> stop{false};
    std::atomic<uint64_t> moved{0}, chunks{0};
    int fill = 0;       // pump thread's cursor
    int take = 0;       // learner thread's cursor
    uint32_t linger_us;
    std::thread thread;
    IngestPump(int n) : state(n), count(n, 0) {}
};

void pump_loop(IngestPump* P) {
    using clock = std::chrono::steady_clock;
    while (!P->stop.load(std::memory_order_acquire)) {
        int i = P->fill;
        if (P->state[i].load(std::memory_order_acquire) != 0) {
            std::this_thread::sleep_for(std::chrono::microseconds(50));
            continue;
        }
        uint32_t got = 0;
        clock::time_point first;
        while (got < P->chunk && !P->stop.load(std::memory_order_relaxed)) {
            int n = pda_ring_pop_many(P->ring, P->bufs[i] + (size_t)got * P->stride, P->stride, P->chunk - got);
            if (n > 0) {
                if (got == 0) first = clock::now();
                got += (uint32_t)n;
                continue;
            }
            if (got > 0 && std::chrono::duration_cast<std::chrono::microseconds>(clock::now() - first).count()
                               >= (long)P->linger_us)
                break;          // partial chunk has waited long enough: hand it over
            std::this_thread::sleep_for(std::chrono::microseconds(got ? 20 : 100));
        }
        if (got == 0) continue;
        P->count[i] = (int)got;
        P->state[i].store(1, std::memory_order_release);
        P->moved.fetch_add(got, std::memory_order_relaxed);
        P->chunks.fetch_add(1, std::memory_order_relaxed);
        P->fill = (i + 1) % P->nbuf;
    }
}

}  // namespace

extern "C" {

void* pda_pump_create(void* ring, int nbuf, void* const* bufs, uint32_t chunk_records, uint32_t record_stride,
                      uint32_t linger_us) {
    if (nbuf <= 0 || !pda_ring_valid(ring)) return nullptr;
    IngestPump* P = new IngestPump(nbuf);
    P->ring = ring;
    P->nbuf = nbuf;
    P->chunk = chunk_records;
    P->stride = record_stride;
    P->linger_us = linger_us;
    for (int i = 0; i < nbuf; ++i) {
        P->bufs.push_back(static_cast<char*>(bufs[i]));
        P->state[i].store(0);
    }
    P->thread = std::thread(pump_loop, P);
    return P;
}
// Learner: next full buffer in order.  Returns its index (count in *n) and marks it BUSY, or -1.
int pda_pump_take(void* h, int* n) {
    IngestPump* P = static_cast<IngestPump*>(h);
    int i = P->take;
    if (P->state[i].load(std::memory_order_acquire) != 1) return -1;
    *n = P->count[i];
    P->state[i].store(2, std::memory_order_relaxed);
    P->take = (i + 1) % P->nbuf;
    return i;
}
// Learner: the DMA out of buffer i has completed.
void pda_pump_release(void* h, int i) {
    static_cast<IngestPump*>(h)->state[i].store(0, std::memory_order_release);
}
uint64_t pda_pump_moved(void* h) { return static_cast<IngestPump*>(h)->moved.load(std::memory_order_relaxed); }
void pda_pump_destroy(void* h) {
    IngestPump* P = static_cast<IngestPump*>(h);
    P->stop.store(true, std::memory_order_release);
    if (P->thread.joinable()) P->thread.join();
    delete P;
}

// ---- n-step transition assembly (kernel K11, host form) ------------------------------------
// The reference assembles n-step transitions in Python deques inside every actor
// (dqn_actor.py:55-61,110-163).  Here one C object per env keeps the window of the last
// nstep+2 states / nstep+1 (action, reward, terminal, q, max q) and writes finished
// transitions straight into the ingestion ring as packed records (scatter-gather from the
// window: no intermediate arrays, no Python per transition).  Bookkeeping is the reference's:
//   * >= 3 states buffered: every step emits the LAGGED transition
//       (s[0], a[0], sum_{i < len(r)-1} gamma^i r_i, gamma^(len(s)-2), s[-2], t[-2])
//   * episode end: exactly one more, reaching the true last state, from s[1] if the window is
//     full, else from s[0]
//   * PER initial priority |R + gamma^k max_a Q(s_k) - Q(s0, a0)| from the actor's own network
}  // extern "C"

namespace {

struct NStep {
    int nstep; double gamma; bool per;
    uint32_t state_bytes, state_stride; int A, T; uint32_t record_bytes;
    int cap_s, cap_o;                    // window capacities: nstep + 2 states, nstep + 1 others
    std::vector<char> pool;              // cap_s state slots of state_stride bytes (pad stays zero)
    std::vector<int> s_slot;             // window order -> pool slot
    int n_s = 0, n_o = 0;
    std::vector<float> actions;          // [cap_o][A]
    std::vector<double> rewards, qs, maxqs;
    std::vector<float> terms;
    std::vector<char> has_q;
    std::vector<float> tail;
    uint64_t emitted = 0;
    struct RunPacker* run = nullptr;     // frame-deduplicated records (frame_runs.h); null = one full record per transition
};

// The open run of one env: planes appended in stream order, the window states as offsets into them.
struct RunPacker {
    uint32_t plane_bytes, plane_stride, trans_stride, tail_floats;
    int C, plane_cap, trans_cap;
    uint64_t planes_off;
    std::vector<char> rec;               // header | transition table | planes
    int np = 0, nt = 0;
    std::vector<int> s_off;              // window order -> first plane of that state in the open run
    int placed = 0;                      // leading window states that are in the open run
    uint64_t opened_ms = 0; uint32_t max_age_ms = 250;
    uint64_t runs = 0, planes = 0, transitions = 0;
};

char* ns_state(NStep* N, int i) { return N->pool.data() + (size_t)N->s_slot[i] * N->state_stride; }

void ns_append_state(NStep* N, const void* st) {
    int slot;
    if (N->n_s == N->cap_s) {            // drop the oldest, reuse its slot
        slot = N->s_slot[0];
        for (int i = 1; i < N->n_s; ++i) N->s_slot[i - 1] = N->s_slot[i];
        N->n_s -= 1;
        if (N->run) {
            for (int i = 1; i < N->cap_s; ++i) N->run->s_off[i - 1] = N->run->s_off[i];
            if (N->run->placed > 0) N->run->placed -= 1;
        }
    } else {
        // free slot = one not referenced by the window
        std::vector<char> used(N->cap_s, 0);
        for (int i = 0; i < N->n_s; ++i) used[N->s_slot[i]] = 1;
        slot = 0;
        while (used[slot]) ++slot;
    }
    std::memcpy(N->pool.data() + (size_t)slot * N->state_stride, st, N->state_bytes);
    N->s_slot[N->n_s++] = slot;
}

int ring_push_wait(void* ring, const void* s0, uint32_t n0, const void* s1, uint32_t n1, const void* s2, uint32_t n2,
                   uint32_t timeout_ms) {
    uint32_t waited_us = 0, delay = 10;
    for (;;) {
        int rc = pda_ring_push(ring, s0, n0, s1, n1, s2, n2, nullptr, 0);
        if (rc == 1) return 1;
        if (rc < 0) return -2;
        if (waited_us / 1000 >= timeout_ms) return -1;      // ring stayed full: the learner is not draining
        std::this_thread::sleep_for(std::chrono::microseconds(delay));
        waited_us += delay;
        if (delay < 2000) delay *= 2;
    }
}

// ---- frame-deduplicated runs (frame_runs.h) ----
// Send the open run (if it holds a transition) and start an empty one; the window states are re-placed by run_sync.
int run_ship(NStep* N, void* ring, uint32_t timeout_ms) {
    RunPacker* R = N->run;
    int rc = 1;
    if (R->nt > 0) {
        PdaRunHeader* h = reinterpret_cast<PdaRunHeader*>(R->rec.data());
        h->magic = PDA_RUN_MAGIC; h->num_planes = R->np; h->num_transitions = R->nt; h->window = R->C;
        rc = ring_push_wait(ring, R->rec.data(), (uint32_t)R->planes_off, R->rec.data() + R->planes_off,
                            (uint32_t)R->np * R->plane_stride, nullptr, 0, timeout_ms);
        if (rc == 1) { R->runs += 1; R->planes += (uint64_t)R->np; R->transitions += (uint64_t)R->nt; }
    }
    R->np = 0; R->nt = 0; R->placed = 0;
    return rc;
}
// Planes that window state i adds behind the states before it: 1 when it is the previous state slid by one frame
// (and that one is the newest thing in the run), else all C.
int run_need(NStep* N, int i) {
    RunPacker* R = N->run;
    if (i > 0 && R->s_off[i - 1] + R->C == R->np &&
        std::memcmp(ns_state(N, i - 1) + R->plane_bytes, ns_state(N, i), (size_t)(R->C - 1) * R->plane_bytes) == 0)
        return 1;
    return R->C;
}
// Place every window state that is not in the open run yet; a run without room is sent first (a fresh run always
// holds the whole window: plane_cap >= 2 * (C + nstep + 2)).
int run_sync(NStep* N, void* ring, uint32_t timeout_ms) {
    RunPacker* R = N->run;
    while (R->placed < N->n_s) {
        int i = R->placed;
        int need = run_need(N, i);
        if (R->np + need > R->plane_cap) {
            int rc = run_ship(N, ring, timeout_ms);
            if (rc < 0) return rc;
            continue;                       // placed == 0 again
        }
        if (R->np == 0 && R->nt == 0) R->opened_ms = now_ms();
        char* dst = R->rec.data() + R->planes_off + (size_t)R->np * R->plane_stride;
        const char* src = ns_state(N, i);
        if (need == 1) {
            std::memcpy(dst, src + (size_t)(R->C - 1) * R->plane_bytes, R->plane_bytes);
            R->s_off[i] = R->s_off[i - 1] + 1;
        } else {
            for (int c = 0; c < R->C; ++c)
                std::memcpy(dst + (size_t)c * R->plane_stride, src + (size_t)c * R->plane_bytes, R->plane_bytes);
            R->s_off[i] = R->np;
        }
        R->np += need;
        R->placed += 1;
    }
    return 1;
}
int run_emit(NStep* N, void* ring, int first, int last_state, const float* tail, uint32_t timeout_ms) {
    RunPacker* R = N->run;
    if (R->nt >= R->trans_cap) {
        int rc = run_ship(N, ring, timeout_ms);
        if (rc < 0) return rc;
    }
    int rc = run_sync(N, ring, timeout_ms);
    if (rc < 0) return rc;
    char* e = R->rec.data() + sizeof(PdaRunHeader) + (size_t)R->nt * R->trans_stride;
    std::memcpy(e, tail, 4u * R->tail_floats);
    int32_t offs[2] = {R->s_off[first], R->s_off[last_state]};
    std::memcpy(e + 4u * R->tail_floats, offs, 8);
    R->nt += 1;
    N->emitted += 1;
    return 1;
}
// after a step: a run that has been open for max_age_ms goes out even if it is not full (fresh data for the learner)
int run_age_check(NStep* N, void* ring, uint32_t timeout_ms) {
    RunPacker* R = N->run;
    if (R->nt > 0 && now_ms() - R->opened_ms >= R->max_age_ms) return run_ship(N, ring, timeout_ms);
    return 1;
}

int ns_emit(NStep* N, void* ring, int first, int last_state, int term_idx, double ret, double gamma_sn,
            double prio, uint32_t timeout_ms) {
    float* t = N->tail.data();
    int A = N->A, T = N->T;
    for (int j = 0; j < A; ++j) t[j] = N->actions[(size_t)first * A + j];
    for (int j = 0; j < T; ++j) {
        t[A + j] = (float)ret;
        t[A + T + j] = (float)gamma_sn;
        t[A + 2 * T + j] = N->terms[term_idx];
    }
    t[A + 3 * T] = (float)prio;
    if (N->run) return run_emit(N, ring, first, last_state, t, timeout_ms);
    uint32_t tail_bytes = N->record_bytes - 2 * N->state_stride;
    int rc = ring_push_wait(ring, ns_state(N, first), N->state_stride, ns_state(N, last_state), N->state_stride,
                            t, tail_bytes, timeout_ms);
    if (rc == 1) N->emitted += 1;
    return rc;
}

}  // namespace

extern "C" {

void* pda_nstep_create(int nstep, double gamma, int enable_per, uint32_t state_bytes, uint32_t state_stride,
                       int action_floats, int num_tasks, uint32_t record_bytes) {
    if (nstep < 1 || state_stride < state_bytes) return nullptr;
    uint32_t tail_floats = (uint32_t)(action_floats + 3 * num_tasks + 1);
    if (record_bytes < 2 * state_stride + 4 * tail_floats) return nullptr;
    NStep* N = new NStep();
    N->nstep = nstep; N->gamma = gamma; N->per = enable_per != 0;
    N->state_bytes = state_bytes; N->state_stride = state_stride; N->A = action_floats; N->T = num_tasks;
    N->record_bytes = record_bytes;
    N->cap_s = nstep + 2; N->cap_o = nstep + 1;
    N->pool.assign((size_t)N->cap_s * state_stride, 0);
    N->s_slot.assign(N->cap_s, 0);
    N->actions.assign((size_t)N->cap_o * action_floats, 0.f);
    N->rewards.assign(N->cap_o, 0.0); N->qs.assign(N->cap_o, 0.0); N->maxqs.assign(N->cap_o, 0.0);
    N->terms.assign(N->cap_o, 0.f); N->has_q.assign(N->cap_o, 0);
    N->tail.assign((record_bytes - 2 * state_stride) / 4 + 4, 0.f);
    return N;
}
// The same assembler writing frame-deduplicated run records (frame_runs.h).  window: planes per state (C of a
// [C, H, W] frame stack); the ring's payload must be pda_run_record_bytes(...).
void* pda_nstep_create_dedup(int nstep, double gamma, int enable_per, uint32_t state_bytes, int window,
                             int action_floats, int num_tasks, int plane_cap, int trans_cap, uint32_t max_age_ms) {
    if (window < 1 || state_bytes % (uint32_t)window != 0) return nullptr;
    uint32_t plane_bytes = state_bytes / (uint32_t)window;
    if (plane_bytes % 16 != 0) return nullptr;          // a state is read as `window` planes at plane stride
    if (plane_cap < 2 * (window + nstep + 2) || trans_cap < 2) return nullptr;
    uint32_t tail_floats = (uint32_t)(action_floats + 3 * num_tasks + 1);
    uint32_t state_stride = (state_bytes + 15u) / 16u * 16u;
    NStep* N = static_cast<NStep*>(pda_nstep_create(nstep, gamma, enable_per, state_bytes, state_stride, action_floats,
                                                    num_tasks, 2 * state_stride + 4 * tail_floats + 16));
    if (!N) return nullptr;
    RunPacker* R = new RunPacker();
    R->plane_bytes = plane_bytes; R->plane_stride = pda_run_plane_stride(plane_bytes);
    R->tail_floats = tail_floats; R->trans_stride = pda_run_trans_stride(tail_floats);
    R->C = window; R->plane_cap = plane_cap; R->trans_cap = trans_cap;
    R->planes_off = pda_run_planes_offset(tail_floats, (uint32_t)trans_cap);
    R->rec.assign(pda_run_record_bytes(tail_floats, (uint32_t)trans_cap, plane_bytes, (uint32_t)plane_cap), 0);
    R->s_off.assign(N->cap_s, 0);
    R->max_age_ms = max_age_ms;
    N->run = R;
    return N;
}
uint64_t pda_run_record_size(int action_floats, int num_tasks, int trans_cap, uint32_t plane_bytes, int plane_cap) {
    return pda_run_record_bytes((uint32_t)(action_floats + 3 * num_tasks + 1), (uint32_t)trans_cap, plane_bytes,
                                (uint32_t)plane_cap);
}
// Send the open run now (tests, shutdown).  Returns 1, or -1 / -2 as pda_nstep_push.
int pda_nstep_ship(void* h, void* ring, uint32_t timeout_ms) {
    NStep* N = static_cast<NStep*>(h);
    return N->run ? run_ship(N, ring, timeout_ms) : 1;
}
// out[0..2] = runs, planes, transitions sent so far
void pda_nstep_run_stats(void* h, uint64_t* out) {
    NStep* N = static_cast<NStep*>(h);
    out[0] = out[1] = out[2] = 0;
    if (N->run) { out[0] = N->run->runs; out[1] = N->run->planes; out[2] = N->run->transitions; }
}
void pda_nstep_destroy(void* h) {
    NStep* N = static_cast<NStep*>(h);
    delete N->run;
    delete N;
}
uint64_t pda_nstep_emitted(void* h) { return static_cast<NStep*>(h)->emitted; }

void pda_nstep_reset(void* h, const void* state) {
    NStep* N = static_cast<NStep*>(h);
    N->n_s = 0; N->n_o = 0;
    if (N->run) N->run->placed = 0;         // (what the open run already holds stays; the new episode's states follow)
    ns_append_state(N, state);
}

// One env step.  Returns the number of transitions written to the ring (0 or 1), -1 if the ring
// stayed full for timeout_ms, -2 on a size mismatch.
int pda_nstep_push(void* h, void* ring, const float* action, double reward, const void* state1, int terminal,
                   double q, double maxq, int has_q, uint32_t timeout_ms) {
    NStep* N = static_cast<NStep*>(h);
    ns_append_state(N, state1);
    if (N->n_o == N->cap_o) {            // slide the (action, reward, ...) window
        int A = N->A;
        for (int i = 1; i < N->n_o; ++i) {
            for (int j = 0; j < A; ++j) N->actions[(size_t)(i - 1) * A + j] = N->actions[(size_t)i * A + j];
            N->rewards[i - 1] = N->rewards[i]; N->qs[i - 1] = N->qs[i]; N->maxqs[i - 1] = N->maxqs[i];
            N->terms[i - 1] = N->terms[i]; N->has_q[i - 1] = N->has_q[i];
        }
        N->n_o -= 1;
    }
    int k = N->n_o++;
    for (int j = 0; j < N->A; ++j) N->actions[(size_t)k * N->A + j] = action[j];
    N->rewards[k] = reward; N->terms[k] = terminal ? 1.f : 0.f;
    N->qs[k] = q; N->maxqs[k] = maxq; N->has_q[k] = (char)(has_q != 0);
    if (N->n_s < 3) return 0;
    double ret = 0.0, g = 1.0;
    for (int i = 0; i < N->n_o - 1; ++i) { ret += N->rewards[i] * g; g *= N->gamma; }
    double gamma_sn = 1.0;
    for (int i = 0; i < N->n_s - 2; ++i) gamma_sn *= N->gamma;
    double prio = 0.0;
    if (N->per && N->has_q[N->n_o - 1] && N->has_q[0]) {
        prio = ret + gamma_sn * N->maxqs[N->n_o - 1] - N->qs[0];
        if (prio < 0) prio = -prio;
    }
    int rc = ns_emit(N, ring, 0, N->n_s - 2, N->n_o - 2, ret, gamma_sn, prio, timeout_ms);
    if (rc > 0 && N->run) { int a = run_age_check(N, ring, timeout_ms); if (a < 0) return a; }
    return rc;
}

// Episode ended: emit the one closing transition (see the header comment).
int pda_nstep_flush(void* h, void* ring, double maxq_last, int has_maxq, uint32_t timeout_ms) {
    NStep* N = static_cast<NStep*>(h);
    if (N->n_o == 0) return 0;
    int first;
    double ret = 0.0, g = 1.0, gamma_sn = 1.0;
    if (N->n_s >= N->nstep + 2) {
        first = 1;
        for (int i = 1; i < N->n_o; ++i) { ret += N->rewards[i] * g; g *= N->gamma; }
        for (int i = 0; i < N->n_s - 2; ++i) gamma_sn *= N->gamma;
    } else {
        first = 0;
        for (int i = 0; i < N->n_o; ++i) { ret += N->rewards[i] * g; g *= N->gamma; }
        for (int i = 0; i < N->n_s - 1; ++i) gamma_sn *= N->gamma;
    }
    double prio = 0.0;
    if (N->per && has_maxq && N->has_q[first]) {
        prio = ret + gamma_sn * maxq_last - N->qs[first];
        if (prio < 0) prio = -prio;
    }
    int rc = ns_emit(N, ring, first, N->n_s - 1, N->n_o - 1, ret, gamma_sn, prio, timeout_ms);
    if (rc > 0 && N->run) { int a = run_age_check(N, ring, timeout_ms); if (a < 0) return a; }
    return rc;
}

// ---- inference mailbox (protocol in mailbox.h) ------------------------------------------
// Actor: copy the observation into its slot, post request number `k`, wait for the answer.
// Returns the last resp value seen (>= k once answered); on a smaller value the caller checks
// for shutdown and calls pda_mailbox_wait again.
uint64_t pda_mailbox_request(void* seq_base, int slot, uint64_t k, void* obs_dst, const void* obs_src,
                             uint64_t obs_bytes, uint32_t spins, uint32_t timeout_us) {
    if (obs_bytes) std::memcpy(obs_dst, obs_src, obs_bytes);
    pda_mailbox::req_word(seq_base, slot)->store(k, std::memory_order_release);
    return pda_mailbox::wait_response(seq_base, slot, k, spins, timeout_us);
}
// Split form for actors that drive several envs: send() posts without waiting, so the server works on
// env A's request while the actor steps env B; pda_mailbox_wait collects the answer later.
void pda_mailbox_send(void* seq_base, int slot, uint64_t k, void* obs_dst, const void* obs_src, uint64_t obs_bytes) {
    if (obs_bytes) std::memcpy(obs_dst, obs_src, obs_bytes);
    pda_mailbox::req_word(seq_base, slot)->store(k, std::memory_order_release);
}
uint64_t pda_mailbox_wait(void* seq_base, int slot, uint64_t k, uint32_t spins, uint32_t timeout_us) {
    return pda_mailbox::wait_response(seq_base, slot, k, spins, timeout_us);
}
// Server (host fallback; the GPU server runs the same protocol natively in libpda_hip.so)
int pda_mailbox_collect(void* seq_base, int nslots, int* ids, uint64_t* req_seen, int max_out) {
    return pda_mailbox::collect_pending(seq_base, nslots, ids, req_seen, max_out);
}
void pda_mailbox_post(void* seq_base, const int* ids, const uint64_t* req_seen, int n) {
    for (int i = 0; i < n; ++i) pda_mailbox::post_response(seq_base, ids[i], req_seen[i]);
}

}  // extern "C"
