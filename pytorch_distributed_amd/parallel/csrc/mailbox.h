// Actor <-> inference-server mailbox protocol over shared memory.
//
// One 128-byte control block per slot (two cache lines, so the actor's word and
// the server's word never share a line):
//     +0   req   (u64)  written by the actor:   k = number of requests posted
//     +64  resp  (u64)  written by the server:  k = last request answered
//     +72  wait  (u32)  actor sets it before it sleeps on `resp`, server clears it and wakes
// An actor posts observation k with a release store to req, then waits until
// resp >= k: a short spin (the server usually answers within ~100 us), then a
// futex sleep on the low word of resp, so 256 waiting actors cost no CPU.  The
// server answers with a release store to resp and one futex wake if (and only
// if) the slot's wait flag is up.  No locks, no syscalls on the fast path.
//
// Used by libpda_host.so (actor side, pure host code) and by the native
// inference loop inside libpda_hip.so (server side).
#pragma once
#include <atomic>
#include <cerrno>
#include <climits>
#include <cstdint>
#include <ctime>
#include <linux/futex.h>
#include <sys/syscall.h>
#include <unistd.h>

namespace pda_mailbox {

constexpr int kSlotBytes = 128;
constexpr int kRespOff = 64;
constexpr int kWaitOff = 72;

inline std::atomic<uint64_t>* req_word(void* seq_base, int slot) {
    return reinterpret_cast<std::atomic<uint64_t>*>(static_cast<char*>(seq_base) + (long)slot * kSlotBytes);
}
inline std::atomic<uint64_t>* resp_word(void* seq_base, int slot) {
    return reinterpret_cast<std::atomic<uint64_t>*>(static_cast<char*>(seq_base) + (long)slot * kSlotBytes + kRespOff);
}
inline std::atomic<uint32_t>* wait_word(void* seq_base, int slot) {
    return reinterpret_cast<std::atomic<uint32_t>*>(static_cast<char*>(seq_base) + (long)slot * kSlotBytes + kWaitOff);
}

inline long futex_call(void* addr, int op, uint32_t val, const timespec* ts) {
    return syscall(SYS_futex, addr, op, val, ts, nullptr, 0);
}

inline void cpu_relax() {
#if defined(__x86_64__)
    __builtin_ia32_pause();
#endif
}

// Actor: wait until resp >= target.  Returns the last value seen (>= target on success,
// smaller when timeout_us elapsed; the caller then checks for shutdown and calls again).
inline uint64_t wait_response(void* seq_base, int slot, uint64_t target, uint32_t spins, uint32_t timeout_us) {
    std::atomic<uint64_t>* resp = resp_word(seq_base, slot);
    uint64_t v = resp->load(std::memory_order_acquire);
    for (uint32_t i = 0; i < spins && v < target; ++i) {
        cpu_relax();
        v = resp->load(std::memory_order_acquire);
    }
    if (v >= target) return v;
    std::atomic<uint32_t>* flag = wait_word(seq_base, slot);
    flag->store(1, std::memory_order_seq_cst);
    v = resp->load(std::memory_order_seq_cst);      // re-check after raising the flag (no lost wake-up)
    if (v < target) {
        timespec ts;
        ts.tv_sec = timeout_us / 1000000u;
        ts.tv_nsec = (long)(timeout_us % 1000000u) * 1000L;
        // sleeps only while the low word still holds the value we saw
        futex_call(resp, FUTEX_WAIT, (uint32_t)v, &ts);
        v = resp->load(std::memory_order_acquire);
    }
    flag->store(0, std::memory_order_relaxed);
    return v;
}

// Server: publish the answer to request k of `slot` and wake its actor if it sleeps.
inline void post_response(void* seq_base, int slot, uint64_t k) {
    std::atomic<uint64_t>* resp = resp_word(seq_base, slot);
    resp->store(k, std::memory_order_seq_cst);
    std::atomic<uint32_t>* flag = wait_word(seq_base, slot);
    if (flag->load(std::memory_order_seq_cst)) futex_call(resp, FUTEX_WAKE, 1, nullptr);
}

// Server: collect slots with req > resp.  Returns how many; ids/req_seen get the slot
// numbers and the request numbers that will be answered.
inline int collect_pending(void* seq_base, int nslots, int* ids, uint64_t* req_seen, int max_out) {
    int n = 0;
    for (int s = 0; s < nslots && n < max_out; ++s) {
        uint64_t r = req_word(seq_base, s)->load(std::memory_order_acquire);
        uint64_t a = resp_word(seq_base, s)->load(std::memory_order_relaxed);
        if (r > a) {
            ids[n] = s;
            req_seen[n] = r;
            ++n;
        }
    }
    return n;
}

}  // namespace pda_mailbox
