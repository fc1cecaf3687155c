// Thread-sanitizer stress of the host runtime (parallel/csrc/shm_ring.cpp): many producers and two
// consumers on the MPMC ring, the ingestion pump, and actor/server threads on the mailbox protocol.
// Built by tests/test_race.py with -fsanitize=thread; exits non-zero on lost/duplicated records.
#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "../../pytorch_distributed_amd/parallel/csrc/shm_ring.cpp"

int main() {
    const uint32_t slots = 64, payload = 256;
    const int producers = 6, per_producer = 20000;
    uint64_t bytes = pda_ring_bytes(slots, payload);
    void* region = aligned_alloc(64, (bytes + 63) / 64 * 64);
    if (pda_ring_init(region, bytes, slots, payload) != 0) return 2;

    // 1) producers -> ring -> two raw consumers
    std::vector<std::atomic<int>> seen(producers * per_producer);
    for (auto& s : seen) s.store(0);
    std::atomic<int> popped{0};
    std::vector<std::thread> th;
    for (int p = 0; p < producers; ++p)
        th.emplace_back([&, p] {
            std::vector<uint32_t> rec(payload / 4);
            for (int i = 0; i < per_producer; ++i) {
                uint32_t id = (uint32_t)(p * per_producer + i);
                for (auto& w : rec) w = id;
                while (pda_ring_push(region, rec.data(), payload, nullptr, 0, nullptr, 0, nullptr, 0) != 1)
                    std::this_thread::yield();
            }
        });
    const int total = producers * per_producer;
    for (int c = 0; c < 2; ++c)
        th.emplace_back([&] {
            std::vector<uint32_t> buf(16 * payload / 4);
            while (popped.load() < total) {
                int n = pda_ring_pop_many(region, buf.data(), payload, 16);
                for (int i = 0; i < n; ++i) {
                    uint32_t id = buf[i * payload / 4];
                    for (uint32_t w = 0; w < payload / 4; ++w)
                        if (buf[i * payload / 4 + w] != id) { std::fprintf(stderr, "torn record\n"); std::exit(3); }
                    seen[id].fetch_add(1);
                }
                popped.fetch_add(n);
                if (n == 0) std::this_thread::yield();
            }
        });
    for (auto& t : th) t.join();
    th.clear();
    for (int i = 0; i < total; ++i)
        if (seen[i].load() != 1) { std::fprintf(stderr, "record %d seen %d times\n", i, seen[i].load()); return 4; }

    // 2) pump thread: ring -> staging buffers -> "learner" takes and releases in order
    const int nbuf = 3, chunk = 32;
    std::vector<std::vector<char>> stage(nbuf, std::vector<char>(chunk * payload));
    void* ptrs[nbuf];
    for (int i = 0; i < nbuf; ++i) ptrs[i] = stage[i].data();
    void* pump = pda_pump_create(region, nbuf, ptrs, chunk, payload, 200);
    if (!pump) return 5;
    const int pumped_total = 30000;
    std::thread prod([&] {
        std::vector<uint32_t> rec(payload / 4);
        for (int i = 0; i < pumped_total; ++i) {
            for (auto& w : rec) w = (uint32_t)i;
            while (pda_ring_push(region, rec.data(), payload, nullptr, 0, nullptr, 0, nullptr, 0) != 1)
                std::this_thread::yield();
        }
    });
    int got = 0;
    uint64_t sum = 0;
    while (got < pumped_total) {
        int n = 0;
        int i = pda_pump_take(pump, &n);
        if (i < 0) { std::this_thread::yield(); continue; }
        for (int k = 0; k < n; ++k) sum += *reinterpret_cast<uint32_t*>(stage[i].data() + (size_t)k * payload);
        got += n;
        pda_pump_release(pump, i);
    }
    prod.join();
    pda_pump_destroy(pump);
    if (sum != (uint64_t)pumped_total * (pumped_total - 1) / 2) { std::fprintf(stderr, "pump lost records\n"); return 6; }

    // 3) mailbox: 8 actor threads, one server thread echoing obs[0] + 1
    const int nslots = 8, rounds = 3000;
    std::vector<char> seq(nslots * pda_mailbox::kSlotBytes, 0);
    std::vector<std::atomic<uint32_t>> obs(nslots), out(nslots);
    std::atomic<bool> stop{false};
    std::thread server([&] {
        int ids[nslots];
        uint64_t reqs[nslots];
        while (!stop.load()) {
            int n = pda_mailbox_collect(seq.data(), nslots, ids, reqs, nslots);
            for (int i = 0; i < n; ++i) out[ids[i]].store(obs[ids[i]].load(std::memory_order_relaxed) + 1,
                                                          std::memory_order_relaxed);
            pda_mailbox_post(seq.data(), ids, reqs, n);
            if (n == 0) std::this_thread::yield();
        }
    });
    std::atomic<int> bad{0};
    for (int s = 0; s < nslots; ++s)
        th.emplace_back([&, s] {
            for (int k = 1; k <= rounds; ++k) {
                obs[s].store((uint32_t)(s * 100000 + k), std::memory_order_relaxed);
                uint64_t v = pda_mailbox_request(seq.data(), s, (uint64_t)k, nullptr, nullptr, 0, 50, 500);
                while (v < (uint64_t)k) v = pda_mailbox_wait(seq.data(), s, (uint64_t)k, 0, 500);
                if (out[s].load(std::memory_order_relaxed) != (uint32_t)(s * 100000 + k + 1)) bad.fetch_add(1);
            }
        });
    for (auto& t : th) t.join();
    stop.store(true);
    server.join();
    if (bad.load()) { std::fprintf(stderr, "mailbox: %d wrong answers\n", bad.load()); return 7; }
    std::printf("ring_stress ok: %d records, %d pumped, %d mailbox round trips\n", total, pumped_total, nslots * rounds);
    free(region);
    return 0;
}
