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
    th.clear();

    // 4) frame-deduplicated runs (frame_runs.h): 4 env streams, each its own n-step assembler with a run packer, into one
    //    ring; the consumer checks every run record -- counts within capacity, both windows of every transition inside
    //    the run's planes, and every plane of a state equal to the byte pattern its stream gave that frame
    {
        const int streams = 4, C = 4, nstep = 3, plane_cap = 24, steps = 2500;
        const uint32_t plane_bytes = 64, state_bytes = C * plane_bytes, tail_floats = 1 + 3 + 1;
        const uint64_t rec_bytes = pda_run_record_size(1, 1, plane_cap, plane_bytes, plane_cap);
        const uint32_t rslots = 16;
        uint64_t rbytes = pda_ring_bytes(rslots, (uint32_t)rec_bytes);
        void* rring = aligned_alloc(64, (rbytes + 63) / 64 * 64);
        if (pda_ring_init(rring, rbytes, rslots, (uint32_t)rec_bytes) != 0) return 8;
        std::atomic<int> done{0};
        std::atomic<long> sent{0}, got{0}, badrun{0};
        for (int s = 0; s < streams; ++s)
            th.emplace_back([&, s] {
                void* h = pda_nstep_create_dedup(nstep, 0.99, 1, state_bytes, C, 1, 1, plane_cap, plane_cap, 5);
                if (!h) { badrun.fetch_add(1000); done.fetch_add(1); return; }
                std::vector<uint8_t> st(state_bytes);
                uint32_t frame = 0;         // frame f of stream s is filled with the byte (s * 64 + f) & 255
                auto fill = [&](bool reset) {
                    if (reset) { for (int c = 0; c < C; ++c) std::memset(&st[c * plane_bytes], (s * 64 + frame) & 255, plane_bytes); }
                    else {
                        std::memmove(&st[0], &st[plane_bytes], (C - 1) * plane_bytes);
                        std::memset(&st[(C - 1) * plane_bytes], (s * 64 + frame) & 255, plane_bytes);
                    }
                    ++frame;
                };
                fill(true);
                pda_nstep_reset(h, st.data());
                float act = 1.f;
                for (int t = 1; t <= steps; ++t) {
                    const bool term = (t % (17 + s)) == 0;
                    fill(false);
                    int rc = pda_nstep_push(h, rring, &act, 0.5, st.data(), term ? 1 : 0, 0.1, 0.2, 1, 20000);
                    if (rc < 0) { badrun.fetch_add(1000); break; }
                    sent.fetch_add(rc);
                    if (term) {
                        rc = pda_nstep_flush(h, rring, 0.3, 1, 20000);
                        if (rc < 0) { badrun.fetch_add(1000); break; }
                        sent.fetch_add(rc);
                        fill(true);
                        pda_nstep_reset(h, st.data());
                    }
                }
                pda_nstep_ship(h, rring, 20000);
                pda_nstep_destroy(h);
                done.fetch_add(1);
            });
        std::thread consumer([&] {
            std::vector<uint8_t> rec(rec_bytes);
            const uint64_t poff = pda_run_planes_offset(tail_floats, plane_cap);
            const uint32_t tstride = pda_run_trans_stride(tail_floats);
            for (;;) {
                int n = pda_ring_pop_many(rring, rec.data(), (uint32_t)rec_bytes, 1);
                if (n == 0) {
                    if (done.load() == streams && pda_ring_backlog(rring) == 0) break;
                    std::this_thread::yield();
                    continue;
                }
                const PdaRunHeader* h = reinterpret_cast<const PdaRunHeader*>(rec.data());
                if (h->magic != PDA_RUN_MAGIC || h->num_planes < 1 || h->num_planes > plane_cap ||
                    h->num_transitions < 1 || h->num_transitions > plane_cap || h->window != C) { badrun.fetch_add(1); continue; }
                for (int j = 0; j < h->num_transitions; ++j) {
                    const uint8_t* e = rec.data() + sizeof(PdaRunHeader) + (size_t)j * tstride;
                    int32_t offs[2];
                    std::memcpy(offs, e + 4 * tail_floats, 8);
                    for (int w = 0; w < 2; ++w) {
                        if (offs[w] < 0 || offs[w] + C > h->num_planes) { badrun.fetch_add(1); continue; }
                        // a state's planes are consecutive frames of one stream, or one frame repeated (episode start)
                        const uint8_t* p0 = rec.data() + poff + (size_t)offs[w] * plane_bytes;
                        for (int c = 1; c < C; ++c) {
                            const uint8_t a = p0[(size_t)(c - 1) * plane_bytes], b = p0[(size_t)c * plane_bytes];
                            if (b != a && b != (uint8_t)(a + 1)) badrun.fetch_add(1);
                        }
                    }
                    if (offs[1] < offs[0]) badrun.fetch_add(1);
                }
                got.fetch_add(h->num_transitions);
            }
        });
        for (auto& t : th) t.join();
        consumer.join();
        std::free(rring);
        if (badrun.load() || got.load() != sent.load() || sent.load() < streams * (steps - 200)) {
            std::fprintf(stderr, "runs: %ld bad, %ld sent, %ld received\n", badrun.load(), sent.load(), got.load());
            return 9;
        }
        std::printf("runs ok: %ld transitions in frame-deduplicated runs\n", got.load());
    }
    std::printf("ring_stress ok: %d records, %d pumped, %d mailbox round trips\n", total, pumped_total, nslots * rounds);
    free(region);
    return 0;
}
