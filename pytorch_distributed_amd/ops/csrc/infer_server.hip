// Native batched-inference loop: the GPU side of the actor <-> server mailbox.
//
// The reference gives every actor process its own CUDA model and runs batch-1
// forwards with two .item() syncs per env step (dqn_actor.py:23-30,84;
// dqn_cnn_model.py:58-78).  Here actors are CPU env loops that post observations
// into shared-memory slots (parallel/csrc/mailbox.h); ONE host thread owned by
// this library polls the slots, DMAs the pending observations straight out of the
// (host-registered) shared-memory block, replays a pre-captured hipGraph of the
// policy network for the batch size at hand, and posts the answers back -- no
// Python, no allocator, no per-request launches in the loop:
//
//   collect pending slots -> H2D of their rows (merged runs) -> graph[bucket]:
//       conv1(u8, gathered through the slot indices in mapped host memory) -> conv2 -> conv3 -> fc1 split-K
//       -> fc2 + epsilon-greedy head writing {action, q[a], max q} into mapped host memory
//   -> stream sync -> scatter to the slots' out rows -> release-store resp (+ futex wake)
//
// Batch sizes are padded up to power-of-two buckets (duplicate rows of the first
// pending slot), one graph per bucket, captured once at creation on the loop's own
// stream.  The stream can be created with the highest priority so that, hosted inside
// the learner process, inference waves overtake the learner's queued step graphs.
//
// Two programs: kind 0 = Nature-CNN DQN (uint8 frames), kind 1 = deterministic MLP
// actor (DDPG: fp32 observation rows -> tanh MLP -> A action floats).
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "api.h"
#include "../../parallel/csrc/mailbox.h"

namespace {

struct InferConfig {
    int kind;                   // 0 cnn-dqn, 1 mlp actor
    int nslots;
    void* seq;                  // mailbox control blocks [nslots][128 B]
    void* obs;                  // mailbox observation block [nslots][obs_stride]
    long obs_stride;            // bytes
    float* out;                 // mailbox answers [nslots][out_ld]
    int out_ld;
    const float* eps;           // per-slot epsilon (kind 0), may be null
    // kind 0 geometry
    int C, H, W, HID, A;
    float inv_norm;
    // kind 1 geometry (padded pitches are multiples of 8)
    int D, Dp, H1, H1p, H2, H2p, Ap;
    const void* w[5];           // device: bf16 shadows (cnn: conv1..fc1, fc2 fp32; mlp: 3 bf16 matrices)
    const void* b[5];           // device fp32 biases
    unsigned long long seed;
    int high_priority;
    int max_bucket;             // 0: nslots rounded up
    int window_us;              // batching window of the loop thread: once a request is pending, wait up to this long
                                // for the other live slots to post theirs (0: answer whatever is pending at once)
};

struct Stats { unsigned long long batches, requests, busy_ns, max_batch, registered, idle_polls, window_ns, live_slots; };

#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
    std::fprintf(stderr, "[pda infer] %s failed: %s\n", #x, hipGetErrorString(e_)); ok = false; } } while (0)

struct Server {
    InferConfig c;
    hipStream_t stream = nullptr;
    bool registered = false;
    bool ok = true;
    std::vector<int> buckets;
    std::vector<hipGraphExec_t> graphs;
    // device
    uint8_t* d_obs = nullptr;
    bf16_t *a1 = nullptr, *a2 = nullptr, *a3 = nullptr;     // cnn activations / mlp x,h1,h2
    bf16_t* a4 = nullptr;                                   // mlp action rows
    float* h1pre = nullptr;
    // pinned + mapped host
    int* h_idx = nullptr; float* h_eps = nullptr; float* h_out = nullptr; uint8_t* h_stage = nullptr;
    unsigned long long* h_counter = nullptr;
    int* hd_idx = nullptr; float* hd_eps = nullptr; float* hd_out = nullptr;   // device views of the above
    unsigned long long* hd_counter = nullptr;
    int out_cols = 0;
    // loop
    std::thread thread;
    std::atomic<bool> stop{false};
    std::atomic<unsigned long long> batches{0}, requests{0}, busy_ns{0}, max_batch{0}, idle_polls{0}, window_ns{0};
    std::atomic<unsigned long long> live_slots{0};
    std::vector<int> ids;
    std::vector<uint64_t> req_seen;
};

int conv_out(int h, int k, int s) { return (h - k) / s + 1; }

void enqueue_program(Server& S, int n) {
    const InferConfig& c = S.c;
    hipStream_t st = S.stream;
    // The batch's slot indices, epsilons and the RNG counter are read by the kernels straight out of mapped host
    // memory (a PCIe read per workgroup, under its prologue): no staging launch in front of the network and no counter
    // launch behind it -- three kernels per batch instead of five, each of which contends with the learner's.
    if (c.kind == 0) {
        int H1 = conv_out(c.H, 8, 4), W1 = conv_out(c.W, 8, 4);
        int H2 = conv_out(H1, 4, 2), W2 = conv_out(W1, 4, 2);
        int H3 = conv_out(H2, 3, 1), W3 = conv_out(W2, 3, 1);
        int F = 64 * H3 * W3;
        const void* frames[1] = {S.d_obs};
        const void* idx[1] = {S.hd_idx};
        const void* w0[1] = {c.w[0]}; const void* b0[1] = {c.b[0]};
        const void* w1[1] = {c.w[1]}; const void* b1[1] = {c.b[1]};
        const void* w2[1] = {c.w[2]}; const void* b2[1] = {c.b[2]};
        const void* w3[1] = {c.w[3]};
        void* o1[1] = {S.a1}; void* o2[1] = {S.a2}; void* o3[1] = {S.a3};
        const void* x1[1] = {S.a1}; const void* x2[1] = {S.a2}; const void* x3[1] = {S.a3};
        void* o32[1] = {S.h1pre};
        if (pda_cnn_fwd_stack_supported(c.H, c.W, c.C)) {
            // the three conv layers in one launch, intermediates in LDS (nature_fwd_stack.hip)
            void* none[1] = {nullptr};
            pda_cnn_fwd_stack(frames, idx, nullptr, 0, w0, b0, w1, b1, w2, b2, none, none, o3, 1, n, c.H, c.W, c.C, c.inv_norm, st);
        } else {
            pda_conv_fwd_u8(frames, idx, w0, b0, o1, 1, n, c.H, c.W, c.C, 8, 8, 4, 32, c.inv_norm, ACT_RELU, st);
            pda_conv_fwd(x1, w1, b1, o2, 1, n, H1, W1, 32, 4, 4, 2, 64, ACT_RELU, st);
            pda_conv_fwd(x2, w2, b2, o3, 1, n, H2, W2, 64, 3, 3, 1, 64, ACT_RELU, st);
        }
        // fc1 split-K into the fp32 pre-activation buffer (the head zeroes it again after reading)
        int tiles = ((n + 63) / 64) * (c.HID / 64);
        int slices = 256 / (tiles > 0 ? tiles : 1);    // measured best for the learner's fc1 too (fewer fp32 atomics)
        if (slices < 1) slices = 1;
        int ks = (F + slices - 1) / slices;
        ks = (ks + 31) / 32 * 32;
        if (ks < 128) ks = 128;
        pda_linear_fwd(x3, w3, nullptr, nullptr, o32, 1, n, c.HID, F, F, F, 0, c.HID, ACT_NONE, ks, st);
        ActHeadArgs a;
        a.h1pre = S.h1pre; a.b4 = (const float*)c.b[3]; a.w5 = (const float*)c.w[4]; a.b5 = (const float*)c.b[4];
        a.eps = S.hd_eps; a.out = S.hd_out; a.out_ld = S.out_cols; a.q_out = nullptr;
        a.B = n; a.HID = c.HID; a.A = c.A; a.seed = c.seed; a.counter = S.hd_counter;    // (bumped by the host per batch)
        pda_act_head(&a, st);
    } else {
        StageJobs jobs;
        jobs.j[0] = {(const float*)S.d_obs, c.obs_stride / 4, 0, c.D, S.a1, (long)c.Dp, 0, 1.f};
        pda_mlp_stage(&jobs, 1, S.hd_idx, n, c.D, st);
        const void* x0[1] = {S.a1}; const void* x1[1] = {S.a2}; const void* x2[1] = {S.a3};
        void* o0[1] = {S.a2}; void* o1[1] = {S.a3}; void* o2[1] = {S.a4};
        const void* w0[1] = {c.w[0]}; const void* w1[1] = {c.w[1]}; const void* w2[1] = {c.w[2]};
        const void* b0[1] = {c.b[0]}; const void* b1[1] = {c.b[1]}; const void* b2[1] = {c.b[2]};
        pda_linear_fwd(x0, w0, b0, o0, nullptr, 1, n, c.H1, c.D, c.Dp, c.Dp, c.H1p, 0, ACT_TANH, 0, st);
        pda_linear_fwd(x1, w1, b1, o1, nullptr, 1, n, c.H2, c.H1, c.H1p, c.H1p, c.H2p, 0, ACT_TANH, 0, st);
        pda_linear_fwd(x2, w2, b2, o2, nullptr, 1, n, c.A, c.H2, c.H2p, c.H2p, c.Ap, 0, ACT_TANH, 0, st);
        pda_bf16_rows_to_f32(S.a4, c.Ap, S.hd_out, S.out_cols, n, c.A, st);
    }
}

template <class T>
bool dev_alloc(T** p, size_t count, bool& ok) {
    HIPCHK(hipMalloc((void**)p, count * sizeof(T)));
    if (ok) HIPCHK(hipMemset(*p, 0, count * sizeof(T)));
    return ok;
}
template <class T>
bool host_alloc(T** h, T** d, size_t count, bool& ok) {
    HIPCHK(hipHostMalloc((void**)h, count * sizeof(T), hipHostMallocMapped));
    if (ok) {
        std::memset(*h, 0, count * sizeof(T));
        HIPCHK(hipHostGetDevicePointer((void**)d, *h, 0));
    }
    return ok;
}

bool build(Server& S) {
    InferConfig& c = S.c;
    bool& ok = S.ok;
    int maxb = c.max_bucket > 0 ? c.max_bucket : c.nslots;
    int b = 8;
    while (true) {
        S.buckets.push_back(b);
        if (b >= maxb) break;
        b *= 2;
    }
    maxb = S.buckets.back();
    S.out_cols = c.kind == 0 ? 4 : (c.A + 2 + 3) / 4 * 4;
    if (c.high_priority) {
        int lo = 0, hi = 0;
        HIPCHK(hipDeviceGetStreamPriorityRange(&lo, &hi));
        HIPCHK(hipStreamCreateWithPriority(&S.stream, hipStreamNonBlocking, hi));
    } else {
        HIPCHK(hipStreamCreateWithFlags(&S.stream, hipStreamNonBlocking));
    }
    if (!ok) return false;
    size_t obs_bytes = (size_t)c.nslots * c.obs_stride;
    dev_alloc(&S.d_obs, obs_bytes, ok);
    if (c.kind == 0) {
        int H1 = conv_out(c.H, 8, 4), W1 = conv_out(c.W, 8, 4);
        int H2 = conv_out(H1, 4, 2), W2 = conv_out(W1, 4, 2);
        int H3 = conv_out(H2, 3, 1), W3 = conv_out(W2, 3, 1);
        dev_alloc(&S.a1, (size_t)maxb * H1 * W1 * 32, ok);
        dev_alloc(&S.a2, (size_t)maxb * H2 * W2 * 64, ok);
        dev_alloc(&S.a3, (size_t)maxb * H3 * W3 * 64, ok);
        dev_alloc(&S.h1pre, (size_t)maxb * c.HID, ok);
    } else {
        dev_alloc(&S.a1, (size_t)maxb * c.Dp, ok);
        dev_alloc(&S.a2, (size_t)maxb * c.H1p, ok);
        dev_alloc(&S.a3, (size_t)maxb * c.H2p, ok);
        dev_alloc(&S.a4, (size_t)maxb * c.Ap, ok);
    }
    host_alloc(&S.h_idx, &S.hd_idx, maxb, ok);
    host_alloc(&S.h_eps, &S.hd_eps, maxb, ok);
    host_alloc(&S.h_out, &S.hd_out, (size_t)maxb * S.out_cols, ok);
    host_alloc(&S.h_counter, &S.hd_counter, 8, ok);
    if (!ok) return false;
    // DMA straight out of the shared-memory mailbox when the runtime lets us pin it
    hipError_t reg = hipHostRegister(c.obs, obs_bytes, hipHostRegisterDefault);
    S.registered = (reg == hipSuccess);
    if (!S.registered) {
        (void)hipGetLastError();
        uint8_t* dummy = nullptr;
        host_alloc(&S.h_stage, &dummy, obs_bytes, ok);
    }
    S.ids.resize(c.nslots);
    S.req_seen.resize(c.nslots);
    // one eager pass per bucket (loads every code object), then capture
    for (int bk : S.buckets) enqueue_program(S, bk);
    HIPCHK(hipStreamSynchronize(S.stream));
    if (!ok) return false;
    for (int bk : S.buckets) {
        hipGraph_t g = nullptr;
        hipGraphExec_t ge = nullptr;
        HIPCHK(hipStreamBeginCapture(S.stream, hipStreamCaptureModeRelaxed));
        if (!ok) return false;
        enqueue_program(S, bk);
        HIPCHK(hipStreamEndCapture(S.stream, &g));
        if (ok) HIPCHK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        if (g) (void)hipGraphDestroy(g);
        if (!ok) return false;
        S.graphs.push_back(ge);
    }
    return ok;
}

// Slots that have posted at least one request: what a full batch is measured against.
int live_slots(Server& S) {
    int live = 0;
    for (int s = 0; s < S.c.nslots; ++s)
        live += pda_mailbox::req_word(S.c.seq, s)->load(std::memory_order_relaxed) > 0 ? 1 : 0;
    return live;
}

// One poll of the mailbox.  Returns the number of requests answered (0: nothing pending, -1: GPU error).
// window_us > 0: a partial batch waits that long (polling) for the rest of the live slots -- a batch of 8 costs the
// GPU and the host what a batch of 4 does (launch-bound), so when the server is the busy party, answering half the
// actors twice as often is the slower way round.
int serve_once(Server& S, int window_us = 0) {
    InferConfig& c = S.c;
    int n = pda_mailbox::collect_pending(c.seq, c.nslots, S.ids.data(), S.req_seen.data(), S.buckets.back());
    if (n == 0) return 0;
    if (window_us > 0) {
        int want = live_slots(S);
        S.live_slots.store((unsigned long long)want, std::memory_order_relaxed);
        if (n < want) {
            auto w0 = std::chrono::steady_clock::now();
            auto deadline = w0 + std::chrono::microseconds(window_us);
            while (n < want && std::chrono::steady_clock::now() < deadline &&
                   !S.stop.load(std::memory_order_relaxed)) {
                for (int i = 0; i < 16; ++i) pda_mailbox::cpu_relax();
                n = pda_mailbox::collect_pending(c.seq, c.nslots, S.ids.data(), S.req_seen.data(), S.buckets.back());
            }
            S.window_ns.fetch_add((unsigned long long)std::chrono::duration_cast<std::chrono::nanoseconds>(
                                      std::chrono::steady_clock::now() - w0).count(), std::memory_order_relaxed);
        }
    }
    auto t0 = std::chrono::steady_clock::now();
    bool& ok = S.ok;
    size_t bi = 0;
    while (S.buckets[bi] < n) ++bi;
    int bucket = S.buckets[bi];
    const uint8_t* obs = static_cast<const uint8_t*>(c.obs);
    if (S.registered) {
        int i = 0;
        while (i < n) {             // merge runs of neighbouring slots (gaps of <= 2 rows ride along)
            int j = i;
            while (j + 1 < n && S.ids[j + 1] - S.ids[j] <= 3) ++j;
            size_t off = (size_t)S.ids[i] * c.obs_stride;
            size_t len = (size_t)(S.ids[j] - S.ids[i] + 1) * c.obs_stride;
            HIPCHK(hipMemcpyAsync(S.d_obs + off, obs + off, len, hipMemcpyHostToDevice, S.stream));
            i = j + 1;
        }
        for (int k = 0; k < bucket; ++k) S.h_idx[k] = S.ids[k < n ? k : 0];
    } else {
        for (int k = 0; k < n; ++k)
            std::memcpy(S.h_stage + (size_t)k * c.obs_stride, obs + (size_t)S.ids[k] * c.obs_stride, c.obs_stride);
        HIPCHK(hipMemcpyAsync(S.d_obs, S.h_stage, (size_t)n * c.obs_stride, hipMemcpyHostToDevice, S.stream));
        for (int k = 0; k < bucket; ++k) S.h_idx[k] = k < n ? k : 0;
    }
    for (int k = 0; k < bucket; ++k) S.h_eps[k] = (c.eps && k < n) ? c.eps[S.ids[k]] : 0.f;
    S.h_counter[0] += 1;            // the epsilon-greedy draws of this batch (the stream is idle: last batch was waited for)
    HIPCHK(hipGraphLaunch(S.graphs[bi], S.stream));
    HIPCHK(hipStreamSynchronize(S.stream));
    if (!ok) return -1;
    int cols = c.kind == 0 ? 3 : c.A;
    for (int k = 0; k < n; ++k) {
        float* dst = c.out + (long)S.ids[k] * c.out_ld;
        const float* src = S.h_out + (long)k * S.out_cols;
        for (int j = 0; j < cols && j < c.out_ld; ++j) dst[j] = src[j];
        if (c.kind == 1)
            for (int j = cols; j < c.out_ld; ++j) dst[j] = 0.f;
        pda_mailbox::post_response(c.seq, S.ids[k], S.req_seen[k]);
    }
    auto t1 = std::chrono::steady_clock::now();
    S.batches.fetch_add(1, std::memory_order_relaxed);
    S.requests.fetch_add((unsigned long long)n, std::memory_order_relaxed);
    S.busy_ns.fetch_add((unsigned long long)std::chrono::duration_cast<std::chrono::nanoseconds>(t1 - t0).count(),
                        std::memory_order_relaxed);
    if ((unsigned long long)n > S.max_batch.load(std::memory_order_relaxed)) S.max_batch.store(n);
    return n;
}

void loop(Server* S, int device) {
    (void)hipSetDevice(device);
    int idle = 0;
    while (!S->stop.load(std::memory_order_acquire)) {
        int n = serve_once(*S, S->c.window_us);
        if (n < 0) break;
        if (n > 0) { idle = 0; continue; }
        ++idle;
        S->idle_polls.fetch_add(1, std::memory_order_relaxed);
        if (idle < 64) {
            for (int i = 0; i < 64; ++i) pda_mailbox::cpu_relax();
        } else {
            std::this_thread::sleep_for(std::chrono::microseconds(idle < 2000 ? 20 : 200));
        }
    }
}

}  // namespace

// ---- C ABI -------------------------------------------------------------------------------
PDA_EXPORT void* pda_infer_create(const InferConfig* cfg) {
    Server* S = new Server();
    S->c = *cfg;
    if (!build(*S)) {
        std::fprintf(stderr, "[pda infer] server build failed\n");
        delete S;
        return nullptr;
    }
    return S;
}
PDA_EXPORT int pda_infer_serve_once(void* h) { return serve_once(*static_cast<Server*>(h)); }
PDA_EXPORT void* pda_infer_stream(void* h) { return static_cast<Server*>(h)->stream; }
PDA_EXPORT int pda_infer_start(void* h) {
    Server* S = static_cast<Server*>(h);
    if (S->thread.joinable()) return 0;
    int dev = 0;
    (void)hipGetDevice(&dev);
    S->stop.store(false);
    S->thread = std::thread(loop, S, dev);
    return 1;
}
PDA_EXPORT void pda_infer_stop(void* h) {
    Server* S = static_cast<Server*>(h);
    S->stop.store(true, std::memory_order_release);
    if (S->thread.joinable()) S->thread.join();
}
PDA_EXPORT void pda_infer_stats(void* h, Stats* out) {
    Server* S = static_cast<Server*>(h);
    out->batches = S->batches.load();
    out->requests = S->requests.load();
    out->busy_ns = S->busy_ns.load();
    out->max_batch = S->max_batch.load();
    out->registered = S->registered ? 1 : 0;
    out->idle_polls = S->idle_polls.load();
    out->window_ns = S->window_ns.load();
    out->live_slots = S->live_slots.load();
}
PDA_EXPORT void pda_infer_destroy(void* h) {
    Server* S = static_cast<Server*>(h);
    pda_infer_stop(h);
    (void)hipStreamSynchronize(S->stream);
    for (hipGraphExec_t g : S->graphs) (void)hipGraphExecDestroy(g);
    if (S->registered) (void)hipHostUnregister(S->c.obs);
    void* dev[] = {S->d_obs, S->a1, S->a2, S->a3, S->a4, S->h1pre};
    for (void* p : dev) if (p) (void)hipFree(p);
    void* host[] = {S->h_idx, S->h_eps, S->h_out, S->h_stage, S->h_counter};
    for (void* p : host) if (p) (void)hipHostFree(p);
    (void)hipStreamDestroy(S->stream);
    delete S;
}
