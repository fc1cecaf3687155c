// DDPG learner step, forward + loss head + data-gradient chain in ONE launch (kernels K5 / K6 / K12 / K16).
//
// The reference's step is two autograd passes over five Linear layers each
// (/root/reference/core/single_processes/ddpg_learner.py:66-86, core/models/ddpg_mlp_model.py:16-35); round 1 ran
// it as ~20 launches of a few kFLOP, i.e. at launch latency.  The layers of an MLP are independent across batch
// rows, so here one workgroup takes 16 rows (one MFMA M-tile) through EVERYTHING that does not reduce over the
// batch:
//
//   stage     s0, s_n, a (fp32 replay rows through the sampled indices) -> bf16 in LDS
//   forward   actor(s0) | actor'(s_n)            waves 0-3 run the online networks, waves 4-7 the target networks:
//             critic.0(s0) | critic'.0(s_n)      same layer shapes, so the two halves meet at every barrier
//             critic.1([h|pi(s0)], [h|a]) | critic'.1([h'|pi'(s_n)])     torch.cat is two column ranges of one LDS row
//   head      Q = w3.h2 + b3 for the three passes, y = r + gamma^n Q' (1 - done), both losses, dL/dh2 of both passes
//   backward  d[h|a] = dh2 W_c1 (x tanh'), dha2 = dz3 W_a4, dha1 = dha2 W_a2     (all 8 waves)
//
// Activations never leave LDS between layers; what the weight gradients need (layer inputs and the dL/d(pre-
// activation) of every layer, bf16) stays in LDS too and is written to global in one coalesced pass at the end;
// ONE grouped GEMM launch (pda_linear_wgrad_n) reduces over the batch afterwards.  Weights stream from L2 / MALL as
// MFMA B fragments, 16 bytes per lane.  The kernel is a latency chain, not a bandwidth or MFMA problem (366 KB of
// bf16 weights per pass, ~0.1 GFLOP), and what decides its duration is (a) how many memory latencies a wave pays
// per layer and (b) how much code it walks through once:
//   v1  one 16-column tile of prefetch, global stores + bias loads in the tile epilogue: a latency per tile, 66 us
//   v2  a 40-fragment register ring refilled slot by slot: the compiler drains it (vmcnt(0)) once per unrolled
//       chunk anyway, and 80 inlined epilogue copies per layer made 500 KB of straight-line code: 73 us, fetch-bound
//   v3  (this) a wave issues the loads of up to 40 fragments = ALL its tiles of most layers in one batch, then
//       multiplies tile by tile; k-steps per tile are a template parameter so the batch is a static [tile][k] array
//       and there is one epilogue copy per tile slot.  No global access inside the batch except the weight loads.
//
// The backward B operands are W^T: the optimizer keeps transposed bf16 copies of critic.1.0 and actor.2 next to the
// row-major shadows (optim.h: AdamExtras::tinfo).
#include "api.h"

namespace {

constexpr int FUSED_THREADS = 512;
constexpr int KSMAX = 13;                   // k-steps of 32: layer inputs up to 416 wide

struct Layer { const bf16_t* w; const float* b; int N, K, Kp; };      // w: bf16 [N][Kp], zero beyond K

// branch-free tanh on v_exp_f32 / v_rcp_f32 (|error| ~1e-7 absolute: far below the bf16 the result is rounded to;
// libm's tanhf is ~40 instructions with three branches, inlined into every tile epilogue)
__device__ __forceinline__ float fast_tanh(float x) {
    const float e = __expf(2.f * x);
    return 1.f - 2.f * __builtin_amdgcn_rcpf(e + 1.f);
}

__host__ __device__ inline int ks_class(int K);
__host__ __device__ inline int lds_pitch(int width) { return ks_class(width) * 32 + 8; }    // rows 4 banks apart

}  // namespace

struct DdpgFusedArgs {
    // replay
    const float* s0; const float* s1; long ld_s; const float* tails; int tail_ld;
    int off_action, off_reward, off_gamma, off_term;
    const int* idx; const float* isw;
    // networks: [0] online, [1] target
    Layer a0[2], a2[2], a4[2], c0[2], c1[2];
    const float* w3[2]; const float* b3[2];         // critic output layer, fp32 [C2]
    Layer c1T, a2T;                                 // transposed online copies: w = [K_in][pad8(N_out)] (N = K_in, K = N_out)
    // what the weight gradients read (global bf16)
    bf16_t* x0; long ld_x0;                         // [2B][Dp]: rows b and B + b
    bf16_t* ha1; long ld_ha1; bf16_t* ha2; long ld_ha2;
    bf16_t* cat; long ld_cat;                       // [2B][CKp]: rows b = [h | pi(s0)], B + b = [h | a]
    bf16_t* h2; long ld_h2;                         // [2B][C2p]
    bf16_t* dh2; bf16_t* dq; int dq_ld;             // [2B][C2p], [2B][dq_ld] (column 0)
    bf16_t* dcat; bf16_t* dha2; bf16_t* dha1;
    float* q_out; float* td_out; float* loss_sum; float* leaf_sum; float* leaf_min; float prio_eps, prio_alpha;
    int B, D, A, C1, leak, actor_warmup;
    unsigned long long* rng_counter; int* opt_step;
    float actor_l2;         // + actor_l2 * mean_b |z_b|^2 in the actor loss, z = the actor's pre-tanh output (0: off)
    long long* prof;        // optional: wall_clock64() of workgroup 0 at every phase boundary (tools/bench_mlp_step.py --phases)
};

namespace {

// Columns [16 t, 16 t + 16) for the tiles t = wv, wv + nwv, .. of  A[16][K] W^T  for NPASS activation blocks in LDS
// (row pitch pa) sharing the weights.  epi(n, row0, acc[NPASS]) gets column n and rows row0 .. row0+3 of each pass;
// it must not touch global memory.  wv / nwv are wave-uniform.  KS = k-steps per tile (>= ceil(K / 32): the
// surplus steps load nothing and multiply the zero pad columns of A, whose pitch covers 32 KS columns).
constexpr int BATCH_FRAGS = 40;
template <int KS> struct BatchTiles { static constexpr int v = (BATCH_FRAGS / KS) < 8 ? (BATCH_FRAGS / KS) : 8; };

template <int NPASS, int KS, class Epi>
__device__ __forceinline__ void batch_layer(const bf16_t* const (&A)[NPASS], int pa, const Layer& L, int wv, int nwv,
                                            int lane, Epi&& epi) {
    constexpr int TB = BatchTiles<KS>::v;
    const int ntiles = (L.N + 15) >> 4;
    const int col = lane & 15, kg = lane >> 4;
    for (int t0 = wv; t0 < ntiles; t0 += TB * nwv) {
        uint4 f[TB][KS];
#pragma unroll
        for (int tb = 0; tb < TB; ++tb) {
            const int n = (t0 + tb * nwv) * 16 + col;
            const bf16_t* row = L.w + (long)n * L.Kp + kg * 8;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const bool ok = n < L.N && ks * 32 + kg * 8 < L.Kp;     // (n < N also rules out tiles past the last)
                f[tb][ks] = ok ? *reinterpret_cast<const uint4*>(row + ks * 32) : make_uint4(0u, 0u, 0u, 0u);
            }
        }
#pragma unroll
        for (int tb = 0; tb < TB; ++tb) {
            const int t = t0 + tb * nwv;
            if (t < ntiles) {
                f32x4 acc[NPASS];
#pragma unroll
                for (int p = 0; p < NPASS; ++p) acc[p] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const bf16x8 fb = __builtin_bit_cast(bf16x8, f[tb][ks]);
#pragma unroll
                    for (int p = 0; p < NPASS; ++p) {
                        const bf16x8 fa = *reinterpret_cast<const bf16x8*>(A[p] + col * pa + ks * 32 + kg * 8);
                        acc[p] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb, acc[p], 0, 0, 0);
                    }
                }
                epi(t * 16 + col, kg * 4, acc);
            }
        }
    }
}

// k-steps per tile are compiled for these sizes; a layer uses the next one up
__host__ __device__ inline int ks_class(int K) {
    const int ks = (K + 31) >> 5;
    return ks <= 1 ? 1 : ks <= 2 ? 2 : ks <= 4 ? 4 : ks <= 7 ? 7 : ks <= 10 ? 10 : 13;
}

template <int NPASS, class Epi>
__device__ __forceinline__ void tile_layer(const bf16_t* const (&A)[NPASS], int pa, const Layer& L, int wv, int nwv,
                                           int lane, Epi&& epi) {
    switch (ks_class(L.K)) {
        case 1: batch_layer<NPASS, 1>(A, pa, L, wv, nwv, lane, epi); break;
        case 2: batch_layer<NPASS, 2>(A, pa, L, wv, nwv, lane, epi); break;
        case 4: batch_layer<NPASS, 4>(A, pa, L, wv, nwv, lane, epi); break;
        case 7: batch_layer<NPASS, 7>(A, pa, L, wv, nwv, lane, epi); break;
        case 10: batch_layer<NPASS, 10>(A, pa, L, wv, nwv, lane, epi); break;
        default: batch_layer<NPASS, 13>(A, pa, L, wv, nwv, lane, epi); break;
    }
}

// LDS -> global, 16 bytes per lane: rows r (batch row r0 + r < B) of a [16][pitch] block, `width8` columns (multiple of 8)
__device__ __forceinline__ void spill_rows(const bf16_t* src, int pitch, bf16_t* dst, long ld, int row_off, int width8,
                                           int r0, int B, int tid) {
    const int per_row = width8 >> 3;
    for (int t = tid; t < 16 * per_row; t += FUSED_THREADS) {
        const int r = t / per_row, c = (t - r * per_row) * 8;
        if (r0 + r < B)
            *reinterpret_cast<uint4*>(dst + (long)(row_off + r0 + r) * ld + c) =
                *reinterpret_cast<const uint4*>(src + r * pitch + c);
    }
}

__global__ __launch_bounds__(FUSED_THREADS) void ddpg_fused_kernel(DdpgFusedArgs p) {
    extern __shared__ __attribute__((aligned(16))) bf16_t lds[];
#define STAMP(k) do { if (p.prof && blockIdx.x == 0 && threadIdx.x == 0) p.prof[k] = (long long)wall_clock64(); } while (0)
    STAMP(31);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // an SGPR: tile_layer's cursors stay scalar
    const int grp = wave >> 2, wg = wave & 3;         // forward: 0 = online networks, 1 = target networks
    const int r0 = blockIdx.x * 16;
    const int B = p.B, D = p.D, A = p.A, C1 = p.C1;
    const int H1 = p.a0[0].N, H2 = p.a2[0].N, C2 = p.c1[0].N, CK = C1 + A;
    const int pD = lds_pitch(D), p1 = lds_pitch(H1), p2 = lds_pitch(H2), pc = lds_pitch(CK), ph = lds_pitch(C2);
    // blocks that live to the end (the weight gradients read them) ...
    bf16_t* X0 = lds;
    bf16_t* HA1 = X0 + 16 * pD;
    bf16_t* HA2 = HA1 + 16 * p1;
    bf16_t* CATP = HA2 + 16 * p2;
    bf16_t* CATC = CATP + 16 * pc;
    bf16_t* H2P = CATC + 16 * pc;
    bf16_t* H2C = H2P + 16 * ph;
    bf16_t* DH2P = H2C + 16 * ph;
    bf16_t* DH2C = DH2P + 16 * ph;
    bf16_t* DCATP = DH2C + 16 * ph;
    bf16_t* DCATC = DCATP + 16 * pc;
    bf16_t* DHA2 = DCATC + 16 * pc;
    bf16_t* DHA1 = DHA2 + 16 * p2;
    bf16_t* X1 = DHA1 + 16 * p1;
    bf16_t* endb = X1 + 16 * pD;
    // ... and the target networks' activations, which die before the backward blocks they share storage with are
    // written (TA1 after actor.2, TA2 after actor.4, CATT after critic.1, H2T after the head)
    bf16_t *TA1 = DHA1, *TA2 = DHA2, *CATT = DCATP, *H2T = DCATC;
    float* dz3 = reinterpret_cast<float*>(endb);       // [16][8] fp32
    float* rowv = dz3 + 16 * 8;                          // [16][4]: dq_pi, dq_c per row
    const int lds_bytes = (int)((endb - lds) * 2) + 16 * 8 * 4 + 16 * 4 * 4;
    // every bias of both networks (a global load inside a tile epilogue would make the compiler drain the weight
    // ring with s_waitcnt vmcnt(0) once per tile)
    float* BIAS = rowv + 16 * 4;
    const int SB = H1 + H2 + A + C1 + C2;
    // ... and everything else the head and the small backward stages would fetch on their own (each such fetch is a
    // full memory latency on the kernel's critical path: ~1.5 us, counter profile profiles/r2_ddpg_fused_*.txt)
    float* W3 = BIAS + 2 * SB;              // [2][C2] critic output weights, online | target
    float* TL = W3 + 2 * C2;                // [16][4] reward, gamma^n, done, importance weight of the tile's rows
    float* A4W = TL + 16 * 4;               // [A][H2] online actor output weights (from the bf16 shadow)
    float* ZA = A4W + A * H2;               // [16][8] the online actor's pre-tanh output (actor_l2 penalty)
    const int oA0 = 0, oA2 = H1, oA4 = H1 + H2, oC0 = H1 + H2 + A, oC1 = H1 + H2 + A + C1;

    // ---- zero LDS (pad columns multiply zero weights, but must be finite), then stage the replay rows ----
    {
        uint4* z = reinterpret_cast<uint4*>(lds);
        for (int i = tid; i < lds_bytes / 16; i += FUSED_THREADS) z[i] = make_uint4(0u, 0u, 0u, 0u);
    }
    __syncthreads();
    STAMP(0);
    for (int t = tid; t < 16 * D; t += FUSED_THREADS) {
        const int r = t / D, d = t - r * D, b = r0 + r;
        if (b < B) {
            const long row = p.idx ? (long)p.idx[b] : (long)b;
            X0[r * pD + d] = f2bf(p.s0[row * p.ld_s + d]);
            X1[r * pD + d] = f2bf(p.s1[row * p.ld_s + d]);
        }
    }
    for (int t = tid; t < 16 * A; t += FUSED_THREADS) {
        const int r = t / A, a = t - r * A, b = r0 + r;
        if (b < B) {
            const long row = p.idx ? (long)p.idx[b] : (long)b;
            CATC[r * pc + C1 + a] = f2bf(p.tails[row * p.tail_ld + p.off_action + a]);
        }
    }
    for (int t = tid; t < C2; t += FUSED_THREADS) { W3[t] = p.w3[0][t]; W3[C2 + t] = p.w3[1][t]; }
    for (int t = tid; t < A * H2; t += FUSED_THREADS) {
        const int a = t / H2, k = t - a * H2;
        A4W[t] = bf2f(p.a4[0].w[(long)a * p.a4[0].Kp + k]);
    }
    if (tid < 16 && r0 + tid < B) {
        const int b = r0 + tid;
        const long row = p.idx ? (long)p.idx[b] : (long)b;
        const float* tl = p.tails + row * p.tail_ld;
        TL[tid * 4] = tl[p.off_reward];
        TL[tid * 4 + 1] = tl[p.off_gamma];
        TL[tid * 4 + 2] = tl[p.off_term];
        TL[tid * 4 + 3] = p.isw ? p.isw[b] : 1.f;
    }
    for (int g = 0; g < 2; ++g) {
        float* bg = BIAS + g * SB;
        for (int t = tid; t < H1; t += FUSED_THREADS) bg[oA0 + t] = p.a0[g].b[t];
        for (int t = tid; t < H2; t += FUSED_THREADS) bg[oA2 + t] = p.a2[g].b[t];
        for (int t = tid; t < A; t += FUSED_THREADS) bg[oA4 + t] = p.a4[g].b[t];
        for (int t = tid; t < C1; t += FUSED_THREADS) bg[oC0 + t] = p.c0[g].b[t];
        for (int t = tid; t < C2; t += FUSED_THREADS) bg[oC1 + t] = p.c1[g].b[t];
    }
    __syncthreads();
    STAMP(1);

    // ---- actor layer 1 and 2 ----
    {
        const bf16_t* const in[1] = {grp ? X1 : X0};
        bf16_t* out = grp ? TA1 : HA1;
        const Layer L = p.a0[grp];
        tile_layer<1>(in, pD, L, wg, 4, lane, [&](int n, int row, const f32x4 (&acc)[1]) {
            if (n >= L.N) return;
            const float bias = BIAS[grp * SB + oA0 + n];
#pragma unroll
            for (int i = 0; i < 4; ++i) out[(row + i) * p1 + n] = f2bf(fast_tanh(acc[0][i] + bias));
        });
    }
    __syncthreads();
    STAMP(2);
    {
        const bf16_t* const in[1] = {grp ? TA1 : HA1};
        bf16_t* out = grp ? TA2 : HA2;
        const Layer L = p.a2[grp];
        tile_layer<1>(in, p1, L, wg, 4, lane, [&](int n, int row, const f32x4 (&acc)[1]) {
            if (n >= L.N) return;
            const float bias = BIAS[grp * SB + oA2 + n];
#pragma unroll
            for (int i = 0; i < 4; ++i) out[(row + i) * p2 + n] = f2bf(fast_tanh(acc[0][i] + bias));
        });
    }
    __syncthreads();
    STAMP(3);
    // ---- actor output layer (wave 0 of each half) and critic layer 1 (all four) into the concat rows ----
    if (wg == 0) {
        const bf16_t* const in[1] = {grp ? TA2 : HA2};
        bf16_t* out = grp ? CATT : CATP;
        const Layer L = p.a4[grp];
        tile_layer<1>(in, p2, L, 0, 1, lane, [&](int n, int row, const f32x4 (&acc)[1]) {
            if (n >= L.N) return;
            const float bias = BIAS[grp * SB + oA4 + n];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                out[(row + i) * pc + C1 + n] = f2bf(fast_tanh(acc[0][i] + bias));
                if (!grp && n < 8) ZA[(row + i) * 8 + n] = acc[0][i] + bias;
            }
        });
    }
    {
        const bf16_t* const in[1] = {grp ? X1 : X0};
        const Layer L = p.c0[grp];
        tile_layer<1>(in, pD, L, wg, 4, lane, [&](int n, int row, const f32x4 (&acc)[1]) {
            if (n >= L.N) return;
            const float bias = BIAS[grp * SB + oC0 + n];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const bf16_t v = f2bf(fast_tanh(acc[0][i] + bias));
                if (grp) {
                    CATT[(row + i) * pc + n] = v;
                } else {
                    CATP[(row + i) * pc + n] = v;
                    CATC[(row + i) * pc + n] = v;
                }
            }
        });
    }
    __syncthreads();
    STAMP(4);
    // ---- critic layer 2: the online half runs both passes against one stream of weights ----
    if (grp == 0) {
        const bf16_t* const in[2] = {CATP, CATC};
        const Layer L = p.c1[0];
        tile_layer<2>(in, pc, L, wg, 4, lane, [&](int n, int row, const f32x4 (&acc)[2]) {
            if (n >= L.N) return;
            const float bias = BIAS[oC1 + n];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                H2P[(row + i) * ph + n] = f2bf(fast_tanh(acc[0][i] + bias));
                H2C[(row + i) * ph + n] = f2bf(fast_tanh(acc[1][i] + bias));
            }
        });
    } else {
        const bf16_t* const in[1] = {CATT};
        const Layer L = p.c1[1];
        tile_layer<1>(in, pc, L, wg, 4, lane, [&](int n, int row, const f32x4 (&acc)[1]) {
            if (n >= L.N) return;
            const float bias = BIAS[SB + oC1 + n];
#pragma unroll
            for (int i = 0; i < 4; ++i) H2T[(row + i) * pc + n] = f2bf(fast_tanh(acc[0][i] + bias));
        });
    }
    __syncthreads();
    STAMP(5);

    // ---- head: wave w takes rows 2w, 2w+1 ----
    const float invB = 1.f / (float)B;
    const bool actor_on = !(p.actor_warmup > 0 && p.opt_step && *p.opt_step < p.actor_warmup);
    for (int rr = 0; rr < 2; ++rr) {
        const int r = wave * 2 + rr, b = r0 + r;
        float sp = 0.f, sc = 0.f, st = 0.f;
        for (int j = lane; j < C2; j += 64) {
            const float wo = W3[j];
            sp += wo * bf2f(H2P[r * ph + j]);
            sc += wo * bf2f(H2C[r * ph + j]);
            st += W3[C2 + j] * bf2f(H2T[r * pc + j]);
        }
        const float q_pi = wave_sum(sp) + p.b3[0][0], q_c = wave_sum(sc) + p.b3[0][0], q_t = wave_sum(st) + p.b3[1][0];
        float dq_pi = 0.f, dq_c = 0.f;
        if (b < B) {
            const float y = TL[r * 4] + TL[r * 4 + 1] * q_t * (1.f - TL[r * 4 + 2]);
            const float td = q_c - y;
            const float w = TL[r * 4 + 3];
            dq_c = 2.f * w * td * invB;
            dq_pi = actor_on ? -invB : 0.f;
            if (lane < p.dq_ld) {
                p.dq[(long)b * p.dq_ld + lane] = f2bf((lane == 0 && p.leak) ? dq_pi : 0.f);
                p.dq[(long)(B + b) * p.dq_ld + lane] = f2bf(lane == 0 ? dq_c : 0.f);
            }
            if (lane == 0) {
                if (p.q_out) { p.q_out[b] = q_pi; p.q_out[B + b] = q_c; p.q_out[2 * B + b] = q_t; }
                if (p.td_out) p.td_out[b] = td;
                float pen = 0.f;
                if (p.actor_l2 > 0.f)
                    for (int a = 0; a < A; ++a) pen += ZA[r * 8 + a] * ZA[r * 8 + a];
                atomicAdd(p.loss_sum, (-q_pi + p.actor_l2 * pen) * invB);
                atomicAdd(p.loss_sum + 1, w * td * td * invB);
                if (p.leaf_sum) {
                    const long row = p.idx ? (long)p.idx[b] : (long)b;
                    const float leaf = __powf(fabsf(td) + p.prio_eps, p.prio_alpha);
                    p.leaf_sum[row] = leaf;
                    p.leaf_min[row] = leaf;
                }
            }
        }
        if (lane == 0) { rowv[r * 4] = dq_pi; rowv[r * 4 + 1] = dq_c; }
    }
    __syncthreads();
    STAMP(6);
    // dL/d(pre-activation of critic.1.0) for both passes (rows beyond the batch stay zero)
    for (int t = tid; t < 16 * C2; t += FUSED_THREADS) {
        const int r = t / C2, j = t - r * C2;
        const float wo = W3[j];
        const float a = bf2f(H2P[r * ph + j]), c = bf2f(H2C[r * ph + j]);
        DH2P[r * ph + j] = f2bf(rowv[r * 4] * wo * (1.f - a * a));
        DH2C[r * ph + j] = f2bf(rowv[r * 4 + 1] * wo * (1.f - c * c));
    }
    __syncthreads();
    STAMP(7);

    // ---- d[h | a] = dh2 W_c1, both passes, times tanh' of the concat row (overwrites CATT / H2T) ----
    {
        const bf16_t* const in[2] = {DH2P, DH2C};
        const Layer L = p.c1T;
        tile_layer<2>(in, ph, L, wave, 8, lane, [&](int n, int row, const f32x4 (&acc)[2]) {
            if (n >= L.N) return;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int r = row + i;
                const float yp = bf2f(CATP[r * pc + n]), yc = bf2f(CATC[r * pc + n]);
                float gp = acc[0][i] * (1.f - yp * yp);
                const float gc = acc[1][i] * (1.f - yc * yc);
                if (n >= C1 && n - C1 < 8) {
                    // the penalty's gradient does not pass through tanh': it is what brings a saturated actor back
                    if (actor_on && r0 + r < B) gp += 2.f * p.actor_l2 * invB * ZA[r * 8 + (n - C1)];
                    dz3[r * 8 + (n - C1)] = gp;
                }
                DCATP[r * pc + n] = f2bf(gp);
                DCATC[r * pc + n] = f2bf(gc);
            }
        });
    }
    __syncthreads();
    STAMP(8);
    // CATT / H2T were wider than what the backward wrote over them: their leftovers must not reach global
    for (int t = tid; t < 16 * (pc - CK); t += FUSED_THREADS) {
        const int r = t / (pc - CK), j = CK + t - r * (pc - CK);
        DCATP[r * pc + j] = 0;
        DCATC[r * pc + j] = 0;
    }
    // ---- dha2 = dz3 W_a4 x tanh'(ha2): A is tiny, plain FMAs on the row-major shadow ----
    {
        for (int t = tid; t < 16 * H2; t += FUSED_THREADS) {
            const int r = t / H2, k = t - r * H2;
            float s = 0.f;
            for (int a = 0; a < A; ++a) s += bf2f(f2bf(dz3[r * 8 + a])) * A4W[a * H2 + k];
            const float y = bf2f(HA2[r * p2 + k]);
            DHA2[r * p2 + k] = f2bf(s * (1.f - y * y));     // (TA2 only ever wrote columns < H2: the pad is still zero)
        }
    }
    __syncthreads();
    STAMP(9);
    // ---- dha1 = dha2 W_a2 x tanh'(ha1) (overwrites TA1) ----
    {
        const bf16_t* const in[1] = {DHA2};
        const Layer L = p.a2T;
        tile_layer<1>(in, p2, L, wave, 8, lane, [&](int n, int row, const f32x4 (&acc)[1]) {
            if (n >= L.N) return;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float y = bf2f(HA1[(row + i) * p1 + n]);
                DHA1[(row + i) * p1 + n] = f2bf(acc[0][i] * (1.f - y * y));
            }
        });
    }
    __syncthreads();
    STAMP(10);
    // ---- everything the weight gradients read, in one coalesced pass ----
    const int Dp8 = (D + 7) & ~7, H1p = (H1 + 7) & ~7, H2p = (H2 + 7) & ~7, CKp = (CK + 7) & ~7, C2p = (C2 + 7) & ~7;
    spill_rows(X0, pD, p.x0, p.ld_x0, 0, Dp8, r0, B, tid);
    spill_rows(X0, pD, p.x0, p.ld_x0, B, Dp8, r0, B, tid);
    spill_rows(HA1, p1, p.ha1, p.ld_ha1, 0, H1p, r0, B, tid);
    spill_rows(HA2, p2, p.ha2, p.ld_ha2, 0, H2p, r0, B, tid);
    spill_rows(CATP, pc, p.cat, p.ld_cat, 0, CKp, r0, B, tid);
    spill_rows(CATC, pc, p.cat, p.ld_cat, B, CKp, r0, B, tid);
    spill_rows(H2P, ph, p.h2, p.ld_h2, 0, C2p, r0, B, tid);
    spill_rows(H2C, ph, p.h2, p.ld_h2, B, C2p, r0, B, tid);
    spill_rows(DH2P, ph, p.dh2, p.ld_h2, 0, C2p, r0, B, tid);
    spill_rows(DH2C, ph, p.dh2, p.ld_h2, B, C2p, r0, B, tid);
    spill_rows(DCATP, pc, p.dcat, p.ld_cat, 0, CKp, r0, B, tid);
    spill_rows(DCATC, pc, p.dcat, p.ld_cat, B, CKp, r0, B, tid);
    spill_rows(DHA2, p2, p.dha2, p.ld_ha2, 0, H2p, r0, B, tid);
    spill_rows(DHA1, p1, p.dha1, p.ld_ha1, 0, H1p, r0, B, tid);
    STAMP(30);
    if (blockIdx.x == 0 && tid == 0) {
        if (p.rng_counter) *p.rng_counter += 1;
        if (p.opt_step) *p.opt_step += 1;
    }
}

}  // namespace

PDA_EXPORT int pda_ddpg_fused_supported(int D, int H1, int H2, int A, int C1, int C2) {
    const int lim = KSMAX * 32;
    if (!(D <= lim && H1 <= lim && H2 <= lim && C1 + A <= lim && C2 <= lim && A <= 8 && C1 % 8 == 0)) return 0;
    const int pD = lds_pitch(D), p1 = lds_pitch(H1), p2 = lds_pitch(H2), pc = lds_pitch(C1 + A), ph = lds_pitch(C2);
    const long bytes = 16L * 2 * (2 * pD + 2 * p1 + 2 * p2 + 4 * pc + 4 * ph) + 16 * 8 * 4 + 16 * 4 * 4 +
                       2L * 4 * (H1 + H2 + A + C1 + C2) + 4L * (2 * C2 + 16 * 4 + A * H2 + 16 * 8);
    return bytes <= 160 * 1024;
}

PDA_EXPORT void pda_ddpg_fused(const DdpgFusedArgs* args, hipStream_t stream) {
    DdpgFusedArgs p = *args;
    const int H1 = p.a0[0].N, H2 = p.a2[0].N, C2 = p.c1[0].N;
    const int pD = lds_pitch(p.D), p1 = lds_pitch(H1), p2 = lds_pitch(H2), pc = lds_pitch(p.C1 + p.A), ph = lds_pitch(C2);
    const size_t lds = 16u * 2 * (2 * pD + 2 * p1 + 2 * p2 + 4 * pc + 4 * ph) + 16 * 8 * 4 + 16 * 4 * 4 +
                       2u * 4 * (H1 + H2 + p.A + p.C1 + C2) + 4u * (2 * C2 + 16 * 4 + p.A * H2 + 16 * 8);
    static size_t granted = 0;
    if (lds > granted) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(ddpg_fused_kernel),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        granted = lds;
    }
    hipLaunchKernelGGL(ddpg_fused_kernel, dim3((p.B + 15) / 16), dim3(FUSED_THREADS), lds, stream, p);
}
