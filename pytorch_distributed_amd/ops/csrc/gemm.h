// Tiled bf16 MFMA GEMM skeleton for gfx950:  C[M,N] = sum_k A(m,k) * B(n,k)
//
// One template serves every matmul-shaped op in the framework -- conv forward
// (implicit GEMM, im2col done by the A loader), conv/linear weight gradients
// (both operands transposed, split-K, fp32 atomics into the flat gradient
// buffer), data gradients, linear layers -- by plugging in operand *sources*
// and an *epilogue*:
//
//   K-contiguous source ("N" mode): lane reads 8 consecutive k of one row with a
//       16-byte global load, stages it [row][k] in LDS, fragments come back with
//       ds_read_b128 exactly in the v_mfma_f32_16x16x32_bf16 A/B layout
//       (lane l: row l&15, k = 8*(l>>4)+j).
//   row-contiguous source ("T" mode: the reduction index is the slow axis in
//       memory -- every weight-gradient operand): lane reads 8 consecutive ROWS at
//       one k, stages the tile [k][row] untransposed with ds_write_b128, and
//       fragments come back through ds_read_b64_tr_b16, CDNA4's transposing LDS
//       read (two 4x16 blocks per fragment), so no scatter of 2-byte LDS writes.
//
// Block = 256 threads = 4 wave64s laid out WM x WN over a BM x BN tile (sides of 32, 64 or 128: the 128-row
// tiles serve the large-batch fc launches, see gemm.hip), BK = 64 or 32
// (two / one MFMA k-steps per tile), LDS double-buffered with a register prefetch of the
// next k-tile, so there is one barrier per k-tile and the global loads of tile t+1
// fly under the MFMAs of tile t.  LDS tiles are unpadded and XOR-swizzled so that
// both ds_read_b128 (N mode) and ds_read_b64_tr_b16 (T mode) are bank-conflict free
// (the first counter profile, profiles/r1_b128_counters.txt, showed 33-50 % of the LDS
// cycles lost to conflicts with padded rows: no row padding can separate the 16-lane
// groups ds_read_b128 is serviced in).  blockIdx.y = split-K slice, blockIdx.z = problem group (e.g. online
// net on s0 and target net on s1 in ONE launch).
#pragma once
#include "common.h"
#include <cstdio>
#include <cstdlib>
#include <type_traits>

#ifndef GEMM_BK
#define GEMM_BK 64          // default k-tile (elements); a template parameter of the kernel
#endif
#define GEMM_PF 3           // k-tiles of global loads kept in flight ahead of the MFMAs (register ring)
#define GEMM_THREADS 256
// The k-loop of both workgroup programs: whole rounds of the three-slot ring in a loop with ONE exit, then the
// 0-2 left-over tiles (which prefetch nothing).  Written with `break`s between the tiles, the loop's exits are
// unified into its latch block and the compiler's s_waitcnt pass merges "left after the first tile" (slot 0
// holds the NEWEST loads) into the loop header: the first commit of every round then waited for vmcnt(0),
// draining the ring once per three tiles.
#define GEMM_RING_LOOP                                                                               \
    {                                                                                                \
        typedef std::integral_constant<int, 0> S0;                                                   \
        typedef std::integral_constant<int, 1> S1;                                                   \
        typedef std::integral_constant<int, 2> S2;                                                   \
        const int nt = (kend - kbeg + BK - 1) / BK, rounds = nt / GEMM_PF;                           \
        int buf = 0, k0 = kbeg;                                                                      \
        for (int r = 0; r < rounds; ++r) {                                                           \
            tile(S0{}, std::true_type{}, k0, buf);                                                   \
            tile(S1{}, std::true_type{}, k0 + BK, buf);                                              \
            tile(S2{}, std::true_type{}, k0 + 2 * BK, buf);                                          \
            k0 += GEMM_PF * BK;                                                                      \
        }                                                                                            \
        const int rem = nt - rounds * GEMM_PF;                                                       \
        if (rem > 0) {                                                                               \
            tile(S0{}, std::false_type{}, k0, buf);                                                  \
            if (rem > 1) tile(S1{}, std::false_type{}, k0 + BK, buf);                                \
        }                                                                                            \
    }
#define GEMM_MAX_GROUPS 8

// ----------------------------------------------------------------------------
// operand sources
// ----------------------------------------------------------------------------
// Every load is issued unconditionally (an out-of-range lane reads the tensor's first 16 bytes and
// the value is replaced by zero afterwards): a load under a divergent branch makes the compiler fall
// back to s_waitcnt vmcnt(0) before each LDS store, which drains the whole prefetch ring.
// A source therefore splits a load in two: raw(ctx, k, ok) only issues it (Raw goes into the prefetch
// ring untouched, so nothing waits on it), finish(raw, ok) masks / converts when the tile is committed
// to LDS, GEMM_PF tiles later.
__device__ __forceinline__ uint4 mask16(uint4 v, bool ok) {
    v.x = ok ? v.x : 0u; v.y = ok ? v.y : 0u; v.z = ok ? v.z : 0u; v.w = ok ? v.w : 0u;
    return v;
}
#define PDA_BF16_SOURCE_FINISH                                                              \
    typedef uint4 Raw;                                                                       \
    __device__ __forceinline__ static uint4 finish(Raw v, bool ok) { return mask16(v, ok); }
#define PDA_U8_SOURCE_FINISH                                                                 \
    typedef uint2 Raw;                                                                       \
    __device__ __forceinline__ static uint4 finish(Raw v, bool ok) {                         \
        uint2 lo = u8x4_to_bf16x4(ok ? v.x : 0u), hi = u8x4_to_bf16x4(ok ? v.y : 0u);        \
        return make_uint4(lo.x, lo.y, hi.x, hi.y);                                           \
    }
// raw_at(ctx, k, lim, ok, off) issues the load `off` elements past the operand's base (the lo plane of a split
// operand, SrcSplit below); raw() is the plain form.  bias_lim(): reduction indices below it count towards the
// bias gradient (column sums of the A operand) -- everything, for a plain source.
#define PDA_SOURCE_RAW                                                                       \
    __device__ __forceinline__ Raw raw(const Ctx& c, int k, int lim, bool& ok) const {       \
        return raw_at(c, k, lim, ok, 0);                                                     \
    }                                                                                        \
    __device__ __forceinline__ int bias_lim() const { return 0x7fffffff; }
struct SrcDenseK {      // element(row, k) = p[row*ld + k]; K-contiguous
    static constexpr bool kT = false;
    const bf16_t* p; long ld; int rows; int red;
    struct Ctx { const bf16_t* base; bool ok; };
    __device__ __forceinline__ Ctx prep(int row) const { return {p + (long)row * ld, row < rows}; }
    PDA_BF16_SOURCE_FINISH
    __device__ __forceinline__ Raw raw_at(const Ctx& c, int k, int lim, bool& ok, long off) const {
        ok = c.ok && k < lim;
        return *reinterpret_cast<const uint4*>((ok ? c.base + k : p) + off);
    }
    PDA_SOURCE_RAW
};

struct SrcDenseT {      // element(row, k) = p[k*ld + row]; row-contiguous
    static constexpr bool kT = true;
    const bf16_t* p; long ld; int rows; int red;
    struct Ctx { const bf16_t* base; bool ok; };
    __device__ __forceinline__ Ctx prep(int row0) const { return {p + row0, row0 < rows}; }
    PDA_BF16_SOURCE_FINISH
    __device__ __forceinline__ Raw raw_at(const Ctx& c, int k, int lim, bool& ok, long off) const {
        ok = c.ok && k < lim;
        return *reinterpret_cast<const uint4*>((ok ? c.base + (long)k * ld : p) + off);
    }
    PDA_SOURCE_RAW
};

// NHWC bf16 activations viewed as im2col rows: row m = (b, oy, ox), k = (ky, kx, c)
// Exact n / d by one multiply-high: m = ceil(2^32 / d), valid while n * d < 2^32 (row and reduction indices of
// every launch here: the launchers check).  The loaders used to divide at run time (v_rcp + ~25 VALU
// instructions per quotient, two quotients per 16-byte load): at batch 128 the gather-form data gradients
// spent their time on that arithmetic, not on memory or MFMA (profiles/r2_b128_fp32x3_v3_*: 3.5 % MFMA busy,
// 34 % wait, ~1000 instructions per k-tile and wave).
__host__ __device__ inline unsigned fastdiv_magic(int d) {
    return d <= 1 ? 0u : (unsigned)(((1ull << 32) + (unsigned)d - 1) / (unsigned)d);
}
// d == 1 has no 32-bit magic (m == 0 there): its quotient is added in arithmetically -- a uniform branch per
// division otherwise
__device__ __forceinline__ int fdiv(int n, int /*d*/, unsigned m) {
    const unsigned one = m == 0u ? 0xffffffffu : 0u;
    return (int)(__umulhi((unsigned)n, m) + ((unsigned)n & one));
}
// e when ok, else 0, as arithmetic: `ok ? e : 0` compiles to a divergent branch around e's multiplications
__device__ __forceinline__ int keep_if(int e, bool ok) { return e & -(int)ok; }

struct ConvGeom { int H, W, C, KH, KW, S, OH, OW; unsigned m_run, m_ohw, m_ow; };
// the launchers' guard for fdiv's validity range (largest index the loaders form x divisor < 2^32)
inline void check_fastdiv(long n, long d, const char* what) {
    if ((unsigned long long)n * (unsigned long long)d >= (1ull << 32)) {
        fprintf(stderr, "pda: %s: index range %ld x divisor %ld exceeds the multiply-high division's bound (2^32)\n",
                what, n, d);
        abort();
    }
}
inline ConvGeom make_geom(int H, int W, int C, int KH, int KW, int S) {
    ConvGeom g;
    g.H = H; g.W = W; g.C = C; g.KH = KH; g.KW = KW; g.S = S;
    g.OH = (H - KH) / S + 1;
    g.OW = (W - KW) / S + 1;
    g.m_run = fastdiv_magic(KW * C); g.m_ohw = fastdiv_magic(g.OH * g.OW); g.m_ow = fastdiv_magic(g.OW);
    check_fastdiv((long)C * KH * KW + 4 * GEMM_BK, (long)KW * C, "conv geometry (reduction index)");
    return g;
}

struct SrcIm2colK {
    static constexpr bool kT = false;
    const bf16_t* p; ConvGeom g; int rows; int red;
    struct Ctx { const bf16_t* base; bool ok; };
    __device__ __forceinline__ Ctx prep(int m) const {
        bool ok = m < rows;
        m = ok ? m : 0;
        int ohw = g.OH * g.OW;
        int b = fdiv(m, ohw, g.m_ohw), r = m - b * ohw;
        int oy = fdiv(r, g.OW, g.m_ow), ox = r - oy * g.OW;
        return {p + (((long)b * g.H + oy * g.S) * g.W + ox * g.S) * g.C, ok};
    }
    PDA_BF16_SOURCE_FINISH
    // branch-free: an out-of-range lane reads its row's first 16 bytes (row 0's for a row past the end)
    __device__ __forceinline__ Raw raw_at(const Ctx& c, int k, int lim, bool& ok, long poff) const {
        const bool kin = k < lim;
        ok = c.ok & kin;
        int run = g.KW * g.C;
        int ky = fdiv(k, run, g.m_run), off = k - ky * run;
        int e = keep_if(ky * (g.W * g.C) + off, kin);
        return *reinterpret_cast<const uint4*>(c.base + poff + e);
    }
    PDA_SOURCE_RAW
};

struct SrcIm2colT {     // rows = k-columns of the im2col matrix (8 consecutive), reduction = m
    static constexpr bool kT = true;
    const bf16_t* p; ConvGeom g; int rows; int red;
    struct Ctx { int off; bool ok; };
    __device__ __forceinline__ Ctx prep(int kc0) const {
        bool ok = kc0 < rows;
        kc0 = ok ? kc0 : 0;
        int run = g.KW * g.C;
        int ky = fdiv(kc0, run, g.m_run), off = kc0 - ky * run;
        return {ky * g.W * g.C + off, ok};
    }
    PDA_BF16_SOURCE_FINISH
    __device__ __forceinline__ Raw raw_at(const Ctx& c, int m, int lim, bool& ok, long poff) const {
        ok = c.ok & (m < lim);
        m = ok ? m : 0;
        int ohw = g.OH * g.OW;
        int b = fdiv(m, ohw, g.m_ohw), r = m - b * ohw;
        int oy = fdiv(r, g.OW, g.m_ow), ox = r - oy * g.OW;
        int e = ((b * g.H + oy * g.S) * g.W + ox * g.S) * g.C + c.off;      // < 2^31 elements (launcher check)
        return *reinterpret_cast<const uint4*>(p + poff + keep_if(e, ok));
    }
    PDA_SOURCE_RAW
};

// ---- conv data gradient as a gather (no dcol matrix, no col2im pass) ------------------------
// dx[b,iy,ix,c] = sum_{ky,kx,o} dy[b,(iy-ky)/S,(ix-kx)/S,o] * w[o][ky][kx][c] over the taps with
// ky = iy mod S (+ S a), kx = ix mod S (+ S b).  Input pixels of one parity class (py,px) all use the
// same TH x TW = (KH/S) x (KW/S) taps, so each class is one GEMM (a problem group of the launch):
//   rows m = (b, iy2, ix2) with iy = S*iy2 + py, ix = S*ix2 + px;   k = (a, b, o);   N = C.
struct DgradGeom { int H, W, C, O, KH, KW, S, OH, OW, HC, WC, TH, TW, py, px; unsigned m_hw, m_wc, m_o, m_tw; };
inline DgradGeom make_dgrad_geom(const ConvGeom& cg, int O, int py, int px) {
    DgradGeom g;
    g.H = cg.H; g.W = cg.W; g.C = cg.C; g.O = O; g.KH = cg.KH; g.KW = cg.KW; g.S = cg.S; g.OH = cg.OH; g.OW = cg.OW;
    g.HC = cg.H / cg.S; g.WC = cg.W / cg.S; g.TH = cg.KH / cg.S; g.TW = cg.KW / cg.S; g.py = py; g.px = px;
    g.m_hw = fastdiv_magic(g.HC * g.WC); g.m_wc = fastdiv_magic(g.WC); g.m_o = fastdiv_magic(O);
    g.m_tw = fastdiv_magic(g.TW);
    return g;
}

struct SrcDgradDyK {    // A operand: element(m, k) = dy[b, iy2 - a, ix2 - b, o]  (zero outside the output grid)
    static constexpr bool kT = false;
    const bf16_t* p; DgradGeom g; int rows; int red;
    struct Ctx { const bf16_t* base; int iy2, ix2; bool ok; };
    __device__ __forceinline__ Ctx prep(int m) const {
        int hw = g.HC * g.WC;
        bool ok = m < rows;
        m = ok ? m : 0;
        int b = fdiv(m, hw, g.m_hw);
        int r = m - b * hw;
        int iy2 = fdiv(r, g.WC, g.m_wc), ix2 = r - iy2 * g.WC;
        return {p + (long)b * g.OH * g.OW * g.O, iy2, ix2, ok};
    }
    PDA_BF16_SOURCE_FINISH
    __device__ __forceinline__ Raw raw_at(const Ctx& c, int k, int lim, bool& ok, long poff) const {
        int tap = fdiv(k, g.O, g.m_o), o = k - tap * g.O;
        int a = fdiv(tap, g.TW, g.m_tw), bb = tap - a * g.TW;
        int oy = c.iy2 - a, ox = c.ix2 - bb;
        ok = c.ok & (k < lim) & ((unsigned)oy < (unsigned)g.OH) & ((unsigned)ox < (unsigned)g.OW);
        int e = keep_if((oy * g.OW + ox) * g.O + o, ok);
        return *reinterpret_cast<const uint4*>(c.base + poff + e);
    }
    PDA_SOURCE_RAW
};

struct SrcDgradWT {     // B operand: element(row = c, k = (a,b,o)) = w[o][(ky,kx,c)], 8 consecutive c per load
    static constexpr bool kT = true;
    const bf16_t* p; DgradGeom g; int rows; int red;
    struct Ctx { const bf16_t* base; bool ok; };
    __device__ __forceinline__ Ctx prep(int c0) const { bool ok = c0 < rows; return {p + (ok ? c0 : 0), ok}; }
    PDA_BF16_SOURCE_FINISH
    __device__ __forceinline__ Raw raw_at(const Ctx& c, int k, int lim, bool& ok, long poff) const {
        ok = c.ok & (k < lim);
        int tap = fdiv(k, g.O, g.m_o), o = k - tap * g.O;
        int a = fdiv(tap, g.TW, g.m_tw), bb = tap - a * g.TW;
        int ky = g.py + g.S * a, kx = g.px + g.S * bb;
        int e = keep_if(o * (g.KH * g.KW * g.C) + (ky * g.KW + kx) * g.C, ok);
        return *reinterpret_cast<const uint4*>(c.base + poff + e);
    }
    PDA_SOURCE_RAW
};

// uint8 planar frames (B rows of C*H*W bytes, gathered through idx when non-null):
// row m = (b, oy, ox), k = (c, ky, kx) -- PyTorch's own conv-weight order, KW == 8.
struct SrcFrameU8K {
    static constexpr bool kT = false;
    const uint8_t* p; const int* idx; long frame_bytes; ConvGeom g; int rows; int red;
    struct Ctx { const uint8_t* base; bool ok; };
    __device__ __forceinline__ Ctx prep(int m) const {
        int ohw = g.OH * g.OW;
        bool ok = m < rows;
        int b = ok ? m / ohw : 0;
        int r = m - b * ohw;
        int oy = r / g.OW, ox = r - oy * g.OW;
        long f = idx ? (long)idx[b] : (long)b;
        return {p + f * frame_bytes + (long)(oy * g.S) * g.W + ox * g.S, ok};
    }
    PDA_U8_SOURCE_FINISH
    PDA_SOURCE_RAW
    __device__ __forceinline__ Raw raw_at(const Ctx& c, int k, int lim, bool& ok, long) const {
        ok = c.ok && k < lim;
        int khw = g.KH * g.KW;
        int ch = k / khw, r = k - ch * khw;
        int ky = r / g.KW;      // kx == 0: k is a multiple of 8 == KW
        const uint8_t* src = ok ? c.base + ((long)ch * g.H + ky) * g.W : p;
        return make_uint2(*reinterpret_cast<const uint32_t*>(src), *reinterpret_cast<const uint32_t*>(src + 4));
    }
};

struct SrcFrameU8T {    // rows = (c, ky, kx0..kx0+7), reduction = m
    static constexpr bool kT = true;
    const uint8_t* p; const int* idx; long frame_bytes; ConvGeom g; int rows; int red;
    struct Ctx { long off; bool ok; };
    __device__ __forceinline__ Ctx prep(int kc0) const {
        int khw = g.KH * g.KW;
        int ch = kc0 / khw, r = kc0 - ch * khw;
        int ky = r / g.KW;
        return {((long)ch * g.H + ky) * g.W, kc0 < rows};
    }
    PDA_U8_SOURCE_FINISH
    PDA_SOURCE_RAW
    __device__ __forceinline__ Raw raw_at(const Ctx& c, int m, int lim, bool& ok, long) const {
        ok = c.ok && m < lim;
        int mm = ok ? m : 0;
        int ohw = g.OH * g.OW;
        int b = mm / ohw, r = mm - b * ohw;
        int oy = r / g.OW, ox = r - oy * g.OW;
        long f = idx ? (long)idx[b] : (long)b;
        const uint8_t* src = ok ? p + f * frame_bytes + (long)(oy * g.S) * g.W + ox * g.S + c.off : p;
        return make_uint2(*reinterpret_cast<const uint32_t*>(src), *reinterpret_cast<const uint32_t*>(src + 4));
    }
};

// ---- split-bf16 operands: fp32-equivalent products on the bf16 matrix cores ---------------------------
// A logical fp32 tensor x is kept as two bf16 planes, hi = bf16(x) at the source's base and lo = bf16(x - hi)
// `plane` elements further on (16 significand bits together).  A product over a reduction of length red1 then
// runs as ONE reduction of length NT * red1 in which operand copy t reads the lo plane iff bit t of LOMASK is set:
//     A = (hi, lo, hi), B = (hi, hi, lo):   sum a_hi b_hi + a_lo b_hi + a_hi b_lo     (NT = 3; the a_lo b_lo term,
//                                            2^-18 relative, is dropped; accumulation is the MFMA's fp32)
//     one operand exact in bf16 (uint8 frames): (x, x) against (hi, lo)               (NT = 2)
// so every kernel of the family serves the fp32-precision learner unchanged.  Needs red1 % 8 == 0 for
// K-contiguous sources (a 16-byte load must not straddle two copies).
template <class S, int NT, int LOMASK>
struct SrcSplit {
    static constexpr bool kT = S::kT;
    S s; long plane; int red1;
    typedef typename S::Ctx Ctx;
    typedef typename S::Raw Raw;
    __device__ __forceinline__ Ctx prep(int row) const { return s.prep(row); }
    __device__ __forceinline__ static uint4 finish(Raw v, bool ok) { return S::finish(v, ok); }
    __device__ __forceinline__ Raw raw(const Ctx& c, int k, int lim, bool& ok) const {
        int t = (k >= red1 ? 1 : 0) + ((NT > 2 && k >= 2 * red1) ? 1 : 0);
        int kk = k - t * red1;
        bool lo = (LOMASK >> t) & 1;
        return s.raw_at(c, kk, (k < lim && kk < red1) ? red1 : 0, ok, lo ? plane : 0);
    }
    // column sums of a split A operand = sum over its (hi, lo) copies, which are always the first two
    __device__ __forceinline__ int bias_lim() const { return 2 * red1; }
};
#define SPLIT_A3(S) SrcSplit<S, 3, 2>      /* (hi, lo, hi) */
#define SPLIT_B3(S) SrcSplit<S, 3, 4>      /* (hi, hi, lo) */
#define SPLIT_X2(S) SrcSplit<S, 2, 0>      /* exact operand, twice */
#define SPLIT_HL2(S) SrcSplit<S, 2, 2>     /* (hi, lo) against an exact operand */

// One plane of a split operand as a plain source (the x3 body below stages hi and lo tiles side by side).
template <class S>
struct SrcPlane {
    static constexpr bool kT = S::kT;
    S s; long off;
    typedef typename S::Ctx Ctx;
    typedef typename S::Raw Raw;
    __device__ __forceinline__ Ctx prep(int row) const { return s.prep(row); }
    __device__ __forceinline__ static uint4 finish(Raw v, bool ok) { return S::finish(v, ok); }
    __device__ __forceinline__ Raw raw(const Ctx& c, int k, int lim, bool& ok) const { return s.raw_at(c, k, lim, ok, off); }
};
template <class T> struct SplitTraits { static constexpr bool kSplit = false; static constexpr bool kHasLo = false; typedef T Base; };
template <class S, int NT, int LOMASK>
struct SplitTraits<SrcSplit<S, NT, LOMASK>> {
    static constexpr bool kSplit = true;
    static constexpr bool kHasLo = LOMASK != 0;     // X2 (exact operand, e.g. uint8 frames): one plane only
    typedef S Base;
};

// hi / lo planes of an fp32 value (lo written `plane` elements after hi; plane == 0: plain bf16 output)
__device__ __forceinline__ void store_split(bf16_t* dst, long plane, float x) {
    bf16_t h = f2bf(x);
    *dst = h;
    if (plane) dst[plane] = f2bf(x - bf2f(h));
}

// ----------------------------------------------------------------------------
// epilogues: operator()(m, n, v) receives rows m..m+3 of column n
// ----------------------------------------------------------------------------
__device__ __forceinline__ float apply_act(float v, int act) {
    if (act == ACT_RELU) return fmaxf(v, 0.f);
    if (act == ACT_TANH) return tanhf(v);
    return v;
}

struct EpiBiasAct {     // out = act(acc*scale + bias[n]) -> bf16 (and/or fp32) row-major
    bf16_t* out; float* out32; const float* bias; float scale; int act; long ld; long ld32; long plane;
    static constexpr bool kBiasGrad = false;
    __device__ __forceinline__ void operator()(int m, int n, f32x4 v, int M, int N) const {
        if (n >= N) return;
        float b = bias ? bias[n] : 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (m + r < M) {
                float x = apply_act(v[r] * scale + b, act);
                if (out) store_split(out + (long)(m + r) * ld + n, plane, x);
                if (out32) out32[(long)(m + r) * ld32 + n] = x;
            }
        }
    }
    __device__ __forceinline__ void bias_grad(int, float, int) const {}
};

struct EpiAtomicF32 {   // split-K partial sums: atomicAdd acc into fp32 out[m*ld + n]
    float* out; long ld; float scale;
    static constexpr bool kBiasGrad = false;
    __device__ __forceinline__ void operator()(int m, int n, f32x4 v, int M, int N) const {
        if (n >= N) return;
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (m + r < M) atomicAdd(&out[(long)(m + r) * ld + n], v[r] * scale);
    }
    __device__ __forceinline__ void bias_grad(int, float, int) const {}
};

// split-K partial sums with plain stores: slice y of the reduction writes its tile to out + y * slice_stride.
// The consumer adds the slices in a fixed order, so the result is bit-reproducible (fp32 atomics arrive in any
// order) and nothing has to be zeroed between steps.
struct EpiSlabF32 {
    float* out; long ld; long slice_stride;
    static constexpr bool kBiasGrad = false;
    __device__ __forceinline__ void operator()(int m, int n, f32x4 v, int M, int N) const {
        if (n >= N) return;
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (m + r < M) out[(long)(m + r) * ld + n] = v[r];
    }
    __device__ __forceinline__ void bias_grad(int, float, int) const {}
};

struct EpiMaskBf16 {    // dgrad: out = acc * act'(y[m][n]) -> bf16 ; y is the forward output
    bf16_t* out; const bf16_t* y; int act; long ld; long ldy; long plane;
    static constexpr bool kBiasGrad = false;
    __device__ __forceinline__ void operator()(int m, int n, f32x4 v, int M, int N) const {
        if (n >= N) return;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (m + r < M) {
                float g = v[r];
                if (act != ACT_NONE) {
                    float yy = bf2f(y[(long)(m + r) * ldy + n]);
                    g = (act == ACT_RELU) ? (yy > 0.f ? g : 0.f) : g * (1.f - yy * yy);
                }
                store_split(out + (long)(m + r) * ld + n, plane, g);
            }
        }
    }
    __device__ __forceinline__ void bias_grad(int, float, int) const {}
    // columns n .. n + 7 of row m (the staged epilogue, below): one 16-byte load of y and one 16-byte store per
    // plane when the row pieces are 16-byte aligned, element by element otherwise
    __device__ __forceinline__ void store8(int m, int n, f32x4 a, f32x4 b, int M, int N) const {
        if (m >= M || n >= N) return;
        float g[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
        bf16_t* o = out + (long)m * ld + n;
        const bf16_t* yp = y + (long)m * ldy + n;
        const bool vec = n + 8 <= N && ((ld | plane) & 7) == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0 &&
                         (act == ACT_NONE || ((ldy & 7) == 0 && (reinterpret_cast<uintptr_t>(y) & 15) == 0));
        if (vec) {
            if (act != ACT_NONE) {
                const uint4 yv = *reinterpret_cast<const uint4*>(yp);
                const uint32_t yw[4] = {yv.x, yv.y, yv.z, yv.w};
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float yy = bf2f((bf16_t)((yw[e >> 1] >> (16 * (e & 1))) & 0xffffu));
                    g[e] = (act == ACT_RELU) ? (yy > 0.f ? g[e] : 0.f) : g[e] * (1.f - yy * yy);
                }
            }
            uint32_t hw[4], lw[4];
#pragma unroll
            for (int e = 0; e < 8; e += 2) {
                hw[e >> 1] = pack2bf(g[e], g[e + 1]);
                lw[e >> 1] = pack2bf(g[e] - bf2f((bf16_t)(hw[e >> 1] & 0xffffu)), g[e + 1] - bf2f((bf16_t)(hw[e >> 1] >> 16)));
            }
            *reinterpret_cast<uint4*>(o) = make_uint4(hw[0], hw[1], hw[2], hw[3]);
            if (plane) *reinterpret_cast<uint4*>(o + plane) = make_uint4(lw[0], lw[1], lw[2], lw[3]);
            return;
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            if (n + e < N) {
                float gg = g[e];
                if (act != ACT_NONE) {
                    const float yy = bf2f(yp[e]);
                    gg = (act == ACT_RELU) ? (yy > 0.f ? gg : 0.f) : gg * (1.f - yy * yy);
                }
                store_split(o + e, plane, gg);
            }
        }
    }
};

// conv dgrad (gather form): row m of a parity class -> NHWC pixel row; out = acc * act'(y[pixel][n])
struct EpiDgradScatter {
    bf16_t* out; const bf16_t* y; int act; DgradGeom g; long plane;
    static constexpr bool kBiasGrad = false;
    __device__ __forceinline__ void operator()(int m, int n, f32x4 v, int M, int N) const {
        if (n >= N) return;
        int hw = g.HC * g.WC;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            int mm = m + r;
            if (mm < M) {
                int b = fdiv(mm, hw, g.m_hw), q = mm - b * hw;
                int iy2 = fdiv(q, g.WC, g.m_wc), ix2 = q - iy2 * g.WC;
                long pix = ((long)b * g.H + (g.S * iy2 + g.py)) * g.W + (g.S * ix2 + g.px);
                float gr = v[r];
                if (act != ACT_NONE) {
                    float yy = bf2f(y[pix * g.C + n]);
                    gr = (act == ACT_RELU) ? (yy > 0.f ? gr : 0.f) : gr * (1.f - yy * yy);
                }
                store_split(out + pix * g.C + n, plane, gr);
            }
        }
    }
    __device__ __forceinline__ void bias_grad(int, float, int) const {}
};

// weight gradient: plain stores of this block's split-K partial tile into its slab slice
// (epi_for_slice below points gw / gb at it), in GEMM (kernel) layout [m][n].  A single-slice launch writes the gradient
// buffer itself; multi-slice launches write a scratch slab that wgrad_reduce_kernel sums.
// (fp32 atomics execute memory-side on MI355X and the PyTorch-layout scatter made every
// lane its own 64-B request: 134 us per conv wgrad in the first profile.)
struct EpiWgrad {
    float* gw; float* gb; long ld; long slice_stride; long bslice_stride; float scale;
    static constexpr bool kBiasGrad = true;
    __device__ __forceinline__ void operator()(int m, int n, f32x4 v, int M, int N) const {
        if (n >= N) return;
        float* dst = gw;
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (m + r < M) dst[(long)(m + r) * ld + n] = v[r] * scale;
    }
    __device__ __forceinline__ void bias_grad(int m, float v, int M) const {
        if (gb && m < M) gb[m] = v;
    }
};
// the epilogue of split-K slice `by` (the slice index is an argument, not blockIdx.y: launches that carry
// several kernels' workgroups decode their own linear block index)
template <class EP> __device__ __forceinline__ EP epi_for_slice(const EP& e, int) { return e; }
__device__ __forceinline__ EpiSlabF32 epi_for_slice(const EpiSlabF32& e, int by) {
    EpiSlabF32 r = e;
    r.out += (long)by * e.slice_stride;
    return r;
}
__device__ __forceinline__ EpiWgrad epi_for_slice(const EpiWgrad& e, int by) {
    EpiWgrad r = e;
    r.gw += (long)by * e.slice_stride;
    if (r.gb) r.gb += (long)by * e.bslice_stride;
    return r;
}

// Epilogues with a store8() leave the tile through LDS: the accumulators (MFMA C layout: a lane holds 4 rows of ONE
// column, so a direct store is 2 bytes per lane in 32-byte runs, and the mask read the same) are parked as fp32
// [BM][BN + 4] in the k-loop's LDS and come back as 8 consecutive columns per lane -- 16-byte loads of y, 16-byte stores
// per plane, 128 bytes per row and instruction.  (fc1's data gradient at batch 4096 writes 51-77 MB through this
// epilogue.)  Used when the parked tile fits the kernel's LDS.
template <class EP> struct EpiTraits { static constexpr bool kStaged = false; };
template <> struct EpiTraits<EpiMaskBf16> { static constexpr bool kStaged = true; };

template <int BM, int BN, int WM, int WN, int LDS_BYTES, class EP>
constexpr bool epi_staged() {
    return EpiTraits<EP>::kStaged && BN % 8 == 0 && (GEMM_THREADS % (BN / 8)) == 0 &&
           BM % (GEMM_THREADS / (BN / 8)) == 0 && BM * (BN + 4) * 4 <= LDS_BYTES;
}

template <int BM, int BN, int WM, int WN, int FM, int FN, class EP>
__device__ __forceinline__ void staged_epilogue(const f32x4 (&acc)[FM][FN], bf16_t* lds, const EP& ep, int m0, int n0,
                                                int M, int N) {
    constexpr int P = BN + 4, TPR = BN / 8, RPP = GEMM_THREADS / TPR;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    float* st = reinterpret_cast<float*>(lds);
    __syncthreads();            // every wave is done with the last k-tile
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                st[(wm * (BM / WM) + i * 16 + (lane >> 4) * 4 + r) * P + wn * (BN / WN) + j * 16 + (lane & 15)] = acc[i][j][r];
    __syncthreads();
#pragma unroll
    for (int r0 = 0; r0 < BM; r0 += RPP) {
        const int row = r0 + tid / TPR, c = (tid % TPR) * 8;
        const f32x4 a = *reinterpret_cast<const f32x4*>(st + row * P + c);
        const f32x4 b = *reinterpret_cast<const f32x4*>(st + row * P + c + 4);
        ep.store8(m0 + row, n0 + c, a, b, M, N);
    }
}

// ----------------------------------------------------------------------------
// kernel
// ----------------------------------------------------------------------------
template <class SA, class SB, class EP>
struct GemmGroup { SA a; SB b; EP e; int M, N, K; };

template <class SA, class SB, class EP>
struct GemmArgs { GemmGroup<SA, SB, EP> g[GEMM_MAX_GROUPS]; int ksplit; int tiles_n; };

// ---- LDS tile layouts ---------------------------------------------------------------
// N mode: [ROWS][BK] bf16, 16-byte chunk c of row r stored at chunk position c ^ swz(r).
//   ds_read_b128 is serviced in four 16-lane groups that mix rows {0-3,12-15} at one chunk
//   with rows {4-11} at the neighbouring chunk; the swizzles below make every group touch 16
//   distinct 16-byte slots of the 256-byte bank row.
// T mode: [BK][ROWS] bf16, 8-byte chunk j of k-row kk stored at j ^ (h(kk) << 2).
//   ds_read_b64_tr_b16 is serviced per 32-lane half = 8 k-rows x 4 chunks; h() spreads the
//   k-rows that share banks over different 4-chunk groups.
template <int BK>
__device__ __forceinline__ int swz_n(int r) {
    if (BK == 32) return (0x78 >> (2 * ((r >> 2) & 3))) & 3;       // {0,2,3,1}[(r>>2)&3]
    int t = (r >> 1) & 7;                                            // BK == 64
    return ((t & 1) << 1) | ((t >> 2) << 2);
}
template <int ROWS>
__device__ __forceinline__ int swz_t(int kk) {
    if (ROWS == 32) return ((kk >> 3) & 1) << 2;
    if (ROWS == 64) return (((kk >> 1) & 1) | (((kk >> 3) & 1) << 1)) << 2;
    // ROWS == 128: a k-row is one whole 256-byte bank row, so the 8 k-rows of a 32-lane half (k & 3, bit 3 of k)
    // all start at bank 0: each goes to a 4-chunk group of its own
    return ((kk & 3) | (((kk >> 3) & 1) << 2)) << 2;
}

template <bool T, int ROWS, int BK>
struct LdsShape {
    static constexpr int kElems = ROWS * BK;
};

template <bool T, int ROWS, int BK, class SRC>
struct Stager {
    static_assert(BK == 32 || BK == 64, "k-tile of 32 or 64");
    static_assert(ROWS == 32 || ROWS == 64 || ROWS == 128, "tile side of 32, 64 or 128");
    static constexpr int kItems = ROWS * BK / 8;
    static constexpr int kIters = (kItems + GEMM_THREADS - 1) / GEMM_THREADS;
    typename SRC::Ctx ctx[kIters];
    int koff[kIters];
    int lds_off[kIters];
    typename SRC::Raw reg[GEMM_PF][kIters];  // ring of prefetched k-tiles (slot indices are compile-time after unrolling)
    bool okr[GEMM_PF][kIters];

    __device__ __forceinline__ void init(const SRC& s, int row0, int tid) {
#pragma unroll
        for (int i = 0; i < kIters; ++i) {
            int item = tid + i * GEMM_THREADS;
            if (!T) {
                int r = item / (BK / 8), c = item % (BK / 8);
                ctx[i] = s.prep(row0 + r);
                koff[i] = c * 8;
                lds_off[i] = r * BK + ((c ^ swz_n<BK>(r)) << 3);
            } else {
                int rg = item % (ROWS / 8), kk = item / (ROWS / 8);
                ctx[i] = s.prep(row0 + rg * 8);
                koff[i] = kk;
                lds_off[i] = kk * ROWS + (((2 * rg) ^ swz_t<ROWS>(kk)) << 2);
            }
            if (item >= kItems) ctx[i].ok = false;
        }
    }
    // always issues kIters loads (lanes / tiles past `lim` read a dummy address and are zeroed at
    // commit), so every path through the k-loop has the same number of loads in flight and the
    // compiler can wait with exact vmcnt counts instead of draining the ring
    template <int SLOT>
    __device__ __forceinline__ void fetch(const SRC& s, int k0, int lim) {
#pragma unroll
        for (int i = 0; i < kIters; ++i) reg[SLOT][i] = s.raw(ctx[i], k0 + koff[i], lim, okr[SLOT][i]);
    }
    template <int SLOT>
    __device__ __forceinline__ void commit(bf16_t* lds, int tid) const {
#pragma unroll
        for (int i = 0; i < kIters; ++i)
            if (kItems % GEMM_THREADS == 0 || tid + i * GEMM_THREADS < kItems)
                *reinterpret_cast<uint4*>(lds + lds_off[i]) = SRC::finish(reg[SLOT][i], okr[SLOT][i]);
    }
};

// fragment for rows r0..r0+15 (lane&15), k = 32*ks + 8*(lane>>4) .. +7 of the current tile
template <bool T, int ROWS, int BK>
__device__ __forceinline__ bf16x8 read_frag(const bf16_t* lds, int r0, int ks, int lane) {
    if (!T) {
        int r = r0 + (lane & 15), c = ks * 4 + (lane >> 4);
        const bf16_t* p = lds + r * BK + ((c ^ swz_n<BK>(r)) << 3);
        return *reinterpret_cast<const bf16x8*>(p);
    } else {
        // 16-lane group g reads k = 8g..8g+7 as two 4(k) x 16(row) blocks; lane 4q+p of the
        // group supplies the address of block row q, columns 4p..4p+3 and receives column
        // (lane&15) of the 4 rows.
        int g = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
        int kk = ks * 32 + 8 * g + q, j = (r0 >> 2) + pp;
        const bf16_t* p0 = lds + kk * ROWS + ((j ^ swz_t<ROWS>(kk)) << 2);
        const bf16_t* p1 = lds + (kk + 4) * ROWS + ((j ^ swz_t<ROWS>(kk + 4)) << 2);
        typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
        s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p0));
        s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p1));
        typedef __attribute__((ext_vector_type(8))) short s16x8;
        s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        return __builtin_bit_cast(bf16x8, v);
    }
}

template <int BM, int BN, int BK, class SA, class SB>
constexpr int gemm_lds_elems() {
    return 2 * (LdsShape<SA::kT, BM, BK>::kElems * (SplitTraits<SA>::kHasLo ? 2 : 1) +
                LdsShape<SB::kT, BN, BK>::kElems * (SplitTraits<SB>::kHasLo ? 2 : 1));
}

template <int BM, int BN, int WM, int WN, int BK, class SA, class SB, class EP>
__device__ __forceinline__ void gemm_body_x3(const GemmArgs<SA, SB, EP>& args, int bx, int by, int bz, bf16_t* lds);

// The work of one workgroup.  (bx, by, bz) = (tile, k-slice, problem group): blockIdx for a plain launch, a
// decoded linear index when two GEMMs share one launch (gemm_duo_kernel).
template <int BM, int BN, int WM, int WN, int BK, class SA, class SB, class EP>
__device__ __forceinline__ void gemm_body(const GemmArgs<SA, SB, EP>& args, int bx, int by, int bz, bf16_t* lds) {
    static_assert(WM * WN == 4, "4 waves per block");
    if constexpr (SplitTraits<SA>::kSplit || SplitTraits<SB>::kSplit) {
        gemm_body_x3<BM, BN, WM, WN, BK, SA, SB, EP>(args, bx, by, bz, lds);
        return;
    } else {
    constexpr int FM = BM / WM / 16, FN = BN / WN / 16;
    static_assert(FM >= 1 && FN >= 1, "tile too small for the wave grid");
    constexpr bool TA = SA::kT, TB = SB::kT;
    constexpr int A_ELEMS = LdsShape<TA, BM, BK>::kElems, B_ELEMS = LdsShape<TB, BN, BK>::kElems;

    const GemmGroup<SA, SB, EP>& G = args.g[bz];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int tile_m = bx / args.tiles_n, tile_n = bx % args.tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    if (m0 >= G.M || n0 >= G.N) return;     // groups may differ in size
    int kbeg = 0, kend = G.K;
    if (args.ksplit > 0) {
        kbeg = by * args.ksplit;
        kend = min(G.K, kbeg + args.ksplit);
        if (kbeg >= kend) return;
    }

    Stager<TA, BM, BK, SA> sa;
    Stager<TB, BN, BK, SB> sb;
    sa.init(G.a, m0, tid);
    sb.init(G.b, n0, tid);

    f32x4 acc[FM][FN];
    f32x4 accb[FM];
#pragma unroll
    for (int i = 0; i < FM; ++i) {
        accb[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    const bool do_bias = EP::kBiasGrad && tile_n == 0 && wn == 0;
    const int bias_lim = G.a.bias_lim();        // INT_MAX for plain operands: `ones` below folds to a constant

    // The loads of k-tiles t+1 .. t+GEMM_PF are in flight while tile t is multiplied: at batch 128 these
    // launches are a few hundred blocks of 8-16 k-tiles, i.e. bound by memory latency, not by MFMA or
    // LDS rate (counter profile: 58 % of wave cycles parked in s_waitcnt with one tile of prefetch).
    auto tile = [&](auto slot_tag, auto fetch_tag, int kk, int& buf) {
        constexpr int SLOT = decltype(slot_tag)::value;
        bf16_t* la = lds + buf * (A_ELEMS + B_ELEMS);
        bf16_t* lb = la + A_ELEMS;
        sa.template commit<SLOT>(la, tid);
        sb.template commit<SLOT>(lb, tid);
        __syncthreads();
        if constexpr (decltype(fetch_tag)::value) {
            sa.template fetch<SLOT>(G.a, kk + GEMM_PF * BK, kend);
            sb.template fetch<SLOT>(G.b, kk + GEMM_PF * BK, kend);
        }
#pragma unroll
        for (int ks = 0; ks < BK / 32; ++ks) {
            bf16x8 fa[FM], fb[FN];
            bf16x8 ones;        // B fragment of the bias column sums: k = kk + 32 ks + 8 (lane >> 4) + j
#pragma unroll
            for (int j = 0; j < 8; ++j)
                ones[j] = (kk + ks * 32 + 8 * (lane >> 4) + j < bias_lim) ? (__bf16)1.0f : (__bf16)0.0f;
#pragma unroll
            for (int i = 0; i < FM; ++i) fa[i] = read_frag<TA, BM, BK>(la, wm * (BM / WM) + i * 16, ks, lane);
#pragma unroll
            for (int j = 0; j < FN; ++j) fb[j] = read_frag<TB, BN, BK>(lb, wn * (BN / WN) + j * 16, ks, lane);
#pragma unroll
            for (int i = 0; i < FM; ++i) {
#pragma unroll
                for (int j = 0; j < FN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
                if (do_bias) accb[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], ones, accb[i], 0, 0, 0);
            }
        }
        buf ^= 1;
    };
    static_assert(GEMM_PF == 3, "the ring below is written out for three slots");
    sa.template fetch<0>(G.a, kbeg, kend);
    sb.template fetch<0>(G.b, kbeg, kend);
    sa.template fetch<1>(G.a, kbeg + BK, kend);
    sb.template fetch<1>(G.b, kbeg + BK, kend);
    sa.template fetch<2>(G.a, kbeg + 2 * BK, kend);
    sb.template fetch<2>(G.b, kbeg + 2 * BK, kend);
    GEMM_RING_LOOP

    // C/D layout of 16x16x32: col = lane & 15, row = (lane >> 4) * 4 + reg
    const EP ep = epi_for_slice(G.e, args.ksplit > 0 ? by : 0);
    if constexpr (epi_staged<BM, BN, WM, WN, 2 * 2 * (A_ELEMS + B_ELEMS), EP>()) {
        staged_epilogue<BM, BN, WM, WN, FM, FN>(acc, lds, ep, m0, n0, G.M, G.N);
        return;
    }
#pragma unroll
    for (int i = 0; i < FM; ++i) {
        int m = m0 + wm * (BM / WM) + i * 16 + (lane >> 4) * 4;
#pragma unroll
        for (int j = 0; j < FN; ++j) {
            int n = n0 + wn * (BN / WN) + j * 16 + (lane & 15);
            ep(m, n, acc[i][j], G.M, G.N);
        }
        if (do_bias && (lane & 15) == 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r) ep.bias_grad(m + r, accb[i][r], G.M);
        }
    }
    }   // !split
}

// The same workgroup program for split-bf16 operands (SrcSplit): fp32-equivalent products as
//     a b ~= a_hi b_hi + a_lo b_hi + a_hi b_lo        (one operand exact in bf16: a b_hi + a b_lo)
// with the hi and lo tiles of every k-tile staged ONCE, side by side in LDS, and the two or three MFMAs issued
// per fragment pair.  (Round 2's first fp32-precision path ran the products as one reduction of three times the
// length -- SrcSplit::raw, still used by nothing else -- which fetched and staged every hi tile twice and paid the
// k-loop's barrier + commit three times per logical k-tile; at batch 128 those launches are bound by exactly that.)
// GemmGroup::K, ksplit and the slice layout are in LOGICAL reduction units here.
template <int BM, int BN, int WM, int WN, int BK, class SA, class SB, class EP>
__device__ __forceinline__ void gemm_body_x3(const GemmArgs<SA, SB, EP>& args, int bx, int by, int bz, bf16_t* lds) {
    constexpr int FM = BM / WM / 16, FN = BN / WN / 16;
    static_assert(FM >= 1 && FN >= 1, "tile too small for the wave grid");
    static_assert(SplitTraits<SA>::kSplit && SplitTraits<SB>::kSplit, "both operands are given as SrcSplit");
    typedef typename SplitTraits<SA>::Base BA;
    typedef typename SplitTraits<SB>::Base BB;
    constexpr bool LA = SplitTraits<SA>::kHasLo, LB = SplitTraits<SB>::kHasLo;
    static_assert(LA || LB, "at least one operand has a lo plane");
    constexpr bool TA = BA::kT, TB = BB::kT;
    constexpr int A_ELEMS = LdsShape<TA, BM, BK>::kElems, B_ELEMS = LdsShape<TB, BN, BK>::kElems;
    constexpr int BUF = A_ELEMS * (LA ? 2 : 1) + B_ELEMS * (LB ? 2 : 1);

    const GemmGroup<SA, SB, EP>& G = args.g[bz];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int tile_m = bx / args.tiles_n, tile_n = bx % args.tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    if (m0 >= G.M || n0 >= G.N) return;
    int kbeg = 0, kend = G.K;
    if (args.ksplit > 0) {
        kbeg = by * args.ksplit;
        kend = min(G.K, kbeg + args.ksplit);
        if (kbeg >= kend) return;
    }
    const SrcPlane<BA> ah = {G.a.s, 0}, al = {G.a.s, G.a.plane};
    const SrcPlane<BB> bh = {G.b.s, 0}, bl = {G.b.s, G.b.plane};
    Stager<TA, BM, BK, SrcPlane<BA>> sah, sal;
    Stager<TB, BN, BK, SrcPlane<BB>> sbh, sbl;
    sah.init(ah, m0, tid);
    sbh.init(bh, n0, tid);
    if (LA) sal.init(al, m0, tid);
    if (LB) sbl.init(bl, n0, tid);

    f32x4 acc[FM][FN];
    f32x4 accb[FM];
#pragma unroll
    for (int i = 0; i < FM; ++i) {
        accb[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    const bool do_bias = EP::kBiasGrad && tile_n == 0 && wn == 0;
    bf16x8 ones;
#pragma unroll
    for (int j = 0; j < 8; ++j) ones[j] = (__bf16)1.0f;

    auto fetch_all = [&](auto slot_tag, int kk) {
        constexpr int SLOT = decltype(slot_tag)::value;
        sah.template fetch<SLOT>(ah, kk, kend);
        sbh.template fetch<SLOT>(bh, kk, kend);
        if (LA) sal.template fetch<SLOT>(al, kk, kend);
        if (LB) sbl.template fetch<SLOT>(bl, kk, kend);
    };
    auto tile = [&](auto slot_tag, auto fetch_tag, int kk, int& buf) {
        constexpr int SLOT = decltype(slot_tag)::value;
        bf16_t* lah = lds + buf * BUF;
        bf16_t* lal = lah + A_ELEMS;
        bf16_t* lbh = lah + A_ELEMS * (LA ? 2 : 1);
        bf16_t* lbl = lbh + B_ELEMS;
        sah.template commit<SLOT>(lah, tid);
        sbh.template commit<SLOT>(lbh, tid);
        if (LA) sal.template commit<SLOT>(lal, tid);
        if (LB) sbl.template commit<SLOT>(lbl, tid);
        __syncthreads();
        if constexpr (decltype(fetch_tag)::value) fetch_all(slot_tag, kk + GEMM_PF * BK);
#pragma unroll
        for (int ks = 0; ks < BK / 32; ++ks) {
            bf16x8 fah[FM], fal[FM], fbh[FN], fbl[FN];
#pragma unroll
            for (int i = 0; i < FM; ++i) {
                fah[i] = read_frag<TA, BM, BK>(lah, wm * (BM / WM) + i * 16, ks, lane);
                if (LA) fal[i] = read_frag<TA, BM, BK>(lal, wm * (BM / WM) + i * 16, ks, lane);
            }
#pragma unroll
            for (int j = 0; j < FN; ++j) {
                fbh[j] = read_frag<TB, BN, BK>(lbh, wn * (BN / WN) + j * 16, ks, lane);
                if (LB) fbl[j] = read_frag<TB, BN, BK>(lbl, wn * (BN / WN) + j * 16, ks, lane);
            }
#pragma unroll
            for (int i = 0; i < FM; ++i) {
#pragma unroll
                for (int j = 0; j < FN; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fah[i], fbh[j], acc[i][j], 0, 0, 0);
                    if (LA) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fal[i], fbh[j], acc[i][j], 0, 0, 0);
                    if (LB) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fah[i], fbl[j], acc[i][j], 0, 0, 0);
                }
                if (do_bias) {
                    accb[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fah[i], ones, accb[i], 0, 0, 0);
                    if (LA) accb[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fal[i], ones, accb[i], 0, 0, 0);
                }
            }
        }
        buf ^= 1;
    };
    static_assert(GEMM_PF == 3, "the ring below is written out for three slots");
    fetch_all(std::integral_constant<int, 0>{}, kbeg);
    fetch_all(std::integral_constant<int, 1>{}, kbeg + BK);
    fetch_all(std::integral_constant<int, 2>{}, kbeg + 2 * BK);
    GEMM_RING_LOOP

    const EP ep = epi_for_slice(G.e, args.ksplit > 0 ? by : 0);
    if constexpr (epi_staged<BM, BN, WM, WN, 2 * 2 * BUF, EP>()) {
        staged_epilogue<BM, BN, WM, WN, FM, FN>(acc, lds, ep, m0, n0, G.M, G.N);
        return;
    }
#pragma unroll
    for (int i = 0; i < FM; ++i) {
        int m = m0 + wm * (BM / WM) + i * 16 + (lane >> 4) * 4;
#pragma unroll
        for (int j = 0; j < FN; ++j) {
            int n = n0 + wn * (BN / WN) + j * 16 + (lane & 15);
            ep(m, n, acc[i][j], G.M, G.N);
        }
        if (do_bias && (lane & 15) == 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r) ep.bias_grad(m + r, accb[i][r], G.M);
        }
    }
}

template <int BM, int BN, int WM, int WN, int BK, class SA, class SB, class EP>
__global__ __launch_bounds__(GEMM_THREADS) void gemm_kernel(GemmArgs<SA, SB, EP> args) {
    __shared__ __attribute__((aligned(16))) bf16_t lds[gemm_lds_elems<BM, BN, BK, SA, SB>()];
    gemm_body<BM, BN, WM, WN, BK, SA, SB, EP>(args, blockIdx.x, blockIdx.y, blockIdx.z, lds);
}

// Two independent GEMMs in ONE launch (horizontal fusion): workgroups [0, n1) run the first, the rest the second.
// For pairs of under-filled launches with no dependence on each other (fc1's data gradient: 196 workgroups, and
// the two fc weight gradients) that would otherwise run back to back.  grid1 = (tiles, k-slices, groups) of the
// first problem; the second is decoded the same way from the remaining indices.
// XCD-aware order for un-split problems (grid.y == 1): workgroups go to the 8 XCDs round-robin by linear id; the
// m-tiles of one column block (n-tile, group) -- which all read the same columns of the B operand -- are given ids
// that are equal mod 8, so they share one XCD's L2.  Local id l -> (bx, bz); false = padding workgroup.
__host__ __device__ inline int xcd_cols_blocks(int3 grid, int tiles_n) {
    return (tiles_n * grid.z + 7) / 8 * 8 * (grid.x / tiles_n);
}
__device__ __forceinline__ bool xcd_cols_decode(int l, int3 grid, int tiles_n, int& bx, int& bz) {
    const int tm = grid.x / tiles_n, x = l & 7, j = l >> 3;
    const int unit = (j / tm) * 8 + x;
    if (unit >= tiles_n * grid.z) return false;
    bz = unit / tiles_n;
    bx = (j % tm) * tiles_n + unit % tiles_n;
    return true;
}

template <int BM1, int BN1, int WM1, int WN1, class SA1, class SB1, class EP1,
          int BM2, int BN2, int WM2, int WN2, class SA2, class SB2, class EP2, int BK = GEMM_BK>
__global__ __launch_bounds__(GEMM_THREADS) void gemm_duo_kernel(GemmArgs<SA1, SB1, EP1> a1, int3 grid1,
                                                                GemmArgs<SA2, SB2, EP2> a2, int3 grid2, int xcd) {
    constexpr int L1 = gemm_lds_elems<BM1, BN1, BK, SA1, SB1>(), L2 = gemm_lds_elems<BM2, BN2, BK, SA2, SB2>();
    __shared__ __attribute__((aligned(16))) bf16_t lds[L1 > L2 ? L1 : L2];
    int b = blockIdx.x;
    if (xcd) {          // both problems un-split (grid.y == 1, checked by the launcher)
        const int p1 = xcd_cols_blocks(grid1, a1.tiles_n);
        int bx, bz;
        if (b < p1) {
            if (xcd_cols_decode(b, grid1, a1.tiles_n, bx, bz))
                gemm_body<BM1, BN1, WM1, WN1, BK, SA1, SB1, EP1>(a1, bx, 0, bz, lds);
        } else if (xcd_cols_decode(b - p1, grid2, a2.tiles_n, bx, bz)) {
            gemm_body<BM2, BN2, WM2, WN2, BK, SA2, SB2, EP2>(a2, bx, 0, bz, lds);
        }
        return;
    }
    const int n1 = grid1.x * grid1.y * grid1.z;
    if (b < n1) {
        gemm_body<BM1, BN1, WM1, WN1, BK, SA1, SB1, EP1>(a1, b % grid1.x, (b / grid1.x) % grid1.y, b / (grid1.x * grid1.y), lds);
    } else {
        b -= n1;
        gemm_body<BM2, BN2, WM2, WN2, BK, SA2, SB2, EP2>(a2, b % grid2.x, (b / grid2.x) % grid2.y, b / (grid2.x * grid2.y), lds);
    }
}

// The same work with an XCD-aware block order.  Workgroups go to the 8 XCDs round-robin by linear id and every XCD
// has an L2 of its own; in the plain launch the tiles of one split-K slice -- which all read the same rows of both
// operands -- therefore land on 8 different L2s and each fetches those rows from HBM again (conv2/conv3 weight
// gradients at batch 128: 33 MB fetched per launch for 6.7 MB of operands).  Here all tiles of a slice run on one XCD.
// Launch with 8 * ceil(slices * groups / 8) * tiles workgroups.
template <int BM, int BN, int WM, int WN, int BK, class SA, class SB, class EP>
__global__ __launch_bounds__(GEMM_THREADS) void gemm_xcd_kernel(GemmArgs<SA, SB, EP> args, int3 grid) {
    __shared__ __attribute__((aligned(16))) bf16_t lds[gemm_lds_elems<BM, BN, BK, SA, SB>()];
    // unit = (slice, group); XCD x takes the units u = x (mod 8) -- empty tail slices of the smaller problem spread
    // evenly that way -- and runs each unit's tiles back to back
    const int x = (int)blockIdx.x & 7, j = (int)blockIdx.x >> 3;
    const int unit = (j / grid.x) * 8 + x, tile = j % grid.x;
    if (unit >= grid.y * grid.z) return;
    gemm_body<BM, BN, WM, WN, BK, SA, SB, EP>(args, tile, unit % grid.y, unit / grid.y, lds);
}

// grid of launch_gemm for these sizes (also fills tiles_n / ksplit): for launches that share a kernel
template <int BM, int BN, class SA, class SB, class EP, int BK = GEMM_BK>
static int3 plan_gemm(GemmArgs<SA, SB, EP>& args, int ngroups, int maxM, int maxN, int maxK, int ksplit) {
    int tiles_m = (maxM + BM - 1) / BM, tiles_n = (maxN + BN - 1) / BN;
    args.tiles_n = tiles_n;
    int nsplit = 1;
    if (ksplit > 0) {
        ksplit = (ksplit + BK - 1) / BK * BK;
        nsplit = (maxK + ksplit - 1) / ksplit;
    }
    args.ksplit = ksplit > 0 ? ksplit : 0;
    return make_int3(tiles_m * tiles_n, nsplit, ngroups);
}

template <int BM, int BN, int WM, int WN, class SA, class SB, class EP, int BK = GEMM_BK>
static void launch_gemm(GemmArgs<SA, SB, EP>& args, int ngroups, int maxM, int maxN, int maxK, int ksplit,
                        hipStream_t stream) {
    int tiles_m = (maxM + BM - 1) / BM, tiles_n = (maxN + BN - 1) / BN;
    args.tiles_n = tiles_n;
    int nsplit = 1;
    if (ksplit > 0) {
        ksplit = (ksplit + BK - 1) / BK * BK;
        nsplit = (maxK + ksplit - 1) / ksplit;
    }
    args.ksplit = ksplit > 0 ? ksplit : 0;
    dim3 grid(tiles_m * tiles_n, nsplit, ngroups);
    hipLaunchKernelGGL((gemm_kernel<BM, BN, WM, WN, BK, SA, SB, EP>), grid, dim3(GEMM_THREADS), 0, stream, args);
}
