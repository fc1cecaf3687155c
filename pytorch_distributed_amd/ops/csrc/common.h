// Shared device helpers for the gfx950 kernels (wave64, bf16 MFMA 16x16x32).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define PDA_EXPORT extern "C" __attribute__((visibility("default")))

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;   // one MFMA A/B fragment (4 VGPRs)
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;    // one 16x16 accumulator fragment
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef unsigned short bf16_t;                               // raw bf16 bits in memory

__device__ __forceinline__ bf16_t f2bf(float f) {
    __bf16 h = (__bf16)f;   // v_cvt_pk_bf16_f32: round-to-nearest-even, NaN preserving
    return __builtin_bit_cast(bf16_t, h);
}
__device__ __forceinline__ float bf2f(bf16_t b) {
    return __builtin_bit_cast(float, ((uint32_t)b) << 16);
}
__device__ __forceinline__ uint32_t pack2bf(float lo, float hi) {   // one v_cvt_pk_bf16_f32
    typedef __attribute__((ext_vector_type(2))) float f32x2_;
    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_;
    f32x2_ v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_));
}
// 4 uint8 (one dword) -> 4 bf16 bit patterns; exact (0..255 fits 8 mantissa bits)
__device__ __forceinline__ uint2 u8x4_to_bf16x4(uint32_t w) {
    uint32_t f0 = __builtin_bit_cast(uint32_t, (float)(w & 0xffu));
    uint32_t f1 = __builtin_bit_cast(uint32_t, (float)((w >> 8) & 0xffu));
    uint32_t f2 = __builtin_bit_cast(uint32_t, (float)((w >> 16) & 0xffu));
    uint32_t f3 = __builtin_bit_cast(uint32_t, (float)(w >> 24));
    return make_uint2((f0 >> 16) | (f1 & 0xffff0000u), (f2 >> 16) | (f3 & 0xffff0000u));
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
// inclusive prefix sum across the 64 lanes
__device__ __forceinline__ float wave_scan_incl(float v, int lane) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        float t = __shfl_up(v, o, 64);
        if (lane >= o) v += t;
    }
    return v;
}

// counter-based RNG (PCG-style hash of a 64-bit counter): reproducible per (seed, idx)
__device__ __forceinline__ uint32_t hash_u32(uint64_t x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
    return (uint32_t)x;
}
__device__ __forceinline__ float u01(uint32_t r) { return (r >> 8) * (1.0f / 16777216.0f); }
// Uniform replay row from a 32-bit random word.  window: int32[4] = {rows to draw from, first row, ring length, -}:
// row = first + floor(r * rows / 2^32), wrapped at the ring length when that is non-zero (frame-deduplicated replay
// draws from the newest `rows` transitions of its ring; plain replay has first = 0, ring length = 0).
__device__ __forceinline__ int window_pick(const int* __restrict__ window, uint32_t r) {
    const int rows = max(window[0], 1), first = window[1], ring = window[2];
    int i = first + (int)(((unsigned long long)r * (unsigned long long)rows) >> 32);
    if (ring > 0 && i >= ring) i -= ring;
    return i;
}

#define ACT_NONE 0
#define ACT_RELU 1
#define ACT_TANH 2
