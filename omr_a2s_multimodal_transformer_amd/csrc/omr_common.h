// Shared device helpers for the gfx950 (MI355X / CDNA4) kernels of libomr_hip.so.
// Wave = 64 lanes.  All matrix work goes through 32x32 MFMA tiles:
//   bf16 : v_mfma_f32_32x32x16_bf16  (8 k-values per lane per operand)
//   fp32 : v_mfma_f32_32x32x2_f32 x4 (4 k-values per lane per operand; exact f32 fma chain)
// Both consume ONE 16-byte fragment per lane per operand, so every tile loop below is written
// once over Frag<T> and works for the fp32 parity path and the bf16 throughput path.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define OMR_F32 0
#define OMR_BF16 1

#define OMR_OK 0
#define OMR_ERR_ARG (-1)
#define OMR_ERR_LAUNCH (-2)
#define OMR_ERR_UNSUPPORTED (-3)

#define OMR_CHECK_LAUNCH()                                    \
    do {                                                      \
        hipError_t e__ = hipGetLastError();                   \
        if (e__ != hipSuccess) return OMR_ERR_LAUNCH;         \
    } while (0)

template <typename T> struct Frag;
template <> struct Frag<bf16> {
    typedef bf16x8 type;
    static constexpr int N = 8;      // elements per 16-byte fragment
};
template <> struct Frag<float> {
    typedef f32x4 type;
    static constexpr int N = 4;
};
// k-values consumed by one mma32 call: lanes 0-31 hold k = [0,N), lanes 32-63 hold k = [N,2N)
template <typename T> struct KStep { static constexpr int value = 2 * Frag<T>::N; };

__device__ __forceinline__ void mma32(f32x16& acc, bf16x8 a, bf16x8 b) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
}
__device__ __forceinline__ void mma32(f32x16& acc, f32x4 a, f32x4 b) {
    // lane half h holds k = 4h+e (e=0..3); instruction e sums k in {e, 4+e}.  A and B use the
    // same (arbitrary) k order, so the result is the exact f32 dot product.
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0], b[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[1], b[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[2], b[2], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[3], b[3], acc, 0, 0, 0);
}

// 32x32 accumulator layout (dtype independent on gfx950): column = lane & 31,
// row(reg) = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5).
__device__ __forceinline__ int acc_row(int reg, int lane) { return (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5); }

__device__ __forceinline__ float to_f32(float x) { return x; }
__device__ __forceinline__ float to_f32(bf16 x) { return (float)x; }
template <typename T> __device__ __forceinline__ T from_f32(float x);
template <> __device__ __forceinline__ float from_f32<float>(float x) { return x; }
template <> __device__ __forceinline__ bf16 from_f32<bf16>(float x) { return (bf16)x; }

template <typename T> __device__ __forceinline__ typename Frag<T>::type frag_zero() {
    typename Frag<T>::type z;
#pragma unroll
    for (int i = 0; i < Frag<T>::N; ++i) z[i] = from_f32<T>(0.f);
    return z;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
// LayerNorm of one row held by a wave, lane = PER consecutive columns: v <- (v - mean) * rstd * gamma + beta.  One definition
// for add_ln_fwd_kernel (norm.hip) and the decode row kernel's prologue (decode.hip): the KV-cached step has to build the rows
// the training forward builds, to the bit.
template <int PER>
__device__ __forceinline__ void ln_row(float (&v)[PER], const float* __restrict__ gamma, const float* __restrict__ beta, int lane, float eps, float& mu,
                                       float& rs) {
    constexpr int d = PER * 64;
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < PER; ++i) s += v[i];
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    mu = s * (1.f / d);
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < PER; ++i) { const float t = v[i] - mu; q += t * t; }
    for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o, 64);
    rs = rsqrtf(q * (1.f / d) + eps);
#pragma unroll
    for (int i = 0; i < PER; ++i) v[i] = (v[i] - mu) * rs * gamma[lane * PER + i] + beta[lane * PER + i];
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// Counter-based RNG for dropout masks: one 32-bit hash per (seed, element index), so a mask is
// regenerated in backward instead of being stored (never materialised in HBM).
__device__ __forceinline__ uint32_t hash_u32(uint64_t seed, uint64_t idx) {
    uint64_t z = idx * 0x9E3779B97F4A7C15ull + seed;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z = z ^ (z >> 31);
    return (uint32_t)(z >> 32);
}
// 32-bit variant (two rounds of a multiply-xorshift mixer keyed by both seed halves) for indices < 2^32:
// ~10 VALU ops per element instead of ~30 for the 64-bit mixer.
__device__ __forceinline__ uint32_t hash32(uint32_t seed_lo, uint32_t seed_hi, uint32_t idx) {
    uint32_t h = idx * 0x9E3779B1u + seed_lo;
    h ^= h >> 16; h *= 0x85EBCA6Bu;
    h ^= h >> 13; h += seed_hi; h *= 0xC2B2AE35u;
    h ^= h >> 16;
    return h;
}
// Dropout keep decision of element `idx` under (seed, p): ONE 32-bit hash serves the element PAIR (idx & ~1, idx | 1) with 16
// random bits each (low half: even element), keep iff bits >= p * 2^16 (thresh16 = OMR_DROP_THRESH16(p); p is realised to
// 1.5e-5, 0.5 and 0.25 exactly).  Every dropout site of the library uses this convention (omr_dropout, the conv / GEMM /
// add+LayerNorm epilogues), so a mask can be regenerated anywhere from (seed, index); vector code hashes once per pair.
#define OMR_DROP_THRESH16(p) ((uint32_t)((double)(p) * 65536.0 + 0.5))
__device__ __forceinline__ uint32_t drop_pair_bits(uint64_t seed, uint64_t pair) {
    return hash32((uint32_t)seed, (uint32_t)(seed >> 32), (uint32_t)pair ^ (uint32_t)(pair >> 32) * 0x27D4EB2Fu);
}
__device__ __forceinline__ bool drop_keep(uint64_t seed, uint64_t idx, uint32_t thresh16) {
    const uint32_t h = drop_pair_bits(seed, idx >> 1);
    return ((idx & 1) ? (h >> 16) : (h & 0xffffu)) >= thresh16;
}

__host__ __device__ static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }
