// Shared device helpers for the gfx950 kernels.  wave = 64 lanes everywhere.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/deephumor_hip.h"

#define DH_WAVE 64

#define DH_REQUIRE(cond) do { if (!(cond)) return DH_ERR_BAD_ARG; } while (0)
#define DH_LAUNCH_CHECK() do { return hipGetLastError() == hipSuccess ? DH_OK : DH_ERR_LAUNCH; } while (0)

static inline int dh_cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ int wave_sum_i(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// Philox4x32-10 counter-based generator: one call gives four 32-bit words that depend only on
// (key, counter), so noise is reproducible for any launch geometry / rank layout.
__device__ __forceinline__ void philox4x32(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                           uint32_t k0, uint32_t k1, uint32_t out[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// Exp(1) sample for (seed, image, step, stream kind, row, index).
__device__ __forceinline__ float philox_exp1(uint64_t seed, uint32_t img, uint32_t step, uint32_t kind,
                                             uint32_t row, uint32_t idx) {
    uint32_t o[4];
    philox4x32(idx, row, step, kind, (uint32_t)seed ^ (img * 0x9E3779B1u), (uint32_t)(seed >> 32) + img, o);
    float u = ((float)(o[0] >> 8) + 0.5f) * (1.0f / 16777216.0f);   // (0,1)
    return -logf(u);
}
