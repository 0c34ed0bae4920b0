// Shared device helpers for the gfx950 kernels.  wave = 64 lanes everywhere.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/deephumor_hip.h"

#define DH_WAVE 64

#define DH_TRY(call) do { const int rc_ = (call); if (rc_ != DH_OK) return rc_; } while (0)
#define DH_REQUIRE(cond) do { if (!(cond)) return DH_ERR_BAD_ARG; } while (0)
#define DH_LAUNCH_CHECK() do { return hipGetLastError() == hipSuccess ? DH_OK : DH_ERR_LAUNCH; } while (0)

static inline int dh_cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }

// Wave-wide all-reduce without the LDS crossbar: hipcc lowers __shfl_xor to ds_bpermute_b32 + s_waitcnt lgkmcnt(0)
// (six dependent ~100-cycle round trips per reduction); here the 16 lanes of a DPP row are folded with four DPP moves
// (quad_perm [1,0,3,2], quad_perm [2,3,0,1], row_half_mirror, row_mirror) and the four rows with v_readlane.
// Must be called by all 64 lanes (wave-uniform control flow), like the shuffles they replace.
template <int CTRL>
__device__ __forceinline__ float dpp_get(float v) {
    const int x = __builtin_bit_cast(int, v);
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(x, x, CTRL, 0xF, 0xF, false));
}
template <int CTRL>
__device__ __forceinline__ int dpp_get_i(int x) { return __builtin_amdgcn_update_dpp(x, x, CTRL, 0xF, 0xF, false); }
__device__ __forceinline__ float lane_get(float v, int l) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l));
}
// sum / max over aligned groups of 8 lanes (every lane of the group gets the result)
__device__ __forceinline__ float sum8(float v) {
    v += dpp_get<0xB1>(v); v += dpp_get<0x4E>(v); v += dpp_get<0x141>(v);
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {
    v = sum8(v);
    v += dpp_get<0x140>(v);                               // the other 8 lanes of the 16-lane row
    return (lane_get(v, 0) + lane_get(v, 16)) + (lane_get(v, 32) + lane_get(v, 48));
}
__device__ __forceinline__ float wave_max(float v) {
    v = fmaxf(v, dpp_get<0xB1>(v)); v = fmaxf(v, dpp_get<0x4E>(v));
    v = fmaxf(v, dpp_get<0x141>(v)); v = fmaxf(v, dpp_get<0x140>(v));
    return fmaxf(fmaxf(lane_get(v, 0), lane_get(v, 16)), fmaxf(lane_get(v, 32), lane_get(v, 48)));
}
__device__ __forceinline__ int wave_sum_i(int v) {
    v += dpp_get_i<0xB1>(v); v += dpp_get_i<0x4E>(v); v += dpp_get_i<0x141>(v); v += dpp_get_i<0x140>(v);
    return (__builtin_amdgcn_readlane(v, 0) + __builtin_amdgcn_readlane(v, 16)) +
           (__builtin_amdgcn_readlane(v, 32) + __builtin_amdgcn_readlane(v, 48));
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
    // 23 random bits: (i + 0.5) / 2^23 is exact in fp32 for every i < 2^23, so u stays strictly inside (0, 1)
    // (with 24 bits, i = 2^24 - 1 rounds to u == 1.0f and the Exp(1) sample would be 0)
    float u = ((float)(o[0] >> 9) + 0.5f) * (1.0f / 8388608.0f);
    return -logf(u);
}

// ---- LDS-DMA (global -> LDS without staging VGPRs) issued through inline assembly ----------------------------------
// hipcc's own builtin (__builtin_amdgcn_global_load_lds) is tracked by the compiler's waitcnt insertion as a store to
// LDS that may alias ANY later LDS read: it then puts an `s_waitcnt vmcnt(0)` in front of the ds_reads of every loop
// iteration, which drains the whole LDS ring each time -- a ring of N slabs behaves like a ring of one, the loads of
// slab t+1 never overlap the MFMAs of slab t, and the counted `s_waitcnt vmcnt(N)` of the kernels is dead code
// (measured: the K = 2048 decoder GEMM 25 us -> the pure ring transfer takes 7 us).  As inline assembly the transfer is
// invisible to that pass; ordering is the kernels' own counted vmcnt waits + barriers (loads, stores and LDS-DMA retire
// in issue order).  Compiler-generated waits for ordinary loads stay correct: they can only over-wait.
// `lds_dst` must be wave-uniform (the hardware adds lane * size itself); M0 carries it (one wait state before use).
typedef void __attribute__((address_space(3)))* dh_lptr_t;
__device__ __forceinline__ void dh_lds_dma16(const void* gsrc, void* lds_dst) {
    const unsigned m = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(dh_lptr_t)lds_dst);
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" :: "s"(m), "v"(gsrc) : "memory");
}
__device__ __forceinline__ void dh_lds_dma4(const void* gsrc, void* lds_dst) {
    const unsigned m = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(dh_lptr_t)lds_dst);
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dword %1, off" :: "s"(m), "v"(gsrc) : "memory");
}

// ---- storage-type helpers: T = float (parity path), __bf16 or _Float16 (throughput paths); math is fp32 ---------
typedef __bf16 bf16_t;
typedef _Float16 f16_t;
template <typename T> struct Vec16;                 // elements per 16-byte access
template <> struct Vec16<float> { static constexpr int N = 4; };
template <> struct Vec16<bf16_t> { static constexpr int N = 8; };
template <> struct Vec16<f16_t> { static constexpr int N = 8; };

__device__ __forceinline__ float bf16_to_f32(uint16_t b) { return __uint_as_float((uint32_t)b << 16); }
__device__ __forceinline__ uint16_t f32_to_bf16(float f) { return __builtin_bit_cast(uint16_t, (bf16_t)f); }

// 16-byte vector load of Vec16<T>::N elements into fp32 registers
__device__ __forceinline__ void load16(const float* p, float (&v)[4]) {
    const float4 t = *reinterpret_cast<const float4*>(p);
    v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
}
__device__ __forceinline__ void load16(const bf16_t* p, float (&v)[8]) {
    const uint4 t = *reinterpret_cast<const uint4*>(p);
    v[0] = __uint_as_float(t.x << 16); v[1] = __uint_as_float(t.x & 0xFFFF0000u);
    v[2] = __uint_as_float(t.y << 16); v[3] = __uint_as_float(t.y & 0xFFFF0000u);
    v[4] = __uint_as_float(t.z << 16); v[5] = __uint_as_float(t.z & 0xFFFF0000u);
    v[6] = __uint_as_float(t.w << 16); v[7] = __uint_as_float(t.w & 0xFFFF0000u);
}
__device__ __forceinline__ void store16(float* p, const float (&v)[4]) {
    *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
}
__device__ __forceinline__ void store16(bf16_t* p, const float (&v)[8]) {
    uint4 t;
    t.x = (uint32_t)f32_to_bf16(v[0]) | ((uint32_t)f32_to_bf16(v[1]) << 16);
    t.y = (uint32_t)f32_to_bf16(v[2]) | ((uint32_t)f32_to_bf16(v[3]) << 16);
    t.z = (uint32_t)f32_to_bf16(v[4]) | ((uint32_t)f32_to_bf16(v[5]) << 16);
    t.w = (uint32_t)f32_to_bf16(v[6]) | ((uint32_t)f32_to_bf16(v[7]) << 16);
    *reinterpret_cast<uint4*>(p) = t;
}
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float f16_to_f32(uint16_t h) { return (float)__builtin_bit_cast(f16_t, h); }
__device__ __forceinline__ uint16_t f32_to_f16(float f) { return __builtin_bit_cast(uint16_t, (f16_t)f); }
// two packed fp16 values of one 32-bit word -> fp32 (low half first)
__device__ __forceinline__ void unpack_f16x2(uint32_t w, float& lo, float& hi) {
    const f16x2_t h = __builtin_bit_cast(f16x2_t, w);
    lo = (float)h.x; hi = (float)h.y;
}
__device__ __forceinline__ uint32_t pack_f16x2(float lo, float hi) {
    f16x2_t h; h.x = (f16_t)lo; h.y = (f16_t)hi;
    return __builtin_bit_cast(uint32_t, h);
}
__device__ __forceinline__ void load16(const f16_t* p, float (&v)[8]) {
    const uint4 t = *reinterpret_cast<const uint4*>(p);
    unpack_f16x2(t.x, v[0], v[1]); unpack_f16x2(t.y, v[2], v[3]);
    unpack_f16x2(t.z, v[4], v[5]); unpack_f16x2(t.w, v[6], v[7]);
}
__device__ __forceinline__ void store16(f16_t* p, const float (&v)[8]) {
    uint4 t;
    t.x = pack_f16x2(v[0], v[1]); t.y = pack_f16x2(v[2], v[3]);
    t.z = pack_f16x2(v[4], v[5]); t.w = pack_f16x2(v[6], v[7]);
    *reinterpret_cast<uint4*>(p) = t;
}
__device__ __forceinline__ float ldf(const f16_t* p) { return (float)*p; }
__device__ __forceinline__ void stf(f16_t* p, float v) { *p = (f16_t)v; }
__device__ __forceinline__ float ldf(const float* p) { return *p; }
__device__ __forceinline__ float ldf(const bf16_t* p) { return (float)*p; }
__device__ __forceinline__ void stf(float* p, float v) { *p = v; }
__device__ __forceinline__ void stf(bf16_t* p, float v) { *p = (bf16_t)v; }

// ---- deferred LayerNorm: per-(row, 64-column tile) partial statistics {mean, M2} -> the row's mean / rstd ------------------
// mean and rstd of a row from its nt <= 8 tile partials (each over 64 elements): Chan's parallel combination in a fixed order.
// Split in two so that the loads can be issued at the top of a kernel and the arithmetic (which waits for them) run in its
// epilogue: called back to back they would put a full memory round trip in front of the first operand slab.
__device__ __forceinline__ void ln_load(const float2* st, int nt, float4 (&raw)[4]) {
#pragma unroll
    for (int t = 0; t < 4; ++t) raw[t] = *reinterpret_cast<const float4*>(st + (2 * t < nt ? 2 * t : 0));   // 16-byte loads, clamped
}
__device__ __forceinline__ void ln_math(const float4 (&raw)[4], int nt, float eps, float& mu, float& rstd) {
    float2 v[8];
#pragma unroll
    for (int t = 0; t < 4; ++t) { v[2 * t] = make_float2(raw[t].x, raw[t].y); v[2 * t + 1] = make_float2(raw[t].z, raw[t].w); }
    float ms = 0.f;
#pragma unroll
    for (int t = 0; t < 8; ++t) ms += t < nt ? v[t].x : 0.f;
    mu = ms / (float)nt;
    float m2 = 0.f;
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        const float d = v[t].x - mu;
        m2 += t < nt ? fmaf(64.f * d, d, v[t].y) : 0.f;
    }
    rstd = 1.0f / sqrtf(m2 / (float)(nt * 64) + eps);
}

typedef const void __attribute__((address_space(1)))* gptr_t;
typedef void __attribute__((address_space(3)))* lptr_t;

#define DH_DISPATCH_T(dtype, ...)                                             \
    if ((dtype) == DH_F32) { using T = float; __VA_ARGS__; }                  \
    else if ((dtype) == DH_BF16) { using T = bf16_t; __VA_ARGS__; }           \
    else if ((dtype) == DH_F16) { using T = f16_t; __VA_ARGS__; }             \
    else return DH_ERR_UNSUPPORTED;

// the two 16-bit storage types (bf16 / fp16 in HBM, MFMA operands, fp32 accumulation)
#define DH_IS_16BIT(dtype) ((dtype) == DH_BF16 || (dtype) == DH_F16)
#define DH_DISPATCH_16(dtype, ...)                                            \
    if ((dtype) == DH_BF16) { using T = bf16_t; __VA_ARGS__; }                \
    else if ((dtype) == DH_F16) { using T = f16_t; __VA_ARGS__; }             \
    else return DH_ERR_UNSUPPORTED;

// Matrix-core operand traits of the 16-bit types: 8 k-values per lane of v_mfma_f32_16x16x32_{bf16,f16}; raw 16-bit
// patterns <-> fp32 (kernels that move operands as uint16_t / uint4 and only convert at the edges)
typedef float dh_f32x4 __attribute__((ext_vector_type(4)));
template <typename T> struct Op16;
template <> struct Op16<bf16_t> {
    typedef __bf16 vec8 __attribute__((ext_vector_type(8)));
    static __device__ __forceinline__ dh_f32x4 mfma(uint4 a, uint4 b, dh_f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(vec8, a), __builtin_bit_cast(vec8, b), c, 0, 0, 0);
    }
    static __device__ __forceinline__ uint16_t from_f32(float f) { return f32_to_bf16(f); }
    static __device__ __forceinline__ float to_f32(uint16_t h) { return bf16_to_f32(h); }
    static __device__ __forceinline__ void unpack2(uint32_t w, float& lo, float& hi) {
        lo = __uint_as_float(w << 16); hi = __uint_as_float(w & 0xFFFF0000u);
    }
};
template <> struct Op16<f16_t> {
    typedef _Float16 vec8 __attribute__((ext_vector_type(8)));
    static __device__ __forceinline__ dh_f32x4 mfma(uint4 a, uint4 b, dh_f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(vec8, a), __builtin_bit_cast(vec8, b), c, 0, 0, 0);
    }
    static __device__ __forceinline__ uint16_t from_f32(float f) { return f32_to_f16(f); }
    static __device__ __forceinline__ float to_f32(uint16_t h) { return f16_to_f32(h); }
    static __device__ __forceinline__ void unpack2(uint32_t w, float& lo, float& hi) { unpack_f16x2(w, lo, hi); }
};
