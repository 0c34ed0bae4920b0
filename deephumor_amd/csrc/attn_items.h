// Arithmetic shared by the attention kernels of attention.hip and by the attention phases of the persistent decoder-layer kernel
// (decode_layers.hip): ONE definition of the softmax arithmetic, the cross-lane reductions, the packed-operand helpers and the
// matrix-core cross-attention core, so that every form stays bit-identical to the others.
#pragma once
#include "common.h"

// Softmax arithmetic.  fp32 path: the reference's exact formulation -- energy / scale (transformers.py:106), expf, e / sum as
// torch.softmax (:114) -- its greedy ids are bit-exact against the CPU reference.  16-bit paths: the hardware reciprocal and
// exp2 (1 ulp of fp32 each; the attention output is rounded to 8 / 11 significant bits anyway): the exact division and expf are
// ~10 and ~15 VALU instructions each, 48 of them per lane on a wave that runs alone on its SIMD.  One definition for every
// kernel, so fused and unfused forms stay bit-identical to each other.
template <typename T> struct SmMath {
    static __device__ __forceinline__ float div(float a, float b) { return a / b; }
    static __device__ __forceinline__ float exp(float x) { return expf(x); }
};
struct SmFast {
    static __device__ __forceinline__ float div(float a, float b) { return a * __builtin_amdgcn_rcpf(b); }
    static __device__ __forceinline__ float exp(float x) { return __expf(x); }
};
template <> struct SmMath<bf16_t> : SmFast {};
template <> struct SmMath<f16_t> : SmFast {};

// max / sum over the four lanes {l, l ^ 16, l ^ 32, l ^ 48} (the lanes sharing an MFMA accumulator row) without the LDS crossbar:
// gfx950's v_permlane16_swap / v_permlane32_swap exchange 16- / 32-lane halves between two registers (x, x) -> (lower copies, upper
// copies); __shfl_xor is a ds_bpermute round trip (~100 cycles + a wait) each
__device__ __forceinline__ float quad_rows_max(float v) {
    auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
    auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
}
__device__ __forceinline__ float quad_rows_sum(float v) {
    auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = __uint_as_float(a[0]) + __uint_as_float(a[1]);
    auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}

// sum over the 8 key slots of a (key slot, 8-dim chunk) lane layout with 8 chunk lanes per key (lanes l, l ^ 8, l ^ 16, ..., l ^ 56), in
// the order the xor-shuffle tree takes them: DPP row_ror:8 (lane ^ 8 inside a 16-lane row), then the row swaps above
__device__ __forceinline__ float key_slots_sum8(float v) {
    v += dpp_get<0x128>(v);
    return quad_rows_sum(v);
}

__device__ __forceinline__ void load8(const float* p, float (&v)[8]) {
    const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
__device__ __forceinline__ void load8(const bf16_t* p, float (&v)[8]) { load16(p, v); }
__device__ __forceinline__ void load8(const f16_t* p, float (&v)[8]) { load16(p, v); }
__device__ __forceinline__ void store8(float* p, const float (&v)[8]) {
    *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
    *reinterpret_cast<float4*>(p + 4) = make_float4(v[4], v[5], v[6], v[7]);
}
__device__ __forceinline__ void store8(bf16_t* p, const float (&v)[8]) { store16(p, v); }
__device__ __forceinline__ void store8(f16_t* p, const float (&v)[8]) { store16(p, v); }
__device__ __forceinline__ void copy8(float* d, const float* s) {
    *reinterpret_cast<float4*>(d) = *reinterpret_cast<const float4*>(s);
    *reinterpret_cast<float4*>(d + 4) = *reinterpret_cast<const float4*>(s + 4);
}
__device__ __forceinline__ void copy8(bf16_t* d, const bf16_t* s) {
    *reinterpret_cast<uint4*>(d) = *reinterpret_cast<const uint4*>(s);
}
__device__ __forceinline__ void copy8(f16_t* d, const f16_t* s) {
    *reinterpret_cast<uint4*>(d) = *reinterpret_cast<const uint4*>(s);
}

template <typename T> struct Raw8;                       // 8 elements kept packed until use
template <> struct Raw8<float> { float4 a, b; };
template <> struct Raw8<bf16_t> { uint4 a; };
template <> struct Raw8<f16_t> { uint4 a; };
__device__ __forceinline__ void raw_load(const float* p, Raw8<float>& r) {
    r.a = *reinterpret_cast<const float4*>(p); r.b = *reinterpret_cast<const float4*>(p + 4);
}
__device__ __forceinline__ void raw_load(const bf16_t* p, Raw8<bf16_t>& r) { r.a = *reinterpret_cast<const uint4*>(p); }
__device__ __forceinline__ void raw_load(const f16_t* p, Raw8<f16_t>& r) { r.a = *reinterpret_cast<const uint4*>(p); }
__device__ __forceinline__ void raw_unpack(const Raw8<f16_t>& r, float (&v)[8]) {
    unpack_f16x2(r.a.x, v[0], v[1]); unpack_f16x2(r.a.y, v[2], v[3]);
    unpack_f16x2(r.a.z, v[4], v[5]); unpack_f16x2(r.a.w, v[6], v[7]);
}
__device__ __forceinline__ void raw_unpack(const Raw8<float>& r, float (&v)[8]) {
    v[0] = r.a.x; v[1] = r.a.y; v[2] = r.a.z; v[3] = r.a.w; v[4] = r.b.x; v[5] = r.b.y; v[6] = r.b.z; v[7] = r.b.w;
}
__device__ __forceinline__ void raw_unpack(const Raw8<bf16_t>& r, float (&v)[8]) {
    v[0] = __uint_as_float(r.a.x << 16); v[1] = __uint_as_float(r.a.x & 0xFFFF0000u);
    v[2] = __uint_as_float(r.a.y << 16); v[3] = __uint_as_float(r.a.y & 0xFFFF0000u);
    v[4] = __uint_as_float(r.a.z << 16); v[5] = __uint_as_float(r.a.z & 0xFFFF0000u);
    v[6] = __uint_as_float(r.a.w << 16); v[7] = __uint_as_float(r.a.w & 0xFFFF0000u);
}

// scores -> softmax -> P V of one (image, head) from preloaded fragments (shared by the two matrix-core kernels below)
template <typename T>
__device__ __forceinline__ void cross_core(const uint4 (&kf)[4][2], const uint4 (&vf)[4][2], const uint4 (&qf)[2], const uint64_t kbits,
                                           int S, float scale, bool live, uint16_t* orow, int lq) {
    dh_f32x4 sacc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        sacc[j] = dh_f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) sacc[j] = Op16<T>::mfma(kf[j][kk], qf[kk], sacc[j]);     // S[m = l15][key = 16j + 4lq + r]
    }
    float e[4][4];
    float mx = -INFINITY;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int key = 16 * j + 4 * lq + r;
            const bool masked = (kbits >> key) & 1ull;
            e[j][r] = key < S ? (masked ? -1e8f : SmMath<T>::div(sacc[j][r], scale)) : -INFINITY;
            mx = fmaxf(mx, e[j][r]);
        }
    mx = quad_rows_max(mx);
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) { e[j][r] = SmMath<T>::exp(e[j][r] - mx); sum += e[j][r]; }
    sum = quad_rows_sum(sum);
    // the lane's own 16 weights, rounded to the operand type, in key-slot order: k-step kk, element e <-> (j = 2kk + (e >> 2), r = e & 3)
    uint4 pf[2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        uint32_t w[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int e0 = 2 * u, e1 = 2 * u + 1;
            const float p0 = SmMath<T>::div(e[2 * kk + (e0 >> 2)][e0 & 3], sum), p1 = SmMath<T>::div(e[2 * kk + (e1 >> 2)][e1 & 3], sum);   // as torch.softmax
            w[u] = (uint32_t)Op16<T>::from_f32(p0) | ((uint32_t)Op16<T>::from_f32(p1) << 16);
        }
        pf[kk] = make_uint4(w[0], w[1], w[2], w[3]);
    }
#pragma unroll
    for (int jd = 0; jd < 4; ++jd) {
        dh_f32x4 o = dh_f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) o = Op16<T>::mfma(vf[jd][kk], pf[kk], o);                 // out[m = l15][d = 16jd + 4lq + r]
        if (live) {
            uint2 pk;
            pk.x = (uint32_t)Op16<T>::from_f32(o[0]) | ((uint32_t)Op16<T>::from_f32(o[1]) << 16);
            pk.y = (uint32_t)Op16<T>::from_f32(o[2]) | ((uint32_t)Op16<T>::from_f32(o[3]) << 16);
            *reinterpret_cast<uint2*>(orow + 16 * jd + 4 * lq) = pk;
        }
    }
}

