// LSTM single-time-step kernels (nn.LSTM of rnn_models.py:23-24, one step of :80 / :108).
// The four gate products run on the matrix cores through dh_linear over the concatenated
// [x | h] operand; these two HBM-bound kernels gather that operand (including the beam reorder of
// the recurrent state -- an index gather, the state arrays are never permuted in place) and apply
// the pointwise cell update.  Storage type T (fp32 or bf16) applies to embeddings and hidden
// states; gate pre-activations and the cell state c are always fp32.
#include "common.h"
#include "prof.h"

template <typename T>
struct LstmPrepParams {
    const T* emb; const T* img_emb; const int32_t* tokens; int tok_ld, tok_pos;
    const int32_t* hparent; const T* h_prev; const float* c_prev;
    T* xcat0; T* xcatl; float* c_cur;
    int rows, rows_per_img, row_mult, rows_total, n_layers, E, Hh;
    // fp32 rows on the split-operand path (dh_lstm_prepare_f32x): the same rows ALSO as the fp16 planes of the gate GEMM's operand
    // (hi = fp16(x), lo = fp16((x - hi) * 2^11)): xcat0p [2][rows][E + Hh], xcatlp [n_layers - 1][2][rows][2 Hh]
    uint16_t* xcat0p; uint16_t* xcatlp; unsigned* range_flag;
};

// 4 fp32 values -> the hi / lo quads of the split-operand kernels
__device__ __forceinline__ void lstm_split4(const uint4& raw, uint2& hi, uint2& lo, float& amax) {
    const float v[4] = {__uint_as_float(raw.x), __uint_as_float(raw.y), __uint_as_float(raw.z), __uint_as_float(raw.w)};
    uint32_t h[2], l[2];
#pragma unroll
    for (int j = 0; j < 4; j += 2) {
        const f16_t ha = (f16_t)v[j], hb = (f16_t)v[j + 1];
        const f16_t la = (f16_t)((v[j] - (float)ha) * 2048.0f), lb = (f16_t)((v[j + 1] - (float)hb) * 2048.0f);
        h[j / 2] = (uint32_t)__builtin_bit_cast(uint16_t, ha) | ((uint32_t)__builtin_bit_cast(uint16_t, hb) << 16);
        l[j / 2] = (uint32_t)__builtin_bit_cast(uint16_t, la) | ((uint32_t)__builtin_bit_cast(uint16_t, lb) << 16);
        amax = fmaxf(amax, fmaxf(fabsf(v[j]), fabsf(v[j + 1])));
    }
    hi = make_uint2(h[0], h[1]); lo = make_uint2(l[0], l[1]);
}

template <typename T>
__global__ __launch_bounds__(256) void lstm_prepare_kernel(LstmPrepParams<T> p) {
    constexpr int VN = Vec16<T>::N;
    const int rc = blockIdx.x, tid = threadIdx.x;
    const int rl = rc * p.row_mult;
    int hp = p.hparent ? p.hparent[rl] : rl;
    if (!p.h_prev) hp = -1;
    const T* xin = p.tokens ? p.emb + (size_t)p.tokens[(size_t)rl * p.tok_ld + p.tok_pos] * p.E
                            : p.img_emb + (size_t)(rc / p.rows_per_img) * p.E;
    const int E = p.E, Hh = p.Hh;
    T* x0 = p.xcat0 + (size_t)rc * (E + Hh);
    const uint4 zero = make_uint4(0u, 0u, 0u, 0u);
    const bool planes = sizeof(T) == 4 && p.xcat0p;
    float amax = 0.f;
    for (int d = tid * VN; d < E; d += 256 * VN) {
        const uint4 raw = *reinterpret_cast<const uint4*>(xin + d);
        *reinterpret_cast<uint4*>(x0 + d) = raw;
        if (planes) {
            uint2 hi, lo;
            lstm_split4(raw, hi, lo, amax);
            const size_t o = (size_t)rc * (E + Hh) + d;
            *reinterpret_cast<uint2*>(p.xcat0p + o) = hi;
            *reinterpret_cast<uint2*>(p.xcat0p + (size_t)p.rows * (E + Hh) + o) = lo;
        }
    }
    for (int l = 0; l < p.n_layers; ++l) {
        const T* hs = hp >= 0 ? p.h_prev + ((size_t)l * p.rows_total + hp) * Hh : nullptr;
        const float* cs = hp >= 0 ? p.c_prev + ((size_t)l * p.rows_total + hp) * Hh : nullptr;
        T* hd = l == 0 ? x0 + E : p.xcatl + ((size_t)(l - 1) * p.rows + rc) * (2 * Hh) + Hh;
        float* cd = p.c_cur + ((size_t)l * p.rows + rc) * Hh;
        for (int d = tid * VN; d < Hh; d += 256 * VN) {
            const uint4 raw = hs ? *reinterpret_cast<const uint4*>(hs + d) : zero;
            *reinterpret_cast<uint4*>(hd + d) = raw;
            if (planes) {
                uint2 hi, lo;
                lstm_split4(raw, hi, lo, amax);
                const int ld = l == 0 ? E + Hh : 2 * Hh;
                uint16_t* base = l == 0 ? p.xcat0p : p.xcatlp + (size_t)(l - 1) * 2 * p.rows * 2 * Hh;
                const size_t o = (size_t)rc * ld + (l == 0 ? E : Hh) + d;
                *reinterpret_cast<uint2*>(base + o) = hi;
                *reinterpret_cast<uint2*>(base + (size_t)p.rows * ld + o) = lo;
            }
        }
        for (int d = tid * 4; d < Hh; d += 1024)
            *reinterpret_cast<uint4*>(cd + d) = cs ? *reinterpret_cast<const uint4*>(cs + d) : zero;
    }
    if (planes && amax >= 65504.0f) atomicOr(p.range_flag, 1u);
}

extern "C" int dh_lstm_prepare(const void* emb, const void* img_emb, const int32_t* tokens, int tok_ld, int tok_pos,
                               const int32_t* hparent, const void* h_prev, const float* c_prev,
                               void* xcat0, void* xcatl, float* c_cur, int rows, int rows_per_img, int row_mult,
                               int rows_total, int n_layers, int E, int Hh, int dtype, void* stream) {
    DH_REQUIRE(xcat0 && c_cur && rows > 0 && rows_per_img > 0 && row_mult > 0 && n_layers > 0);
    DH_REQUIRE((tokens && emb) || img_emb);
    DH_REQUIRE((E % 8) == 0 && (Hh % 8) == 0 && (n_layers == 1 || xcatl) && ((h_prev == nullptr) == (c_prev == nullptr)));
    DhProfScope prof("dh_lstm_prepare", 0.0, 0.0, stream);
    DH_DISPATCH_T(dtype, {
        LstmPrepParams<T> p{(const T*)emb, (const T*)img_emb, tokens, tok_ld, tok_pos, hparent, (const T*)h_prev,
                            c_prev, (T*)xcat0, (T*)xcatl, c_cur, rows, rows_per_img, row_mult, rows_total,
                            n_layers, E, Hh, nullptr, nullptr, nullptr};
        hipLaunchKernelGGL(lstm_prepare_kernel<T>, dim3(rows), dim3(256), 0, (hipStream_t)stream, p);
    });
    DH_LAUNCH_CHECK();
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

// gates row layout: [i | f | g | o], each Hh wide (PyTorch order), fp32.
template <typename T>
__global__ __launch_bounds__(256) void lstm_cell_kernel(
    const float* __restrict__ gates, const float* __restrict__ c_cur, T* __restrict__ h_new,
    float* __restrict__ c_new, T* __restrict__ h_out, int ld_out, int rows, int row_mult, int Hh,
    uint16_t* __restrict__ h_planes = nullptr, size_t plane = 0, int ld_planes = 0, unsigned* range_flag = nullptr) {
    const int rc = blockIdx.x;
    const int rl = rc * row_mult;
    const float* g = gates + (size_t)rc * 4 * Hh;
    for (int d = threadIdx.x; d < Hh; d += 256) {
        const float c1 = sigmoidf_(g[Hh + d]) * c_cur[(size_t)rc * Hh + d] + sigmoidf_(g[d]) * tanhf(g[2 * Hh + d]);
        const float h1 = sigmoidf_(g[3 * Hh + d]) * tanhf(c1);
        c_new[(size_t)rl * Hh + d] = c1;
        stf(h_new + (size_t)rl * Hh + d, h1);
        stf(h_out + (size_t)rc * ld_out + d, h1);
        if (sizeof(T) == 4 && h_planes) {               // |h| < 1: the range word cannot trip here
            const f16_t hi = (f16_t)h1;
            const f16_t lo = (f16_t)((h1 - (float)hi) * 2048.0f);
            h_planes[(size_t)rc * ld_planes + d] = __builtin_bit_cast(uint16_t, hi);
            h_planes[plane + (size_t)rc * ld_planes + d] = __builtin_bit_cast(uint16_t, lo);
        }
    }
    (void)range_flag;
}

extern "C" int dh_lstm_cell(const float* gates, const float* c_cur, void* h_new, float* c_new, void* h_out,
                            int ld_out, int rows, int row_mult, int Hh, int dtype, void* stream) {
    DH_REQUIRE(gates && c_cur && h_new && c_new && h_out && rows > 0 && row_mult > 0 && Hh > 0);
    DhProfScope prof("dh_lstm_cell", 0.0, 0.0, stream);
    DH_DISPATCH_T(dtype, hipLaunchKernelGGL(lstm_cell_kernel<T>, dim3(rows), dim3(256), 0, (hipStream_t)stream, gates,
                                            c_cur, (T*)h_new, c_new, (T*)h_out, ld_out, rows, row_mult, Hh, (uint16_t*)nullptr, (size_t)0, 0,
                                            (unsigned*)nullptr));
    DH_LAUNCH_CHECK();
}

// The two row kernels of the fp32 step on the split-operand path with the gate GEMM's operand ALSO stored as fp16 planes (hi, lo * 2^11;
// options "f32_split" + "f32_planes"): dh_lstm_prepare_f32x = dh_lstm_prepare(DH_F32) + xcat0_planes [2][rows][E + Hh], xcatl_planes
// [n_layers - 1][2][rows][2 Hh]; dh_lstm_cell_f32x = dh_lstm_cell(DH_F32) + the new hidden row into h_planes (hi at h_planes, lo at
// h_planes + plane elements, row stride ld_planes: the x half of the next layer's operand, or the classifier's).
unsigned* dh_f32x_range_flag_of(hipStream_t s);      // gemm_f32x.hip
extern "C" int dh_lstm_prepare_f32x(const float* emb, const float* img_emb, const int32_t* tokens, int tok_ld, int tok_pos,
                                    const int32_t* hparent, const float* h_prev, const float* c_prev, float* xcat0, float* xcatl, float* c_cur,
                                    void* xcat0_planes, void* xcatl_planes, int rows, int rows_per_img, int row_mult, int rows_total,
                                    int n_layers, int E, int Hh, void* stream) {
    DH_REQUIRE(xcat0 && c_cur && xcat0_planes && rows > 0 && rows_per_img > 0 && row_mult > 0 && n_layers > 0);
    DH_REQUIRE((tokens && emb) || img_emb);
    DH_REQUIRE((E % 8) == 0 && (Hh % 8) == 0 && (n_layers == 1 || (xcatl && xcatl_planes)) && ((h_prev == nullptr) == (c_prev == nullptr)));
    DhProfScope prof("dh_lstm_prepare", 0.0, 0.0, stream);
    unsigned* flag = dh_f32x_range_flag_of((hipStream_t)stream);
    if (!flag) return DH_ERR_LAUNCH;
    LstmPrepParams<float> p{emb, img_emb, tokens, tok_ld, tok_pos, hparent, h_prev, c_prev, xcat0, xcatl, c_cur, rows, rows_per_img, row_mult,
                            rows_total, n_layers, E, Hh, (uint16_t*)xcat0_planes, (uint16_t*)xcatl_planes, flag};
    hipLaunchKernelGGL(lstm_prepare_kernel<float>, dim3(rows), dim3(256), 0, (hipStream_t)stream, p);
    DH_LAUNCH_CHECK();
}

extern "C" int dh_lstm_cell_f32x(const float* gates, const float* c_cur, float* h_new, float* c_new, float* h_out, int ld_out,
                                 void* h_planes, long long plane, int ld_planes, int rows, int row_mult, int Hh, void* stream) {
    DH_REQUIRE(gates && c_cur && h_new && c_new && h_out && h_planes && plane > 0 && ld_planes >= Hh && rows > 0 && row_mult > 0 && Hh > 0);
    DhProfScope prof("dh_lstm_cell", 0.0, 0.0, stream);
    hipLaunchKernelGGL(lstm_cell_kernel<float>, dim3(rows), dim3(256), 0, (hipStream_t)stream, gates, c_cur, h_new, c_new, h_out, ld_out, rows,
                       row_mult, Hh, (uint16_t*)h_planes, (size_t)plane, ld_planes, (unsigned*)nullptr);
    DH_LAUNCH_CHECK();
}
