// LSTM single-time-step kernels (nn.LSTM of rnn_models.py:23-24, one step of :80 / :108).
// The four gate products run on the matrix cores through dh_linear over the concatenated
// [x | h] operand; these two HBM-bound kernels gather that operand (including the beam reorder of
// the recurrent state -- an index gather, the state arrays are never permuted in place) and apply
// the pointwise cell update.
#include "common.h"

struct LstmPrepParams {
    const float* emb; const float* img_emb; const int32_t* tokens; int tok_ld, tok_pos;
    const int32_t* hparent; const float* h_prev; const float* c_prev;
    float* xcat0; float* xcatl; float* c_cur;
    int rows, rows_per_img, row_mult, rows_total, n_layers, E, Hh;
};

__global__ __launch_bounds__(256) void lstm_prepare_kernel(LstmPrepParams p) {
    const int rc = blockIdx.x, tid = threadIdx.x;
    const int rl = rc * p.row_mult;
    int hp = p.hparent ? p.hparent[rl] : rl;
    if (!p.h_prev) hp = -1;
    const float* xin = p.tokens ? p.emb + (size_t)p.tokens[(size_t)rl * p.tok_ld + p.tok_pos] * p.E
                                : p.img_emb + (size_t)(rc / p.rows_per_img) * p.E;
    const int E = p.E, Hh = p.Hh;
    float* x0 = p.xcat0 + (size_t)rc * (E + Hh);
    for (int d = tid * 4; d < E; d += 1024) *reinterpret_cast<float4*>(x0 + d) = *reinterpret_cast<const float4*>(xin + d);
    const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int l = 0; l < p.n_layers; ++l) {
        const float* hs = hp >= 0 ? p.h_prev + ((size_t)l * p.rows_total + hp) * Hh : nullptr;
        const float* cs = hp >= 0 ? p.c_prev + ((size_t)l * p.rows_total + hp) * Hh : nullptr;
        float* hd = l == 0 ? x0 + E : p.xcatl + ((size_t)(l - 1) * p.rows + rc) * (2 * Hh) + Hh;
        float* cd = p.c_cur + ((size_t)l * p.rows + rc) * Hh;
        for (int d = tid * 4; d < Hh; d += 1024) {
            *reinterpret_cast<float4*>(hd + d) = hs ? *reinterpret_cast<const float4*>(hs + d) : zero;
            *reinterpret_cast<float4*>(cd + d) = cs ? *reinterpret_cast<const float4*>(cs + d) : zero;
        }
    }
}

extern "C" int dh_lstm_prepare(const void* emb, const void* img_emb, const int32_t* tokens, int tok_ld, int tok_pos,
                               const int32_t* hparent, const void* h_prev, const void* c_prev,
                               void* xcat0, void* xcatl, void* c_cur, int rows, int rows_per_img, int row_mult,
                               int rows_total, int n_layers, int E, int Hh, int dtype, void* stream) {
    if (dtype != DH_F32) return DH_ERR_UNSUPPORTED;
    DH_REQUIRE(xcat0 && c_cur && rows > 0 && rows_per_img > 0 && row_mult > 0 && n_layers > 0);
    DH_REQUIRE((tokens && emb) || img_emb);
    DH_REQUIRE((E % 4) == 0 && (Hh % 4) == 0 && (n_layers == 1 || xcatl) && ((h_prev == nullptr) == (c_prev == nullptr)));
    LstmPrepParams p{(const float*)emb, (const float*)img_emb, tokens, tok_ld, tok_pos, hparent,
                     (const float*)h_prev, (const float*)c_prev, (float*)xcat0, (float*)xcatl, (float*)c_cur,
                     rows, rows_per_img, row_mult, rows_total, n_layers, E, Hh};
    hipLaunchKernelGGL(lstm_prepare_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, p);
    DH_LAUNCH_CHECK();
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

// gates row layout: [i | f | g | o], each Hh wide (PyTorch order).
__global__ __launch_bounds__(256) void lstm_cell_kernel(
    const float* __restrict__ gates, const float* __restrict__ c_cur, float* __restrict__ h_new,
    float* __restrict__ c_new, float* __restrict__ h_out, int ld_out, int rows, int row_mult, int Hh) {
    const int rc = blockIdx.x;
    const int rl = rc * row_mult;
    const float* g = gates + (size_t)rc * 4 * Hh;
    for (int d = threadIdx.x * 4; d < Hh; d += 1024) {
        const float4 gi = *reinterpret_cast<const float4*>(g + d);
        const float4 gf = *reinterpret_cast<const float4*>(g + Hh + d);
        const float4 gg = *reinterpret_cast<const float4*>(g + 2 * Hh + d);
        const float4 go = *reinterpret_cast<const float4*>(g + 3 * Hh + d);
        const float4 c0 = *reinterpret_cast<const float4*>(c_cur + (size_t)rc * Hh + d);
        float4 c1, h1;
        c1.x = sigmoidf_(gf.x) * c0.x + sigmoidf_(gi.x) * tanhf(gg.x); h1.x = sigmoidf_(go.x) * tanhf(c1.x);
        c1.y = sigmoidf_(gf.y) * c0.y + sigmoidf_(gi.y) * tanhf(gg.y); h1.y = sigmoidf_(go.y) * tanhf(c1.y);
        c1.z = sigmoidf_(gf.z) * c0.z + sigmoidf_(gi.z) * tanhf(gg.z); h1.z = sigmoidf_(go.z) * tanhf(c1.z);
        c1.w = sigmoidf_(gf.w) * c0.w + sigmoidf_(gi.w) * tanhf(gg.w); h1.w = sigmoidf_(go.w) * tanhf(c1.w);
        *reinterpret_cast<float4*>(c_new + (size_t)rl * Hh + d) = c1;
        *reinterpret_cast<float4*>(h_new + (size_t)rl * Hh + d) = h1;
        *reinterpret_cast<float4*>(h_out + (size_t)rc * ld_out + d) = h1;
    }
}

extern "C" int dh_lstm_cell(const void* gates, const void* c_cur, void* h_new, void* c_new, void* h_out,
                            int ld_out, int rows, int row_mult, int Hh, int dtype, void* stream) {
    if (dtype != DH_F32) return DH_ERR_UNSUPPORTED;
    DH_REQUIRE(gates && c_cur && h_new && c_new && h_out && rows > 0 && row_mult > 0 && (Hh % 4) == 0 && (ld_out % 4) == 0);
    hipLaunchKernelGGL(lstm_cell_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, (const float*)gates,
                       (const float*)c_cur, (float*)h_new, (float*)c_new, (float*)h_out, ld_out, rows, row_mult, Hh);
    DH_LAUNCH_CHECK();
}
