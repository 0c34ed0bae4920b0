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
};

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
    for (int d = tid * VN; d < E; d += 256 * VN)
        *reinterpret_cast<uint4*>(x0 + d) = *reinterpret_cast<const uint4*>(xin + d);
    for (int l = 0; l < p.n_layers; ++l) {
        const T* hs = hp >= 0 ? p.h_prev + ((size_t)l * p.rows_total + hp) * Hh : nullptr;
        const float* cs = hp >= 0 ? p.c_prev + ((size_t)l * p.rows_total + hp) * Hh : nullptr;
        T* hd = l == 0 ? x0 + E : p.xcatl + ((size_t)(l - 1) * p.rows + rc) * (2 * Hh) + Hh;
        float* cd = p.c_cur + ((size_t)l * p.rows + rc) * Hh;
        for (int d = tid * VN; d < Hh; d += 256 * VN)
            *reinterpret_cast<uint4*>(hd + d) = hs ? *reinterpret_cast<const uint4*>(hs + d) : zero;
        for (int d = tid * 4; d < Hh; d += 1024)
            *reinterpret_cast<uint4*>(cd + d) = cs ? *reinterpret_cast<const uint4*>(cs + d) : zero;
    }
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
                            n_layers, E, Hh};
        hipLaunchKernelGGL(lstm_prepare_kernel<T>, dim3(rows), dim3(256), 0, (hipStream_t)stream, p);
    });
    DH_LAUNCH_CHECK();
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

// gates row layout: [i | f | g | o], each Hh wide (PyTorch order), fp32.
template <typename T>
__global__ __launch_bounds__(256) void lstm_cell_kernel(
    const float* __restrict__ gates, const float* __restrict__ c_cur, T* __restrict__ h_new,
    float* __restrict__ c_new, T* __restrict__ h_out, int ld_out, int rows, int row_mult, int Hh) {
    const int rc = blockIdx.x;
    const int rl = rc * row_mult;
    const float* g = gates + (size_t)rc * 4 * Hh;
    for (int d = threadIdx.x; d < Hh; d += 256) {
        const float c1 = sigmoidf_(g[Hh + d]) * c_cur[(size_t)rc * Hh + d] + sigmoidf_(g[d]) * tanhf(g[2 * Hh + d]);
        const float h1 = sigmoidf_(g[3 * Hh + d]) * tanhf(c1);
        c_new[(size_t)rl * Hh + d] = c1;
        stf(h_new + (size_t)rl * Hh + d, h1);
        stf(h_out + (size_t)rc * ld_out + d, h1);
    }
}

extern "C" int dh_lstm_cell(const float* gates, const float* c_cur, void* h_new, float* c_new, void* h_out,
                            int ld_out, int rows, int row_mult, int Hh, int dtype, void* stream) {
    DH_REQUIRE(gates && c_cur && h_new && c_new && h_out && rows > 0 && row_mult > 0 && Hh > 0);
    DhProfScope prof("dh_lstm_cell", 0.0, 0.0, stream);
    DH_DISPATCH_T(dtype, hipLaunchKernelGGL(lstm_cell_kernel<T>, dim3(rows), dim3(256), 0, (hipStream_t)stream, gates,
                                            c_cur, (T*)h_new, c_new, (T*)h_out, ld_out, rows, row_mult, Hh));
    DH_LAUNCH_CHECK();
}
