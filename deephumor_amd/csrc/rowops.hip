// Row-wise decoder kernels: token/position embedding lookup, residual + LayerNorm, encoder key mask.
// All are HBM/L2-bound streaming kernels: one wave per row, 16-byte loads, wave-shuffle reductions.
#include "common.h"

// x[rc,:] = (start slot ? start_emb[img] : tok_emb[token]) / scale + pos_emb[pos]
// (transformers.py:455-469: the image embedding is divided by sqrt(hid_dim) together with the tokens)
__global__ __launch_bounds__(256) void embed_rows_kernel(
    const float* __restrict__ tok_emb, const float* __restrict__ pos_emb, const float* __restrict__ start_emb,
    const int32_t* __restrict__ tokens, int tok_ld, float* __restrict__ x, int rows, int rows_per_img,
    int row_mult, int pos, int D, float scale) {
    const int rc = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (rc >= rows) return;
    const float* src;
    if (start_emb && pos == 0) {
        src = start_emb + (size_t)(rc / rows_per_img) * D;
    } else {
        const int rl = rc * row_mult;
        const int tok = tokens[(size_t)rl * tok_ld + pos - (start_emb ? 1 : 0)];
        src = tok_emb + (size_t)tok * D;
    }
    const float* pe = pos_emb + (size_t)pos * D;
    float* dst = x + (size_t)rc * D;
    for (int d = lane * 4; d < D; d += 256) {
        const float4 a = *reinterpret_cast<const float4*>(src + d);
        const float4 p = *reinterpret_cast<const float4*>(pe + d);
        float4 o;
        o.x = a.x / scale + p.x; o.y = a.y / scale + p.y; o.z = a.z / scale + p.z; o.w = a.w / scale + p.w;
        *reinterpret_cast<float4*>(dst + d) = o;
    }
}

extern "C" int dh_embed_rows(const void* tok_emb, const void* pos_emb, const void* start_emb,
                             const int32_t* tokens, int tok_ld, void* x, int rows, int rows_per_img,
                             int row_mult, int pos, int D, float scale, int dtype, void* stream) {
    if (dtype != DH_F32) return DH_ERR_UNSUPPORTED;
    DH_REQUIRE(tok_emb && pos_emb && x && rows > 0 && rows_per_img > 0 && row_mult > 0 && pos >= 0);
    DH_REQUIRE((D % 4) == 0 && (tokens || (start_emb && pos == 0)));
    hipLaunchKernelGGL(embed_rows_kernel, dim3(dh_cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream,
                       (const float*)tok_emb, (const float*)pos_emb, (const float*)start_emb, tokens, tok_ld,
                       (float*)x, rows, rows_per_img, row_mult, pos, D, scale);
    DH_LAUNCH_CHECK();
}

// out = LayerNorm(x + y): mean and biased variance over the row in two in-register passes
// (same formulation as torch's RowwiseMoments: var = E[(v-mean)^2]), rstd = 1/sqrt(var+eps).
template <int NV>   // float4 per lane kept in registers: D <= NV*256
__global__ __launch_bounds__(256) void add_layernorm_kernel(
    const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ gamma,
    const float* __restrict__ beta, float* __restrict__ out, int rows, int D, float eps) {
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (r >= rows) return;
    const float* xr = x + (size_t)r * D;
    const float* yr = y ? y + (size_t)r * D : nullptr;
    float4 v[NV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int d = lane * 4 + i * 256;
        if (d < D) {
            v[i] = *reinterpret_cast<const float4*>(xr + d);
            if (yr) {
                const float4 b = *reinterpret_cast<const float4*>(yr + d);
                v[i].x += b.x; v[i].y += b.y; v[i].z += b.z; v[i].w += b.w;
            }
            s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
        }
    }
    const float mean = wave_sum(s) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int d = lane * 4 + i * 256;
        if (d < D) {
            const float a = v[i].x - mean, b = v[i].y - mean, c = v[i].z - mean, e = v[i].w - mean;
            q += (a * a + b * b) + (c * c + e * e);
        }
    }
    const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)D + eps);
    float* o = out + (size_t)r * D;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int d = lane * 4 + i * 256;
        if (d < D) {
            const float4 g = *reinterpret_cast<const float4*>(gamma + d);
            const float4 b = *reinterpret_cast<const float4*>(beta + d);
            float4 w;
            w.x = (v[i].x - mean) * rstd * g.x + b.x; w.y = (v[i].y - mean) * rstd * g.y + b.y;
            w.z = (v[i].z - mean) * rstd * g.z + b.z; w.w = (v[i].w - mean) * rstd * g.w + b.w;
            *reinterpret_cast<float4*>(o + d) = w;
        }
    }
}

extern "C" int dh_add_layernorm(const void* x, const void* y, const float* gamma, const float* beta,
                                void* out, int rows, int D, float eps, int dtype, void* stream) {
    if (dtype != DH_F32) return DH_ERR_UNSUPPORTED;
    DH_REQUIRE(x && gamma && beta && out && rows > 0 && D > 0 && (D % 4) == 0 && D <= 4096);
    const dim3 grid(dh_cdiv(rows, 4)), block(256);
    hipStream_t s = (hipStream_t)stream;
#define DH_LN(NV) hipLaunchKernelGGL((add_layernorm_kernel<NV>), grid, block, 0, s, (const float*)x, \
                                     (const float*)y, gamma, beta, (float*)out, rows, D, eps)
    if (D <= 512) DH_LN(2); else if (D <= 1024) DH_LN(4); else if (D <= 2048) DH_LN(8); else DH_LN(16);
#undef DH_LN
    DH_LAUNCH_CHECK();
}

// keymask[r] = any(enc_out[r,:] == 0)  (transformers.py:480-481)
__global__ __launch_bounds__(256) void enc_key_mask_kernel(const float* __restrict__ e, uint8_t* __restrict__ m,
                                                            int rows, int D) {
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (r >= rows) return;
    int z = 0;
    for (int d = lane; d < D; d += 64) z |= (e[(size_t)r * D + d] == 0.f);
    z = __any(z);
    if (lane == 0) m[r] = (uint8_t)(z != 0);
}

extern "C" int dh_enc_key_mask(const void* enc_out, uint8_t* keymask, int rows, int D, int dtype, void* stream) {
    if (dtype != DH_F32) return DH_ERR_UNSUPPORTED;
    DH_REQUIRE(enc_out && keymask && rows > 0 && D > 0);
    hipLaunchKernelGGL(enc_key_mask_kernel, dim3(dh_cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream,
                       (const float*)enc_out, keymask, rows, D);
    DH_LAUNCH_CHECK();
}

// out[n, 0..E) = mean_j emb[labels[n, j], :]  (LabelEncoder, encoders.py:104: mean over ALL positions)
__global__ __launch_bounds__(256) void label_mean_kernel(const float* __restrict__ emb, const int64_t* __restrict__ labels,
                                                          float* __restrict__ out, int ld_out, int L, int E) {
    const int n = blockIdx.x;
    for (int d = threadIdx.x; d < E; d += 256) {
        float s = 0.f;
        for (int j = 0; j < L; ++j) s += emb[(size_t)labels[(size_t)n * L + j] * E + d];
        out[(size_t)n * ld_out + d] = s / (float)L;
    }
}

extern "C" int dh_label_mean(const void* emb, const int64_t* labels, void* out, int ld_out, int N, int L, int E,
                             int dtype, void* stream) {
    if (dtype != DH_F32) return DH_ERR_UNSUPPORTED;
    DH_REQUIRE(emb && labels && out && N > 0 && L > 0 && E > 0 && ld_out >= E);
    hipLaunchKernelGGL(label_mean_kernel, dim3(N), dim3(256), 0, (hipStream_t)stream, (const float*)emb, labels,
                       (float*)out, ld_out, L, E);
    DH_LAUNCH_CHECK();
}
