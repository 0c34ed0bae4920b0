// Row-wise decoder kernels: token/position embedding lookup, residual + LayerNorm, encoder key mask,
// label-embedding mean.  HBM/L2-bound streaming kernels: one wave per row, 16-byte accesses,
// wave-shuffle reductions, fp32 math on fp32 or bf16 storage.
#include "common.h"
#include "prof.h"

// x[rc,:] = (start slot ? start_emb[img] : tok_emb[token]) / scale + pos_emb[pos]
// (transformers.py:455-469: the image embedding is divided by sqrt(hid_dim) together with the tokens)
template <typename T>
__global__ __launch_bounds__(256) void embed_rows_kernel(
    const T* __restrict__ tok_emb, const T* __restrict__ pos_emb, const T* __restrict__ start_emb,
    const int32_t* __restrict__ tokens, int tok_ld, T* __restrict__ x, int rows, int rows_per_img,
    int row_mult, int pos, int D, float scale) {
    constexpr int VN = Vec16<T>::N;
    const int rc = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (rc >= rows) return;
    const T* src;
    if (start_emb && pos == 0) {
        src = start_emb + (size_t)(rc / rows_per_img) * D;
    } else {
        const int rl = rc * row_mult;
        const int tok = tokens[(size_t)rl * tok_ld + pos - (start_emb ? 1 : 0)];
        src = tok_emb + (size_t)tok * D;
    }
    const T* pe = pos_emb + (size_t)pos * D;
    T* dst = x + (size_t)rc * D;
    for (int d = lane * VN; d < D; d += 64 * VN) {
        float a[VN], b[VN], o[VN];
        load16(src + d, a);
        load16(pe + d, b);
#pragma unroll
        for (int i = 0; i < VN; ++i) o[i] = a[i] / scale + b[i];
        store16(dst + d, o);
    }
}

extern "C" int dh_embed_rows(const void* tok_emb, const void* pos_emb, const void* start_emb,
                             const int32_t* tokens, int tok_ld, void* x, int rows, int rows_per_img,
                             int row_mult, int pos, int D, float scale, int dtype, void* stream) {
    DH_REQUIRE(tok_emb && pos_emb && x && rows > 0 && rows_per_img > 0 && row_mult > 0 && pos >= 0);
    DH_REQUIRE((D % 8) == 0 && (tokens || (start_emb && pos == 0)));
    DhProfScope prof("dh_embed_rows", 0.0, 0.0, stream);
    DH_DISPATCH_T(dtype, hipLaunchKernelGGL(embed_rows_kernel<T>, dim3(dh_cdiv(rows, 4)), dim3(256), 0,
                                            (hipStream_t)stream, (const T*)tok_emb, (const T*)pos_emb,
                                            (const T*)start_emb, tokens, tok_ld, (T*)x, rows, rows_per_img,
                                            row_mult, pos, D, scale));
    DH_LAUNCH_CHECK();
}

// Teacher-forced (prefill) form of embed_rows: every position of every sequence, rows sequence-major
// (row n * n_pos + t): x = (t == 0 ? start_emb[n] : tok_emb[tokens[n, t-1]]) / scale + pos_emb[t].
template <typename T>
__global__ __launch_bounds__(256) void embed_prefill_kernel(
    const T* __restrict__ tok_emb, const T* __restrict__ pos_emb, const T* __restrict__ start_emb,
    const int32_t* __restrict__ tokens, int tok_ld, T* __restrict__ x, int rows, int n_pos, int D, float scale) {
    constexpr int VN = Vec16<T>::N;
    const int rc = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (rc >= rows) return;
    const int n = rc / n_pos, t = rc - n * n_pos;
    // with a start embedding the image occupies slot 0 and token j sits at position j + 1 (transformers.py:455-458)
    const T* src = start_emb ? (t == 0 ? start_emb + (size_t)n * D : tok_emb + (size_t)tokens[(size_t)n * tok_ld + t - 1] * D)
                             : tok_emb + (size_t)tokens[(size_t)n * tok_ld + t] * D;
    const T* pe = pos_emb + (size_t)t * D;
    T* dst = x + (size_t)rc * D;
    for (int d = lane * VN; d < D; d += 64 * VN) {
        float a[VN], b[VN], o[VN];
        load16(src + d, a);
        load16(pe + d, b);
#pragma unroll
        for (int i = 0; i < VN; ++i) o[i] = a[i] / scale + b[i];
        store16(dst + d, o);
    }
}

extern "C" int dh_embed_prefill(const void* tok_emb, const void* pos_emb, const void* start_emb, const int32_t* tokens,
                                int tok_ld, void* x, int n_seq, int n_pos, int D, float scale, int dtype, void* stream) {
    DH_REQUIRE(tok_emb && pos_emb && x && n_seq > 0 && n_pos > 0 && ((start_emb && n_pos == 1) || tokens) && (D % 8) == 0);
    DhProfScope prof("dh_embed_prefill", 0.0, 0.0, stream);
    const int rows = n_seq * n_pos;
    DH_DISPATCH_T(dtype, hipLaunchKernelGGL(embed_prefill_kernel<T>, dim3(dh_cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream,
                                            (const T*)tok_emb, (const T*)pos_emb, (const T*)start_emb, tokens, tok_ld, (T*)x,
                                            rows, n_pos, D, scale));
    DH_LAUNCH_CHECK();
}

// out = LayerNorm(x + y): mean and biased variance over the row in two in-register passes
// (same formulation as torch's RowwiseMoments: var = E[(v-mean)^2]), rstd = 1/sqrt(var+eps).
template <typename T, int NV>   // 16-byte vectors per lane kept in registers
__global__ __launch_bounds__(256) void add_layernorm_kernel(
    const T* __restrict__ x, const T* __restrict__ y, const float* __restrict__ gamma,
    const float* __restrict__ beta, T* __restrict__ out, int rows, int D, float eps,
    uint16_t* __restrict__ planes = nullptr, size_t plane = 0, unsigned* range_flag = nullptr) {
    constexpr int VN = Vec16<T>::N;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (r >= rows) return;
    const T* xr = x + (size_t)r * D;
    const T* yr = y ? y + (size_t)r * D : xr;
    // every load of the kernel is requested up front at a clamped (always valid) offset, without per-vector branches:
    // one memory round trip instead of one per conditional load (x, y, gamma, beta used to cost four)
    float v[NV][VN], b[NV][VN], gm[NV][VN], bt[NV][VN];
    bool live[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int d = (lane + i * 64) * VN;
        live[i] = d < D;
        const int dd = live[i] ? d : 0;
        load16(xr + dd, v[i]);
        load16(yr + dd, b[i]);
#pragma unroll
        for (int j = 0; j < VN; j += 4) {
            const float4 g4 = *reinterpret_cast<const float4*>(gamma + dd + j), b4 = *reinterpret_cast<const float4*>(beta + dd + j);
            gm[i][j] = g4.x; gm[i][j + 1] = g4.y; gm[i][j + 2] = g4.z; gm[i][j + 3] = g4.w;
            bt[i][j] = b4.x; bt[i][j + 1] = b4.y; bt[i][j + 2] = b4.z; bt[i][j + 3] = b4.w;
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        if (live[i]) {
#pragma unroll
            for (int j = 0; j < VN; ++j) { if (y) v[i][j] += b[i][j]; s += v[i][j]; }
        }
    }
    const float mean = wave_sum(s) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        if (live[i]) {
#pragma unroll
            for (int j = 0; j < VN; ++j) { const float a = v[i][j] - mean; q += a * a; }
        }
    }
    const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)D + eps);
    T* o = out + (size_t)r * D;
    float amax = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        if (live[i]) {
            float w[VN];
#pragma unroll
            for (int j = 0; j < VN; ++j) w[j] = (v[i][j] - mean) * rstd * gm[i][j] + bt[i][j];
            store16(o + (lane + i * 64) * VN, w);
            if (VN == 4 && planes) {                     // (fp32 rows only) the same values as the split planes of the next GEMM operand
                uint32_t h[2], l[2];
#pragma unroll
                for (int j = 0; j < 4; j += 2) {
                    const f16_t ha = (f16_t)w[j], hb = (f16_t)w[j + 1];
                    const f16_t la = (f16_t)((w[j] - (float)ha) * 2048.0f), lb = (f16_t)((w[j + 1] - (float)hb) * 2048.0f);
                    h[j / 2] = (uint32_t)__builtin_bit_cast(uint16_t, ha) | ((uint32_t)__builtin_bit_cast(uint16_t, hb) << 16);
                    l[j / 2] = (uint32_t)__builtin_bit_cast(uint16_t, la) | ((uint32_t)__builtin_bit_cast(uint16_t, lb) << 16);
                    amax = fmaxf(amax, fmaxf(fabsf(w[j]), fabsf(w[j + 1])));
                }
                const size_t po = (size_t)r * D + (lane + i * 64) * 4;
                *reinterpret_cast<uint2*>(planes + po) = make_uint2(h[0], h[1]);
                *reinterpret_cast<uint2*>(planes + plane + po) = make_uint2(l[0], l[1]);
            }
        }
    }
    if (VN == 4 && planes && amax >= 65504.0f) atomicOr(range_flag, 1u);
}

extern "C" int dh_add_layernorm(const void* x, const void* y, const float* gamma, const float* beta,
                                void* out, int rows, int D, float eps, int dtype, void* stream) {
    DH_REQUIRE(x && gamma && beta && out && rows > 0 && D > 0 && (D % 8) == 0 && D <= 4096);
    DH_REQUIRE(((uintptr_t)gamma % 16) == 0 && ((uintptr_t)beta % 16) == 0);
    DhProfScope prof("dh_add_layernorm", 0.0, 0.0, stream);
    const dim3 grid(dh_cdiv(rows, 4)), block(256);
    hipStream_t s = (hipStream_t)stream;
#define DH_LN(NV) hipLaunchKernelGGL((add_layernorm_kernel<T, NV>), grid, block, 0, s, (const T*)x, \
                                     (const T*)y, gamma, beta, (T*)out, rows, D, eps, (uint16_t*)nullptr, (size_t)0, (unsigned*)nullptr)
    DH_DISPATCH_T(dtype, {
        const int per_pass = 64 * Vec16<T>::N;
        if (D <= per_pass) DH_LN(1); else if (D <= 2 * per_pass) DH_LN(2); else if (D <= 4 * per_pass) DH_LN(4);
        else if (D <= 8 * per_pass) DH_LN(8); else DH_LN(16);
    });
#undef DH_LN
    DH_LAUNCH_CHECK();
}

// dh_add_layernorm on fp32 rows with the result ALSO stored as the fp16 planes [2][rows][D] of the split-operand GEMMs (hi, lo * 2^11):
// the LayerNorm's output is the residual of the next sublayer (fp32) and the operand of its first GEMM (planes, csrc/linear_f32x_wreg.hip)
unsigned* dh_f32x_range_flag_of(hipStream_t s);      // gemm_f32x.hip
extern "C" int dh_add_layernorm_f32x(const float* x, const float* y, const float* gamma, const float* beta, float* out, void* out_planes,
                                     int rows, int D, float eps, void* stream) {
    DH_REQUIRE(x && gamma && beta && out && out_planes && rows > 0 && D > 0 && (D % 8) == 0 && D <= 4096);
    DH_REQUIRE(((uintptr_t)gamma % 16) == 0 && ((uintptr_t)beta % 16) == 0 && ((uintptr_t)out_planes % 16) == 0);
    DhProfScope prof("dh_add_layernorm", 0.0, 0.0, stream);
    const dim3 grid(dh_cdiv(rows, 4)), block(256);
    hipStream_t s = (hipStream_t)stream;
    unsigned* flag = dh_f32x_range_flag_of(s);
    if (!flag) return DH_ERR_LAUNCH;
#define DH_LN(NV) hipLaunchKernelGGL((add_layernorm_kernel<float, NV>), grid, block, 0, s, x, y, gamma, beta, out, rows, D, eps, \
                                     (uint16_t*)out_planes, (size_t)rows * D, flag)
    if (D <= 256) DH_LN(1); else if (D <= 512) DH_LN(2); else if (D <= 1024) DH_LN(4); else if (D <= 2048) DH_LN(8); else DH_LN(16);
#undef DH_LN
    DH_LAUNCH_CHECK();
}

// keymask[r] = any(enc_out[r,:] == 0)  (transformers.py:480-481)
template <typename T>
__global__ __launch_bounds__(256) void enc_key_mask_kernel(const T* __restrict__ e, uint8_t* __restrict__ m,
                                                            int rows, int D) {
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (r >= rows) return;
    int z = 0;
    for (int d = lane; d < D; d += 64) z |= (ldf(e + (size_t)r * D + d) == 0.f);
    z = __any(z);
    if (lane == 0) m[r] = (uint8_t)(z != 0);
}

extern "C" int dh_enc_key_mask(const void* enc_out, uint8_t* keymask, int rows, int D, int dtype, void* stream) {
    DH_REQUIRE(enc_out && keymask && rows > 0 && D > 0);
    DhProfScope prof("dh_enc_key_mask", 0.0, 0.0, stream);
    DH_DISPATCH_T(dtype, hipLaunchKernelGGL(enc_key_mask_kernel<T>, dim3(dh_cdiv(rows, 4)), dim3(256), 0,
                                            (hipStream_t)stream, (const T*)enc_out, keymask, rows, D));
    DH_LAUNCH_CHECK();
}

// ---- mask helpers of the reference's module API (transformers.py:12-40, 480-481) ------------------------------------
// get_pad_mask: mask[b, q, k] = (key[b, k] == pad_index)
__global__ __launch_bounds__(256) void pad_mask_kernel(const int64_t* __restrict__ key, uint8_t* __restrict__ mask, int Lq, int Lk,
                                                        long long pad_index, size_t total) {
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < total; i += (size_t)gridDim.x * 256ull) {
        const size_t b = i / ((size_t)Lq * Lk);
        const int k = (int)(i % Lk);
        mask[i] = (uint8_t)(key[b * Lk + k] == pad_index);
    }
}

extern "C" int dh_pad_mask(const int64_t* key, uint8_t* mask, int bs, int Lq, int Lk, long long pad_index, void* stream) {
    DH_REQUIRE(key && mask && bs > 0 && Lq > 0 && Lk > 0);
    DhProfScope prof("dh_pad_mask", 0.0, 0.0, stream);
    const size_t total = (size_t)bs * Lq * Lk;
    const int grid = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
    hipLaunchKernelGGL(pad_mask_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, key, mask, Lq, Lk, pad_index, total);
    DH_LAUNCH_CHECK();
}

// get_autoregressive_mask: mask[b, q, k] = (k > q)   (torch.triu(ones, 1))
__global__ __launch_bounds__(256) void autoregressive_mask_kernel(uint8_t* __restrict__ mask, int L, size_t total) {
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < total; i += (size_t)gridDim.x * 256ull) {
        const int k = (int)(i % L), q = (int)((i / L) % L);
        mask[i] = (uint8_t)(k > q);
    }
}

extern "C" int dh_autoregressive_mask(uint8_t* mask, int bs, int L, void* stream) {
    DH_REQUIRE(mask && bs > 0 && L > 0);
    DhProfScope prof("dh_autoregressive_mask", 0.0, 0.0, stream);
    const size_t total = (size_t)bs * L * L;
    const int grid = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
    hipLaunchKernelGGL(autoregressive_mask_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, mask, L, total);
    DH_LAUNCH_CHECK();
}

// a |= b over n bytes (input_mask = pad_mask | autoregressive_mask, transformers.py:477)
__global__ __launch_bounds__(256) void mask_or_kernel(uint8_t* __restrict__ a, const uint8_t* __restrict__ b, size_t n) {
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += (size_t)gridDim.x * 256ull) a[i] = (uint8_t)((a[i] | b[i]) != 0);
}

extern "C" int dh_mask_or(uint8_t* a, const uint8_t* b, long long n, void* stream) {
    DH_REQUIRE(a && b && n > 0);
    DhProfScope prof("dh_mask_or", 0.0, 0.0, stream);
    const int grid = (int)((n + 255) / 256 < 16384 ? (n + 255) / 256 : 16384);
    hipLaunchKernelGGL(mask_or_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, a, b, (size_t)n);
    DH_LAUNCH_CHECK();
}

// enc_inp_mask[r] = all(enc_out[r, :] != 0) as int64 0/1  (transformers.py:480)
template <typename T>
__global__ __launch_bounds__(256) void enc_nonzero_rows_kernel(const T* __restrict__ e, int64_t* __restrict__ m, int rows, int D) {
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (r >= rows) return;
    int z = 0;
    for (int d = lane; d < D; d += 64) z |= (ldf(e + (size_t)r * D + d) == 0.f);
    z = __any(z);
    if (lane == 0) m[r] = z ? 0 : 1;
}

extern "C" int dh_enc_nonzero_rows(const void* enc_out, int64_t* out, int rows, int D, int dtype, void* stream) {
    DH_REQUIRE(enc_out && out && rows > 0 && D > 0);
    DhProfScope prof("dh_enc_nonzero_rows", 0.0, 0.0, stream);
    DH_DISPATCH_T(dtype, hipLaunchKernelGGL(enc_nonzero_rows_kernel<T>, dim3(dh_cdiv(rows, 4)), dim3(256), 0,
                                            (hipStream_t)stream, (const T*)enc_out, out, rows, D));
    DH_LAUNCH_CHECK();
}

// out[n, 0..E) = mean_j emb[labels[n, j], :]  (LabelEncoder, encoders.py:104: mean over ALL positions)
template <typename T>
__global__ __launch_bounds__(256) void label_mean_kernel(const T* __restrict__ emb, const int64_t* __restrict__ labels,
                                                          T* __restrict__ out, int ld_out, int L, int E) {
    const int n = blockIdx.x;
    for (int d = threadIdx.x; d < E; d += 256) {
        float s = 0.f;
        for (int j = 0; j < L; ++j) s += ldf(emb + (size_t)labels[(size_t)n * L + j] * E + d);
        stf(out + (size_t)n * ld_out + d, s / (float)L);
    }
}

extern "C" int dh_label_mean(const void* emb, const int64_t* labels, void* out, int ld_out, int N, int L, int E,
                             int dtype, void* stream) {
    DH_REQUIRE(emb && labels && out && N > 0 && L > 0 && E > 0 && ld_out >= E);
    DhProfScope prof("dh_label_mean", 0.0, 0.0, stream);
    DH_DISPATCH_T(dtype, hipLaunchKernelGGL(label_mean_kernel<T>, dim3(N), dim3(256), 0, (hipStream_t)stream,
                                            (const T*)emb, labels, (T*)out, ld_out, L, E));
    DH_LAUNCH_CHECK();
}

// ---- teacher-forced scoring (deephumor/experiments/metrics.py:4-9) ---------------------------------------------
// logp[r] = log_softmax(logits[r, :])[targets[r]] = logits[r, t] - max - log(sum(exp(logits - max))).
// One workgroup per row, the V logits are read once (HBM-bound: rows*V*4 B), two block reductions.
__global__ __launch_bounds__(256) void token_logprob_kernel(const float* __restrict__ logits, int ldl, int V,
                                                             const int64_t* __restrict__ targets, float* __restrict__ logp) {
    __shared__ float red[256];
    const int r = blockIdx.x, tid = threadIdx.x;
    const float* row = logits + (size_t)r * ldl;
    float m = -INFINITY;
    for (int i = tid; i < V; i += 256) m = fmaxf(m, row[i]);
    red[tid] = m;
    __syncthreads();
    for (int s2 = 128; s2 > 0; s2 >>= 1) { if (tid < s2) red[tid] = fmaxf(red[tid], red[tid + s2]); __syncthreads(); }
    m = red[0];
    __syncthreads();
    float s = 0.f;
    for (int i = tid; i < V; i += 256) s += expf(row[i] - m);
    red[tid] = s;
    __syncthreads();
    for (int s2 = 128; s2 > 0; s2 >>= 1) { if (tid < s2) red[tid] += red[tid + s2]; __syncthreads(); }
    if (tid == 0) {
        const int64_t t = targets[r];
        logp[r] = (t >= 0 && t < V) ? (row[t] - m) - logf(red[0]) : 0.f;
    }
}

extern "C" int dh_token_logprob(const float* logits, int ldl, int V, const int64_t* targets, float* logp, int rows,
                                void* stream) {
    DH_REQUIRE(logits && targets && logp && rows > 0 && V > 0 && ldl >= V);
    DhProfScope prof("dh_token_logprob", 0.0, 4.0 * rows * V, stream);
    hipLaunchKernelGGL(token_logprob_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, logits, ldl, V, targets, logp);
    DH_LAUNCH_CHECK();
}

// pp[b] = exp(-sum_t [targets[b,t] != pad] * logp[b,t] / lengths[b])   (metrics.py:5-8), one wave per sequence
__global__ __launch_bounds__(64) void seq_perplexity_kernel(const float* __restrict__ logp, const int64_t* __restrict__ targets,
                                                             const int64_t* __restrict__ lengths, float* __restrict__ pp,
                                                             int L, int pad_index) {
    const int b = blockIdx.x, lane = threadIdx.x;
    const float len = (float)lengths[b];
    float s = 0.f;
    for (int t = lane; t < L; t += 64)
        if (targets[(size_t)b * L + t] != pad_index) s += logp[(size_t)b * L + t] / len;
    s = wave_sum(s);
    if (lane == 0) pp[b] = expf(-s);
}

extern "C" int dh_seq_perplexity(const float* logp, const int64_t* targets, const int64_t* lengths, float* pp,
                                 int n_seq, int L, int pad_index, void* stream) {
    DH_REQUIRE(logp && targets && lengths && pp && n_seq > 0 && L > 0);
    DhProfScope prof("dh_seq_perplexity", 0.0, 0.0, stream);
    hipLaunchKernelGGL(seq_perplexity_kernel, dim3(n_seq), dim3(64), 0, (hipStream_t)stream, logp, targets, lengths, pp, L,
                       pad_index);
    DH_LAUNCH_CHECK();
}
