// dh_linear: C = act((A * W^T + bias) * scale + shift), fp32 storage, exact-fp32 accumulation on the
// matrix cores with v_mfma_f32_32x32x2_f32 (a k-ordered fmaf chain, bit-identical to VALU fma).
//
// Replaces the nn.Linear / nn.LSTM gate products of the reference (see include/deephumor_hip.h).
// Tiling (wave64): a workgroup = 4 waves; tile BMxBN with each wave owning a (BM/2)x(BN/2) quadrant
// made of 32x32 MFMA tiles; K is consumed in BK=32 slabs staged through LDS as [row][BK+1]
// (odd stride -> the per-lane operand reads A[row=l&31][k=l>>5] hit 32 distinct banks per half-wave).
// Global->LDS goes through registers so the next slab's loads are in flight during the MFMAs.
// Grid: one block per tile, block ids remapped so that consecutive ids (which share a W panel)
// stay on one XCD and find it in that XCD's L2.
#include "common.h"
#include "prof.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int BM, int BN>
__global__ __launch_bounds__(256) void linear_f32_kernel(
    const float* __restrict__ A, int lda, const float* __restrict__ W, int ldw,
    const float* __restrict__ bias, const float* __restrict__ scale, const float* __restrict__ shift,
    const float* __restrict__ res, int ldres,
    float* __restrict__ C, int ldc, int M, int N, int K, int relu, int tiles_m, int tiles_n) {
    constexpr int BK = 32, LD = BK + 1;
    constexpr int TM = BM / 64, TN = BN / 64;          // 32x32 tiles per wave in each direction
    constexpr int PA = BM / 32, PB = BN / 32;          // float4 loads per thread per slab
    __shared__ float As[BM * LD];
    __shared__ float Bs[BN * LD];

    // XCD-aware remap: blocks b and b+8 share an XCD; give each XCD a contiguous run of tiles.
    const int nblk = tiles_m * tiles_n;
    int bid = blockIdx.x;
    {
        const int q = nblk / 8, r = nblk % 8, xcd = bid % 8, idx = bid / 8;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int tm = bid % tiles_m, tn = bid / tiles_m;   // row tiles fastest: neighbours share the W panel
    const int m0 = tm * BM, n0 = tn * BN;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm0 = (wave & 1) * (BM / 2), wn0 = (wave >> 1) * (BN / 2);
    const int lrow = tid >> 3, lk = (tid & 7) * 4;     // loader coordinates inside a slab

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    float4 ra[PA], rb[PB];
    auto load_slab = [&](int k0) {
#pragma unroll
        for (int p = 0; p < PA; ++p) {
            // unconditional load at a clamped offset, then select: conditional loads are waited for one by one
            const int m = m0 + lrow + 32 * p, k = k0 + lk;
            const bool ok = m < M && k < K;
            const float4 t = *reinterpret_cast<const float4*>(A + (ok ? (size_t)m * lda + k : 0));
            ra[p] = ok ? t : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int p = 0; p < PB; ++p) {
            const int n = n0 + lrow + 32 * p, k = k0 + lk;
            const bool ok = n < N && k < K;
            const float4 t = *reinterpret_cast<const float4*>(W + (ok ? (size_t)n * ldw + k : 0));
            rb[p] = ok ? t : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto store_slab = [&]() {
#pragma unroll
        for (int p = 0; p < PA; ++p) {
            float* d = As + (lrow + 32 * p) * LD + lk;
            d[0] = ra[p].x; d[1] = ra[p].y; d[2] = ra[p].z; d[3] = ra[p].w;
        }
#pragma unroll
        for (int p = 0; p < PB; ++p) {
            float* d = Bs + (lrow + 32 * p) * LD + lk;
            d[0] = rb[p].x; d[1] = rb[p].y; d[2] = rb[p].z; d[3] = rb[p].w;
        }
    };

    load_slab(0);
    store_slab();
    __syncthreads();
    const int l31 = lane & 31, lhi = lane >> 5;
    for (int k0 = 0; k0 < K; k0 += BK) {
        const bool more = k0 + BK < K;
        if (more) load_slab(k0 + BK);
#pragma unroll
        for (int kk = 0; kk < BK; kk += 2) {
            float a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) a[i] = As[(wm0 + 32 * i + l31) * LD + kk + lhi];
#pragma unroll
            for (int j = 0; j < TN; ++j) b[j] = Bs[(wn0 + 32 * j + l31) * LD + kk + lhi];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
        if (more) { store_slab(); __syncthreads(); }
    }

    // epilogue: accumulator register r of lane l is C[row=(r&3)+8*(r>>2)+4*(l>>5)][col=l&31]
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + wn0 + 32 * j + l31;
        if (n >= N) continue;
        const float bv = bias ? bias[n] : 0.f;
        const float sc = scale ? scale[n] : 1.f, sh = scale ? shift[n] : 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm0 + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * lhi;
                if (m >= M) continue;
                float v = acc[i][j][r] + bv;
                if (scale) v = v * sc + sh;
                if (res) v += res[(size_t)m * ldres + n];
                if (relu) v = fmaxf(v, 0.f);
                C[(size_t)m * ldc + n] = v;
            }
        }
    }
}

int dh_linear_bf16_impl(const void* A, int lda, const void* W, int ldw, const float* bias, const float* scale,
                        const float* shift, const void* residual, int ldres, void* C, int ldc, int M, int N, int K,
                        int relu, int out_f32, int f16, hipStream_t s);

extern "C" int dh_linear(const void* A, int lda, const void* W, int ldw, const float* bias,
                         const float* scale, const float* shift, const void* residual, int ldres,
                         void* C, int ldc, int M, int N, int K, int relu, int dtype, void* stream) {
    DH_REQUIRE(A && W && C && M > 0 && N > 0 && K > 0);
    DH_REQUIRE((scale == nullptr) == (shift == nullptr));
    DH_REQUIRE(!residual || ldres >= N);
    hipStream_t s = (hipStream_t)stream;
    const double esz = dtype == DH_F32 ? 4.0 : 2.0;
    dh_prof_set_dims(M, N, K);
    DhProfScope prof("dh_linear", 2.0 * M * N * K, esz * ((double)M * K + (double)N * K) + (DH_IS_16BIT(dtype) ? 2.0 : 4.0) * M * N, stream);
    if (dtype == DH_BF16 || dtype == DH_BF16_OUT_F32 || dtype == DH_F16 || dtype == DH_F16_OUT_F32)
        return dh_linear_bf16_impl(A, lda, W, ldw, bias, scale, shift, residual, ldres, C, ldc, M, N, K, relu,
                                   dtype == DH_BF16_OUT_F32 || dtype == DH_F16_OUT_F32, dtype == DH_F16 || dtype == DH_F16_OUT_F32, s);
    if (dtype != DH_F32) return DH_ERR_UNSUPPORTED;
    DH_REQUIRE((K % 4) == 0 && (lda % 4) == 0 && (ldw % 4) == 0 && lda >= K && ldw >= K && ldc >= N);
    DH_REQUIRE(((uintptr_t)A % 16) == 0 && ((uintptr_t)W % 16) == 0);
    const long long big_tiles = (long long)dh_cdiv(M, 128) * dh_cdiv(N, 128);
    if (big_tiles >= 192 && M >= 96) {
        const int tm = dh_cdiv(M, 128), tn = dh_cdiv(N, 128);
        hipLaunchKernelGGL((linear_f32_kernel<128, 128>), dim3(tm * tn), dim3(256), 0, s,
                           (const float*)A, lda, (const float*)W, ldw, bias, scale, shift, (const float*)residual, ldres,
                           (float*)C, ldc, M, N, K, relu, tm, tn);
    } else {
        const int tm = dh_cdiv(M, 64), tn = dh_cdiv(N, 64);
        hipLaunchKernelGGL((linear_f32_kernel<64, 64>), dim3(tm * tn), dim3(256), 0, s,
                           (const float*)A, lda, (const float*)W, ldw, bias, scale, shift, (const float*)residual, ldres,
                           (float*)C, ldc, M, N, K, relu, tm, tn);
    }
    DH_LAUNCH_CHECK();
}
