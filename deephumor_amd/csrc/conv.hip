// Encoder trunk kernels (torchvision ResNet-50 as used at encoders.py:34-38,56): direct NCHW
// convolution as an implicit GEMM on the vector ALUs with folded eval-mode BatchNorm, optional
// residual add and ReLU fused in the epilogue; 3x3/2 max-pool; global average pool; NCHW->rows.
//
// conv tile: TM output channels x 128 output pixels (pixels run over n,oh,ow so small feature maps
// of different images share a tile), reduction over k=(ci,kh,kw) in slabs of 16 staged through LDS:
//   As[k][co]  weights, transposed on the way in (global rows are k-contiguous)
//   Bs[k][px]  the input window ("im2col" on the fly: each lane owns ONE output pixel for the whole
//              reduction, so its (n,oh,ow) decode is done once and every k is a wave-uniform offset;
//              a wave's 64 lanes read 64 consecutive ow -> coalesced 256-B rows for stride 1)
// Each thread accumulates an 8x8 (TM=128) or 4x8 (TM=64) register tile with 16-byte LDS reads
// (conflict-free: 16 lanes x 16 B cover one 256-B bank row, the co operand is a broadcast).
// Loads for slab s+1 are issued before the FMAs of slab s and written to LDS after them.
#include "common.h"
#include "prof.h"

template <int KS> struct KDecode {
    __device__ static __forceinline__ void run(int k, int& ci, int& kh, int& kw) {
        ci = k / (KS * KS); const int r = k - ci * (KS * KS); kh = r / KS; kw = r - kh * KS;
    }
};
template <> struct KDecode<1> {
    __device__ static __forceinline__ void run(int k, int& ci, int& kh, int& kw) { ci = k; kh = 0; kw = 0; }
};

struct ConvParams {
    const float* x; const float* w; const float* scale; const float* shift; const float* res; float* y;
    int N, Cin, H, W, Cout, Ho, Wo, stride, pad, relu, K, P;   // K = Cin*KS*KS, P = N*Ho*Wo
};

// OUT16 = void: NCHW fp32 output (parity path); bf16_t / f16_t: channels-last 16-bit output (stem of the 16-bit paths)
template <int TM, int KS, typename OUT16>
__global__ __launch_bounds__(256) void conv_bn_act_kernel(ConvParams p) {
    constexpr bool NHWC_BF16_OUT = !__is_same(OUT16, void);
    constexpr int TN = 128, BK = 16, LDA = TM + 4;
    constexpr int CO_T = TM / 16;            // output channels per thread (8 or 4)
    constexpr int AK = (TM * BK) / 256;      // weights loaded per thread per slab (8 or 4)
    __shared__ __attribute__((aligned(16))) float As[BK * LDA];
    __shared__ __attribute__((aligned(16))) float Bs[BK * TN];

    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
    const int co0 = blockIdx.y * TM, p0 = blockIdx.x * TN;
    const int HoWo = p.Ho * p.Wo, HW = p.H * p.W;

    // ---- loader roles -------------------------------------------------------------------------
    const int a_co = tid / (BK / AK), a_k = (tid % (BK / AK)) * AK;     // weight row / first k of this thread
    const bool a_ok = co0 + a_co < p.Cout;
    const float* a_ptr = p.w + (size_t)(a_ok ? co0 + a_co : 0) * p.K;
    const int b_px = tid & 127, b_k = (tid >> 7) * 8;
    const int pp = p0 + b_px;
    const bool b_ok = pp < p.P;
    int ih0 = 0, iw0 = 0;
    const float* b_ptr = p.x;
    if (b_ok) {
        const int n = pp / HoWo, r = pp - n * HoWo, oh = r / p.Wo, ow = r - oh * p.Wo;
        ih0 = oh * p.stride - p.pad; iw0 = ow * p.stride - p.pad;
        b_ptr = p.x + (size_t)n * p.Cin * HW;
    }
    float ra[AK], rb[8];
    auto load_slab = [&](int k0) {
#pragma unroll
        for (int i = 0; i < AK; ++i) {
            // unconditional loads at clamped offsets, then select: conditional loads are waited for one by one
            const int k = k0 + a_k + i;
            const bool ok = a_ok && k < p.K;
            const float t = a_ptr[ok ? k : 0];
            ra[i] = ok ? t : 0.f;
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int k = k0 + b_k + i;
            int ci, kh, kw;
            KDecode<KS>::run(k, ci, kh, kw);
            const int ih = ih0 + kh, iw = iw0 + kw;
            const bool ok = b_ok && k < p.K && (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
            const float t = b_ptr[ok ? (size_t)ci * HW + ih * p.W + iw : 0];
            rb[i] = ok ? t : 0.f;
        }
    };
    auto store_slab = [&]() {
#pragma unroll
        for (int i = 0; i < AK; ++i) As[(a_k + i) * LDA + a_co] = ra[i];
#pragma unroll
        for (int i = 0; i < 8; ++i) Bs[(b_k + i) * TN + b_px] = rb[i];
    };

    float acc[CO_T][8];
#pragma unroll
    for (int i = 0; i < CO_T; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = 0.f;

    load_slab(0);
    store_slab();
    __syncthreads();
    for (int k0 = 0; k0 < p.K; k0 += BK) {
        const bool more = k0 + BK < p.K;
        if (more) load_slab(k0 + BK);
#pragma unroll
        for (int k = 0; k < BK; ++k) {
            float a[CO_T], b[8];
            *reinterpret_cast<float4*>(&a[0]) = *reinterpret_cast<const float4*>(&As[k * LDA + ty * 4]);
            if (CO_T == 8) *reinterpret_cast<float4*>(&a[4]) = *reinterpret_cast<const float4*>(&As[k * LDA + 64 + ty * 4]);
            *reinterpret_cast<float4*>(&b[0]) = *reinterpret_cast<const float4*>(&Bs[k * TN + tx * 4]);
            *reinterpret_cast<float4*>(&b[4]) = *reinterpret_cast<const float4*>(&Bs[k * TN + 64 + tx * 4]);
#pragma unroll
            for (int i = 0; i < CO_T; ++i)
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[i][j] = fmaf(a[i], b[j], acc[i][j]);
        }
        __syncthreads();
        if (more) { store_slab(); __syncthreads(); }
    }

    // ---- epilogue: y = act(acc*scale + shift (+ residual)) --------------------------------------
    if constexpr (NHWC_BF16_OUT) {
        // stem of the 16-bit paths: NCHW fp32 image in, channels-last bf16 / fp16 out (4 consecutive channels per store)
        uint16_t* yo = reinterpret_cast<uint16_t*>(p.y);
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int pj = p0 + g * 64 + tx * 4 + j;
                if (pj >= p.P) continue;
#pragma unroll
                for (int ig = 0; ig < CO_T / 4; ++ig) {
                    const int cb = co0 + ig * 64 + ty * 4;
                    if (cb >= p.Cout) continue;
                    uint16_t h[4];
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        float v = acc[ig * 4 + c][g * 4 + j] * p.scale[cb + c] + p.shift[cb + c];
                        if (p.relu) v = fmaxf(v, 0.f);
                        h[c] = Op16<OUT16>::from_f32(v);
                    }
                    uint2 pk;
                    pk.x = (uint32_t)h[0] | ((uint32_t)h[1] << 16);
                    pk.y = (uint32_t)h[2] | ((uint32_t)h[3] << 16);
                    *reinterpret_cast<uint2*>(yo + (size_t)pj * p.Cout + cb) = pk;
                }
            }
        return;
    }
    const bool vec = (HoWo % 4) == 0;
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        const int pq = p0 + g * 64 + tx * 4;          // first of 4 consecutive output pixels
        if (pq >= p.P) continue;
        const int n = pq / HoWo, r = pq - n * HoWo;
#pragma unroll
        for (int i = 0; i < CO_T; ++i) {
            const int co = co0 + (i >> 2) * 64 + ty * 4 + (i & 3);
            if (co >= p.Cout) continue;
            const float sc = p.scale[co], sh = p.shift[co];
            if (vec) {      // 4 pixels stay inside one image and the address is 16-B aligned
                const size_t o = ((size_t)n * p.Cout + co) * HoWo + r;
                float4 v;
                v.x = acc[i][g * 4 + 0] * sc + sh; v.y = acc[i][g * 4 + 1] * sc + sh;
                v.z = acc[i][g * 4 + 2] * sc + sh; v.w = acc[i][g * 4 + 3] * sc + sh;
                if (p.res) {
                    const float4 q = *reinterpret_cast<const float4*>(p.res + o);
                    v.x += q.x; v.y += q.y; v.z += q.z; v.w += q.w;
                }
                if (p.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                *reinterpret_cast<float4*>(p.y + o) = v;
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int pj = pq + j;
                    if (pj >= p.P) break;
                    const int nj = pj / HoWo, rj = pj - nj * HoWo;
                    const size_t o = ((size_t)nj * p.Cout + co) * HoWo + rj;
                    float v = acc[i][g * 4 + j] * sc + sh;
                    if (p.res) v += p.res[o];
                    if (p.relu) v = fmaxf(v, 0.f);
                    p.y[o] = v;
                }
            }
        }
    }
}

template <int TM, int KS, typename OUT16 = void>
static void launch_conv(const ConvParams& p, hipStream_t s) {
    hipLaunchKernelGGL((conv_bn_act_kernel<TM, KS, OUT16>), dim3(dh_cdiv(p.P, 128), dh_cdiv(p.Cout, TM)),
                       dim3(256), 0, s, p);
}

extern "C" int dh_conv2d_bn_act(const void* x, const void* w, const float* scale, const float* shift,
                                const void* residual, void* y, int N, int Cin, int H, int W, int Cout,
                                int KH, int KW, int stride, int pad, int relu, int dtype, void* stream) {
    if (dtype != DH_F32) return DH_ERR_UNSUPPORTED;
    DH_REQUIRE(x && w && scale && shift && y && N > 0 && Cin > 0 && H > 0 && W > 0 && Cout > 0);
    DH_REQUIRE(KH == KW && (KH == 1 || KH == 3 || KH == 7) && stride >= 1 && pad >= 0);
    ConvParams p{(const float*)x, (const float*)w, scale, shift, (const float*)residual, (float*)y,
                 N, Cin, H, W, Cout, (H + 2 * pad - KH) / stride + 1, (W + 2 * pad - KW) / stride + 1,
                 stride, pad, relu, Cin * KH * KW, 0};
    DH_REQUIRE(p.Ho > 0 && p.Wo > 0 && (long long)N * p.Ho * p.Wo < (1ll << 31));
    p.P = N * p.Ho * p.Wo;
    hipStream_t s = (hipStream_t)stream;
    dh_prof_set_tag(KH == 1 ? "1x1" : KH == 3 ? "3x3" : "7x7");
    dh_prof_set_dims(p.P, Cout, p.K);
    DhProfScope prof("dh_conv2d_bn_act", 2.0 * p.P * Cout * p.K,
                     4.0 * ((double)N * Cin * H * W + (double)Cout * p.K + (double)p.P * Cout * (residual ? 2 : 1)), stream);
    const bool big = Cout >= 128;
    if (KH == 1) { if (big) launch_conv<128, 1>(p, s); else launch_conv<64, 1>(p, s); }
    else if (KH == 3) { if (big) launch_conv<128, 3>(p, s); else launch_conv<64, 3>(p, s); }
    else { if (big) launch_conv<128, 7>(p, s); else launch_conv<64, 7>(p, s); }
    DH_LAUNCH_CHECK();
}

extern "C" int dh_stem_conv_nhwc(const float* x, const float* w, const float* scale, const float* shift, void* y,
                                 int N, int Cin, int H, int W, int Cout, int KS, int stride, int pad, int relu,
                                 int dtype, void* stream) {
    DH_REQUIRE(x && w && scale && shift && y && N > 0 && Cin > 0 && H > 0 && W > 0 && Cout > 0 && (Cout % 4) == 0);
    DH_REQUIRE((KS == 7 || KS == 3) && stride >= 1 && pad >= 0);
    ConvParams p{x, w, scale, shift, nullptr, (float*)y, N, Cin, H, W, Cout, (H + 2 * pad - KS) / stride + 1,
                 (W + 2 * pad - KS) / stride + 1, stride, pad, relu, Cin * KS * KS, 0};
    DH_REQUIRE(p.Ho > 0 && p.Wo > 0 && (long long)N * p.Ho * p.Wo < (1ll << 31));
    p.P = N * p.Ho * p.Wo;
    hipStream_t s = (hipStream_t)stream;
    DhProfScope prof("dh_stem_conv_nhwc", 2.0 * p.P * Cout * p.K, 4.0 * N * Cin * H * W + 2.0 * p.P * Cout, stream);
    DH_DISPATCH_16(dtype, {
        if (KS == 7) { if (Cout >= 128) launch_conv<128, 7, T>(p, s); else launch_conv<64, 7, T>(p, s); }
        else { if (Cout >= 128) launch_conv<128, 3, T>(p, s); else launch_conv<64, 3, T>(p, s); }
    });
    DH_LAUNCH_CHECK();
}

// ---- image packing for the matrix-core stem: NCHW fp32 [N,C,H,W] (C <= 8) -> NHWC bf16 [N,H,W,8], channels C..7 zero ----
template <typename T>
__global__ __launch_bounds__(256) void pack_nchw_to_nhwc8_kernel(const float* __restrict__ x, T* __restrict__ y,
                                                                   int C, int HW, size_t total) {
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < total; i += (size_t)gridDim.x * 256ull) {
        const size_t n = i / HW, p = i - n * HW;
        float v[8];
        // channels 0..3 at a clamped channel index without a branch (RGB: one memory round trip for the pixel instead
        // of one per conditional load), 4..7 only if the input really has them
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float t = x[(n * C + min(c, C - 1)) * HW + p];
            v[c] = c < C ? t : 0.f;
        }
#pragma unroll
        for (int c = 4; c < 8; ++c) v[c] = c < C ? x[(n * C + c) * HW + p] : 0.f;
        store16(y + i * 8, v);
    }
}

extern "C" int dh_pack_nchw_to_nhwc8(const float* x, void* y, int N, int C, int H, int W, int dtype, void* stream) {
    DH_REQUIRE(x && y && N > 0 && C > 0 && C <= 8 && H > 0 && W > 0);
    DhProfScope prof("dh_pack_nchw_to_nhwc8", 0.0, (double)N * H * W * (4.0 * C + 16.0), stream);
    const size_t total = (size_t)N * H * W;
    const int grid = (int)((total + 255) / 256 < 32768 ? (total + 255) / 256 : 32768);
    DH_DISPATCH_16(dtype, hipLaunchKernelGGL(pack_nchw_to_nhwc8_kernel<T>, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, (T*)y, C, H * W, total));
    DH_LAUNCH_CHECK();
}

// ---- image preprocessing on device (the notebook's ToTensor + Normalize, ipynb:565-567) ----
// u8 [N,H,W,C] (the decoded image as PIL / numpy hold it) -> fp32 NCHW (x / 255 - mean[c]) / std[c]: the same
// IEEE operations in the same order as torchvision's ToTensor().div(255) and Normalize's sub_().div_() -> bit-exact.
__global__ __launch_bounds__(256) void normalize_u8_hwc_kernel(const uint8_t* __restrict__ x, const float* __restrict__ mean,
                                                                const float* __restrict__ stdv, float* __restrict__ y,
                                                                int C, int HW, size_t total) {
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < total; i += (size_t)gridDim.x * 256ull) {
        const size_t n = i / HW, p = i - n * HW;
        for (int c = 0; c < C; ++c)
            y[(n * C + c) * HW + p] = ((float)x[i * C + c] / 255.0f - mean[c]) / stdv[c];
    }
}

extern "C" int dh_normalize_u8_hwc(const uint8_t* x, const float* mean, const float* stdv, float* y, int N, int H, int W,
                                   int C, void* stream) {
    DH_REQUIRE(x && mean && stdv && y && N > 0 && C > 0 && C <= 8 && H > 0 && W > 0);
    DhProfScope prof("dh_normalize_u8_hwc", 0.0, (double)N * H * W * C * 5.0, stream);
    const size_t total = (size_t)N * H * W;
    const int grid = (int)((total + 255) / 256 < 32768 ? (total + 255) / 256 : 32768);
    hipLaunchKernelGGL(normalize_u8_hwc_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, mean, stdv, y, C, H * W, total);
    DH_LAUNCH_CHECK();
}

// ---- channels-last bf16 pools (bf16 path) ------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void maxpool3x3s2_nhwc_kernel(const T* __restrict__ x, T* __restrict__ y,
                                                                 int N, int H, int W, int C, int Ho, int Wo) {
    const int c8 = C / 8;
    const size_t total = (size_t)N * Ho * Wo * c8;
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < total; i += (size_t)gridDim.x * 256ull) {
        const int cc = (int)(i % c8);
        size_t r = i / c8;
        const int ow = (int)(r % Wo); r /= Wo;
        const int oh = (int)(r % Ho);
        const int n = (int)(r / Ho);
        float m[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) m[u] = -INFINITY;
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
            const int ih = oh * 2 - 1 + kh;
            if ((unsigned)ih >= (unsigned)H) continue;
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const int iw = ow * 2 - 1 + kw;
                if ((unsigned)iw >= (unsigned)W) continue;
                float v[8];
                load16(x + (((size_t)n * H + ih) * W + iw) * C + cc * 8, v);
#pragma unroll
                for (int u = 0; u < 8; ++u) m[u] = fmaxf(m[u], v[u]);
            }
        }
        store16(y + (((size_t)n * Ho + oh) * Wo + ow) * C + cc * 8, m);
    }
}

extern "C" int dh_maxpool3x3s2_nhwc(const void* x, void* y, int N, int H, int W, int C, int dtype, void* stream) {
    if (!DH_IS_16BIT(dtype)) return DH_ERR_UNSUPPORTED;
    DH_REQUIRE(x && y && N > 0 && C > 0 && H > 0 && W > 0 && (C % 8) == 0);
    DhProfScope prof("dh_maxpool3x3s2_nhwc", 0.0, 0.0, stream);
    const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
    const size_t total = (size_t)N * Ho * Wo * (C / 8);
    const int grid = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
    DH_DISPATCH_16(dtype, hipLaunchKernelGGL(maxpool3x3s2_nhwc_kernel<T>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const T*)x,
                                             (T*)y, N, H, W, C, Ho, Wo));
    DH_LAUNCH_CHECK();
}

// x [N, HW, C] -> y [N, C]: mean over the HW positions, fp32 accumulation
template <typename T>
__global__ __launch_bounds__(256) void avgpool_nhwc_kernel(const T* __restrict__ x, T* __restrict__ y,
                                                            int N, int HW, int C) {
    const int c8 = C / 8;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N * c8) return;
    const int n = i / c8, cc = i - n * c8;
    float s[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) s[u] = 0.f;
    // eight positions per round, all requested before the first is used (a one-load-at-a-time loop is a chain of 49 memory
    // round trips: 29 us per 256 images for 51 MB); same summation order as the plain loop
    const T* src = x + (size_t)n * HW * C + cc * 8;
    for (int j0 = 0; j0 < HW; j0 += 8) {
        float v[8][8];
#pragma unroll
        for (int r = 0; r < 8; ++r) load16(src + (size_t)min(j0 + r, HW - 1) * C, v[r]);
#pragma unroll
        for (int r = 0; r < 8; ++r)
            if (j0 + r < HW) {
#pragma unroll
                for (int u = 0; u < 8; ++u) s[u] += v[r][u];
            }
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) s[u] /= (float)HW;
    store16(y + (size_t)n * C + cc * 8, s);
}

extern "C" int dh_avgpool_nhwc(const void* x, void* y, int N, int HW, int C, int dtype, void* stream) {
    if (!DH_IS_16BIT(dtype)) return DH_ERR_UNSUPPORTED;
    DH_REQUIRE(x && y && N > 0 && HW > 0 && C > 0 && (C % 8) == 0);
    DhProfScope prof("dh_avgpool_nhwc", 0.0, 0.0, stream);
    DH_DISPATCH_16(dtype, hipLaunchKernelGGL(avgpool_nhwc_kernel<T>, dim3(dh_cdiv((long long)N * (C / 8), 256)), dim3(256), 0,
                                             (hipStream_t)stream, (const T*)x, (T*)y, N, HW, C));
    DH_LAUNCH_CHECK();
}

// ---- MaxPool2d(3, stride 2, padding 1): padding cells never win (they are -inf in torch) -----------
__global__ __launch_bounds__(256) void maxpool3x3s2_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                            int NC, int H, int W, int Ho, int Wo) {
    const size_t total = (size_t)NC * Ho * Wo;
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < total; i += (size_t)gridDim.x * 256ull) {
        const int ow = (int)(i % Wo), oh = (int)((i / Wo) % Ho);
        const size_t nc = i / ((size_t)Wo * Ho);
        const float* src = x + nc * H * W;
        float m = -INFINITY;
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
            const int ih = oh * 2 - 1 + kh;
            if ((unsigned)ih >= (unsigned)H) continue;
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const int iw = ow * 2 - 1 + kw;
                if ((unsigned)iw < (unsigned)W) m = fmaxf(m, src[ih * W + iw]);
            }
        }
        y[i] = m;
    }
}

extern "C" int dh_maxpool3x3s2(const void* x, void* y, int N, int C, int H, int W, int dtype, void* stream) {
    if (dtype != DH_F32) return DH_ERR_UNSUPPORTED;
    DH_REQUIRE(x && y && N > 0 && C > 0 && H > 0 && W > 0);
    DhProfScope prof("dh_maxpool3x3s2", 0.0, 0.0, stream);
    const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
    const size_t total = (size_t)N * C * Ho * Wo;
    const int grid = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
    hipLaunchKernelGGL(maxpool3x3s2_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const float*)x,
                       (float*)y, N * C, H, W, Ho, Wo);
    DH_LAUNCH_CHECK();
}

// ---- AdaptiveAvgPool2d(1): one wave per (n, c) row of HW values --------------------------------------
__global__ __launch_bounds__(256) void avgpool_rows_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                            int rows, int HW) {
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (r >= rows) return;
    float s = 0.f;
    for (int i = lane; i < HW; i += 64) s += x[(size_t)r * HW + i];
    s = wave_sum(s);
    if (lane == 0) y[r] = s / (float)HW;
}

extern "C" int dh_avgpool_rows(const void* x, void* y, int rows, int HW, int dtype, void* stream) {
    if (dtype != DH_F32) return DH_ERR_UNSUPPORTED;
    DH_REQUIRE(x && y && rows > 0 && HW > 0);
    DhProfScope prof("dh_avgpool_rows", 0.0, 0.0, stream);
    hipLaunchKernelGGL(avgpool_rows_kernel, dim3(dh_cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream,
                       (const float*)x, (float*)y, rows, HW);
    DH_LAUNCH_CHECK();
}

// ---- [N, C, HW] -> [N, HW, C] through a padded 32x32 LDS tile (coalesced on both sides) -----------
__global__ __launch_bounds__(256) void nchw_to_rows_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                            int C, int HW) {
    __shared__ float tile[32][33];
    const int n = blockIdx.z, c0 = blockIdx.y * 32, s0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const float* xs = x + (size_t)n * C * HW;
    float* ys = y + (size_t)n * C * HW;
    for (int i = ty; i < 32; i += 8) {
        const int c = c0 + i, s = s0 + tx;
        tile[i][tx] = (c < C && s < HW) ? xs[(size_t)c * HW + s] : 0.f;
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
        const int s = s0 + i, c = c0 + tx;
        if (s < HW && c < C) ys[(size_t)s * C + c] = tile[tx][i];
    }
}

extern "C" int dh_nchw_to_rows(const void* x, void* y, int N, int C, int HW, int dtype, void* stream) {
    if (dtype != DH_F32) return DH_ERR_UNSUPPORTED;
    DH_REQUIRE(x && y && N > 0 && C > 0 && HW > 0 && N < 65536);
    DhProfScope prof("dh_nchw_to_rows", 0.0, 0.0, stream);
    hipLaunchKernelGGL(nchw_to_rows_kernel, dim3(dh_cdiv(HW, 32), dh_cdiv(C, 32), N), dim3(256), 0,
                       (hipStream_t)stream, (const float*)x, (float*)y, C, HW);
    DH_LAUNCH_CHECK();
}
