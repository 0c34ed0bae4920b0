// The split-operand GEMM / convolution of gemm_f32x.hip for activations that are STORED split ("planes"): round 6.
//
// gemm_f32x_kernel reads fp32 activations, splits them in registers slab by slab -- once per column tile and, in a 3 x 3 convolution,
// once per filter tap -- and keeps one slab of look-ahead: every 32-k slab (48 MFMAs per wave = 0.3 us) waits for a memory round
// trip.  Measured (profiles/r6/c2_f32_split1_kernel_stats.csv): the classifier at 207 us = 0.28 of its MFMA bound, the encoder's
// convolutions 14.5 of the 25.6 ms of a C2 step.  A tensor whose only consumers are GEMM operands can just as well be stored as the
// two fp16 planes its consumer would make of it -- hi = fp16(x), lo = fp16((x - hi) * 2^11): 4 bytes per element like fp32, and
// the SAME numbers the consumer's split produces, so every product and every sum below is the one gemm_f32x_kernel computes
// (bit-identical results: tests/test_f32x_gpu.py) -- and then both operands are LDS-DMA material:
//   A   planes [2][rows][lda] fp16 (dense rows, or the pixels of a channels-last activation with lda = Cin, Cin % 32 == 0), written
//       by the producing kernel's epilogue (below: out_planes), by dh_maxpool3x3s2_nhwc_f32's planes form or by dh_split_act_f32x;
//   W   planes [2][N][Kp] of dh_split_f32x, as before;
//   256 x 128 tile (8 waves, 64 x 64 each: 4 x 4 MFMA tiles x {main, correction} = 128 accumulator registers), 32-k slabs, an NS = 3
//   ring of 48 KB stages filled by global_load_lds_dwordx4 only (6 per wave per slab; no staging registers, no VALU work in the
//   loop but the convolution's tap addresses), two slabs in flight behind the one being multiplied, one counted wait + one barrier
//   per slab; 256 x 64 (4 waves) for Cout = 64.
//   Epilogue: (main + 2^-11 correction + bias) * scale + shift (+ fp32 residual) (ReLU) -> fp32 and / or planes, and for the
//   classifier the per-row maxima of each 64-column group (what dh_vocab_logits hands the beam sampler on the 16-bit paths).
#include "common.h"
#include "prof.h"
#include <cstdlib>

unsigned* dh_f32x_range_flag_of(hipStream_t s);      // gemm_f32x.hip: the stream's sticky activation-range word

namespace {

constexpr float kLoScale = 2048.0f, kLoInv = 1.0f / 2048.0f, kF16Max = 65504.0f;

struct XpParams {
    const uint16_t* Ap; size_t a_plane; int lda;       // activation planes: hi at Ap, lo at Ap + a_plane; row stride lda elements
    const uint16_t* Wp; size_t w_plane; int Kp;
    const float* bias; const float* scale; const float* shift;
    const float* res; int ldres;
    float* C; int ldc;                                 // fp32 output (optional)
    uint16_t* Cp; size_t c_plane; int ldcp;            // planes output (optional)
    float* gmax; int gmax_ld;                          // per-row maxima of 64-column groups (optional)
    int M, N, relu;
    int H, Wd, Cin, Ho, Wo, KS, stride, pad;           // convolution loader
    int tiles_m, tiles_n, n_fast;
    unsigned* range_flag;
};

__device__ uint4 g_xp_zero[1];                         // source of taps outside the image

__device__ __forceinline__ int swz(int r, int c) { return c ^ ((r >> 2) & 3); }

__device__ __forceinline__ void split4(const float (&v)[4], uint2& hi, uint2& lo) {
    uint16_t h[4], l[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const f16_t hh = (f16_t)v[r];
        const f16_t ll = (f16_t)((v[r] - (float)hh) * kLoScale);
        h[r] = __builtin_bit_cast(uint16_t, hh); l[r] = __builtin_bit_cast(uint16_t, ll);
    }
    hi = make_uint2((uint32_t)h[0] | ((uint32_t)h[1] << 16), (uint32_t)h[2] | ((uint32_t)h[3] << 16));
    lo = make_uint2((uint32_t)l[0] | ((uint32_t)l[1] << 16), (uint32_t)l[2] | ((uint32_t)l[3] << 16));
}

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }

// CONV 0: dense A planes; 1: channels-last convolution, Cin % 32 == 0 (a slab lies inside one filter tap: the tap is wave-uniform)
template <int CONV, int WMW, int WNW, int TM, int NS>
__global__ __launch_bounds__(64 * WMW * WNW, 1) void gemm_f32xp_kernel(XpParams p) {
    constexpr int NW = WMW * WNW, WTM = 16 * TM, BM = WTM * WMW, BN = 64 * WNW;
    constexpr int PLANE_A = BM * 64, PLANE_W = BN * 64, STAGE = 2 * PLANE_A + 2 * PLANE_W;
    constexpr int RPA = BM / 16 / NW, RPW = BN / 16 / NW;                // 16-row pieces of A / of W per wave (one plane)
    static_assert(RPA * NW * 16 == BM && RPW * NW * 16 == BN, "every wave moves the same number of pieces (the counted wait)");
    constexpr int NDMA = 2 * RPA + 2 * RPW;
    constexpr int TN = 4;
    __shared__ __attribute__((aligned(16))) unsigned char lds[NS * STAGE];

    const int nblk = p.tiles_m * p.tiles_n;
    int bid = blockIdx.x;
    {   // XCD-aware bijective remap: consecutive tile ids stay on one XCD's L2
        const int q = nblk / 8, r = nblk % 8, xcd = bid % 8, idx = bid / 8;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int tm = p.n_fast ? bid / p.tiles_n : bid % p.tiles_m, tn = p.n_fast ? bid % p.tiles_n : bid / p.tiles_m;
    const int m0 = tm * BM, n0 = tn * BN;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm0 = (wave % WMW) * WTM, wn0 = (wave / WMW) * 64;
    const int l15 = lane & 15, lq = lane >> 4;

    // ---- loaders: one wave instruction = 16 rows x 64 B of one plane -----------------------------------------------------------------
    const uint16_t* a_src[RPA];
    int a_ih0[RPA], a_iw0[RPA];
#pragma unroll
    for (int i = 0; i < RPA; ++i) {
        const int row = (wave * RPA + i) * 16 + (lane >> 2);
        const int m = min(m0 + row, p.M - 1);           // rows past M repeat the last one (their outputs are not stored)
        const int ch = swz(row, lane & 3) * 8;
        if (CONV == 0) {
            a_src[i] = p.Ap + (size_t)m * p.lda + ch;
            a_ih0[i] = a_iw0[i] = 0;
        } else {
            const int hw = p.Ho * p.Wo, n = m / hw, r = m - n * hw, oh = r / p.Wo, ow = r - oh * p.Wo;
            a_ih0[i] = oh * p.stride - p.pad; a_iw0[i] = ow * p.stride - p.pad;
            a_src[i] = p.Ap + ((ptrdiff_t)((size_t)n * p.H + a_ih0[i]) * p.Wd + a_iw0[i]) * p.Cin + ch;
        }
    }
    const uint16_t* w_src[RPW];
#pragma unroll
    for (int i = 0; i < RPW; ++i) {
        const int row = (wave * RPW + i) * 16 + (lane >> 2);
        const int n = min(n0 + row, p.N - 1);
        w_src[i] = p.Wp + (size_t)n * p.Kp + swz(row, lane & 3) * 8;
    }
    int st_kh = 0, st_kw = 0, st_ci = 0;                // convolution: tap / channel offset of the NEXT slab to issue
    auto issue = [&](int buf, int k0) {
        unsigned char* st = lds + buf * STAGE;
#pragma unroll
        for (int i = 0; i < RPA; ++i) {
            const uint16_t* src;
            bool ok = true;
            if (CONV == 0) src = a_src[i] + k0;
            else {
                ok = (unsigned)(a_ih0[i] + st_kh) < (unsigned)p.H && (unsigned)(a_iw0[i] + st_kw) < (unsigned)p.Wd;
                src = a_src[i] + ((ptrdiff_t)st_kh * p.Wd + st_kw) * p.Cin + st_ci;
            }
            const uint16_t* zero = reinterpret_cast<const uint16_t*>(g_xp_zero);
            dh_lds_dma16(ok ? src : zero, st + (wave * RPA + i) * 1024);
            dh_lds_dma16(ok ? src + p.a_plane : zero, st + PLANE_A + (wave * RPA + i) * 1024);
        }
#pragma unroll
        for (int i = 0; i < RPW; ++i) {
            dh_lds_dma16(w_src[i] + k0, st + 2 * PLANE_A + (wave * RPW + i) * 1024);
            dh_lds_dma16(w_src[i] + p.w_plane + k0, st + 2 * PLANE_A + PLANE_W + (wave * RPW + i) * 1024);
        }
        if (CONV == 1) {
            st_ci += 32;
            if (st_ci == p.Cin) { st_ci = 0; if (++st_kw == p.KS) { st_kw = 0; ++st_kh; } }
        }
    };

    dh_f32x4 acc[TN][TM], cor[TN][TM];
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int i = 0; i < TM; ++i) { acc[j][i] = dh_f32x4{0.f, 0.f, 0.f, 0.f}; cor[j][i] = dh_f32x4{0.f, 0.f, 0.f, 0.f}; }

    const int nslab = p.Kp / 32;
#pragma unroll
    for (int s = 0; s < NS - 1; ++s)
        if (s < nslab) issue(s, s * 32);
    for (int t = 0; t < nslab; ++t) {
        // slab t has landed when at most the (NS - 2) younger slabs' loads are outstanding
        if (t + NS - 2 < nslab) wait_vm<(NS - 2) * NDMA>(); else wait_vm<0>();
        __syncthreads();                                 // ... for every wave; and every wave has finished reading slab t - 1
        if (t + NS - 1 < nslab) issue((t + NS - 1) % NS, (t + NS - 1) * 32);
        const unsigned char* ah = lds + (t % NS) * STAGE;
        const unsigned char* wh = ah + 2 * PLANE_A;
        uint4 fa_h[TM], fa_l[TM];
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int r = wm0 + 16 * i + l15, off = r * 64 + swz(r, lq) * 16;
            fa_h[i] = *reinterpret_cast<const uint4*>(ah + off);
            fa_l[i] = *reinterpret_cast<const uint4*>(ah + PLANE_A + off);
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int r = wn0 + 16 * j + l15, off = r * 64 + swz(r, lq) * 16;
            const uint4 fw_h = *reinterpret_cast<const uint4*>(wh + off);
            const uint4 fw_l = *reinterpret_cast<const uint4*>(wh + PLANE_W + off);
            // (an accumulator's two correction products are TM MFMAs apart: no back-to-back dependent issue)
#pragma unroll
            for (int i = 0; i < TM; ++i) acc[j][i] = Op16<f16_t>::mfma(fw_h, fa_h[i], acc[j][i]);
#pragma unroll
            for (int i = 0; i < TM; ++i) cor[j][i] = Op16<f16_t>::mfma(fw_h, fa_l[i], cor[j][i]);
#pragma unroll
            for (int i = 0; i < TM; ++i) cor[j][i] = Op16<f16_t>::mfma(fw_l, fa_h[i], cor[j][i]);
        }
    }

    // ---- epilogue: acc[j][i][r] = C[m = mw + 16 i + l15][n = nw + 16 j + 4 lq + r] ----------------------------------------------------
    const int mw = m0 + wm0, nw = n0 + wn0;
    const bool vec = !p.C || ((p.ldc & 3) == 0 && (!p.res || (p.ldres & 3) == 0));
    float amax = 0.f;
    float gm[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) gm[i] = -INFINITY;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = nw + 16 * j + 4 * lq;
        float bi[4], mu[4], ad[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int nn = min(n + r, p.N - 1);
            bi[r] = p.bias ? p.bias[nn] : 0.f;
            mu[r] = p.scale ? p.scale[nn] : 1.f;
            ad[r] = p.shift ? p.shift[nn] : 0.f;
        }
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int m = mw + 16 * i + l15;
            if (m >= p.M || n >= p.N) continue;
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = fmaf(fmaf(cor[j][i][r], kLoInv, acc[j][i][r]) + bi[r], mu[r], ad[r]);
            if (vec && n + 3 < p.N) {
                if (p.res) {
                    const float4 rr = *reinterpret_cast<const float4*>(p.res + (size_t)m * p.ldres + n);
                    v[0] += rr.x; v[1] += rr.y; v[2] += rr.z; v[3] += rr.w;
                }
                if (p.relu) { v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f); }
                if (p.C) *reinterpret_cast<float4*>(p.C + (size_t)m * p.ldc + n) = make_float4(v[0], v[1], v[2], v[3]);
                if (p.Cp) {                              // (host: N % 4 == 0, ldcp % 4 == 0)
                    uint2 hi, lo;
                    split4(v, hi, lo);
                    amax = fmaxf(amax, fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))));
                    *reinterpret_cast<uint2*>(p.Cp + (size_t)m * p.ldcp + n) = hi;
                    *reinterpret_cast<uint2*>(p.Cp + p.c_plane + (size_t)m * p.ldcp + n) = lo;
                }
                gm[i] = fmaxf(gm[i], fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3])));
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (n + r < p.N) {
                        float o = v[r];
                        if (p.res) o += p.res[(size_t)m * p.ldres + n + r];
                        o = p.relu ? fmaxf(o, 0.f) : o;
                        if (p.C) p.C[(size_t)m * p.ldc + n + r] = o;
                        gm[i] = fmaxf(gm[i], o);
                    }
            }
        }
    }
    if (p.gmax) {
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            float mx = gm[i];
            mx = fmaxf(mx, __shfl_xor(mx, 16));
            mx = fmaxf(mx, __shfl_xor(mx, 32));
            const int m = mw + 16 * i + l15, g = nw / 64;
            if (lq == 0 && m < p.M && g < p.gmax_ld) p.gmax[(size_t)m * p.gmax_ld + g] = mx;       // -inf for a group past N
        }
    }
    if (p.Cp && amax >= kF16Max) atomicOr(p.range_flag, 1u);
}

// fp32 rows -> planes, with the range guard of an activation split (dh_split_f32x is its unguarded twin for weights)
__global__ __launch_bounds__(256) void split_act_kernel(const float* __restrict__ a, int lda, uint16_t* __restrict__ planes, size_t plane, int ldp,
                                                        int M, int K, int Kp, unsigned* range_flag) {
    const size_t total = (size_t)M * (Kp / 4);
    float amax = 0.f;
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < total; i += (size_t)gridDim.x * 256ull) {
        const int m = (int)(i / (Kp / 4)), k = (int)(i - (size_t)m * (Kp / 4)) * 4;
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        if (k + 3 < K) {
            const float4 q = *reinterpret_cast<const float4*>(a + (size_t)m * lda + k);
            v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (k + r < K) v[r] = a[(size_t)m * lda + k + r];
        }
        amax = fmaxf(amax, fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))));
        uint2 hi, lo;
        split4(v, hi, lo);
        *reinterpret_cast<uint2*>(planes + (size_t)m * ldp + k) = hi;
        *reinterpret_cast<uint2*>(planes + plane + (size_t)m * ldp + k) = lo;
    }
    if (amax >= kF16Max) atomicOr(range_flag, 1u);
}

// 3 x 3 / stride 2 / pad 1 max-pooling of a channels-last fp32 tensor straight into planes (torchvision's maxpool, encoders.py:37: its
// only consumers are layer1.0's conv1 and downsample convolutions)
__global__ __launch_bounds__(256) void maxpool_planes_kernel(const float* __restrict__ x, uint16_t* __restrict__ planes, size_t plane, int N, int H,
                                                             int W, int C, int Ho, int Wo, unsigned* range_flag) {
    const int c4 = C / 4;
    const size_t total = (size_t)N * Ho * Wo * c4;
    float amax = 0.f;
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < total; i += (size_t)gridDim.x * 256ull) {
        const int c = (int)(i % c4) * 4;
        size_t r = i / c4;
        const int ow = (int)(r % Wo); r /= Wo;
        const int oh = (int)(r % Ho);
        const int n = (int)(r / Ho);
        float v[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const int ih = oh * 2 - 1 + kh, iw = ow * 2 - 1 + kw;
                if ((unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W) {
                    const float4 q = *reinterpret_cast<const float4*>(x + (((size_t)n * H + ih) * W + iw) * C + c);
                    v[0] = fmaxf(v[0], q.x); v[1] = fmaxf(v[1], q.y); v[2] = fmaxf(v[2], q.z); v[3] = fmaxf(v[3], q.w);
                }
            }
        amax = fmaxf(amax, fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))));
        uint2 hi, lo;
        split4(v, hi, lo);
        const size_t o = (((size_t)n * Ho + oh) * Wo + ow) * C + c;
        *reinterpret_cast<uint2*>(planes + o) = hi;
        *reinterpret_cast<uint2*>(planes + plane + o) = lo;
    }
    if (amax >= kF16Max) atomicOr(range_flag, 1u);
}

int launch(XpParams& p, int conv, hipStream_t s) {
    p.range_flag = dh_f32x_range_flag_of(s);
    if (!p.range_flag) return DH_ERR_LAUNCH;
    p.n_fast = (double)p.N * p.Kp * 4.0 <= 4.0 * 1048576.0;
    static const int variant = getenv("DH_XP_VARIANT") ? atoi(getenv("DH_XP_VARIANT")) : 0;
    if (p.N <= 64) {
        p.tiles_m = dh_cdiv(p.M, 256); p.tiles_n = 1;
        const dim3 grid((unsigned)p.tiles_m), block(256);
        if (conv) hipLaunchKernelGGL((gemm_f32xp_kernel<1, 4, 1, 4, 3>), grid, block, 0, s, p);
        else hipLaunchKernelGGL((gemm_f32xp_kernel<0, 4, 1, 4, 3>), grid, block, 0, s, p);
    } else if (variant == 1) {
        p.tiles_m = dh_cdiv(p.M, 256); p.tiles_n = dh_cdiv(p.N, 128);
        const dim3 grid((unsigned)(p.tiles_m * p.tiles_n)), block(256);
        if (conv) hipLaunchKernelGGL((gemm_f32xp_kernel<1, 2, 2, 8, 3>), grid, block, 0, s, p);
        else hipLaunchKernelGGL((gemm_f32xp_kernel<0, 2, 2, 8, 3>), grid, block, 0, s, p);
    } else {
        p.tiles_m = dh_cdiv(p.M, 256); p.tiles_n = dh_cdiv(p.N, 128);
        const dim3 grid((unsigned)(p.tiles_m * p.tiles_n)), block(512);
        if (conv) hipLaunchKernelGGL((gemm_f32xp_kernel<1, 4, 2, 4, 3>), grid, block, 0, s, p);
        else hipLaunchKernelGGL((gemm_f32xp_kernel<0, 4, 2, 4, 3>), grid, block, 0, s, p);
    }
    return hipGetLastError() == hipSuccess ? DH_OK : DH_ERR_LAUNCH;
}

}  // namespace

extern "C" int dh_split_act_f32x(const float* A, int lda, void* planes, int M, int K, int Kp, void* stream) {
    DH_REQUIRE(A && planes && M > 0 && K > 0 && lda >= K && Kp >= K && (Kp % 32) == 0 && ((uintptr_t)planes % 16) == 0);
    DH_REQUIRE((lda % 4) == 0 && ((uintptr_t)A % 16) == 0);
    unsigned* flag = dh_f32x_range_flag_of((hipStream_t)stream);
    if (!flag) return DH_ERR_LAUNCH;
    const size_t total = (size_t)M * (Kp / 4);
    const int grid = (int)((total + 255) / 256 < 65536 ? (total + 255) / 256 : 65536);
    DhProfScope prof("dh_split_act_f32x", 0.0, 8.0 * M * K, stream);
    hipLaunchKernelGGL(split_act_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, A, lda, (uint16_t*)planes, (size_t)M * Kp, Kp, M, K, Kp, flag);
    DH_LAUNCH_CHECK();
}

// C[M, N] fp32 and / or planes [2][M][N] = act((A W^T + bias) * scale + shift (+ residual)), A = planes [2][M][Kp] (zero padded to Kp);
// group_max (optional): [M, gm_ld] maxima of the 64-column groups of the stored values.  The arithmetic of dh_linear_f32x (bit-identical).
extern "C" int dh_linear_f32xp(const void* a_planes, const void* w_planes, int Kp, const float* bias, const float* scale, const float* shift,
                               const float* residual, int ldres, float* C, int ldc, void* c_planes, float* group_max, int gm_ld, int M, int N,
                               int relu, void* stream) {
    DH_REQUIRE(a_planes && w_planes && (C || c_planes) && M > 0 && N > 0 && Kp > 0 && (Kp % 32) == 0 && (!C || ldc >= N));
    DH_REQUIRE(((uintptr_t)a_planes % 16) == 0 && ((uintptr_t)w_planes % 16) == 0 && (!residual || ldres >= N) && (!scale) == (!shift));
    DH_REQUIRE((!C || ((uintptr_t)C % 16) == 0) && (!c_planes || ((N % 4) == 0 && ((uintptr_t)c_planes % 16) == 0)));
    DH_REQUIRE(!residual || ((uintptr_t)residual % 16) == 0);
    DH_REQUIRE(!group_max || gm_ld >= (N + 63) / 64);
    XpParams p{};
    p.Ap = (const uint16_t*)a_planes; p.a_plane = (size_t)M * Kp; p.lda = Kp;
    p.Wp = (const uint16_t*)w_planes; p.w_plane = (size_t)N * Kp; p.Kp = Kp;
    p.bias = bias; p.scale = scale; p.shift = shift; p.res = residual; p.ldres = ldres; p.C = C; p.ldc = ldc;
    p.Cp = (uint16_t*)c_planes; p.c_plane = (size_t)M * N; p.ldcp = N; p.gmax = group_max; p.gmax_ld = gm_ld;
    p.M = M; p.N = N; p.relu = relu;
    dh_prof_set_dims(M, N, Kp);
    DhProfScope prof("dh_linear_f32xp", 2.0 * M * N * Kp, 4.0 * ((double)M * Kp + (double)N * Kp + (double)M * N), stream);
    return launch(p, 0, (hipStream_t)stream);
}

// Channels-last convolution + BatchNorm affine (+ fp32 residual) (+ ReLU) of an activation stored as planes [2][N, H, W, Cin]
// (Cin % 32 == 0) -> fp32 [N, Ho, Wo, Cout] and / or planes; the arithmetic of dh_conv2d_nhwc_f32x (bit-identical).
extern "C" int dh_conv2d_nhwc_f32xp(const void* x_planes, const void* w_planes, const float* scale, const float* shift, const float* residual,
                                    float* y, void* y_planes, int N, int H, int W, int Cin, int Cout, int KS, int stride, int pad, int relu,
                                    void* stream) {
    DH_REQUIRE(x_planes && w_planes && (y || y_planes) && scale && shift && N > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0 && KS > 0 &&
               stride > 0 && pad >= 0);
    DH_REQUIRE((Cin % 32) == 0 && (Cout % 4) == 0 && ((uintptr_t)x_planes % 16) == 0 && ((uintptr_t)w_planes % 16) == 0);
    DH_REQUIRE((!y || ((uintptr_t)y % 16) == 0) && (!y_planes || ((uintptr_t)y_planes % 16) == 0) && (!residual || ((uintptr_t)residual % 16) == 0));
    const int Ho = (H + 2 * pad - KS) / stride + 1, Wo = (W + 2 * pad - KS) / stride + 1;
    DH_REQUIRE(Ho > 0 && Wo > 0 && (long long)N * Ho * Wo < (1ll << 31));
    XpParams p{};
    p.Ap = (const uint16_t*)x_planes; p.a_plane = (size_t)N * H * W * Cin; p.lda = Cin;
    p.Wp = (const uint16_t*)w_planes; p.Kp = KS * KS * Cin; p.w_plane = (size_t)Cout * p.Kp;
    p.scale = scale; p.shift = shift; p.res = residual; p.ldres = Cout; p.C = y; p.ldc = Cout;
    p.M = N * Ho * Wo; p.N = Cout; p.relu = relu;
    p.Cp = (uint16_t*)y_planes; p.c_plane = (size_t)p.M * Cout; p.ldcp = Cout;
    p.H = H; p.Wd = W; p.Cin = Cin; p.Ho = Ho; p.Wo = Wo; p.KS = KS; p.stride = stride; p.pad = pad;
    dh_prof_set_tag(KS == 1 ? "1x1" : KS == 3 ? "3x3" : "7x7");
    dh_prof_set_dims(p.M, Cout, p.Kp);
    DhProfScope prof("dh_conv2d_nhwc_f32xp", 2.0 * p.M * Cout * p.Kp,
                     4.0 * ((double)N * H * W * Cin + (double)Cout * p.Kp + (double)p.M * Cout * ((residual ? 1 : 0) + (y ? 1 : 0) + (y_planes ? 1 : 0))),
                     stream);
    return launch(p, 1, (hipStream_t)stream);
}

// dh_maxpool3x3s2_nhwc_f32 with the result stored as planes [2][N, Ho, Wo, C]
extern "C" int dh_maxpool3x3s2_nhwc_f32xp(const float* x, void* y_planes, int N, int H, int W, int C, void* stream) {
    DH_REQUIRE(x && y_planes && N > 0 && H > 0 && W > 0 && C > 0 && (C % 4) == 0 && ((uintptr_t)x % 16) == 0 && ((uintptr_t)y_planes % 16) == 0);
    const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
    unsigned* flag = dh_f32x_range_flag_of((hipStream_t)stream);
    if (!flag) return DH_ERR_LAUNCH;
    const size_t total = (size_t)N * Ho * Wo * (C / 4);
    const int grid = (int)((total + 255) / 256 < 65536 ? (total + 255) / 256 : 65536);
    DhProfScope prof("dh_maxpool3x3s2_nhwc_f32xp", 0.0, 0.0, stream);
    hipLaunchKernelGGL(maxpool_planes_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, (uint16_t*)y_planes, (size_t)N * Ho * Wo * C, N, H, W,
                       C, Ho, Wo, flag);
    DH_LAUNCH_CHECK();
}
