// fp32-class GEMMs / convolutions on the 16-bit matrix cores ("f32x", option f32_split): the arithmetic of the fp32 parity path
// at a multiple of its speed.  The reference is fp32 throughout (rnn_models.py / transformers.py / encoders.py hold no half or
// autocast); `dh_linear(DH_F32)` runs v_mfma_f32_32x32x2_f32 (157 TF peak) and `dh_conv2d_bn_act(DH_F32)` the vector ALUs.  Here
// every fp32 operand x is split into two fp16 numbers
//     x = hi + lo * 2^-11,   hi = fp16(x),   lo = fp16((x - hi) * 2^11)          (22 significand bits; both exact differences)
// and a product sum over k is three v_mfma_f32_16x16x32_f16 per tile-product with fp32 accumulation
//     sum a*w  ~=  [sum a_hi*w_hi]  +  2^-11 * [sum a_hi*w_lo + a_lo*w_hi]          (the dropped a_lo*w_lo term is 2^-22 relative)
// -- every fp16 x fp16 product is exact in fp32, so the result differs from an fp32 FMA chain by the same few ulps two fp32
// summation orders differ by (tests/test_f32x_gpu.py: against fp64).  Range contract: |x| < 65504 (checked for weights when a plan
// is built; activations of this model are O(10)).
//
// One tile kernel, C[M,N] = act((A W^T + bias) * scale + shift (+ residual)):
//   A   fp32, dense [M, lda] or the implicit im2col of a channels-last (NHWC) fp32 activation (row = output pixel, k = (kh, kw, ci));
//       loaded by 16-byte global loads into registers one slab ahead, split there, written to LDS as two fp16 planes;
//   W   two fp16 planes [2][N][Kp] (hi, then lo * 2^11; Kp = K rounded up to 32, zero padded) made ONCE per weight version by
//       dh_split_f32x; LDS-DMA straight into the LDS planes (swizzle on the source address);
//   128 x 128 tile, 32-k slabs, two LDS stages of 32 KB (two workgroups per CU), 4 waves of 64 x 64 = 4 x 4 MFMA tiles with TWO
//   accumulator sets (main, correction), 48 MFMAs per wave per slab, one barrier per slab;
//   weights are the MFMA "A" operand: an accumulator quad is 4 consecutive output columns of one row -> 16-byte fp32 stores.
#include "common.h"
#include "prof.h"
#include <cstdlib>

namespace {

constexpr float kLoScale = 2048.0f, kLoInv = 1.0f / 2048.0f;

struct F32xParams {
    const float* A; int lda;
    const uint16_t* Wp; int Kp; size_t plane;          // plane = elements from the hi plane to the lo plane
    const float* bias; const float* scale; const float* shift;
    const float* res; int ldres;
    float* C; int ldc;
    uint16_t* Cp; size_t c_plane;                      // optional: the result as fp16 planes [2][M][N] INSTEAD of C (its consumer is a GEMM operand)
    int M, N, K, relu;
    int H, Wd, Cin, Ho, Wo, KS, stride, pad;           // conv loader (NHWC input)
    int tiles_m, tiles_n, n_fast;                      // n_fast: consecutive workgroups walk the N tiles of one M tile (small weights)
    unsigned* range_flag;                              // sticky per-stream word: an activation outside the fp16 range was split (see below)
    int diag;
};

// Range guard of the ACTIVATIONS: the weights' range is checked when a plan is built, but an activation with |x| >= 65504
// splits into hi = inf and the GEMM would silently return inf / NaN.  Every thread keeps the largest |x| it splits (one v_max3 per
// pair) and a kernel that saw one out of range sets a sticky word -- one of 256, chosen by the launch stream -- that the Python layer
// reads where it synchronises anyway (dh_f32x_take_overflow) and answers by repeating the call on the exact-fp32 kernels.
// (Below 6.1e-5 an operand's hi part is an fp16 subnormal: its ABSOLUTE error stays <= 2^-36 after the lo part -- negligible next to
// rows of ordinary magnitude -- but a tensor that is tiny THROUGHOUT keeps only ~1e-5 relative accuracy: documented, not guarded.)
constexpr float kF16Max = 65504.0f;
__device__ unsigned g_f32x_range_flags[256];
__device__ __forceinline__ void track(float& amax, float a, float b) { amax = fmaxf(amax, fmaxf(fabsf(a), fabsf(b))); }
__device__ __forceinline__ void report(const F32xParams& p, float amax) {
    if (amax >= kF16Max) atomicOr(p.range_flag, 1u);
}

__device__ __forceinline__ void split2(float a, float b, uint32_t& hi, uint32_t& lo) {
    const f16_t ha = (f16_t)a, hb = (f16_t)b;
    const f16_t la = (f16_t)((a - (float)ha) * kLoScale), lb = (f16_t)((b - (float)hb) * kLoScale);
    hi = (uint32_t)__builtin_bit_cast(uint16_t, ha) | ((uint32_t)__builtin_bit_cast(uint16_t, hb) << 16);
    lo = (uint32_t)__builtin_bit_cast(uint16_t, la) | ((uint32_t)__builtin_bit_cast(uint16_t, lb) << 16);
}

// position of 16-byte chunk c (0..3) of row r inside the row's 64 bytes.  A 16-row fragment read is one ds_read_b128: lane = row +
// 16 chunk, served in the four lane groups {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} (+ 32) over a 256-byte bank row = four 64-byte
// rows -- a group holds the row quartets (chunk 0: rows 0-3, 12-15; chunk 1: rows 4-11), so the XOR key g(row >> 2) must make
// g(0), g(3), 1 ^ g(1), 1 ^ g(2) four different slots: g = 0, 3, 2, 1 (the key r >> 2 itself of rounds 4-5 left every read 2-way
// conflicted: twice the LDS cycles)
__device__ __forceinline__ int swz(int r, int c) { return c ^ ((0 - (r >> 2)) & 3); }

// acc[j][i][r] = C[m = mw + 16 i + l15][n = nw + 16 j + 4 lq + r]; main + 2^-11 * correction, then bias / scale + shift / residual / ReLU
template <int TM, int TN>
__device__ __forceinline__ void epilogue(const F32xParams& p, const dh_f32x4 (&acc)[TN][TM], const dh_f32x4 (&cor)[TN][TM], int mw, int nw, int l15,
                                         int lq) {
    const bool vec = p.Cp || ((p.ldc & 3) == 0 && (((uintptr_t)p.C) & 15) == 0 && (!p.res || ((p.ldres & 3) == 0 && (((uintptr_t)p.res) & 15) == 0)));
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
            float* dst = p.C + (size_t)m * p.ldc + n;
            if (vec && n + 3 < p.N) {
                if (p.res && !(p.diag & 2)) {
                    const float4 rr = *reinterpret_cast<const float4*>(p.res + (size_t)m * p.ldres + n);
                    v[0] += rr.x; v[1] += rr.y; v[2] += rr.z; v[3] += rr.w;
                }
                if (p.relu) { v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f); }
                if (p.Cp) {                              // (host: N % 4 == 0, no residual stride games: the planes are dense [M][N])
                    uint2 hi, lo;
                    split2(v[0], v[1], hi.x, lo.x); split2(v[2], v[3], hi.y, lo.y);
                    if (fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))) >= kF16Max) atomicOr(p.range_flag, 1u);
                    *reinterpret_cast<uint2*>(p.Cp + (size_t)m * p.N + n) = hi;
                    *reinterpret_cast<uint2*>(p.Cp + p.c_plane + (size_t)m * p.N + n) = lo;
                } else if (!(p.diag & 1) || v[0] == 12345.678f)
                *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (n + r < p.N) {
                        float o = v[r];
                        if (p.res) o += p.res[(size_t)m * p.ldres + n + r];
                        dst[r] = p.relu ? fmaxf(o, 0.f) : o;
                    }
            }
        }
    }
}

// MODE 0: dense A; 1: convolution with Cin % 32 == 0 (a 32-k slab lies inside one filter tap: the tap is wave-uniform);
// 2: convolution, any Cin % 4 == 0 (the stem: per-lane tap decode)
// Tiles: 128 x 128 (the large layers), 128 x 64 (Cout = 64: stage 1 and the stem), 64 x 64 (the decode-position linears, whose
// 1,280 x 512 outputs are 40 tiles of 128 x 128 on 256 CUs).  Four waves, 2 x 2 over the tile.
template <int MODE, int BM, int BN>
__global__ __launch_bounds__(256, BM * BN >= 128 * 128 ? 2 : 3) void gemm_f32x_kernel(F32xParams p) {
    constexpr int BK = 32;
    constexpr int PLANE_A = BM * BK * 2, PLANE_W = BN * BK * 2;     // one fp16 plane of one operand
    constexpr int STAGE = 2 * PLANE_A + 2 * PLANE_W;     // A hi | A lo | W hi | W lo
    constexpr int WM = BM / 2, WN = BN / 2, TM = WM / 16, TN = WN / 16, A_IT = BM / 32, W_PW = BN / 64;
    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * STAGE + 1024];     // + 1 KB: the sink of the residual prefetch below

    const int nblk = p.tiles_m * p.tiles_n;
    int bid = blockIdx.x;
    {   // XCD-aware bijective remap: consecutive tile ids (which share a W panel) stay on one XCD's L2
        const int q = nblk / 8, r = nblk % 8, xcd = bid % 8, idx = bid / 8;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    // tile order: consecutive workgroups share a weight panel and walk over M; with small weights (L2-resident anyway) they share
    // the activation tile instead, so that it is fetched from HBM once and not tiles_n times
    const int tm = p.n_fast ? bid / p.tiles_n : bid % p.tiles_m, tn = p.n_fast ? bid % p.tiles_n : bid / p.tiles_m;
    const int m0 = tm * BM, n0 = tn * BN;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm0 = (wave & 1) * WM, wn0 = (wave >> 1) * WN;
    const int l15 = lane & 15, lq = lane >> 4;

    // ---- A loader: thread = (row, 4-float chunk kc) x 4 rows; rows fixed over the reduction -----------------------------------
    const int a_kc = tid & 7;
    const float* a_ptr[A_IT];
    int a_ih0[A_IT], a_iw0[A_IT];
    bool a_ok[A_IT];
#pragma unroll
    for (int it = 0; it < A_IT; ++it) {
        const int row = (tid >> 3) + it * 32, m = m0 + row;
        a_ok[it] = m < p.M;
        const int mm = a_ok[it] ? m : 0;
        a_ih0[it] = a_iw0[it] = 0;
        if (MODE == 0) {
            a_ptr[it] = p.A + (size_t)mm * p.lda + a_kc * 4;
        } else {
            const int hw = p.Ho * p.Wo, n = mm / hw, r = mm - n * hw, oh = r / p.Wo, ow = r - oh * p.Wo;
            a_ih0[it] = oh * p.stride - p.pad; a_iw0[it] = ow * p.stride - p.pad;
            a_ptr[it] = p.A + (((size_t)n * p.H + a_ih0[it]) * p.Wd + a_iw0[it]) * p.Cin + (MODE == 1 ? a_kc * 4 : 0);
        }
    }
    int st_kh = 0, st_kw = 0, st_ci = 0;               // MODE 1: tap of the NEXT slab to load
    float4 areg[A_IT];
    auto load_a = [&](int k0) {
#pragma unroll
        for (int it = 0; it < A_IT; ++it) {
            const float* src = nullptr;
            if (MODE == 0) {
                if (a_ok[it] && k0 + a_kc * 4 < p.K) src = a_ptr[it] + k0;
            } else if (MODE == 1) {
                if (a_ok[it] && (unsigned)(a_ih0[it] + st_kh) < (unsigned)p.H && (unsigned)(a_iw0[it] + st_kw) < (unsigned)p.Wd)
                    src = a_ptr[it] + ((ptrdiff_t)st_kh * p.Wd + st_kw) * p.Cin + st_ci;
            } else {
                const int k = k0 + a_kc * 4;
                const int tap = k / p.Cin, ci = k - tap * p.Cin, kh = tap / p.KS, kw = tap - kh * p.KS;
                if (a_ok[it] && k < p.K && (unsigned)(a_ih0[it] + kh) < (unsigned)p.H && (unsigned)(a_iw0[it] + kw) < (unsigned)p.Wd)
                    src = a_ptr[it] + ((ptrdiff_t)kh * p.Wd + kw) * p.Cin + ci;
            }
            areg[it] = src ? *reinterpret_cast<const float4*>(src) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        if (MODE == 1) {
            st_ci += BK;
            if (st_ci == p.Cin) { st_ci = 0; if (++st_kw == p.KS) { st_kw = 0; ++st_kh; } }
        }
    };
    float amax = 0.f;                                  // largest |activation| this thread has split (range guard)
    auto store_a = [&](int buf) {
        unsigned char* ah = lds + buf * STAGE;
#pragma unroll
        for (int it = 0; it < A_IT; ++it) {
            const int row = (tid >> 3) + it * 32;
            uint2 hi, lo;
            track(amax, areg[it].x, areg[it].y); track(amax, areg[it].z, areg[it].w);
            split2(areg[it].x, areg[it].y, hi.x, lo.x);
            split2(areg[it].z, areg[it].w, hi.y, lo.y);
            const int off = row * 64 + swz(row, a_kc >> 1) * 16 + (a_kc & 1) * 8;
            *reinterpret_cast<uint2*>(ah + off) = hi;
            *reinterpret_cast<uint2*>(ah + PLANE_A + off) = lo;
        }
    };
    // ---- W loader: LDS-DMA, one wave instruction = 16 rows x 64 B of one plane; wave w moves pieces W_PW w ... of both planes ----
    const uint16_t* w_src[W_PW];
#pragma unroll
    for (int i = 0; i < W_PW; ++i) {
        const int row = (wave * W_PW + i) * 16 + (lane >> 2);
        const int n = min(n0 + row, p.N - 1);            // rows past N repeat the last one (their outputs are not stored)
        w_src[i] = p.Wp + (size_t)n * p.Kp + swz(row, lane & 3) * 8;
    }
    auto load_w = [&](int buf, int k0) {
        unsigned char* wh = lds + buf * STAGE + 2 * PLANE_A;
#pragma unroll
        for (int i = 0; i < W_PW; ++i) {
            dh_lds_dma16(w_src[i] + k0, wh + (wave * W_PW + i) * 1024);
            dh_lds_dma16(w_src[i] + p.plane + k0, wh + PLANE_W + (wave * W_PW + i) * 1024);
        }
    };

    dh_f32x4 acc[TN][TM], cor[TN][TM];                   // [tn][tm]: main (hi x hi) and correction (hi x lo + lo x hi, scaled 2^11)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int i = 0; i < TM; ++i) { acc[j][i] = dh_f32x4{0.f, 0.f, 0.f, 0.f}; cor[j][i] = dh_f32x4{0.f, 0.f, 0.f, 0.f}; }

    const int nslab = p.Kp / BK;
    if (p.res && !(p.diag & 4) && (p.ldres & 3) == 0 && (p.N & 3) == 0 && (((uintptr_t)p.res) & 15) == 0) {
        // The residual tile is read in the epilogue, after the reduction: measured (256 x 56 x 56 x 64 -> 256 with residual, 645 us) the
        // reduction, the residual loads and the stores run one after the other -- 226 + 300 + 210 us -- because a CU holds two
        // workgroups.  Touch the tile's cache lines NOW (one 16-byte LDS-DMA per 128-byte line into a sink nobody reads: no register,
        // no wait of its own) so that the epilogue's loads find them in L2 / the Infinity Cache instead of in HBM.
        constexpr int LPR = BN * 4 / 128;                // cache lines per residual row of the tile
        constexpr int LINES = BM * LPR;
#pragma unroll
        for (int q = tid; q < LINES; q += 256) {
            const int row = q / LPR, ln = q - row * LPR;
            const int m = min(m0 + row, p.M - 1), n = min(n0 + ln * 32, p.N - 4);
            dh_lds_dma16(p.res + (size_t)m * p.ldres + n, lds + 2 * STAGE);
        }
    }
    load_w(0, 0);
    load_a(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    store_a(0);
    __syncthreads();
    for (int t = 0; t < nslab; ++t) {
        const int buf = t & 1;
        if (t + 1 < nslab) { load_w(buf ^ 1, (t + 1) * BK); load_a((t + 1) * BK); }
        const unsigned char* ah = lds + buf * STAGE;
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
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                acc[j][i] = Op16<f16_t>::mfma(fw_h, fa_h[i], acc[j][i]);
                cor[j][i] = Op16<f16_t>::mfma(fw_h, fa_l[i], cor[j][i]);
                cor[j][i] = Op16<f16_t>::mfma(fw_l, fa_h[i], cor[j][i]);
            }
        }
        if (t + 1 < nslab) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // slab t + 1: A in registers, W planes in LDS
            store_a(buf ^ 1);
        }
        __syncthreads();
    }

    report(p, amax);
    epilogue<TM, TN>(p, acc, cor, m0 + wm0, n0 + wn0, l15, lq);
}

__device__ uint4 dh_f32x_zero_page[2];              // zero-initialised: source of chunks outside M / K

// The decode-position shapes (a few hundred to a few thousand rows x N = 512 ... 2,048): too few 128 x 128 tiles to fill 256 CUs and,
// with one slab of look-ahead, every 32-k slab costs a memory round trip (first version: 18 us for 1,280 x 512 x 512).  Here: 64 x 64
// tiles and an NS-deep LDS ring filled by LDS-DMA only -- the fp32 activation slab goes into LDS AS fp32 ([64 rows][32 k], chunk slots
// XOR-swizzled by row) and is split into its fp16 planes when a wave reads its fragment (2 x ds_read_b128 + 28 VALU operations per
// fragment; twice redundant across the two column halves, cheap next to a round trip per slab) -- so NS - 1 slabs are in flight with no
// staging registers, one counted wait + one barrier per slab.
template <int NS>
__global__ __launch_bounds__(256, 1) void gemm_f32x_small_kernel(F32xParams p) {
    constexpr int BM = 64, BN = 64, BK = 32;
    constexpr int A_BYTES = BM * BK * 4, PLANE_W = BN * BK * 2, STAGE = A_BYTES + 2 * PLANE_W;     // 8 + 4 + 4 KB
    constexpr int P = 4;                                // LDS-DMA instructions per wave per slab: 2 (A) + 1 + 1 (W planes)
    static_assert((NS - 1) * P <= 60, "vmcnt range");
    __shared__ __attribute__((aligned(16))) unsigned char lds[NS * STAGE];
    const int nblk = p.tiles_m * p.tiles_n;
    int bid = blockIdx.x;
    {
        const int q = nblk / 8, r = nblk % 8, xcd = bid % 8, idx = bid / 8;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int tm = p.n_fast ? bid / p.tiles_n : bid % p.tiles_m, tn = p.n_fast ? bid % p.tiles_n : bid / p.tiles_m;
    const int m0 = tm * BM, n0 = tn * BN;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm0 = (wave & 1) * 32, wn0 = (wave >> 1) * 32;
    const int l15 = lane & 15, lq = lane >> 4;
    const unsigned char* zero = reinterpret_cast<const unsigned char*>(dh_f32x_zero_page);

    // A: a piece = 8 rows x 128 B (fp32); wave w moves pieces 2w, 2w + 1; LDS slot (row, pos) holds source chunk pos ^ (row & 7)
    const float* a_src[2]; int a_chunk[2]; bool a_ok[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = (wave * 2 + i) * 8 + (lane >> 3), m = m0 + row;
        a_ok[i] = m < p.M;
        a_chunk[i] = (lane & 7) ^ (row & 7);
        a_src[i] = p.A + (size_t)(a_ok[i] ? m : 0) * p.lda + a_chunk[i] * 4;
    }
    // W planes: a piece = 16 rows x 64 B; wave w moves piece w of the hi and of the lo plane
    const int w_row = wave * 16 + (lane >> 2);
    const uint16_t* w_src = p.Wp + (size_t)min(n0 + w_row, p.N - 1) * p.Kp + swz(w_row, lane & 3) * 8;
    auto stage = [&](int slab) {
        unsigned char* st = lds + (slab % NS) * STAGE;
        const int k0 = slab * BK;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const void* src = (a_ok[i] && k0 + a_chunk[i] * 4 < p.K) ? (const void*)(a_src[i] + k0) : (const void*)zero;
            dh_lds_dma16(src, st + (wave * 2 + i) * 1024);
        }
        dh_lds_dma16(w_src + k0, st + A_BYTES + wave * 1024);
        dh_lds_dma16(w_src + p.plane + k0, st + A_BYTES + PLANE_W + wave * 1024);
    };
    dh_f32x4 acc[2][2], cor[2][2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int i = 0; i < 2; ++i) { acc[j][i] = dh_f32x4{0.f, 0.f, 0.f, 0.f}; cor[j][i] = dh_f32x4{0.f, 0.f, 0.f, 0.f}; }
    const int nslab = p.Kp / BK;
#pragma unroll
    for (int u = 0; u < NS - 1; ++u)
        if (u < nslab) stage(u);
    float amax = 0.f;                                  // largest |activation| this thread has split (range guard)
    for (int t = 0; t < nslab; ++t) {
        // slab t has landed once at most min(NS - 2, slabs issued after t) newer slabs of this wave are outstanding
        const int newer = min(NS - 2, nslab - 1 - t);
        if (newer == NS - 2) asm volatile("s_waitcnt vmcnt(%0)" :: "n"((NS - 2) * P) : "memory");
        else switch (newer) {
            case 6: asm volatile("s_waitcnt vmcnt(24)" ::: "memory"); break;
            case 5: asm volatile("s_waitcnt vmcnt(20)" ::: "memory"); break;
            case 4: asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); break;
            case 3: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
            case 2: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
            case 1: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
            default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        }
        __syncthreads();                                 // slab t visible to every wave; every wave is done with slab t - 1
        if (t + NS - 1 < nslab) stage(t + NS - 1);       // refills the stage slab t - 1 lived in
        const unsigned char* st = lds + (t % NS) * STAGE;
        uint4 fa_h[2], fa_l[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int r = wm0 + 16 * i + l15;
            const float4 x0 = *reinterpret_cast<const float4*>(st + r * 128 + (((2 * lq) ^ (r & 7)) << 4));
            const float4 x1 = *reinterpret_cast<const float4*>(st + r * 128 + (((2 * lq + 1) ^ (r & 7)) << 4));
            track(amax, x0.x, x0.y); track(amax, x0.z, x0.w); track(amax, x1.x, x1.y); track(amax, x1.z, x1.w);
            split2(x0.x, x0.y, fa_h[i].x, fa_l[i].x); split2(x0.z, x0.w, fa_h[i].y, fa_l[i].y);
            split2(x1.x, x1.y, fa_h[i].z, fa_l[i].z); split2(x1.z, x1.w, fa_h[i].w, fa_l[i].w);
        }
        const unsigned char* wh = st + A_BYTES;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int r = wn0 + 16 * j + l15, off = r * 64 + swz(r, lq) * 16;
            const uint4 fw_h = *reinterpret_cast<const uint4*>(wh + off);
            const uint4 fw_l = *reinterpret_cast<const uint4*>(wh + PLANE_W + off);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                acc[j][i] = Op16<f16_t>::mfma(fw_h, fa_h[i], acc[j][i]);
                cor[j][i] = Op16<f16_t>::mfma(fw_h, fa_l[i], cor[j][i]);
                cor[j][i] = Op16<f16_t>::mfma(fw_l, fa_h[i], cor[j][i]);
            }
        }
    }
    report(p, amax);
    epilogue<2, 2>(p, acc, cor, m0 + wm0, n0 + wn0, l15, lq);
}

__global__ __launch_bounds__(256) void split_f32x_kernel(const float* __restrict__ w, int ldw, uint16_t* __restrict__ planes, int N, int K,
                                                          int Kp) {
    const size_t total = (size_t)N * (Kp / 2), plane = (size_t)N * Kp;
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < total; i += (size_t)gridDim.x * 256ull) {
        const int n = (int)(i / (Kp / 2)), k = (int)(i - (size_t)n * (Kp / 2)) * 2;
        const float a = k < K ? w[(size_t)n * ldw + k] : 0.f, b = k + 1 < K ? w[(size_t)n * ldw + k + 1] : 0.f;
        uint32_t hi, lo;
        split2(a, b, hi, lo);
        *reinterpret_cast<uint32_t*>(planes + (size_t)n * Kp + k) = hi;
        *reinterpret_cast<uint32_t*>(planes + plane + (size_t)n * Kp + k) = lo;
    }
}

// [N, C, H, W] fp32 -> channels-last [N, H, W, Cp] fp32, channels >= C zero (the f32x stem reads 16-byte pixels)
__global__ __launch_bounds__(256) void nchw_to_nhwc_f32_kernel(const float* __restrict__ x, float* __restrict__ y, int N, int C, int H, int W,
                                                                int Cp) {
    const size_t total = (size_t)N * H * W;
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < total; i += (size_t)gridDim.x * 256ull) {
        const size_t hw = (size_t)H * W, n = i / hw, r = i - n * hw;
        for (int c = 0; c < Cp; ++c) y[i * Cp + c] = c < C ? x[(n * C + c) * hw + r] : 0.f;
    }
}

template <int VEC_DUMMY = 0>
__global__ __launch_bounds__(256) void maxpool3x3s2_nhwc_f32_kernel(const float* __restrict__ x, float* __restrict__ y, int N, int H, int W, int C,
                                                                     int Ho, int Wo) {
    const int c4 = C / 4;
    const size_t total = (size_t)N * Ho * Wo * c4;
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < total; i += (size_t)gridDim.x * 256ull) {
        const int cc = (int)(i % c4);
        size_t r = i / c4;
        const int ow = (int)(r % Wo); r /= Wo;
        const int oh = (int)(r % Ho);
        const int n = (int)(r / Ho);
        float4 m = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
            const int ih = oh * 2 - 1 + kh;
            if ((unsigned)ih >= (unsigned)H) continue;
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const int iw = ow * 2 - 1 + kw;
                if ((unsigned)iw >= (unsigned)W) continue;
                const float4 v = *reinterpret_cast<const float4*>(x + (((size_t)n * H + ih) * W + iw) * C + cc * 4);
                m.x = fmaxf(m.x, v.x); m.y = fmaxf(m.y, v.y); m.z = fmaxf(m.z, v.z); m.w = fmaxf(m.w, v.w);
            }
        }
        *reinterpret_cast<float4*>(y + (((size_t)n * Ho + oh) * Wo + ow) * C + cc * 4) = m;
    }
}

// x [N, HW, C] fp32 -> y [N, C]: mean over the positions in index order (the order dh_avgpool_rows sums an NCHW row in)
__global__ __launch_bounds__(256) void avgpool_nhwc_f32_kernel(const float* __restrict__ x, float* __restrict__ y, int N, int HW, int C) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N * C) return;
    const int n = i / C, c = i - n * C;
    const float* src = x + (size_t)n * HW * C + c;
    float s = 0.f;
    for (int j = 0; j < HW; ++j) s += src[(size_t)j * C];
    y[i] = s / (float)HW;
}

template <int BM, int BN>
void launch_tile(F32xParams& p, int mode, hipStream_t s) {
    p.tiles_m = dh_cdiv(p.M, BM); p.tiles_n = dh_cdiv(p.N, BN);
    const dim3 grid((unsigned)(p.tiles_m * p.tiles_n)), block(256);
    if (mode == 0) hipLaunchKernelGGL((gemm_f32x_kernel<0, BM, BN>), grid, block, 0, s, p);
    else if (mode == 1) hipLaunchKernelGGL((gemm_f32x_kernel<1, BM, BN>), grid, block, 0, s, p);
    else hipLaunchKernelGGL((gemm_f32x_kernel<2, BM, BN>), grid, block, 0, s, p);
}

// the sticky range word of a stream: one of 256 by a hash of the stream handle (two streams of one process sharing a word could at
// worst make one of them repeat a call on the exact path without need)
unsigned* range_flag_of(hipStream_t s) {
    static unsigned* base = nullptr;
    if (!base && hipGetSymbolAddress((void**)&base, HIP_SYMBOL(g_f32x_range_flags)) != hipSuccess) return nullptr;
    const uintptr_t h = (uintptr_t)s;
    return base + (((h >> 4) ^ (h >> 12) ^ (h >> 20)) & 255u);
}

__global__ void take_flag_kernel(unsigned* flag, uint32_t* dst) { *dst = atomicExch(flag, 0u); }

int launch(const F32xParams& p0, int mode, hipStream_t s) {
    F32xParams p = p0;
    p.range_flag = range_flag_of(s);
    if (!p.range_flag) return DH_ERR_LAUNCH;
    static const int diag = getenv("DH_F32X_DIAG") ? atoi(getenv("DH_F32X_DIAG")) : 0;      // (tools/f32x_conv1x1_bench.py: phase timing, wrong results)
    p.diag = diag;
    // weight planes of <= 4 MB stay in every XCD's L2: walk the N tiles of one M tile back to back (the activation tile comes
    // from HBM once instead of tiles_n times)
    p.n_fast = (double)p.N * p.Kp * 4.0 <= 4.0 * 1048576.0;
    const long long t128 = (long long)dh_cdiv(p.M, 128) * dh_cdiv(p.N, 128);
    if (p.N <= 64) launch_tile<128, 64>(p, mode, s);
    else if (t128 < 256 && mode == 0 && (p.K % 4) == 0 && (long long)dh_cdiv(p.M, 64) * dh_cdiv(p.N, 64) <= 256) {
        // fewer 64 x 64 tiles than CUs (a decode position's N = 512 layers: 160 tiles): one workgroup per CU at best, so the deep ring
        // (measured: proj 16.4 against 18.2 us per launch); with several workgroups per CU the 2-stage kernel is the faster one
        // (qkv 480 tiles: 21 against 28 us, the LSTM gate product 640 tiles: 34 against 59 us)
        p.tiles_m = dh_cdiv(p.M, 64); p.tiles_n = dh_cdiv(p.N, 64);
        hipLaunchKernelGGL(gemm_f32x_small_kernel<8>, dim3((unsigned)(p.tiles_m * p.tiles_n)), dim3(256), 0, s, p);
    }
    else if (t128 < 256) launch_tile<64, 64>(p, mode, s);          // (a decode position: 1,280 x 512 = 40 tiles of 128 x 128)
    else launch_tile<128, 128>(p, mode, s);
    return hipGetLastError() == hipSuccess ? DH_OK : DH_ERR_LAUNCH;
}

}  // namespace

// *dst (device memory) = 1 when a dh_linear_f32x / dh_conv2d_nhwc_f32x launch on `stream` since the last call split an ACTIVATION
// outside the fp16 range (|x| >= 65504: its results hold inf / NaN), else 0; resets the stream's word.
unsigned* dh_f32x_range_flag_of(hipStream_t s) { return range_flag_of(s); }      // (linear_f32x_wreg.hip)

extern "C" int dh_f32x_take_overflow(uint32_t* dst, void* stream) {
    DH_REQUIRE(dst);
    unsigned* flag = range_flag_of((hipStream_t)stream);
    if (!flag) return DH_ERR_LAUNCH;
    hipLaunchKernelGGL(take_flag_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, flag, dst);
    DH_LAUNCH_CHECK();
}

extern "C" int dh_split_f32x(const float* w, int ldw, void* planes, int N, int K, int Kp, void* stream) {
    DH_REQUIRE(w && planes && N > 0 && K > 0 && ldw >= K && Kp >= K && (Kp % 32) == 0 && ((uintptr_t)planes % 16) == 0);
    const size_t total = (size_t)N * (Kp / 2);
    const int grid = (int)((total + 255) / 256 < 65536 ? (total + 255) / 256 : 65536);
    hipLaunchKernelGGL(split_f32x_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, w, ldw, (uint16_t*)planes, N, K, Kp);
    DH_LAUNCH_CHECK();
}

extern "C" int dh_linear_f32x(const float* A, int lda, const void* w_planes, int Kp, const float* bias, const float* scale, const float* shift,
                              const float* residual, int ldres, float* C, int ldc, int M, int N, int K, int relu, void* stream) {
    DH_REQUIRE(A && w_planes && C && M > 0 && N > 0 && K > 0 && lda >= K && ldc >= N && Kp >= K && (Kp % 32) == 0);
    DH_REQUIRE((lda % 4) == 0 && ((uintptr_t)A % 16) == 0 && ((uintptr_t)w_planes % 16) == 0 && (!residual || ldres >= N));
    DH_REQUIRE((K % 4) == 0 && (!scale) == (!shift));
    F32xParams p{};
    p.A = A; p.lda = lda; p.Wp = (const uint16_t*)w_planes; p.Kp = Kp; p.plane = (size_t)N * Kp;
    p.bias = bias; p.scale = scale; p.shift = shift; p.res = residual; p.ldres = ldres; p.C = C; p.ldc = ldc;
    p.M = M; p.N = N; p.K = K; p.relu = relu;
    dh_prof_set_dims(M, N, K);
    DhProfScope prof("dh_linear_f32x", 2.0 * M * N * K, 4.0 * ((double)M * K + (double)N * K + (double)M * N), stream);
    return launch(p, 0, (hipStream_t)stream);
}

extern "C" int dh_conv2d_nhwc_f32x(const float* x, const void* w_planes, int Kp, const float* scale, const float* shift, const float* residual,
                                   float* y, int N, int H, int W, int Cin, int Cout, int KS, int stride, int pad, int relu, void* stream) {
    DH_REQUIRE(x && w_planes && y && scale && shift && N > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0 && KS > 0 && stride > 0 && pad >= 0);
    const int K = KS * KS * Cin;
    DH_REQUIRE((Cin % 4) == 0 && Kp >= K && (Kp % 32) == 0 && ((uintptr_t)x % 16) == 0 && ((uintptr_t)w_planes % 16) == 0 && ((uintptr_t)y % 16) == 0);
    const int Ho = (H + 2 * pad - KS) / stride + 1, Wo = (W + 2 * pad - KS) / stride + 1;
    DH_REQUIRE(Ho > 0 && Wo > 0 && (long long)N * Ho * Wo < (1ll << 31));
    F32xParams p{};
    p.A = x; p.Wp = (const uint16_t*)w_planes; p.Kp = Kp; p.plane = (size_t)Cout * Kp;
    p.scale = scale; p.shift = shift; p.res = residual; p.ldres = Cout; p.C = y; p.ldc = Cout;
    p.M = N * Ho * Wo; p.N = Cout; p.K = K; p.relu = relu;
    p.H = H; p.Wd = W; p.Cin = Cin; p.Ho = Ho; p.Wo = Wo; p.KS = KS; p.stride = stride; p.pad = pad;
    dh_prof_set_tag(KS == 1 ? "1x1" : KS == 3 ? "3x3" : "7x7");
    dh_prof_set_dims(p.M, Cout, K);
    DhProfScope prof("dh_conv2d_nhwc_f32x", 2.0 * p.M * Cout * K,
                     4.0 * ((double)N * H * W * Cin + (double)Cout * K + (double)p.M * Cout * (residual ? 2 : 1)), stream);
    return launch(p, (Cin % 32) == 0 && Kp == K ? 1 : 2, (hipStream_t)stream);
}

// dh_conv2d_nhwc_f32x with the result stored as fp16 planes [2][N, Ho, Wo, Cout] (hi, lo * 2^11) instead of fp32: Bottleneck.conv1 of
// stages 2 - 4, whose only consumer is conv2 on dh_conv2d_nhwc_f32xp.  Cout % 4 == 0, no residual.
extern "C" int dh_conv2d_nhwc_f32x_planes_out(const float* x, const void* w_planes, int Kp, const float* scale, const float* shift,
                                              void* y_planes, int N, int H, int W, int Cin, int Cout, int KS, int stride, int pad, int relu,
                                              void* stream) {
    DH_REQUIRE(x && w_planes && y_planes && scale && shift && N > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0 && KS > 0 && stride > 0 && pad >= 0);
    const int K = KS * KS * Cin;
    DH_REQUIRE((Cin % 4) == 0 && (Cout % 4) == 0 && Kp >= K && (Kp % 32) == 0 && ((uintptr_t)x % 16) == 0 && ((uintptr_t)w_planes % 16) == 0 &&
               ((uintptr_t)y_planes % 16) == 0);
    const int Ho = (H + 2 * pad - KS) / stride + 1, Wo = (W + 2 * pad - KS) / stride + 1;
    DH_REQUIRE(Ho > 0 && Wo > 0 && (long long)N * Ho * Wo < (1ll << 31));
    F32xParams p{};
    p.A = x; p.Wp = (const uint16_t*)w_planes; p.Kp = Kp; p.plane = (size_t)Cout * Kp;
    p.scale = scale; p.shift = shift; p.ldc = Cout;
    p.M = N * Ho * Wo; p.N = Cout; p.K = K; p.relu = relu;
    p.Cp = (uint16_t*)y_planes; p.c_plane = (size_t)p.M * Cout;
    p.H = H; p.Wd = W; p.Cin = Cin; p.Ho = Ho; p.Wo = Wo; p.KS = KS; p.stride = stride; p.pad = pad;
    dh_prof_set_tag(KS == 1 ? "1x1" : KS == 3 ? "3x3" : "7x7");
    dh_prof_set_dims(p.M, Cout, K);
    DhProfScope prof("dh_conv2d_nhwc_f32x", 2.0 * p.M * Cout * K, 4.0 * ((double)N * H * W * Cin + (double)Cout * K + (double)p.M * Cout), stream);
    return launch(p, (Cin % 32) == 0 && Kp == K ? 1 : 2, (hipStream_t)stream);
}

extern "C" int dh_nchw_to_nhwc_f32(const float* x, float* y, int N, int C, int H, int W, int Cp, void* stream) {
    DH_REQUIRE(x && y && N > 0 && C > 0 && H > 0 && W > 0 && Cp >= C && Cp <= 8);
    const size_t total = (size_t)N * H * W;
    const int grid = (int)((total + 255) / 256 < 65536 ? (total + 255) / 256 : 65536);
    hipLaunchKernelGGL(nchw_to_nhwc_f32_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, y, N, C, H, W, Cp);
    DH_LAUNCH_CHECK();
}

extern "C" int dh_maxpool3x3s2_nhwc_f32(const float* x, float* y, int N, int H, int W, int C, void* stream) {
    DH_REQUIRE(x && y && N > 0 && H > 0 && W > 0 && C > 0 && (C % 4) == 0 && ((uintptr_t)x % 16) == 0 && ((uintptr_t)y % 16) == 0);
    const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
    const size_t total = (size_t)N * Ho * Wo * (C / 4);
    const int grid = (int)((total + 255) / 256 < 65536 ? (total + 255) / 256 : 65536);
    DhProfScope prof("dh_maxpool3x3s2_nhwc_f32", 0.0, 0.0, stream);
    hipLaunchKernelGGL(maxpool3x3s2_nhwc_f32_kernel<0>, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, y, N, H, W, C, Ho, Wo);
    DH_LAUNCH_CHECK();
}

extern "C" int dh_avgpool_nhwc_f32(const float* x, float* y, int N, int HW, int C, void* stream) {
    DH_REQUIRE(x && y && N > 0 && HW > 0 && C > 0);
    DhProfScope prof("dh_avgpool_nhwc_f32", 0.0, 0.0, stream);
    hipLaunchKernelGGL(avgpool_nhwc_f32_kernel, dim3(dh_cdiv((long long)N * C, 256)), dim3(256), 0, (hipStream_t)stream, x, y, N, HW, C);
    DH_LAUNCH_CHECK();
}
