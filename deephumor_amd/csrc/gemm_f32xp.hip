// The split-operand classifier of gemm_f32x.hip for an activation that is STORED split ("planes"): round 6.
//
// gemm_f32x_kernel reads fp32 activations, splits them in registers slab by slab -- once per column tile -- and keeps one slab of
// look-ahead.  A tensor whose only consumers are GEMM operands can just as well be stored as the two fp16 planes its consumer would
// make of it -- hi = fp16(x), lo = fp16((x - hi) * 2^11): 4 bytes per element like fp32, and the SAME numbers the consumer's split
// produces, so every product and every sum below is the one gemm_f32x_kernel computes (bit-identical results:
// tests/test_f32x_gpu.py) -- and then both operands are LDS-DMA material:
//   A   planes [2][M][Kp] fp16, written by the producing kernel (dh_add_layernorm_f32x: the final LayerNorm of a decode position) or
//       by dh_split_act_f32x (the LSTM's top-layer state);
//   W   planes [2][N][Kp] of dh_split_f32x, as before;
//   256 x 128 tile, 8 waves of 64 x 64 (4 x 4 MFMA tiles x {main, correction} = 128 accumulator registers), 32-k slabs, a ring of three
//   48 KB stages filled by global_load_lds_dwordx4 only (6 per wave per slab), counted waits, and the waves in TWO GROUPS A PHASE APART
//   (waves w and w + 4 share a SIMD): while one group multiplies a slab out of registers -- 48 MFMAs at raised priority -- the other
//   reads its 16 fragments of the next slab and requests the slab after, so the matrix pipe and the LDS are busy at the same time.
//   Epilogue: (main + 2^-11 correction + bias) * scale + shift (+ fp32 residual) (ReLU) -> fp32 and / or planes, and the per-row maxima
//   of each 64-column group (what dh_vocab_logits hands the beam sampler on the 16-bit paths: the fp32 path's sampler then reads 3 MB
//   of maxima + the candidate groups instead of 187 MB of logits, 20 instead of 61 us per position).
// Measured (tools/f32xp_kbench.py, profiles/r6/f32xp_kbench_*.txt; 1,280 x 36,541 x 512): gemm_f32x_kernel 237 - 257 us, this kernel
// 187 us with all waves in one phase, 176 us with the two groups = 818 TF of MFMA work (0.33 of the dense fp16 peak; the 16-bit
// classifier's HBM-bound 61 us are out of reach at three MFMAs per product).  The same kernel with an implicit-GEMM loader (CONV) is the
// trunk's 3 x 3 convolution of stages 2 - 4: conv1 writes its output as planes (dh_conv2d_nhwc_f32x_planes_out), this kernel reads them
// -- 13 - 27 % faster than the tile kernel on those layers (l3 conv2 174 against 231 us, l4 conv2 160 against 222 us).  NOT used where
// it loses: layers that write wide outputs (conv3 + residual: 1.5 x slower with one 8-wave workgroup per CU and nothing to overlap
// its epilogue with -- those layers are csrc/conv1x1_f32x.hip's) and stage 1 (Cout = 64).  Also measured: 128 x 64 wave tiles at
// one wave per SIMD (fewer LDS bytes per MFMA): 20 % slower; the bank-conflict-free fragment swizzle: within noise (kept).
#include "common.h"
#include "prof.h"

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
    int H, Wd, Cin, Ho, Wo, KS, stride, pad;           // convolution loader (CONV = 1)
    int tiles_m, tiles_n, n_fast;
    unsigned* range_flag;
};

__device__ uint4 g_xp_zero[1];                         // source of filter taps outside the image

__device__ __forceinline__ int swz(int r, int c) { return c ^ ((0 - (r >> 2)) & 3); }

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
template <int CONV, int WMW, int WNW, int TM, int NS, bool PP>
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
    if constexpr (!PP) {
        for (int t = 0; t < nslab; ++t) {
            // slab t has landed when at most the (NS - 2) younger slabs' loads are outstanding
            if (t + NS - 2 < nslab) wait_vm<(NS - 2) * NDMA>(); else wait_vm<0>();
            __syncthreads();                             // ... for every wave; and every wave has finished reading slab t - 1
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
    } else {
        // Two wave groups a phase apart (waves w and w + NW / 2 share a SIMD): while one group multiplies slab t out of registers
        // (48 MFMAs, raised priority) the other reads its 16 fragments of the next slab and requests the slab after -- the matrix
        // pipe and the LDS are busy at the same time instead of in turns.  Barrier count p: group 0 loads slab p / 2 in even phases,
        // group 1 in the following odd one.  A slab's DMA is waited for (counted) at the end of the issuing wave's previous load phase
        // and read at the earliest two barriers later; its stage is requested again only after both groups' reads were retired
        // (lgkmcnt(0)) in front of a barrier.
        static_assert(NS == 3 && NW % 2 == 0, "phase arithmetic below");
        const int grp = wave / (NW / 2);
        if (nslab > 1) wait_vm<NDMA>(); else wait_vm<0>();
        __syncthreads();                                 // slab 0 is in LDS for everyone
        if (grp == 1) __builtin_amdgcn_s_barrier();      // stagger
        for (int t = 0; t < nslab; ++t) {
            const unsigned char* ah = lds + (t % NS) * STAGE;
            const unsigned char* wh = ah + 2 * PLANE_A;
            uint4 fa_h[TM], fa_l[TM], fw_h[TN], fw_l[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int r = wm0 + 16 * i + l15, off = r * 64 + swz(r, lq) * 16;
                fa_h[i] = *reinterpret_cast<const uint4*>(ah + off);
                fa_l[i] = *reinterpret_cast<const uint4*>(ah + PLANE_A + off);
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int r = wn0 + 16 * j + l15, off = r * 64 + swz(r, lq) * 16;
                fw_h[j] = *reinterpret_cast<const uint4*>(wh + off);
                fw_l[j] = *reinterpret_cast<const uint4*>(wh + PLANE_W + off);
            }
            if (t + 2 < nslab) { issue((t + 2) % NS, (t + 2) * 32); wait_vm<NDMA>(); }      // slab t + 1 complete (my pieces)
            else wait_vm<0>();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int j = 0; j < TN; ++j) {
#pragma unroll
                for (int i = 0; i < TM; ++i) acc[j][i] = Op16<f16_t>::mfma(fw_h[j], fa_h[i], acc[j][i]);
#pragma unroll
                for (int i = 0; i < TM; ++i) cor[j][i] = Op16<f16_t>::mfma(fw_h[j], fa_l[i], cor[j][i]);
#pragma unroll
                for (int i = 0; i < TM; ++i) cor[j][i] = Op16<f16_t>::mfma(fw_l[j], fa_h[i], cor[j][i]);
            }
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
        }
        if (grp == 0) __builtin_amdgcn_s_barrier();
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

int launch(XpParams& p, int conv, hipStream_t s) {
    p.range_flag = dh_f32x_range_flag_of(s);
    if (!p.range_flag) return DH_ERR_LAUNCH;
    p.n_fast = (double)p.N * p.Kp * 4.0 <= 4.0 * 1048576.0;
    p.tiles_m = dh_cdiv(p.M, 256);
    if (p.N <= 64 && !p.gmax && !conv) {     // (group maxima: whole 128-column tiles, so that every group of the caller's table is written)
        p.tiles_n = 1;
        hipLaunchKernelGGL((gemm_f32xp_kernel<0, 4, 1, 4, 3, false>), dim3((unsigned)p.tiles_m), dim3(256), 0, s, p);
    } else {
        p.tiles_n = dh_cdiv(p.N, 128);
        const dim3 grid((unsigned)(p.tiles_m * p.tiles_n));
        if (conv) hipLaunchKernelGGL((gemm_f32xp_kernel<1, 4, 2, 4, 3, true>), grid, dim3(512), 0, s, p);
        else hipLaunchKernelGGL((gemm_f32xp_kernel<0, 4, 2, 4, 3, true>), grid, dim3(512), 0, s, p);
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
// (Cin % 32 == 0, Cout >= 128) -> fp32 [N, Ho, Wo, Cout] and / or planes; the arithmetic of dh_conv2d_nhwc_f32x (bit-identical).  The
// trunk's 3 x 3 layers of stages 2 - 4 (Bottleneck.conv2, encoders.py:56): their input comes from conv1 as planes
// (dh_conv2d_nhwc_f32x_planes_out), 13 - 27 % faster than the fp32-activation tile kernel (profiles/r6/f32xp_kbench_two_phase.txt).
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
    DhProfScope prof("dh_conv2d_nhwc_f32x", 2.0 * p.M * Cout * p.Kp,
                     4.0 * ((double)N * H * W * Cin + (double)Cout * p.Kp + (double)p.M * Cout * ((residual ? 1 : 0) + (y ? 1 : 0) + (y_planes ? 1 : 0))),
                     stream);
    return launch(p, 1, (hipStream_t)stream);
}
