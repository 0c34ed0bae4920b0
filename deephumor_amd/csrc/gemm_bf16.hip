// bf16 matrix-core engine: C[M,N] = act((A[M,K] * W[N,K]^T + bias) * scale + shift (+ residual)),
// bf16 operands, fp32 accumulation in v_mfma_f32_16x16x32_bf16, bf16 or fp32 output.
//
// One kernel, two A-operand loaders:
//   dense  : A is a row-major [M, lda] matrix (every nn.Linear / LSTM gate / vocabulary product);
//   conv   : A is the implicit im2col of a channels-last (NHWC) activation: row m = output pixel
//            (n, oh, ow), column k = (kh, kw, ci) with ci fastest, so every 16-byte chunk of 8
//            consecutive k is 8 consecutive channels of ONE input pixel (Cin % 8 == 0).  The weight
//            operand is [Cout][kh][kw][ci] (repacked once when the model is planned).  Output rows are
//            pixels, i.e. the result is NHWC again.
//
// Structure (MI355X_MICROARCH / cdna_hip_programming guide, "minimum 2-phase" + glds):
//   * both operand slabs ([rows][64 k] bf16 = 128-B rows) go HBM/L2 -> LDS with global_load_lds_dwordx4
//     (no staging VGPRs); a wave instruction fills 8 consecutive rows (1 KiB, lane-linear), and the
//     bank-conflict swizzle chunk' = chunk ^ (row & 7) is applied on the per-lane SOURCE address;
//     rows beyond M/N, chunks beyond K and the convolution halo read a 16-byte zero page instead;
//   * two LDS slabs: the loads of slab t+1 are in flight while slab t feeds the MFMAs, one barrier per slab;
//   * the weight rows are the MFMA "A" operand and the activation rows the "B" operand, so an
//     accumulator register quad is 4 consecutive output columns n of one row m;
//   * the bf16 epilogue stages the fp32 tile through LDS (XOR-swizzled 16-B slots) and finishes with
//     16-byte row-contiguous residual loads / stores -- one rounding, at the very end; fp32 output
//     (logits, odd leading dimension) is stored straight from registers.
#include "common.h"
#include "prof.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ uint4 dh_zero_page[4];        // zero-initialised: source of padding chunks

struct GemmBf16Params {
    const uint16_t* A; int lda;
    const uint16_t* W; int ldw;
    const float* bias; const float* scale; const float* shift;
    const uint16_t* res; int ldres;
    void* C; int ldc;
    int M, N, K, relu, out_f32;
    int conv, H, Wd, Cin, Ho, Wo, KS, stride, pad;     // conv loader: A = NHWC input
    int tiles_m, tiles_n;
    float* gmax; int gmax_ld;                          // optional: per-row maxima of each wave-wide column group
};

typedef const void __attribute__((address_space(1)))* gptr_t;
typedef void __attribute__((address_space(3)))* lptr_t;

// wait until at most `n` of this wave's vector-memory operations (LDS-DMA included) are outstanding
__device__ __forceinline__ void wait_vmcnt(int n) {
    switch (n) {
        case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
        case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
        case 12: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
        case 16: asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); break;
        case 20: asm volatile("s_waitcnt vmcnt(20)" ::: "memory"); break;
        case 24: asm volatile("s_waitcnt vmcnt(24)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
}

// NS = LDS ring depth: slabs t+1 .. t+NS-1 are in flight (LDS-DMA) while slab t feeds the MFMAs.
// NW waves per workgroup, arranged WAVES_M x (NW / WAVES_M) over the BM x BN block.
template <int BM, int BN, int WAVES_M, bool CONV, int NS, int NW = 4>
__global__ __launch_bounds__(64 * NW) void gemm_bf16_kernel(GemmBf16Params p) {
    constexpr int NT = 64 * NW;
    constexpr int BK = 64;
    constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, SLAB = A_BYTES + B_BYTES;
    constexpr int WAVES_N = NW / WAVES_M;
    constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;  // per-wave sub-tile
    constexpr int TM = WM / 16, TN = WN / 16;            // 16x16 MFMA tiles per wave
    constexpr int IA = BM / (8 * NW), IB = BN / (8 * NW);  // glds instructions per wave per slab (8 rows each)
    constexpr int G = IA + IB;                           // LDS-DMA instructions per wave per slab
    constexpr int LDS_BYTES = NS * SLAB;                 // >= the fp32 epilogue tile (BM*BN*4 B = 2 slabs)
    static_assert(NS >= 2 && (NS - 2) * G <= 24, "ring depth vs vmcnt encoding");
    __shared__ __attribute__((aligned(16))) unsigned char lds[LDS_BYTES];

    const int nblk = p.tiles_m * p.tiles_n;
    int bid = blockIdx.x;
    {   // XCD-aware bijective remap: consecutive tile ids (sharing a W panel) stay on one XCD's L2
        const int q = nblk / 8, r = nblk % 8, xcd = bid % 8, idx = bid / 8;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int tm = bid % p.tiles_m, tn = bid / p.tiles_m;
    const int m0 = tm * BM, n0 = tn * BN;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm0 = (wave % WAVES_M) * WM, wn0 = (wave / WAVES_M) * WN;
    const int lr = lane >> 3, lpos = lane & 7;           // loader: row within the 8-row group, LDS chunk slot
    const unsigned char* zero = reinterpret_cast<const unsigned char*>(dh_zero_page);

    // ---- per-lane source rows (fixed over the reduction) ----------------------------------------
    const uint16_t* a_base[IA];
    int a_ih0[IA], a_iw0[IA], a_swz[IA];
    bool a_ok[IA];
#pragma unroll
    for (int i = 0; i < IA; ++i) {
        const int row = (wave * IA + i) * 8 + lr, m = m0 + row;
        a_ok[i] = m < p.M;
        a_swz[i] = lpos ^ (row & 7);                     // global chunk landing in this lane's LDS slot
        a_ih0[i] = a_iw0[i] = 0;
        if (CONV) {
            const int mm = a_ok[i] ? m : 0;
            const int hw = p.Ho * p.Wo, n = mm / hw, r = mm - n * hw, oh = r / p.Wo, ow = r - oh * p.Wo;
            a_ih0[i] = oh * p.stride - p.pad; a_iw0[i] = ow * p.stride - p.pad;
            a_base[i] = p.A + (size_t)n * p.H * p.Wd * p.Cin;
        } else {
            a_base[i] = p.A + (size_t)(a_ok[i] ? m : 0) * p.lda;
        }
    }
    const uint16_t* b_base[IB];
    int b_swz[IB];
    bool b_ok[IB];
#pragma unroll
    for (int i = 0; i < IB; ++i) {
        const int row = (wave * IB + i) * 8 + lr, n = n0 + row;
        b_ok[i] = n < p.N;
        b_swz[i] = lpos ^ (row & 7);
        b_base[i] = p.W + (size_t)(b_ok[i] ? n : 0) * p.ldw;
    }

    // conv loader, Cin % 64 == 0 (every bottleneck conv): a 64-wide k slab lies inside ONE filter tap, so the tap
    // (kh, kw) and its first channel are wave-uniform and advance incrementally with the slabs (stage() is called
    // with k0 = 0, 64, 128, ... in order) -- no per-lane integer divisions in the loop
    const bool tap_uniform = CONV && (p.Cin & 63) == 0;
    int st_kh = 0, st_kw = 0, st_ci = 0, st_off = 0;
    const uint16_t* a_pix[IA];
#pragma unroll
    for (int i = 0; i < IA; ++i)
        a_pix[i] = CONV ? a_base[i] + ((ptrdiff_t)a_ih0[i] * p.Wd + a_iw0[i]) * p.Cin + a_swz[i] * 8 : nullptr;
    auto stage = [&](int buf, int k0) {
        unsigned char* slab = lds + __builtin_amdgcn_readfirstlane(buf) * SLAB;
#pragma unroll
        for (int i = 0; i < IA; ++i) {
            const int k = k0 + a_swz[i] * 8;
            const void* src = zero;
            if (a_ok[i] && k < p.K) {
                if (CONV) {
                    if (tap_uniform) {
                        // a_pix = address of (pixel of tap (0,0), this lane's chunk); the tap offset is wave-uniform
                        if ((unsigned)(a_ih0[i] + st_kh) < (unsigned)p.H && (unsigned)(a_iw0[i] + st_kw) < (unsigned)p.Wd)
                            src = a_pix[i] + st_off;
                    } else {
                        const int tap = k / p.Cin, ci = k - tap * p.Cin, kh = tap / p.KS, kw = tap - kh * p.KS;
                        const int ih = a_ih0[i] + kh, iw = a_iw0[i] + kw;
                        if ((unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.Wd)
                            src = a_base[i] + ((size_t)ih * p.Wd + iw) * p.Cin + ci;
                    }
                } else {
                    src = a_base[i] + k;
                }
            }
            __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(slab + (wave * IA + i) * 1024), 16, 0, 0);
        }
        if (tap_uniform) {
            st_ci += BK;
            if (st_ci == p.Cin) { st_ci = 0; if (++st_kw == p.KS) { st_kw = 0; ++st_kh; } }
            st_off = (st_kh * p.Wd + st_kw) * p.Cin + st_ci;
        }
#pragma unroll
        for (int i = 0; i < IB; ++i) {
            const int k = k0 + b_swz[i] * 8;
            const void* src = (b_ok[i] && k < p.K) ? (const void*)(b_base[i] + k) : (const void*)zero;
            __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(slab + A_BYTES + (wave * IB + i) * 1024), 16, 0, 0);
        }
    };

    f32x4 acc[TN][TM];
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int i = 0; i < TM; ++i) acc[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int l15 = lane & 15, lq = lane >> 4;
    const int nslab = (p.K + BK - 1) / BK;
    // prologue: NS-1 slabs in flight
#pragma unroll
    for (int u = 0; u < NS - 1; ++u)
        if (u < nslab) stage(u, u * BK);
    for (int t = 0; t < nslab; ++t) {
        // slab t has landed once at most min(NS-2, slabs issued after t) newer slabs are still outstanding
        const int newer = min(NS - 2, nslab - 1 - t);
        wait_vmcnt(newer * G);
        __builtin_amdgcn_s_barrier();                     // everyone's part of slab t landed; slab t-1 fully consumed
        if (t + NS - 1 < nslab) stage((t + NS - 1) % NS, (t + NS - 1) * BK);   // refill the buffer slab t-1 used
        const unsigned char* sa = lds + (t % NS) * SLAB;
        const unsigned char* sb = sa + A_BYTES;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int g = kk * 4 + lq;                    // logical 16-B chunk holding k = kk*32 + 8*lq .. +7
            bf16x8 fa[TM], fw[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int r = wm0 + i * 16 + l15;
                fa[i] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(sa + r * 128 + ((g ^ (r & 7)) << 4)));
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int r = wn0 + j * 16 + l15;
                fw[j] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(sb + r * 128 + ((g ^ (r & 7)) << 4)));
            }
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int i = 0; i < TM; ++i)
                    acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[j], fa[i], acc[j][i], 0, 0, 0);
        }
    }
    __syncthreads();                                      // all slabs consumed: LDS is free for the epilogue

    // ---- epilogue: acc[j][i][r] = C[m = m0+wm0+16i+(lane&15)][n = n0+wn0+16j+4*(lane>>4)+r] --------------
    float bv[TN][4], sc[TN][4], sh[TN][4];
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int n = n0 + wn0 + 16 * j + 4 * lq + r;
            const bool ok = n < p.N;
            bv[j][r] = (ok && p.bias) ? p.bias[n] : 0.f;
            sc[j][r] = (ok && p.scale) ? p.scale[n] : 1.f;
            sh[j][r] = (ok && p.scale) ? p.shift[n] : 0.f;
        }
    if (p.gmax) {
        // maximum of this wave's WN consecutive output columns for each of its rows (beam-search pre-filter)
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            float mxv = -INFINITY;
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (n0 + wn0 + 16 * j + 4 * lq + r < p.N) mxv = fmaxf(mxv, (acc[j][i][r] + bv[j][r]) * sc[j][r] + sh[j][r]);
            mxv = fmaxf(mxv, __shfl_xor(mxv, 16, 64));
            mxv = fmaxf(mxv, __shfl_xor(mxv, 32, 64));
            const int m = m0 + wm0 + 16 * i + l15;
            if (lq == 0 && m < p.M) p.gmax[(size_t)m * p.gmax_ld + (n0 + wn0) / WN] = mxv;
        }
    }
    // Stage the fp32 tile in LDS, then finish rows with row-contiguous accesses.  Slot s of row m
    // (16 B = 4 fp32) is stored at slot s ^ (m & (SLOTS-1)): the 16 lanes of a store hit 16 distinct slots.
    constexpr int SLOTS = BN / 4;
    float* ep = reinterpret_cast<float*>(lds);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int row = wm0 + 16 * i + l15, slot = (wn0 + 16 * j) / 4 + lq;
            float4 v;
            v.x = (acc[j][i][0] + bv[j][0]) * sc[j][0] + sh[j][0];
            v.y = (acc[j][i][1] + bv[j][1]) * sc[j][1] + sh[j][1];
            v.z = (acc[j][i][2] + bv[j][2]) * sc[j][2] + sh[j][2];
            v.w = (acc[j][i][3] + bv[j][3]) * sc[j][3] + sh[j][3];
            *reinterpret_cast<float4*>(ep + row * BN + ((slot ^ (row & (SLOTS - 1))) << 2)) = v;
        }
    __syncthreads();
    if (p.out_f32) {
        // fp32 output (logits; leading dimension may be odd): lane <-> consecutive column, so every wave
        // store is one contiguous run of up to 256 B of a row whatever its alignment
        float* Cf = reinterpret_cast<float*>(p.C);
        if ((p.ldc & 3) == 0 && ((uintptr_t)Cf & 15) == 0 && !p.res) {
            // row stride a multiple of 4 floats (the decoders pad the logits rows that way): one 16-byte LDS
            // read + one 16-byte global store per 4 columns, a wave writes whole 512-B row segments
            for (int e = tid; e < BM * SLOTS; e += NT) {
                const int row = e / SLOTS, slot = e - row * SLOTS;
                const int m = m0 + row, n = n0 + slot * 4;
                if (m >= p.M || n >= p.N) continue;
                float4 v = *reinterpret_cast<const float4*>(ep + row * BN + ((slot ^ (row & (SLOTS - 1))) << 2));
                if (p.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                if (n + 4 <= p.N) *reinterpret_cast<float4*>(Cf + (size_t)m * p.ldc + n) = v;
                else {      // last, partial group of the row (static indexing: no scratch)
                    float* dst = Cf + (size_t)m * p.ldc + n;
                    dst[0] = v.x;
                    if (n + 1 < p.N) dst[1] = v.y;
                    if (n + 2 < p.N) dst[2] = v.z;
                }
            }
            return;
        }
        for (int e = tid; e < BM * BN; e += NT) {
            const int row = e / BN, col = e - row * BN;
            const int m = m0 + row, n = n0 + col;
            if (m >= p.M || n >= p.N) continue;
            float x = ep[row * BN + ((((col >> 2) ^ (row & (SLOTS - 1))) << 2) | (col & 3))];
            if (p.res) x += bf16_to_f32(p.res[(size_t)m * p.ldres + n]);
            if (p.relu) x = fmaxf(x, 0.f);
            Cf[(size_t)m * p.ldc + n] = x;
        }
        return;
    }
    constexpr int CHUNKS = BN / 8;                        // 16-byte bf16 chunks per tile row
    uint16_t* C = reinterpret_cast<uint16_t*>(p.C);
    const bool fast = ((p.ldc & 7) == 0) && (!p.res || (p.ldres & 7) == 0);
    for (int c = tid; c < BM * CHUNKS; c += NT) {
        const int row = c / CHUNKS, ch = c - row * CHUNKS;
        const int m = m0 + row, n = n0 + ch * 8;
        if (m >= p.M || n >= p.N) continue;
        const int sw = row & (SLOTS - 1);
        const float4 lo = *reinterpret_cast<const float4*>(ep + row * BN + (((2 * ch) ^ sw) << 2));
        const float4 hi = *reinterpret_cast<const float4*>(ep + row * BN + (((2 * ch + 1) ^ sw) << 2));
        float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
        if (fast && n + 8 <= p.N) {
            if (p.res) {
                float q[8];
                load16(reinterpret_cast<const bf16_t*>(p.res + (size_t)m * p.ldres + n), q);
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] += q[u];
            }
            if (p.relu) {
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = fmaxf(v[u], 0.f);
            }
            store16(reinterpret_cast<bf16_t*>(C + (size_t)m * p.ldc + n), v);
        } else {
            for (int u = 0; u < 8 && n + u < p.N; ++u) {
                float x = v[u];
                if (p.res) x += bf16_to_f32(p.res[(size_t)m * p.ldres + n + u]);
                if (p.relu) x = fmaxf(x, 0.f);
                C[(size_t)m * p.ldc + n + u] = f32_to_bf16(x);
            }
        }
    }
}

template <bool CONV>
static void launch_gemm_bf16(GemmBf16Params& p, hipStream_t s) {
    const long long big_tiles = (long long)dh_cdiv(p.M, 128) * dh_cdiv(p.N, 128);
    // >= 192 big tiles (measured: lowering the threshold to 128/100/40 tiles does not help gates / ffn / qkv / proj)
    if (big_tiles >= 192 && p.M >= 96 && p.N >= 96) {
        p.tiles_m = dh_cdiv(p.M, 128); p.tiles_n = dh_cdiv(p.N, 128);
        // 8 waves (4 x 2, each 32 x 64) on the 128 x 128 tile, 2 workgroups per CU = 4 waves per SIMD: measured
        // 510 TF vs 450 TF with 4 waves per workgroup and 300 TF with one 4-wave workgroup and a deeper ring --
        // the MFMA pipe needs co-resident waves to cover each wave's LDS-read/barrier gaps
        hipLaunchKernelGGL((gemm_bf16_kernel<128, 128, 4, CONV, 2, 8>), dim3(p.tiles_m * p.tiles_n), dim3(512), 0, s, p);
        return;
    }
    if (p.N <= 64 && p.M >= 256 * 512) {
        // narrow outputs with very many rows (stage-1 convolutions): 256 x 64 tiles, 4 waves stacked along M, so the
        // 64 weight rows are staged once per 256 pixels and every wave still owns a 64 x 64 accumulator
        p.tiles_m = dh_cdiv(p.M, 256); p.tiles_n = dh_cdiv(p.N, 64);
        hipLaunchKernelGGL((gemm_bf16_kernel<256, 64, 4, CONV, 2>), dim3(p.tiles_m * p.tiles_n), dim3(256), 0, s, p);
        return;
    }
    p.tiles_m = dh_cdiv(p.M, 64); p.tiles_n = dh_cdiv(p.N, 64);
    const int blocks = p.tiles_m * p.tiles_n;
    // few blocks: one per CU with a deep ring (7 slabs = 112 KB in flight); many blocks: two per CU, 3 in flight each
    if (blocks <= 320 && p.K > 128)
        hipLaunchKernelGGL((gemm_bf16_kernel<64, 64, 2, CONV, 8>), dim3(blocks), dim3(256), 0, s, p);
    else
        hipLaunchKernelGGL((gemm_bf16_kernel<64, 64, 2, CONV, 4>), dim3(blocks), dim3(256), 0, s, p);
}

// called from dh_linear (gemm.hip) for DH_BF16 / DH_BF16_OUT_F32
int dh_linear_bf16_impl(const void* A, int lda, const void* W, int ldw, const float* bias, const float* scale,
                        const float* shift, const void* residual, int ldres, void* C, int ldc, int M, int N, int K,
                        int relu, int out_f32, hipStream_t s) {
    DH_REQUIRE((K % 8) == 0 && (lda % 8) == 0 && (ldw % 8) == 0 && lda >= K && ldw >= K && ldc >= N);
    DH_REQUIRE(((uintptr_t)A % 16) == 0 && ((uintptr_t)W % 16) == 0);
    GemmBf16Params p{};
    p.A = (const uint16_t*)A; p.lda = lda; p.W = (const uint16_t*)W; p.ldw = ldw;
    p.bias = bias; p.scale = scale; p.shift = shift; p.res = (const uint16_t*)residual; p.ldres = ldres;
    p.C = C; p.ldc = ldc; p.M = M; p.N = N; p.K = K; p.relu = relu; p.out_f32 = out_f32;
    launch_gemm_bf16<false>(p, s);
    DH_LAUNCH_CHECK();
}

extern "C" int dh_conv2d_nhwc_bn_act(const void* x, const void* w, const float* scale, const float* shift,
                                     const void* residual, void* y, int N, int H, int W, int Cin, int Cout,
                                     int KS, int stride, int pad, int relu, int dtype, void* stream) {
    if (dtype != DH_BF16) return DH_ERR_UNSUPPORTED;
    DH_REQUIRE(x && w && scale && shift && y && N > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0);
    DH_REQUIRE((Cin % 8) == 0 && KS >= 1 && stride >= 1 && pad >= 0);
    DH_REQUIRE(((uintptr_t)x % 16) == 0 && ((uintptr_t)w % 16) == 0);
    GemmBf16Params p{};
    p.Ho = (H + 2 * pad - KS) / stride + 1; p.Wo = (W + 2 * pad - KS) / stride + 1;
    DH_REQUIRE(p.Ho > 0 && p.Wo > 0 && (long long)N * p.Ho * p.Wo < (1ll << 31));
    p.A = (const uint16_t*)x; p.W = (const uint16_t*)w; p.ldw = KS * KS * Cin;
    p.scale = scale; p.shift = shift; p.res = (const uint16_t*)residual; p.ldres = Cout;
    p.C = y; p.ldc = Cout; p.M = N * p.Ho * p.Wo; p.N = Cout; p.K = KS * KS * Cin; p.relu = relu; p.out_f32 = 0;
    p.H = H; p.Wd = W; p.Cin = Cin; p.KS = KS; p.stride = stride; p.pad = pad;
    hipStream_t s = (hipStream_t)stream;
    static const char* const tags[] = {"?", "1x1", "2x2", "3x3", "4x4", "5x5", "6x6", "7x7"};
    dh_prof_set_tag(tags[KS < 8 ? KS : 0]);
    DhProfScope prof("dh_conv2d_nhwc_bn_act", 2.0 * p.M * Cout * p.K,
                     2.0 * ((double)N * H * W * Cin + (double)Cout * p.K + (double)p.M * Cout * (residual ? 2 : 1)), stream);
    if (KS == 1 && stride == 1 && pad == 0) { p.lda = Cin; p.conv = 0; launch_gemm_bf16<false>(p, s); }
    else { p.conv = 1; launch_gemm_bf16<true>(p, s); }
    DH_LAUNCH_CHECK();
}

// Vocabulary projection for beam search: logits[M,V] fp32 = A[M,K] * W[V,K]^T + bias, plus group_max[m, g] =
// max of logits[m, 64g .. 64g+63] (always the 128x128 tile: its waves own 64-column groups).
extern "C" int dh_vocab_logits(const void* A, int lda, const void* W, int ldw, const float* bias, float* logits, int ldl,
                               float* group_max, int gm_ld, int M, int V, int K, int dtype, void* stream) {
    if (dtype != DH_BF16) return DH_ERR_UNSUPPORTED;
    DH_REQUIRE(A && W && logits && group_max && M > 0 && V > 0 && K > 0 && ldl >= V && gm_ld >= 2 * dh_cdiv(V, 128));
    DH_REQUIRE((K % 8) == 0 && (lda % 8) == 0 && (ldw % 8) == 0 && lda >= K && ldw >= K);
    DH_REQUIRE(((uintptr_t)A % 16) == 0 && ((uintptr_t)W % 16) == 0);
    dh_prof_set_tag("vocab");
    DhProfScope prof("dh_linear", 2.0 * M * V * K, 2.0 * ((double)M * K + (double)V * K) + 4.0 * M * V, stream);
    GemmBf16Params p{};
    p.A = (const uint16_t*)A; p.lda = lda; p.W = (const uint16_t*)W; p.ldw = ldw; p.bias = bias;
    p.C = logits; p.ldc = ldl; p.M = M; p.N = V; p.K = K; p.out_f32 = 1; p.gmax = group_max; p.gmax_ld = gm_ld;
    p.tiles_m = dh_cdiv(M, 128); p.tiles_n = dh_cdiv(V, 128);
    hipLaunchKernelGGL((gemm_bf16_kernel<128, 128, 4, false, 2, 8>), dim3(p.tiles_m * p.tiles_n), dim3(512), 0, (hipStream_t)stream, p);
    DH_LAUNCH_CHECK();
}
