// 16-bit matrix-core engine: C[M,N] = act((A[M,K] * W[N,K]^T + bias) * scale + shift (+ residual)),
// bf16 or fp16 operands (template parameter OT; the data path moves raw 16-bit patterns and is type-agnostic),
// fp32 accumulation in v_mfma_f32_16x16x32_{bf16,f16}, 16-bit or fp32 output.
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
#include <type_traits>
#include "common.h"
#include "prof.h"
#include "options.h"

typedef dh_f32x4 f32x4;

__device__ uint4 dh_zero_page[4];        // zero-initialised: source of padding chunks

struct GemmBf16Params {
    const uint16_t* A; int lda;
    const uint16_t* W; int ldw;
    const float* bias; const float* scale; const float* shift;
    const uint16_t* res; int ldres;
    void* C; int ldc;
    int M, N, K, relu, out_f32;
    int conv, H, Wd, Cin, Ho, Wo, KS, stride, pad;     // conv loader: A = NHWC input
    unsigned cin_magic, ks_magic;                      // k / Cin == (k * cin_magic) >> 20 for k < K; tap / KS likewise
    int tiles_m, tiles_n, n_fast;                      // n_fast: consecutive workgroups walk the N tiles of one M tile
    // dense mode, optional second A source for k >= K1 (the fused "conv3 + downsample" of a ResNet stage's first block):
    // row m = output pixel (n, oh, ow) reads channels of input pixel (n, oh*a2_stride, ow*a2_stride) of the NHWC tensor A2
    const uint16_t* A2; int K1, a2_H, a2_W, a2_C, a2_stride, a2_Ho, a2_Wo;
    float* gmax; int gmax_ld;                          // optional: per-row maxima of each wave-wide column group
    // ---- deferred LayerNorm (template flag LNX; decode / prefill chain of the 16-bit Transformer decoders) -------------
    // Rows travel PRE-LayerNorm together with per-(row, 64-column tile) partial statistics (mean, M2 = sum of squared
    // deviations from that mean) left by the GEMM that produced them; a row's mean / rstd is the fixed-order (Chan)
    // combination of its tiles -- deterministic, no atomics.
    const float2* a_stats; int a_nt; float a_eps;      // A rows are pre-LN: W has gamma folded in (W' = W * gamma[k]), bias has
    const float* a_colsum;                             //   beta folded in; out = rstd * (acc - mu * colsum[n]) + bias[n], colsum[n] = sum_k W'[n,k]
    const float2* r_stats; int r_nt; float r_eps;      // residual rows are pre-LN: res' = (res - mu) * rstd * r_gamma[n] + r_beta[n]
    const float* r_gamma; const float* r_beta;
    float2* o_stats;                                   // out: partial statistics of the OUTPUT rows, [M][N / 64] (requires BN == 64)
};

// wait until at most `n` of this wave's vector-memory operations (LDS-DMA included) are outstanding
#define DH_VMCNT_CASE(x) case x: asm volatile("s_waitcnt vmcnt(" #x ")" ::: "memory"); break;
__device__ __forceinline__ void wait_vmcnt_any(int n) {
    switch (n) {
        DH_VMCNT_CASE(1) DH_VMCNT_CASE(2) DH_VMCNT_CASE(3) DH_VMCNT_CASE(4) DH_VMCNT_CASE(5) DH_VMCNT_CASE(6)
        DH_VMCNT_CASE(7) DH_VMCNT_CASE(8) DH_VMCNT_CASE(9) DH_VMCNT_CASE(10) DH_VMCNT_CASE(11) DH_VMCNT_CASE(12)
        DH_VMCNT_CASE(13) DH_VMCNT_CASE(14) DH_VMCNT_CASE(15) DH_VMCNT_CASE(16) DH_VMCNT_CASE(17) DH_VMCNT_CASE(18)
        DH_VMCNT_CASE(19) DH_VMCNT_CASE(20) DH_VMCNT_CASE(21) DH_VMCNT_CASE(22) DH_VMCNT_CASE(23) DH_VMCNT_CASE(24)
        DH_VMCNT_CASE(25) DH_VMCNT_CASE(26) DH_VMCNT_CASE(27) DH_VMCNT_CASE(28) DH_VMCNT_CASE(29) DH_VMCNT_CASE(30)
        DH_VMCNT_CASE(31) DH_VMCNT_CASE(32) DH_VMCNT_CASE(33) DH_VMCNT_CASE(34) DH_VMCNT_CASE(35) DH_VMCNT_CASE(36)
        DH_VMCNT_CASE(37) DH_VMCNT_CASE(38) DH_VMCNT_CASE(39) DH_VMCNT_CASE(40) DH_VMCNT_CASE(41) DH_VMCNT_CASE(42)
        DH_VMCNT_CASE(43) DH_VMCNT_CASE(44) DH_VMCNT_CASE(45) DH_VMCNT_CASE(46) DH_VMCNT_CASE(47) DH_VMCNT_CASE(48)
        DH_VMCNT_CASE(49) DH_VMCNT_CASE(50) DH_VMCNT_CASE(51) DH_VMCNT_CASE(52) DH_VMCNT_CASE(53) DH_VMCNT_CASE(54)
        DH_VMCNT_CASE(55) DH_VMCNT_CASE(56) DH_VMCNT_CASE(57) DH_VMCNT_CASE(58) DH_VMCNT_CASE(59) DH_VMCNT_CASE(60)
        DH_VMCNT_CASE(61) DH_VMCNT_CASE(62) DH_VMCNT_CASE(63)
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
}

__device__ __forceinline__ void wait_vmcnt(int n) { wait_vmcnt_any(n); }

// The switch above costs a tree of ~6 scalar compares + TAKEN branches per call when `n` is a run-time value -- measured with
// tools/probe/ta_probe: a bare LDS-DMA ring (2 pieces per wave per step, 8 deep, one barrier per step) streams 71.6 GB/s per CU
// with an immediate s_waitcnt and 39.7 GB/s with the switch in front of it: several hundred cycles per K slab, as much as the
// slab's MFMAs.  In steady state the count is a compile-time constant (the ring is full: NS - 2 slabs younger than the one
// waited for); only the last NS - 2 slabs of the reduction need smaller counts.  HOT = that constant: one scalar compare and a
// not-taken branch in the loop, the switch only in the tail.
template <int HOT>
__device__ __forceinline__ void wait_vmcnt_hot(int n) {
    if (__builtin_expect(n == HOT, 1)) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(HOT) : "memory");
    else wait_vmcnt_any(n);
}

// NS = LDS ring depth: slabs t+1 .. t+NS-1 are in flight (LDS-DMA) while slab t feeds the MFMAs.
// NW waves per workgroup, arranged WAVES_M x (NW / WAVES_M) over the BM x BN block.
// EXT: 0 = plain; 1 = deferred LayerNorm on the A rows (a_stats); 3 = deferred LayerNorm on the residual rows and / or
// statistics of the output rows (r_stats, o_stats); 2 = convolution + MaxPool2d(3, 2, 1) in one launch (POOL, the ResNet
// stem): a workgroup's 256 GEMM rows are a 15 x 15 patch of convolution pixels whose BatchNorm + ReLU outputs stay in LDS and
// leave the kernel as the 7 x 7 pooled pixels they cover -- the 4x larger un-pooled activation never goes to memory.
template <typename OT, int BM, int BN, int WAVES_M, bool CONV, int NS, int NW = 4, int EXT = 0>
__global__ __launch_bounds__(64 * NW) void gemm_bf16_kernel(GemmBf16Params p) {
    constexpr bool AFX = EXT == 1, LNX = EXT == 3, POOL = EXT == 2;
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
    // tile order: by default consecutive workgroups share a weight panel (tn) and walk over M; with a small weight
    // matrix (L2-resident anyway) they share the activation tile instead, so it is fetched from HBM once, not tiles_n times
    const int tm = p.n_fast ? bid / p.tiles_n : bid % p.tiles_m, tn = p.n_fast ? bid % p.tiles_n : bid / p.tiles_m;
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
        if constexpr (POOL) {
            // tile tm = (image, 7 x 7 block of pooled pixels); row = pixel (cy, cx) of the 15 x 15 convolution patch under it
            const int pbw = (p.Wo / 2 + 6) / 7, pbh = (p.Ho / 2 + 6) / 7;
            const int n = tm / (pbw * pbh), blk = tm - n * (pbw * pbh), by = blk / pbw, bx = blk - by * pbw;
            const int cy = row / 15, cx = row - cy * 15;
            const int oh = 14 * by - 1 + cy, ow = 14 * bx - 1 + cx;
            a_ok[i] = row < 225 && (unsigned)oh < (unsigned)p.Ho && (unsigned)ow < (unsigned)p.Wo;
            a_ih0[i] = oh * p.stride - p.pad; a_iw0[i] = ow * p.stride - p.pad;
            a_base[i] = p.A + (size_t)n * p.H * p.Wd * p.Cin;
        } else if (CONV) {
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
    // dense operands with K % 64 == 0 (no K tail): one running source pointer per LDS-DMA piece, rows outside M / N
    // parked on the zero page with step 0 -- no selects or multiplies in the loop (the SIMD's instruction issue bounds
    // this loop, see vocab_logits_kernel)
    const bool stepping = (p.K & 63) == 0;
    const uint16_t* a_run[IA]; const uint16_t* b_run[IB];
    const uint16_t* a2_run[IA];
    int a_stp[IA], b_stp[IB];
#pragma unroll
    for (int i = 0; i < IA; ++i) {
        a_run[i] = (!CONV && a_ok[i]) ? a_base[i] + a_swz[i] * 8 : reinterpret_cast<const uint16_t*>(zero);
        a_stp[i] = (!CONV && a_ok[i]) ? BK : 0;
        a2_run[i] = reinterpret_cast<const uint16_t*>(zero);
        if (!CONV && p.A2 && a_ok[i]) {
            const int m = m0 + (wave * IA + i) * 8 + lr;
            const int hw = p.a2_Ho * p.a2_Wo, n = m / hw, r = m - n * hw, oh = r / p.a2_Wo, ow = r - oh * p.a2_Wo;
            a2_run[i] = p.A2 + (((size_t)n * p.a2_H + (size_t)oh * p.a2_stride) * p.a2_W + (size_t)ow * p.a2_stride) * p.a2_C + a_swz[i] * 8;
        }
    }
#pragma unroll
    for (int i = 0; i < IB; ++i) {
        b_run[i] = b_ok[i] ? b_base[i] + b_swz[i] * 8 : reinterpret_cast<const uint16_t*>(zero);
        b_stp[i] = b_ok[i] ? BK : 0;
    }
    auto stage = [&](int buf, int k0) {
        unsigned char* slab = lds + __builtin_amdgcn_readfirstlane(buf) * SLAB;
        if (!CONV && stepping) {
            if (p.A2 && k0 == p.K1) {                     // wave-uniform: the remaining slabs come from the second source
#pragma unroll
                for (int i = 0; i < IA; ++i) a_run[i] = a2_run[i];
            }
#pragma unroll
            for (int i = 0; i < IA; ++i) {
                dh_lds_dma16(a_run[i], slab + (wave * IA + i) * 1024);
                a_run[i] += a_stp[i];
            }
        } else {
#pragma unroll
            for (int i = 0; i < IA; ++i) {
                const int k = k0 + a_swz[i] * 8;
                const void* src = zero;
                if (a_ok[i] && k < p.K) {
                    if (!CONV) {
                        src = a_base[i] + k;
                    } else if (tap_uniform) {
                        // a_pix = address of (pixel of tap (0,0), this lane's chunk); the tap offset is wave-uniform
                        if ((unsigned)(a_ih0[i] + st_kh) < (unsigned)p.H && (unsigned)(a_iw0[i] + st_kw) < (unsigned)p.Wd)
                            src = a_pix[i] + st_off;
                    } else {
                        // k -> (tap, ci), tap -> (kh, kw) by multiply-shift with host-verified magic numbers
                        const int tap = (int)(((unsigned)k * p.cin_magic) >> 20), ci = k - tap * p.Cin;
                        const int kh = (int)(((unsigned)tap * p.ks_magic) >> 20), kw = tap - kh * p.KS;
                        const int ih = a_ih0[i] + kh, iw = a_iw0[i] + kw;
                        if ((unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.Wd)
                            src = a_base[i] + ((size_t)ih * p.Wd + iw) * p.Cin + ci;
                    }
                }
                dh_lds_dma16(src, slab + (wave * IA + i) * 1024);
            }
            if (tap_uniform) {
                st_ci += BK;
                if (st_ci == p.Cin) { st_ci = 0; if (++st_kw == p.KS) { st_kw = 0; ++st_kh; } }
                st_off = (st_kh * p.Wd + st_kw) * p.Cin + st_ci;
            }
        }
        if (stepping) {
#pragma unroll
            for (int i = 0; i < IB; ++i) {
                dh_lds_dma16(b_run[i], slab + A_BYTES + (wave * IB + i) * 1024);
                b_run[i] += b_stp[i];
            }
        } else {
#pragma unroll
            for (int i = 0; i < IB; ++i) {
                const int k = k0 + b_swz[i] * 8;
                const void* src = (b_ok[i] && k < p.K) ? (const void*)(b_base[i] + k) : (const void*)zero;
                dh_lds_dma16(src, slab + A_BYTES + (wave * IB + i) * 1024);
            }
        }
    };

    f32x4 acc[TN][TM];
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int i = 0; i < TM; ++i) acc[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int l15 = lane & 15, lq = lane >> 4;
    // per-column epilogue constants, requested BEFORE the first slab so their latency hides behind the operand
    // loads (a K = 64 convolution has a single slab): out = (acc + bias) * mul + add with
    //   scale given: mul = scale, add = shift (bias, if any, is fetched late -- only the tiny BatchNorm1d linear has both)
    //   otherwise  : mul = 1,     add = bias
    float mulv[TN][4], addv[TN][4];
    const float* addp = p.scale ? p.shift : p.bias;
    const bool vec = (p.N & 3) == 0 && ((((uintptr_t)p.scale) | ((uintptr_t)addp)) & 15) == 0;
    if (vec) {                        // 16-byte loads up front; an odd N (vocabulary) takes 4-byte loads after the loop
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + wn0 + 16 * j + 4 * lq;
            const bool ok = n < p.N;
            const float4 one = make_float4(1.f, 1.f, 1.f, 1.f), zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
            const float4 mv = (ok && p.scale) ? *reinterpret_cast<const float4*>(p.scale + n) : one;
            const float4 av = (ok && addp) ? *reinterpret_cast<const float4*>(addp + n) : zero4;
            mulv[j][0] = mv.x; mulv[j][1] = mv.y; mulv[j][2] = mv.z; mulv[j][3] = mv.w;
            addv[j][0] = av.x; addv[j][1] = av.y; addv[j][2] = av.z; addv[j][3] = av.w;
        }
    }
    // deferred LayerNorm on the A rows: this lane's rows' statistics partials and its columns' folded-weight row sums are
    // REQUESTED here, in front of the operand slabs; the arithmetic on them runs after the reduction
    float4 a_raw[TM][4];
    float csum[TN][4];
    constexpr bool a_fold = AFX;
    if constexpr (AFX) {
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int m = min(m0 + wm0 + 16 * i + l15, p.M - 1);
            ln_load(p.a_stats + (size_t)m * p.a_nt, p.a_nt, a_raw[i]);
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + wn0 + 16 * j + 4 * lq;
            const float4 c4 = n < p.N ? *reinterpret_cast<const float4*>(p.a_colsum + n) : make_float4(0.f, 0.f, 0.f, 0.f);
            csum[j][0] = c4.x; csum[j][1] = c4.y; csum[j][2] = c4.z; csum[j][3] = c4.w;
        }
    }
    // residual path of the deferred-LayerNorm chain: residual chunk, its row's statistics partials, gamma / beta -- requested
    // here as well (one thread = one 16-byte chunk of one row per epilogue iteration)
    constexpr int CHUNKS = BN / 8;                        // 16-byte bf16 chunks per tile row
    constexpr int EP_IT = (BM * CHUNKS + NT - 1) / NT;
    uint4 rq[EP_IT];
    float4 r_raw[EP_IT][4], rg[EP_IT][2], rb[EP_IT][2];
    bool r_ln = false;
    if constexpr (LNX) {
        r_ln = p.r_stats != nullptr;
        if (p.res) {
#pragma unroll
            for (int it = 0; it < EP_IT; ++it) {
                const int c = tid + it * NT, row = c / CHUNKS, ch = c - row * CHUNKS;
                const int m = min(m0 + row, p.M - 1), n = min(n0 + ch * 8, p.N - 8);
                rq[it] = *reinterpret_cast<const uint4*>(p.res + (size_t)m * p.ldres + n);
                if (r_ln) {
                    ln_load(p.r_stats + (size_t)m * p.r_nt, p.r_nt, r_raw[it]);
                    rg[it][0] = *reinterpret_cast<const float4*>(p.r_gamma + n); rg[it][1] = *reinterpret_cast<const float4*>(p.r_gamma + n + 4);
                    rb[it][0] = *reinterpret_cast<const float4*>(p.r_beta + n); rb[it][1] = *reinterpret_cast<const float4*>(p.r_beta + n + 4);
                }
            }
        }
    }
    const int nslab = (p.K + BK - 1) / BK;
    // prologue: NS-1 slabs in flight
#pragma unroll
    for (int u = 0; u < NS - 1; ++u)
        if (u < nslab) stage(u, u * BK);
    for (int t = 0; t < nslab; ++t) {
        // slab t has landed once at most min(NS-2, slabs issued after t) newer slabs are still outstanding
        if constexpr (NS == 2) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // two-slab ring: nothing younger than slab t is in flight
        } else {
            const int newer = min(NS - 2, nslab - 1 - t);
            wait_vmcnt_hot<(NS - 2) * G>(newer * G);
        }
        __builtin_amdgcn_s_barrier();                     // everyone's part of slab t landed; slab t-1 fully consumed
        if (t + NS - 1 < nslab) stage((t + NS - 1) % NS, (t + NS - 1) * BK);   // refill the buffer slab t-1 used
        const unsigned char* sa = lds + (t % NS) * SLAB;
        const unsigned char* sb = sa + A_BYTES;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int g = kk * 4 + lq;                    // logical 16-B chunk holding k = kk*32 + 8*lq .. +7
            uint4 fa[TM], fw[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int r = wm0 + i * 16 + l15;
                fa[i] = *reinterpret_cast<const uint4*>(sa + r * 128 + ((g ^ (r & 7)) << 4));
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int r = wn0 + j * 16 + l15;
                fw[j] = *reinterpret_cast<const uint4*>(sb + r * 128 + ((g ^ (r & 7)) << 4));
            }
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int i = 0; i < TM; ++i)
                    acc[j][i] = Op16<OT>::mfma(fw[j], fa[i], acc[j][i]);
        }
    }
    __syncthreads();                                      // all slabs consumed: LDS is free for the epilogue

    // ---- epilogue: acc[j][i][r] = C[m = m0+wm0+16i+(lane&15)][n = n0+wn0+16j+4*(lane>>4)+r] --------------
    if (!vec) {
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = n0 + wn0 + 16 * j + 4 * lq + r;
                const bool ok = n < p.N;
                mulv[j][r] = (ok && p.scale) ? p.scale[n] : 1.f;
                addv[j][r] = (ok && addp) ? addp[n] : 0.f;
            }
    }
    if (p.bias && p.scale) {                              // rare: bias AND BatchNorm affine -> fold the bias into add
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = n0 + wn0 + 16 * j + 4 * lq + r;
                if (n < p.N) addv[j][r] = fmaf(p.bias[n], mulv[j][r], addv[j][r]);
            }
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
                    if (n0 + wn0 + 16 * j + 4 * lq + r < p.N) mxv = fmaxf(mxv, fmaf(acc[j][i][r], mulv[j][r], addv[j][r]));
            mxv = fmaxf(mxv, __shfl_xor(mxv, 16, 64));
            mxv = fmaxf(mxv, __shfl_xor(mxv, 32, 64));
            const int m = m0 + wm0 + 16 * i + l15;
            if (lq == 0 && m < p.M) p.gmax[(size_t)m * p.gmax_ld + (n0 + wn0) / WN] = mxv;
        }
    }
    // bf16 output with a residual: this thread's residual chunks (one 16-byte chunk per epilogue iteration) are requested
    // NOW, so they are in flight while the tile is staged through LDS -- loaded inside the loop below each of them
    // cost a full memory round trip (load, wait, add, store, next)
    const bool fast = ((p.ldc & 7) == 0) && (!p.res || (p.ldres & 7) == 0);
    float a_mu[TM], a_rs[TM], r_mu[EP_IT], r_rs[EP_IT];
    if constexpr (AFX) {
#pragma unroll
        for (int i = 0; i < TM; ++i) ln_math(a_raw[i], p.a_nt, p.a_eps, a_mu[i], a_rs[i]);
    }
    if constexpr (LNX) {
        if (r_ln) {
#pragma unroll
            for (int it = 0; it < EP_IT; ++it) ln_math(r_raw[it], p.r_nt, p.r_eps, r_mu[it], r_rs[it]);
        }
    } else if (!p.out_f32 && p.res && fast) {
#pragma unroll
        for (int it = 0; it < EP_IT; ++it) {
            const int c = tid + it * NT, row = c / CHUNKS, ch = c - row * CHUNKS;
            const int m = m0 + row, n = n0 + ch * 8;
            const bool ok = c < BM * CHUNKS && m < p.M && n + 8 <= p.N;
            rq[it] = *reinterpret_cast<const uint4*>(p.res + (ok ? (size_t)m * p.ldres + n : 0));
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
            if (a_fold) {               // LayerNorm of the A rows applied on the accumulators: rstd * (acc - mu * colsum) + bias'
                v.x = fmaf(a_rs[i], fmaf(-a_mu[i], csum[j][0], acc[j][i][0]), addv[j][0]);
                v.y = fmaf(a_rs[i], fmaf(-a_mu[i], csum[j][1], acc[j][i][1]), addv[j][1]);
                v.z = fmaf(a_rs[i], fmaf(-a_mu[i], csum[j][2], acc[j][i][2]), addv[j][2]);
                v.w = fmaf(a_rs[i], fmaf(-a_mu[i], csum[j][3], acc[j][i][3]), addv[j][3]);
            } else {
                v.x = fmaf(acc[j][i][0], mulv[j][0], addv[j][0]);
                v.y = fmaf(acc[j][i][1], mulv[j][1], addv[j][1]);
                v.z = fmaf(acc[j][i][2], mulv[j][2], addv[j][2]);
                v.w = fmaf(acc[j][i][3], mulv[j][3], addv[j][3]);
            }
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
            if (p.res) x += Op16<OT>::to_f32(p.res[(size_t)m * p.ldres + n]);
            if (p.relu) x = fmaxf(x, 0.f);
            Cf[(size_t)m * p.ldc + n] = x;
        }
        return;
    }
    uint16_t* C = reinterpret_cast<uint16_t*>(p.C);
    if constexpr (POOL) {
        // MaxPool2d(kernel 3, stride 2, padding 1) over the staged 15 x 15 patch (BatchNorm applied, ReLU here): pooled pixel
        // (py, px) of the block = max over patch pixels (2py + {0,1,2}, 2px + {0,1,2}); pixels outside the image do not exist
        const int Hp = p.Ho / 2, Wp = p.Wo / 2;
        const int pbw = (Wp + 6) / 7, pbh = (Hp + 6) / 7;
        const int n = tm / (pbw * pbh), blk = tm - n * (pbw * pbh), by = blk / pbw, bx = blk - by * pbw;
        for (int e = tid; e < 49 * CHUNKS; e += NT) {
            const int pp = e / CHUNKS, ch = e - pp * CHUNKS, py = pp / 7, px = pp - py * 7;
            const int gy = 7 * by + py, gx = 7 * bx + px;
            if (gy >= Hp || gx >= Wp) continue;
            float m8[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) m8[u] = 0.f;                           // ReLU floor (every window holds >= 1 real pixel)
#pragma unroll
            for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) {
                    const int cy = 2 * py + dy, cx = 2 * px + dx, row = cy * 15 + cx;
                    const int oh = 14 * by - 1 + cy, ow = 14 * bx - 1 + cx;
                    if ((unsigned)oh >= (unsigned)p.Ho || (unsigned)ow >= (unsigned)p.Wo) continue;
                    const int sw = row & (SLOTS - 1);
                    const float4 lo = *reinterpret_cast<const float4*>(ep + row * BN + (((2 * ch) ^ sw) << 2));
                    const float4 hi = *reinterpret_cast<const float4*>(ep + row * BN + (((2 * ch + 1) ^ sw) << 2));
                    m8[0] = fmaxf(m8[0], lo.x); m8[1] = fmaxf(m8[1], lo.y); m8[2] = fmaxf(m8[2], lo.z); m8[3] = fmaxf(m8[3], lo.w);
                    m8[4] = fmaxf(m8[4], hi.x); m8[5] = fmaxf(m8[5], hi.y); m8[6] = fmaxf(m8[6], hi.z); m8[7] = fmaxf(m8[7], hi.w);
                }
            if (n0 + ch * 8 + 8 <= p.N)
                store16(reinterpret_cast<OT*>(C + (((size_t)n * Hp + gy) * Wp + gx) * p.ldc + n0 + ch * 8), m8);
        }
        return;
    }
    if constexpr (LNX) {
        if (p.o_stats) {
            // statistics-emitting form (N % 64 == 0, 16-byte rows: checked by the host): every lane takes part in the 8-lane row
            // reductions, so no early exits; the statistics are those of the ROUNDED values the consumers will read
            static_assert(BN == 64, "one statistics tile = one 64-column GEMM tile");
#pragma unroll
            for (int it = 0; it < EP_IT; ++it) {
                const int c = tid + it * NT, row = c / CHUNKS, ch = c - row * CHUNKS;
                const int m = m0 + row, n = n0 + ch * 8;
                const int sw = row & (SLOTS - 1);
                const float4 lo = *reinterpret_cast<const float4*>(ep + row * BN + (((2 * ch) ^ sw) << 2));
                const float4 hi = *reinterpret_cast<const float4*>(ep + row * BN + (((2 * ch + 1) ^ sw) << 2));
                float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
                if (p.res) {
                    const uint32_t w4[4] = {rq[it].x, rq[it].y, rq[it].z, rq[it].w};
                    float rr[8];
#pragma unroll
                    for (int u = 0; u < 4; ++u) Op16<OT>::unpack2(w4[u], rr[2 * u], rr[2 * u + 1]);
                    if (r_ln) {
                        const float g8[8] = {rg[it][0].x, rg[it][0].y, rg[it][0].z, rg[it][0].w, rg[it][1].x, rg[it][1].y, rg[it][1].z, rg[it][1].w};
                        const float b8[8] = {rb[it][0].x, rb[it][0].y, rb[it][0].z, rb[it][0].w, rb[it][1].x, rb[it][1].y, rb[it][1].z, rb[it][1].w};
#pragma unroll
                        for (int u = 0; u < 8; ++u) rr[u] = fmaf((rr[u] - r_mu[it]) * r_rs[it], g8[u], b8[u]);
                    }
#pragma unroll
                    for (int u = 0; u < 8; ++u) v[u] += rr[u];
                }
                if (p.relu) {
#pragma unroll
                    for (int u = 0; u < 8; ++u) v[u] = fmaxf(v[u], 0.f);
                }
                float s1 = 0.f;
#pragma unroll
                for (int u = 0; u < 8; ++u) { v[u] = Op16<OT>::to_f32(Op16<OT>::from_f32(v[u])); s1 += v[u]; }
                const float mean = sum8(s1) * (1.0f / 64.0f);
                float s2 = 0.f;
#pragma unroll
                for (int u = 0; u < 8; ++u) { const float d = v[u] - mean; s2 = fmaf(d, d, s2); }
                s2 = sum8(s2);
                if (m < p.M) {
                    store16(reinterpret_cast<OT*>(C + (size_t)m * p.ldc + n), v);
                    if (ch == 0) p.o_stats[(size_t)m * p.tiles_n + tn] = make_float2(mean, s2);
                }
            }
            return;
        }
    }
#pragma unroll
    for (int it = 0; it < EP_IT; ++it) {
        const int c = tid + it * NT;
        if (c >= BM * CHUNKS) break;
        const int row = c / CHUNKS, ch = c - row * CHUNKS;
        const int m = m0 + row, n = n0 + ch * 8;
        if (m >= p.M || n >= p.N) continue;
        const int sw = row & (SLOTS - 1);
        const float4 lo = *reinterpret_cast<const float4*>(ep + row * BN + (((2 * ch) ^ sw) << 2));
        const float4 hi = *reinterpret_cast<const float4*>(ep + row * BN + (((2 * ch + 1) ^ sw) << 2));
        float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
        if (fast && n + 8 <= p.N) {
            if (p.res) {
                const uint32_t w4[4] = {rq[it].x, rq[it].y, rq[it].z, rq[it].w};
                float rr[8];
#pragma unroll
                for (int u = 0; u < 4; ++u) Op16<OT>::unpack2(w4[u], rr[2 * u], rr[2 * u + 1]);
                if (LNX && r_ln) {
                    const float g8[8] = {rg[it][0].x, rg[it][0].y, rg[it][0].z, rg[it][0].w, rg[it][1].x, rg[it][1].y, rg[it][1].z, rg[it][1].w};
                    const float b8[8] = {rb[it][0].x, rb[it][0].y, rb[it][0].z, rb[it][0].w, rb[it][1].x, rb[it][1].y, rb[it][1].z, rb[it][1].w};
#pragma unroll
                    for (int u = 0; u < 8; ++u) rr[u] = fmaf((rr[u] - r_mu[it]) * r_rs[it], g8[u], b8[u]);
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] += rr[u];
            }
            if (p.relu) {
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = fmaxf(v[u], 0.f);
            }
            store16(reinterpret_cast<OT*>(C + (size_t)m * p.ldc + n), v);
        } else {
            for (int u = 0; u < 8 && n + u < p.N; ++u) {
                float x = v[u];
                if (p.res) x += Op16<OT>::to_f32(p.res[(size_t)m * p.ldres + n + u]);
                if (p.relu) x = fmaxf(x, 0.f);
                C[(size_t)m * p.ldc + n + u] = Op16<OT>::from_f32(x);
            }
        }
    }
}

template <typename OT, bool CONV>
static void launch_gemm_bf16(GemmBf16Params& p, hipStream_t s) {
    {   // measured on the ResNet-50 layers at 256 images: 4.63 -> 4.45 ms over all bottleneck convolutions
        const double w_bytes = 2.0 * p.N * p.K, a_bytes = 2.0 * p.M * (CONV ? p.Cin : p.K);
        p.n_fast = w_bytes <= 8.0 * 1024 * 1024 && a_bytes > w_bytes;
    }
    // (K <= 128 layers: 64 x 64, 128 x 64 and 64 x 128 tiles were all slower than 128 x 128: 245 / 204 / 205 vs 175 us;
    //  the persistent kernel with a register-only bf16 epilogue (8-byte stores / residual loads in accumulator layout):
    //  K = 64: 166 vs 176 us, K = 128 + residual: 156 vs 127 us, K = 256: 77 vs 64 us -- the LDS-staged 16-byte rows win)
    const long long big_tiles = (long long)dh_cdiv(p.M, 128) * dh_cdiv(p.N, 128);
    const bool lnx = !CONV && (p.a_stats || p.r_stats || p.o_stats);
    // >= 192 big tiles (measured: lowering the threshold to 128/100/40 tiles does not help gates / ffn / qkv / proj)
    if (big_tiles >= 192 && p.M >= 96 && p.N >= 96 && !lnx) {
        p.tiles_m = dh_cdiv(p.M, 128); p.tiles_n = dh_cdiv(p.N, 128);
        // 8 waves (4 x 2, each 32 x 64) on the 128 x 128 tile, 2 workgroups per CU = 4 waves per SIMD: measured
        // 510 TF vs 450 TF with 4 waves per workgroup and 300 TF with one 4-wave workgroup and a deeper ring --
        // the MFMA pipe needs co-resident waves to cover each wave's LDS-read/barrier gaps
        // (measured again in round 2 with the LDS ring really asynchronous: 3- / 4-slab rings at one workgroup per CU
        //  3.6 vs 2.8 ms over the 1x1 convolutions; 256 x 128 tiles with a 3-slab ring 1.80 vs 1.54 ms over the 3x3 ones --
        //  this one-barrier-per-slab loop needs the second co-resident workgroup)
        hipLaunchKernelGGL((gemm_bf16_kernel<OT, 128, 128, 4, CONV, 2, 8>), dim3(p.tiles_m * p.tiles_n), dim3(512), 0, s, p);
        return;
    }
    if (p.N <= 64 && p.M >= 256 * 512 && !lnx) {
        // narrow outputs with very many rows (stage-1 convolutions): 3x3 -> 256 x 64 tiles on 8 waves (the 64 weight
        // rows are staged once per 256 pixels; measured 127 us vs 146 us with 4 waves, 135 us with 128 x 64 tiles)
        p.tiles_m = dh_cdiv(p.M, 256); p.tiles_n = dh_cdiv(p.N, 64);
        if (CONV) {
            hipLaunchKernelGGL((gemm_bf16_kernel<OT, 256, 64, 4, CONV, 2, 8>), dim3(p.tiles_m * p.tiles_n), dim3(512), 0, s, p);
        } else {    // dense 1x1 (HBM-bound, ~5 TB/s): smaller tiles, 3 workgroups per CU
            p.tiles_m = dh_cdiv(p.M, 128);
            hipLaunchKernelGGL((gemm_bf16_kernel<OT, 128, 64, 2, CONV, 2, 4>), dim3(p.tiles_m * p.tiles_n), dim3(256), 0, s, p);
        }
        return;
    }
    p.tiles_m = dh_cdiv(p.M, 64); p.tiles_n = dh_cdiv(p.N, 64);
    const int blocks = p.tiles_m * p.tiles_n;
    if (!CONV && (p.a_stats || p.r_stats || p.o_stats)) {       // deferred-LayerNorm forms: the 64 x 64 kernels, same ring choice
#define DH_LN_LAUNCH(E)                                                                                                             \
        if (blocks > 1280 || (blocks > 320 && blocks <= 512) || p.K <= 128)                                                         \
            hipLaunchKernelGGL((gemm_bf16_kernel<OT, 64, 64, 2, false, 4, 4, E>), dim3(blocks), dim3(256), 0, s, p);                \
        else if (blocks <= 320)                                                                                                     \
            hipLaunchKernelGGL((gemm_bf16_kernel<OT, 64, 64, 2, false, 8, 8, E>), dim3(blocks), dim3(512), 0, s, p);                \
        else if (blocks <= 768)                                                                                                     \
            hipLaunchKernelGGL((gemm_bf16_kernel<OT, 64, 64, 2, false, 3, 4, E>), dim3(blocks), dim3(256), 0, s, p);                \
        else                                                                                                                        \
            hipLaunchKernelGGL((gemm_bf16_kernel<OT, 64, 64, 2, false, 2, 4, E>), dim3(blocks), dim3(256), 0, s, p);
        if (p.a_stats) { DH_LN_LAUNCH(1) } else { DH_LN_LAUNCH(3) }
#undef DH_LN_LAUNCH
        return;
    }
    // ring depth (16 KB per slab) by workgroup count, so that all tiles are co-resident in ONE round where possible:
    // <= 320: one per CU with a deep ring (7 slabs in flight); <= 512: two per CU; <= 768: three; <= 1280: five
    // (the fused LSTM step, 640 workgroups: 20.7 us with 4 slabs / 1.25 rounds, 16.6 us with 3 slabs / one round)
    // (32 x 64 tiles for the 160-workgroup decoder projections: slower -- proj 5.5 -> 6.1 ms, ffn 6.0 -> 7.6 ms per C3 step)
    if (blocks > 1280 || (blocks > 320 && blocks <= 512) || p.K <= 128)
        hipLaunchKernelGGL((gemm_bf16_kernel<OT, 64, 64, 2, CONV, 4>), dim3(blocks), dim3(256), 0, s, p);
    else if (blocks <= 320)
        // 8 waves on the 64 x 64 tile (32 x 16 per wave): twice the waves issuing LDS-DMA for the 7 slabs in flight --
        // these launches are bound by how fast one workgroup per CU can pull its operands (proj 5.75 -> 5.5 ms per C3 step)
        hipLaunchKernelGGL((gemm_bf16_kernel<OT, 64, 64, 2, CONV, 8, 8>), dim3(blocks), dim3(512), 0, s, p);
    else if (blocks <= 768)
        hipLaunchKernelGGL((gemm_bf16_kernel<OT, 64, 64, 2, CONV, 3>), dim3(blocks), dim3(256), 0, s, p);
    else
        hipLaunchKernelGGL((gemm_bf16_kernel<OT, 64, 64, 2, CONV, 2>), dim3(blocks), dim3(256), 0, s, p);
}

template <typename OT> static bool launch_persistent_f32(const GemmBf16Params& p, hipStream_t s);   // defined below vocab_logits_kernel

// called from dh_linear (gemm.hip) for DH_BF16 / DH_BF16_OUT_F32
int dh_linear_bf16_impl(const void* A, int lda, const void* W, int ldw, const float* bias, const float* scale,
                        const float* shift, const void* residual, int ldres, void* C, int ldc, int M, int N, int K,
                        int relu, int out_f32, int f16, hipStream_t s) {
    DH_REQUIRE((K % 8) == 0 && (lda % 8) == 0 && (ldw % 8) == 0 && lda >= K && ldw >= K && ldc >= N);
    DH_REQUIRE(((uintptr_t)A % 16) == 0 && ((uintptr_t)W % 16) == 0);
    GemmBf16Params p{};
    p.A = (const uint16_t*)A; p.lda = lda; p.W = (const uint16_t*)W; p.ldw = ldw;
    p.bias = bias; p.scale = scale; p.shift = shift; p.res = (const uint16_t*)residual; p.ldres = ldres;
    p.C = C; p.ldc = ldc; p.M = M; p.N = N; p.K = K; p.relu = relu; p.out_f32 = out_f32;
    if (f16) {
        if (out_f32 && launch_persistent_f32<f16_t>(p, s)) DH_LAUNCH_CHECK();
        launch_gemm_bf16<f16_t, false>(p, s);
    } else {
        if (out_f32 && launch_persistent_f32<bf16_t>(p, s)) DH_LAUNCH_CHECK();
        launch_gemm_bf16<bf16_t, false>(p, s);
    }
    DH_LAUNCH_CHECK();
}


// (Measured in round 2 and not kept: an output-projection GEMM in which a wave owns <= 16 rows and loads them straight from
//  HBM / L2 into MFMA B fragments while the tile's 64 weight rows are staged once per workgroup in LDS -- the structure of
//  dh_attn_cross_qproj_decode's projection part, 256 workgroups at M = 1280.  Bit-identical results, but 4.4 vs 3.9 ms over the
//  projections of a C3 step and +1.6 ms on the step: fragment-shaped activation loads (20 live lanes x 16 B per instruction)
//  cost more than the LDS-DMA bytes they save.)

// dh_linear with deferred LayerNorm (include/deephumor_hip.h: dh_ln_fold_t)
extern "C" int dh_linear_ln(const void* A, int lda, const void* W, int ldw, const float* bias, const void* residual, int ldres,
                            void* C, int ldc, int M, int N, int K, int relu, const dh_ln_fold_t* ln, int dtype, void* stream) {
    if (!DH_IS_16BIT(dtype)) return DH_ERR_UNSUPPORTED;
    DH_REQUIRE(A && W && C && ln && M > 0 && N > 0 && K > 0 && (K % 8) == 0 && (lda % 8) == 0 && (ldw % 8) == 0 && lda >= K && ldw >= K);
    DH_REQUIRE(ldc >= N && (ldc % 8) == 0 && (N % 8) == 0 && ((uintptr_t)A % 16) == 0 && ((uintptr_t)W % 16) == 0 && ((uintptr_t)C % 16) == 0);
    DH_REQUIRE(!residual || (ldres >= N && (ldres % 8) == 0 && ((uintptr_t)residual % 16) == 0));
    DH_REQUIRE(!ln->a_stats || (ln->a_colsum && bias && ln->a_tiles >= 2 && ln->a_tiles <= 8 && (ln->a_tiles % 2) == 0 && ln->a_tiles * 64 == K &&
                                ((uintptr_t)ln->a_stats % 16) == 0 && ((uintptr_t)ln->a_colsum % 16) == 0 && ((uintptr_t)bias % 16) == 0 && (N % 4) == 0));
    DH_REQUIRE(!ln->r_stats || (residual && ln->r_gamma && ln->r_beta && ln->r_tiles >= 2 && ln->r_tiles <= 8 && (ln->r_tiles % 2) == 0 &&
                                ln->r_tiles * 64 == N && ((uintptr_t)ln->r_stats % 16) == 0 && ((uintptr_t)ln->r_gamma % 16) == 0 &&
                                ((uintptr_t)ln->r_beta % 16) == 0));
    DH_REQUIRE(!ln->o_stats || ((N % 64) == 0 && ((uintptr_t)ln->o_stats % 8) == 0));
    DH_REQUIRE(!(ln->a_stats && (ln->r_stats || ln->o_stats)));      // one kernel flavour per launch (the chain never needs both)
    GemmBf16Params p{};
    p.A = (const uint16_t*)A; p.lda = lda; p.W = (const uint16_t*)W; p.ldw = ldw; p.bias = bias;
    p.res = (const uint16_t*)residual; p.ldres = ldres; p.C = C; p.ldc = ldc; p.M = M; p.N = N; p.K = K; p.relu = relu;
    p.a_stats = (const float2*)ln->a_stats; p.a_nt = ln->a_tiles; p.a_eps = ln->a_eps; p.a_colsum = ln->a_colsum;
    p.r_stats = (const float2*)ln->r_stats; p.r_nt = ln->r_tiles; p.r_eps = ln->r_eps; p.r_gamma = ln->r_gamma; p.r_beta = ln->r_beta;
    p.o_stats = (float2*)ln->o_stats;
    dh_prof_set_dims(M, N, K);
    DhProfScope prof("dh_linear", 2.0 * M * N * K, 2.0 * ((double)M * K + (double)N * K + (double)M * N * (residual ? 2 : 1)), stream);
    DH_DISPATCH_16(dtype, launch_gemm_bf16<T, false>(p, (hipStream_t)stream));
    DH_LAUNCH_CHECK();
}

extern "C" int dh_conv2d_nhwc_bn_act(const void* x, const void* w, const float* scale, const float* shift,
                                     const void* residual, void* y, int N, int H, int W, int Cin, int Cout,
                                     int KS, int stride, int pad, int relu, int dtype, void* stream) {
    if (!DH_IS_16BIT(dtype)) return DH_ERR_UNSUPPORTED;
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
    {   // division by Cin / KS in the generic (not tap-uniform) loader as multiply-shift; verified exhaustively here
        auto magic = [](int d, int limit) -> unsigned {
            const unsigned mg = (unsigned)(((1u << 20) + d - 1) / d);
            if ((unsigned long long)limit * mg >= (1ull << 32)) return 0;
            for (int v = 0; v < limit; ++v)
                if ((int)(((unsigned)v * mg) >> 20) != v / d) return 0;
            return mg;
        };
        p.cin_magic = magic(Cin, p.K); p.ks_magic = magic(KS, KS * KS);
        DH_REQUIRE((Cin % 64) == 0 || (p.cin_magic && p.ks_magic));
    }
    hipStream_t s = (hipStream_t)stream;
    static const char* const tags[] = {"?", "1x1", "2x2", "3x3", "4x4", "5x5", "6x6", "7x7"};
    dh_prof_set_tag(tags[KS < 8 ? KS : 0]);
    dh_prof_set_dims(p.M, Cout, p.K);
    DhProfScope prof("dh_conv2d_nhwc_bn_act", 2.0 * p.M * Cout * p.K,
                     2.0 * ((double)N * H * W * Cin + (double)Cout * p.K + (double)p.M * Cout * (residual ? 2 : 1)), stream);
    DH_DISPATCH_16(dtype, {
        if (KS == 1 && stride == 1 && pad == 0) { p.lda = Cin; p.conv = 0; launch_gemm_bf16<T, false>(p, s); }
        else { p.conv = 1; launch_gemm_bf16<T, true>(p, s); }
    });
    DH_LAUNCH_CHECK();
}

// Convolution + BatchNorm + ReLU + MaxPool2d(3, 2, 1) as ONE launch (the ResNet stem: torchvision resnet.conv1 / bn1 / relu /
// maxpool, encoders.py:37-38).  x NHWC [N,H,W,Cin], w [Cout,KS,KS,Cin], y NHWC [N, Ho/2, Wo/2, Cout] with Ho = conv output
// height (even).  The bf16 rounding happens once, after the pooling (max and rounding commute: rounding is monotonic).
extern "C" int dh_conv2d_nhwc_bn_relu_maxpool(const void* x, const void* w, const float* scale, const float* shift, void* y, int N,
                                              int H, int W, int Cin, int Cout, int KS, int stride, int pad, int dtype, void* stream) {
    if (!DH_IS_16BIT(dtype)) return DH_ERR_UNSUPPORTED;
    DH_REQUIRE(x && w && scale && shift && y && N > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0 && (Cin % 8) == 0 && (Cout % 8) == 0 && Cout <= 64);
    DH_REQUIRE(KS >= 1 && stride >= 1 && pad >= 0 && ((uintptr_t)x % 16) == 0 && ((uintptr_t)w % 16) == 0 && ((uintptr_t)y % 16) == 0);
    GemmBf16Params p{};
    p.Ho = (H + 2 * pad - KS) / stride + 1; p.Wo = (W + 2 * pad - KS) / stride + 1;
    DH_REQUIRE(p.Ho >= 2 && p.Wo >= 2 && (p.Ho % 2) == 0 && (p.Wo % 2) == 0);
    const int Hp = p.Ho / 2, Wp = p.Wo / 2, pbh = (Hp + 6) / 7, pbw = (Wp + 6) / 7;
    DH_REQUIRE((long long)N * pbh * pbw * 256 < (1ll << 31));
    p.A = (const uint16_t*)x; p.W = (const uint16_t*)w; p.ldw = KS * KS * Cin; p.scale = scale; p.shift = shift;
    p.C = y; p.ldc = Cout; p.N = Cout; p.K = KS * KS * Cin; p.relu = 1;
    p.H = H; p.Wd = W; p.Cin = Cin; p.KS = KS; p.stride = stride; p.pad = pad; p.conv = 1;
    p.tiles_m = N * pbh * pbw; p.tiles_n = 1; p.M = p.tiles_m * 256; p.n_fast = 0;
    {
        auto magic = [](int d, int limit) -> unsigned {
            const unsigned mg = (unsigned)(((1u << 20) + d - 1) / d);
            if ((unsigned long long)limit * mg >= (1ull << 32)) return 0;
            for (int v = 0; v < limit; ++v)
                if ((int)(((unsigned)v * mg) >> 20) != v / d) return 0;
            return mg;
        };
        p.cin_magic = magic(Cin, p.K); p.ks_magic = magic(KS, KS * KS);
        DH_REQUIRE((Cin % 64) == 0 || (p.cin_magic && p.ks_magic));
    }
    dh_prof_set_tag("stem+pool");
    dh_prof_set_dims(N * p.Ho * p.Wo, Cout, p.K);
    DhProfScope prof("dh_conv2d_nhwc_bn_act", 2.0 * N * p.Ho * p.Wo * Cout * p.K,
                     2.0 * ((double)N * H * W * Cin + (double)Cout * p.K + (double)N * Hp * Wp * Cout), stream);
    DH_DISPATCH_16(dtype, hipLaunchKernelGGL((gemm_bf16_kernel<T, 256, 64, 4, true, 2, 8, 2>), dim3(p.tiles_m), dim3(512), 0, (hipStream_t)stream, p));
    DH_LAUNCH_CHECK();
}

// ---- persistent vocabulary-projection kernel ---------------------------------------------------------------
// The classifier GEMM is short in K (K = hidden size = 512: 8 slabs per 128 x 128 tile), so a one-tile-per-workgroup
// kernel spends about half of its time filling the ring and draining the epilogue.  Here a workgroup walks a
// sequence of tiles and the LDS ring never drains: the slab loads of tile i+1 are in flight during the last
// MFMAs and the epilogue of tile i.  The epilogue needs no LDS -- an accumulator quad is 4 consecutive logits of
// one row, stored straight from registers as 16-byte stores (a wave covers 256 contiguous bytes of 16 rows over
// its 4 column blocks) -- so there is no barrier between the last MFMA of a tile and the first of the next.
// vmcnt bookkeeping (MI355X_MICROARCH: loads, stores and LDS-DMA retire in issue order): the wait for slab g
// allows the stores of the previous tile's epilogue to stay outstanding when that tile was interior (then their
// number is fixed: TM*TN logits stores + TM group-max stores); after an edge tile it waits for everything.
// Stores the 16-row x 64-column fp32 block a wave holds in accumulator layout -- lane (l15, lq): v[j] = columns 16 j + 4 lq .. + 3
// of row l15 -- as FULL 128-byte lines.  Stored straight from that layout, one instruction covers 16 rows x 64 bytes: every line
// is written as two half-line requests by two instructions, which costs the classifier ~11 us of its ~30 us of store time
// (measured: same bytes, full lines).  Here lane pairs (l15, l15 ^ 1) swap one quad per 32-column half (DPP quad_perm, no LDS), so
// that an instruction covers 8 rows x 128 bytes: line h of the even row = [even lane's v[2h] | even lane's v[2h+1] via the odd
// lane], line h of the odd row = [odd lane's v[2h] via the even lane | odd lane's v[2h+1]].  4 stores, as before.
__device__ __forceinline__ float4 dh_dpp_swap1(float4 v) {
    float4 r;
    r.x = dpp_get<0xB1>(v.x); r.y = dpp_get<0xB1>(v.y); r.z = dpp_get<0xB1>(v.z); r.w = dpp_get<0xB1>(v.w);   // lane ^ 1
    return r;
}
// One 32-column half: va / vb = the lane's quads 2h and 2h + 1; r_even = the even row's address of this lane's slot.
__device__ __forceinline__ void store_half_full_lines(float* r_even, size_t ldc, float4 va, float4 vb, bool odd) {
    // (component-wise selects: a select between float4 values goes through scratch memory)
#define DH_SEL4(c, a, b) make_float4((c) ? (a).x : (b).x, (c) ? (a).y : (b).y, (c) ? (a).z : (b).z, (c) ? (a).w : (b).w)
    const float4 own = DH_SEL4(odd, vb, va);
    const float4 rcv = dh_dpp_swap1(DH_SEL4(odd, va, vb));
    *reinterpret_cast<float4*>(r_even) = DH_SEL4(odd, rcv, own);
    *reinterpret_cast<float4*>(r_even + ldc) = DH_SEL4(odd, own, rcv);
#undef DH_SEL4
}

struct VocabParams {
    const uint16_t* A; int lda;
    const uint16_t* W; int ldw;
    const float* bias;
    float* C; int ldc;
    float* gmax; int gmax_ld;
    int M, N, K, tiles_m, tiles_n;
    // LSE mode (teacher-forced scoring): no logits; per (row, 64-column group) max and sum of exp(logit - max), and the
    // logit of each row's target column
    float* gsum; const int64_t* targets; float* tgt_logit;
};

template <typename OT, int NS, int BM, int BN, int WAVES_M, int NW, bool LSE = false>
__global__ __launch_bounds__(64 * NW, (NS * (BM + BN) * 128 <= 72 * 1024 ? 2 : 1)) void vocab_logits_kernel(VocabParams p) {
    constexpr int BK = 64;
    constexpr int A_BYTES = BM * 128, SLAB = A_BYTES + BN * 128;
    constexpr int WM = BM / WAVES_M, WN = BN / (NW / WAVES_M);      // per-wave sub-tile; WN = 64 = one column group
    static_assert(WN == 64, "a wave owns one 64-column group (group_max, bias strip)");
    constexpr int TM = WM / 16, TN = WN / 16;
    constexpr int IA = BM / (8 * NW), IB = BN / (8 * NW), G = IA + IB;
    constexpr int N_STORE_MAX = TM * TN + TM;                      // stores of one interior-tile epilogue, per wave
    const int N_STORE = (p.C ? TM * TN : 0) + (p.gmax ? TM : 0);   // (group maxima are optional: plain fp32-output GEMM; so are the logits)
    static_assert((NS - 2) * G + N_STORE_MAX + 1 <= 63, "vmcnt encoding");
    __shared__ __attribute__((aligned(16))) unsigned char lds[NS * SLAB + NW * 256];
    unsigned char* bias_lds = lds + NS * SLAB + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6) * 256;   // wave-private

    // this workgroup's tiles: XCD x (= blockIdx % 8) owns one contiguous range of tile ids (tm fastest: the 10 M tiles
    // of a classifier-weight panel are neighbours in time on one L2); its workgroups interleave over that range
    const int ntiles = p.tiles_m * p.tiles_n;
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    const int nbx = ((int)gridDim.x - xcd + 7) >> 3;
    const int q = ntiles / 8, r = ntiles % 8;
    const int first = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    const int count = q + (xcd < r ? 1 : 0);
    const int my_tiles = idx < count ? (count - idx + nbx - 1) / nbx : 0;
    if (my_tiles == 0) return;
    const int nslab = (p.K + BK - 1) / BK, total = my_tiles * nslab;       // host guarantees nslab >= NS

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm0 = (wave % WAVES_M) * WM, wn0 = (wave / WAVES_M) * WN;
    const int lr = lane >> 3, lpos = lane & 7, swz = lpos ^ lr;   // loader: row in the 8-row group, source chunk
    const int l15 = lane & 15, lq = lane >> 4;
    const unsigned char* zero = reinterpret_cast<const unsigned char*>(dh_zero_page);

    // ---- loader state: the tile whose slabs are being staged (runs NS-1 slabs ahead of the MFMAs) ---------------
    // Per LDS-DMA piece one running source pointer and a per-lane step: rows outside M / V point at the zero page with
    // step 0, so the loop carries no bounds selects (K % 64 == 0 is required by the host: no K tail either).  The
    // SIMD's instruction issue, not the MFMA pipe, bounds this loop (PMC: SQ_ACTIVE_INST_ANY ~ all SIMD cycles), so
    // every VALU instruction removed from the staging path shows up in the kernel time.
    const uint16_t* a_ptr[IA]; const uint16_t* b_ptr[IB];
    int a_step[IA], b_step[IB];
    int ld_it = 0, ld_s = 0, ld_g = 0;
    auto set_load_tile = [&](int it) {
        const int tile = first + idx + it * nbx, tm = tile % p.tiles_m, tn = tile / p.tiles_m;
#pragma unroll
        for (int i = 0; i < IA; ++i) {
            const int m = tm * BM + (wave * IA + i) * 8 + lr;
            const bool ok = m < p.M;
            a_ptr[i] = ok ? p.A + (size_t)m * p.lda + swz * 8 : reinterpret_cast<const uint16_t*>(zero);
            a_step[i] = ok ? BK : 0;
        }
#pragma unroll
        for (int i = 0; i < IB; ++i) {
            const int n = tn * BN + (wave * IB + i) * 8 + lr;
            const bool ok = n < p.N;
            b_ptr[i] = ok ? p.W + (size_t)n * p.ldw + swz * 8 : reinterpret_cast<const uint16_t*>(zero);
            b_step[i] = ok ? BK : 0;
        }
    };
    auto stage_next = [&]() {
        unsigned char* slab = lds + __builtin_amdgcn_readfirstlane(ld_g % NS) * SLAB;
#pragma unroll
        for (int i = 0; i < IA; ++i) {
            dh_lds_dma16(a_ptr[i], slab + (wave * IA + i) * 1024);
            a_ptr[i] += a_step[i];
        }
#pragma unroll
        for (int i = 0; i < IB; ++i) {
            dh_lds_dma16(b_ptr[i], slab + A_BYTES + (wave * IB + i) * 1024);
            b_ptr[i] += b_step[i];
        }
        ++ld_g;
        if (++ld_s == nslab) { ld_s = 0; if (++ld_it < my_tiles) set_load_tile(ld_it); }
    };

    set_load_tile(0);
#pragma unroll
    for (int u = 0; u < NS - 1; ++u) stage_next();        // NS - 1 slabs ahead (total >= nslab >= NS)
    int g = 0;
    bool prev_full = false;
    for (int it = 0; it < my_tiles; ++it) {
        const int tile = first + idx + it * nbx, tm = tile % p.tiles_m, tn = tile / p.tiles_m;
        const int m0 = tm * BM, n0 = tn * BN;
        const bool full = m0 + BM <= p.M && n0 + BN <= p.N;           // wave-uniform
        f32x4 acc[TN][TM];
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int i = 0; i < TM; ++i) acc[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int t = 0; t < nslab; ++t, ++g) {
            // Operations of this wave issued AFTER the loads of slab g (staged NS-1 iterations ago), oldest first:
            //   the slabs g+1 .. g+NS-2, the previous tile's epilogue stores when that epilogue ran within the last
            //   NS-1 iterations (t <= NS-2; counted only if that tile was interior, i.e. their number is fixed), and
            //   this tile's bias LDS-DMA (issued at t == 0, so younger than slab g for 1 <= t <= NS-2).
            int allow = min(NS - 2, total - 1 - g) * G;
            if (t <= NS - 2 && it > 0 && prev_full) allow += N_STORE;
            if (t >= 1 && t <= NS - 2) allow += 1;
            wait_vmcnt_hot<(NS - 2) * G>(allow);          // steady state inside a tile: the constant; tile starts / the tail: the switch
            __builtin_amdgcn_s_barrier();                 // slab g complete for every wave; slab g-1 fully consumed
            const unsigned char* sa = lds + (g % NS) * SLAB;
            const unsigned char* sb = sa + A_BYTES;
            // all fragments of the slab first (12 ds_read_b128 back to back, one wait), then the 16 MFMAs back to back:
            // the other waves of the SIMD issue during the single LDS wait instead of during four short ones
            uint4 fa[2][TM], fw[2][TN];
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const int c = kk * 4 + lq;
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    const int rr = wm0 + i * 16 + l15;
                    fa[kk][i] = *reinterpret_cast<const uint4*>(sa + rr * 128 + ((c ^ (rr & 7)) << 4));
                }
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int rr = wn0 + j * 16 + l15;
                    fw[kk][j] = *reinterpret_cast<const uint4*>(sb + rr * 128 + ((c ^ (rr & 7)) << 4));
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            // the LDS-DMA issue (the most expensive instructions of the loop) goes into the LDS-read latency window
            if (t == 0) {
                // this wave's 64 bias values -> its private LDS strip (one 4-byte LDS-DMA per lane); older than the tile's
                // last slab, so the wait in front of that slab's MFMAs also covers it
                const int n = n0 + wn0 + lane;
                const void* src = (p.bias && n < p.N) ? (const void*)(p.bias + n) : (const void*)zero;
                dh_lds_dma4(src, bias_lds);
            }
            if (ld_g < total) stage_next();               // refill the buffer slab g-1 used
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int i = 0; i < TM; ++i)
                        acc[j][i] = Op16<OT>::mfma(fw[kk][j], fa[kk][i], acc[j][i]);
        }
        // ---- epilogue, registers only: acc[j][i][r] = logit[m0+wm0+16i+l15][n0+wn0+16j+4lq+r] -------------------
        float4 b4[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j) b4[j] = *reinterpret_cast<const float4*>(bias_lds + (16 * j + 4 * lq) * 4);
        if constexpr (LSE) {
            // log-sum-exp partials instead of the logits: group max, sum of exp(logit - group max), target logit
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int m = m0 + wm0 + 16 * i + l15;
                const int64_t tcol = m < p.M ? p.targets[m] - (int64_t)(n0 + wn0) : -1;
                float v[TN][4];
                float mxv = -INFINITY;
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const float bj[4] = {b4[j].x, b4[j].y, b4[j].z, b4[j].w};
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr) {
                        const int n = n0 + wn0 + 16 * j + 4 * lq + rr;
                        v[j][rr] = n < p.N ? acc[j][i][rr] + bj[rr] : -INFINITY;
                        mxv = fmaxf(mxv, v[j][rr]);
                        if (tcol == 16 * j + 4 * lq + rr) p.tgt_logit[m] = v[j][rr];
                    }
                }
                mxv = fmaxf(mxv, __shfl_xor(mxv, 16, 64));
                mxv = fmaxf(mxv, __shfl_xor(mxv, 32, 64));
                float se = 0.f;
                if (mxv > -INFINITY) {
#pragma unroll
                    for (int j = 0; j < TN; ++j)
#pragma unroll
                        for (int rr = 0; rr < 4; ++rr) se += expf(v[j][rr] - mxv);       // exp(-inf) = 0 for columns past V
                }
                se += __shfl_xor(se, 16, 64);
                se += __shfl_xor(se, 32, 64);
                if (lq == 0 && m < p.M) {
                    p.gmax[(size_t)m * p.gmax_ld + (n0 + wn0) / WN] = mxv;
                    p.gsum[(size_t)m * p.gmax_ld + (n0 + wn0) / WN] = se;
                }
            }
            prev_full = false;          // (the number of stores is data dependent here: the next wait takes all of them)
            continue;
        }
        if (full) {
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int m = m0 + wm0 + 16 * i + l15;
                float mxv = -INFINITY;
                float* r_even = p.C + (size_t)(m0 + wm0 + 16 * i + (l15 & ~1)) * p.ldc + n0 + wn0 + ((l15 & 1) ? 16 : 0) + 4 * lq;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    float4 va, vb;
                    va.x = acc[2 * h][i][0] + b4[2 * h].x; va.y = acc[2 * h][i][1] + b4[2 * h].y;
                    va.z = acc[2 * h][i][2] + b4[2 * h].z; va.w = acc[2 * h][i][3] + b4[2 * h].w;
                    vb.x = acc[2 * h + 1][i][0] + b4[2 * h + 1].x; vb.y = acc[2 * h + 1][i][1] + b4[2 * h + 1].y;
                    vb.z = acc[2 * h + 1][i][2] + b4[2 * h + 1].z; vb.w = acc[2 * h + 1][i][3] + b4[2 * h + 1].w;
                    mxv = fmaxf(fmaxf(mxv, fmaxf(fmaxf(va.x, va.y), fmaxf(va.z, va.w))), fmaxf(fmaxf(vb.x, vb.y), fmaxf(vb.z, vb.w)));
                    if (p.C) store_half_full_lines(r_even + 32 * h, p.ldc, va, vb, l15 & 1);
                }
                mxv = fmaxf(mxv, __shfl_xor(mxv, 16, 64));
                mxv = fmaxf(mxv, __shfl_xor(mxv, 32, 64));
                if (lq == 0 && p.gmax) p.gmax[(size_t)m * p.gmax_ld + (n0 + wn0) / WN] = mxv;
            }
        } else {
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int m = m0 + wm0 + 16 * i + l15;
                float mxv = -INFINITY;
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const float bj[4] = {b4[j].x, b4[j].y, b4[j].z, b4[j].w};
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr) {
                        const int n = n0 + wn0 + 16 * j + 4 * lq + rr;
                        if (n < p.N) {
                            const float v = acc[j][i][rr] + bj[rr];
                            mxv = fmaxf(mxv, v);
                            if (m < p.M && p.C) p.C[(size_t)m * p.ldc + n] = v;
                        }
                    }
                }
                mxv = fmaxf(mxv, __shfl_xor(mxv, 16, 64));
                mxv = fmaxf(mxv, __shfl_xor(mxv, 32, 64));
                if (lq == 0 && m < p.M && p.gmax) p.gmax[(size_t)m * p.gmax_ld + (n0 + wn0) / WN] = mxv;   // -inf for a group past V
            }
        }
        prev_full = full;
    }
}


// ---- 256 x 256 persistent classifier kernel ---------------------------------------------------------------------------
// Same contract as vocab_logits_kernel (fp32 logits + 64-column group maxima, or the LSE partials), four times the MFMA work
// per barrier.  8 waves as 2 (M) x 4 (N), 128 x 64 outputs per wave (32 accumulator quads = 128 VGPRs); two 64 KB LDS
// slabs (A 256 x 64 k | B 256 x 64 k) in a ring that never drains across tiles.  Per 64-deep K slab a wave issues ONE
// wait + barrier, then four groups of 16 MFMAs (two row tiles x four column tiles x two k halves); the B fragments of the slab
// stay in registers, the A fragments are double-buffered two row tiles at a time, and the eight LDS-DMA pieces of the NEXT slab
// are spread over the four groups -- LDS reads, DMA issue and MFMAs of different groups overlap inside the wave, the next
// slab's transfer overlaps the whole slab.  Rows past M / V are clamped to the last row (finite garbage in never-stored
// outputs), so the loader carries no zero page, no selects and only 32-bit lane offsets from wave-uniform bases.
// vmcnt bookkeeping as in vocab_logits_kernel: loads, stores and LDS-DMA retire in issue order, the wait in front of a tile's
// first slab leaves the previous (interior) tile's epilogue stores outstanding.
__device__ __forceinline__ void dh_lds_dma16_s(const void* base, unsigned off, void* lds_dst) {
    const unsigned m = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(dh_lptr_t)lds_dst);
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" :: "s"(m), "v"(off), "s"(base) : "memory");
}

template <typename OT, bool LSE = false>
__global__ __launch_bounds__(512, 1) void vocab256_kernel(VocabParams p) {
    constexpr int BM = 256, BN = 256, BK = 64, NW = 8;
    constexpr int A_BYTES = BM * 128, SLAB = A_BYTES + BN * 128;          // 64 KB
    constexpr int TM = 8, TN = 4, G = 8;                                  // LDS-DMA pieces per wave per slab: 4 A + 4 B
    constexpr int N_STORE_MAX = TM * TN + TM;
    const int N_STORE = (p.C ? TM * TN : 0) + (p.gmax ? TM : 0);
    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * SLAB + NW * 256];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    unsigned char* bias_lds = lds + 2 * SLAB + wave * 256;
    const int wr = wave & 1, wc = wave >> 1;                              // 2 (M) x 4 (N)
    const int wm0 = wr * 128, wn0 = wc * 64;
    const int lr = lane >> 3, lpos = lane & 7, swz = lpos ^ lr;
    const int l15 = lane & 15, lq = lane >> 4;

    const int ntiles = p.tiles_m * p.tiles_n;
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    const int nbx = ((int)gridDim.x - xcd + 7) >> 3;
    const int q = ntiles / 8, r = ntiles % 8;
    const int first = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    const int count = q + (xcd < r ? 1 : 0);
    const int my_tiles = idx < count ? (count - idx + nbx - 1) / nbx : 0;
    if (my_tiles == 0) return;
    const int nslab = p.K / BK, total = my_tiles * nslab;                 // host: K % 64 == 0, K >= 128

    // ---- loader state (runs one slab ahead of the MFMAs) ---------------------------------------------------------------
    unsigned a_off[4], b_off[4];
    const unsigned char* a_base = reinterpret_cast<const unsigned char*>(p.A);
    const unsigned char* b_base = reinterpret_cast<const unsigned char*>(p.W);
    int ld_it = 0, ld_s = 0, ld_g = 0;
    auto set_load_tile = [&](int it) {
        const int tile = first + idx + it * nbx, tm = tile % p.tiles_m, tn = tile / p.tiles_m;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = (wave * 4 + i) * 8 + lr;
            a_off[i] = (unsigned)min(tm * BM + row, p.M - 1) * (unsigned)(p.lda * 2) + swz * 16;
            b_off[i] = (unsigned)min(tn * BN + row, p.N - 1) * (unsigned)(p.ldw * 2) + swz * 16;
        }
    };
    // pieces j = 0..3: A rows (wave*4 + j)*8 ..; j = 4..7: B rows likewise.  Two pieces per MFMA group.
    auto stage_pair = [&](int pair) {
        unsigned char* slab = lds + (ld_g & 1) * SLAB;
        const unsigned kb = (unsigned)ld_s * 128u;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int j = pair * 2 + u;
            if (j < 4) dh_lds_dma16_s(a_base + kb, a_off[j], slab + (wave * 4 + j) * 1024);
            else dh_lds_dma16_s(b_base + kb, b_off[j - 4], slab + A_BYTES + (wave * 4 + j - 4) * 1024);
        }
    };
    auto stage_done = [&]() {
        ++ld_g;
        if (++ld_s == nslab) { ld_s = 0; if (++ld_it < my_tiles) set_load_tile(ld_it); }
    };
    set_load_tile(0);
#pragma unroll
    for (int pr = 0; pr < 4; ++pr) stage_pair(pr);
    stage_done();

    int g = 0;
    bool prev_full = false;
    for (int it = 0; it < my_tiles; ++it) {
        const int tile = first + idx + it * nbx, tm = tile % p.tiles_m, tn = tile / p.tiles_m;
        const int m0 = tm * BM, n0 = tn * BN;
        const bool full = m0 + BM <= p.M && n0 + BN <= p.N;
        dh_f32x4 acc[TN][TM];
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int i = 0; i < TM; ++i) acc[j][i] = dh_f32x4{0.f, 0.f, 0.f, 0.f};
        for (int t = 0; t < nslab; ++t, ++g) {
            // everything of slab g landed (it was issued one slab ago); younger: the previous tile's epilogue stores (t == 0)
            // -- the bias strip of this tile is issued below, after this wait
            int allow = 0;
            if (t == 0 && it > 0 && prev_full) allow = N_STORE;
            wait_vmcnt_hot<0>(allow);
            __builtin_amdgcn_s_barrier();
            const unsigned char* sa = lds + (g & 1) * SLAB;
            const unsigned char* sb = sa + A_BYTES;
            const bool more = ld_g < total;
            uint4 fw[2][TN], fa[2][2][2];                          // B: [kk][col tile]; A: [buffer][row tile of the pair][kk]
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int rr = wn0 + j * 16 + l15;
                    fw[kk][j] = *reinterpret_cast<const uint4*>(sb + rr * 128 + (((kk * 4 + lq) ^ (rr & 7)) << 4));
                }
            auto read_a = [&](int buf, int c) {
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int kk = 0; kk < 2; ++kk) {
                        const int rr = wm0 + (2 * c + u) * 16 + l15;
                        fa[buf][u][kk] = *reinterpret_cast<const uint4*>(sa + rr * 128 + (((kk * 4 + lq) ^ (rr & 7)) << 4));
                    }
            };
            read_a(0, 0);
            if (t == 0) {       // this wave's 64 bias values -> its LDS strip; older than this tile's later slabs, read in the epilogue
                const int n = n0 + wn0 + lane;
                dh_lds_dma4(p.bias ? p.bias + (n < p.N ? n : 0) : reinterpret_cast<const float*>(dh_zero_page), bias_lds);   // bias == NULL: the zero page, never address 0
            }
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                if (c < 3) read_a((c + 1) & 1, c + 1);             // next pair of row tiles while this pair's MFMAs run
                if (more) stage_pair(c);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                    for (int u = 0; u < 2; ++u)
#pragma unroll
                        for (int j = 0; j < TN; ++j)
                            acc[j][2 * c + u] = Op16<OT>::mfma(fw[kk][j], fa[c & 1][u][kk], acc[j][2 * c + u]);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (more) stage_done();
        }
        // ---- epilogue from registers: acc[j][i][r] = logit[m0 + wm0 + 16 i + l15][n0 + wn0 + 16 j + 4 lq + r] -------------
        float4 b4[TN];
        if (p.bias) {
            asm volatile("s_waitcnt vmcnt(%0)" :: "n"(G) : "memory");      // the bias strip is older than the last slab's 8 pieces
#pragma unroll
            for (int j = 0; j < TN; ++j) b4[j] = *reinterpret_cast<const float4*>(bias_lds + (16 * j + 4 * lq) * 4);
        } else {
#pragma unroll
            for (int j = 0; j < TN; ++j) b4[j] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        if constexpr (LSE) {
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int m = m0 + wm0 + 16 * i + l15;
                const int64_t tcol = m < p.M ? p.targets[m] - (int64_t)(n0 + wn0) : -1;
                float v[TN][4];
                float mxv = -INFINITY;
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const float bj[4] = {b4[j].x, b4[j].y, b4[j].z, b4[j].w};
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr) {
                        const int n = n0 + wn0 + 16 * j + 4 * lq + rr;
                        v[j][rr] = n < p.N ? acc[j][i][rr] + bj[rr] : -INFINITY;
                        mxv = fmaxf(mxv, v[j][rr]);
                        if (tcol == 16 * j + 4 * lq + rr) p.tgt_logit[m] = v[j][rr];
                    }
                }
                mxv = fmaxf(mxv, __shfl_xor(mxv, 16, 64));
                mxv = fmaxf(mxv, __shfl_xor(mxv, 32, 64));
                float se = 0.f;
                if (mxv > -INFINITY) {
#pragma unroll
                    for (int j = 0; j < TN; ++j)
#pragma unroll
                        for (int rr = 0; rr < 4; ++rr) se += expf(v[j][rr] - mxv);
                }
                se += __shfl_xor(se, 16, 64);
                se += __shfl_xor(se, 32, 64);
                // a 256-column tile spans four 64-column groups, the row holds 2 * ceil(V / 128) of them: when the last tile's second
                // half lies past V its groups do not exist (written unguarded they landed in the NEXT row's first slots -- wrong
                // scores for V % 256 in 1..128 and for V < 128; found by tools/fuzz_scoring.py)
                if (lq == 0 && m < p.M && (n0 + wn0) / 64 < p.gmax_ld) {
                    p.gmax[(size_t)m * p.gmax_ld + (n0 + wn0) / 64] = mxv;
                    p.gsum[(size_t)m * p.gmax_ld + (n0 + wn0) / 64] = se;
                }
            }
            prev_full = false;
            continue;
        }
        if (full) {
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int m = m0 + wm0 + 16 * i + l15;
                float mxv = -INFINITY;
                float* r_even = p.C + (size_t)(m0 + wm0 + 16 * i + (l15 & ~1)) * p.ldc + n0 + wn0 + ((l15 & 1) ? 16 : 0) + 4 * lq;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    float4 va, vb;
                    va.x = acc[2 * h][i][0] + b4[2 * h].x; va.y = acc[2 * h][i][1] + b4[2 * h].y;
                    va.z = acc[2 * h][i][2] + b4[2 * h].z; va.w = acc[2 * h][i][3] + b4[2 * h].w;
                    vb.x = acc[2 * h + 1][i][0] + b4[2 * h + 1].x; vb.y = acc[2 * h + 1][i][1] + b4[2 * h + 1].y;
                    vb.z = acc[2 * h + 1][i][2] + b4[2 * h + 1].z; vb.w = acc[2 * h + 1][i][3] + b4[2 * h + 1].w;
                    mxv = fmaxf(fmaxf(mxv, fmaxf(fmaxf(va.x, va.y), fmaxf(va.z, va.w))), fmaxf(fmaxf(vb.x, vb.y), fmaxf(vb.z, vb.w)));
                    if (p.C) store_half_full_lines(r_even + 32 * h, p.ldc, va, vb, l15 & 1);
                }
                mxv = fmaxf(mxv, __shfl_xor(mxv, 16, 64));
                mxv = fmaxf(mxv, __shfl_xor(mxv, 32, 64));
                if (lq == 0 && p.gmax) p.gmax[(size_t)m * p.gmax_ld + (n0 + wn0) / 64] = mxv;
            }
        } else {
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int m = m0 + wm0 + 16 * i + l15;
                float mxv = -INFINITY;
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const float bj[4] = {b4[j].x, b4[j].y, b4[j].z, b4[j].w};
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr) {
                        const int n = n0 + wn0 + 16 * j + 4 * lq + rr;
                        if (n < p.N) {
                            const float v = acc[j][i][rr] + bj[rr];
                            mxv = fmaxf(mxv, v);
                            if (m < p.M && p.C) p.C[(size_t)m * p.ldc + n] = v;
                        }
                    }
                }
                mxv = fmaxf(mxv, __shfl_xor(mxv, 16, 64));
                mxv = fmaxf(mxv, __shfl_xor(mxv, 32, 64));
                if (lq == 0 && m < p.M && p.gmax && n0 + wn0 < ((p.N + 63) / 64) * 64 + 64) {
                    const int gidx = (n0 + wn0) / 64;
                    if (gidx < p.gmax_ld) p.gmax[(size_t)m * p.gmax_ld + gidx] = mxv;       // -inf for a group past V
                }
            }
        }
        prev_full = full;
    }
}

#include "vocab_areg.h"

// fp32-output dense GEMMs with many tiles (teacher-forced classifier: [bs*T, V] logits): the persistent kernel without
// the group maxima (600 vs 385 TF for the one-tile-per-workgroup kernel with its LDS-staged fp32 epilogue).
template <typename OT>
static bool launch_persistent_f32(const GemmBf16Params& p, hipStream_t s) {
    if (p.scale || p.res || p.relu || p.gmax || (p.K & 63) || p.K < 128) return false;
    if ((p.ldc & 3) || ((uintptr_t)p.C & 15)) return false;
    const int tiles_m = dh_cdiv(p.M, 128), tiles_n = dh_cdiv(p.N, 128);
    if ((long long)tiles_m * tiles_n < 1024) return false;
    VocabParams v{};
    v.A = p.A; v.lda = p.lda; v.W = p.W; v.ldw = p.ldw; v.bias = p.bias; v.C = (float*)p.C; v.ldc = p.ldc;
    v.M = p.M; v.N = p.N; v.K = p.K; v.tiles_m = tiles_m; v.tiles_n = tiles_n;
    hipLaunchKernelGGL((vocab_logits_kernel<OT, 2, 128, 128, 4, 8>), dim3(512), dim3(512), 0, s, v);
    return true;
}

// A ResNet stage's first bottleneck ends in relu(bn3(conv3(y)) + bn_d(downsample(x))): two 1x1 convolutions into the same
// output.  With the BatchNorm scales folded into the (bf16) weights -- w = [W3 * s3 | Wd * sd] over K = C1 + C2 -- this
// is ONE GEMM over the virtual operand [y | x at the strided pixels]: the 411 MB (stage 1) identity tensor is neither
// written by a downsample launch nor read back as a residual.  y: NHWC [N, Ho, Wo, C1]; x: NHWC [N, H, W, C2].
extern "C" int dh_conv1x1_dual_nhwc(const void* y, const void* x, const void* w, const float* shift, void* out, int N, int Ho,
                                    int Wo, int C1, int H, int W, int C2, int stride, int Cout, int relu, int dtype,
                                    void* stream) {
    if (!DH_IS_16BIT(dtype)) return DH_ERR_UNSUPPORTED;
    DH_REQUIRE(y && x && w && shift && out && N > 0 && Ho > 0 && Wo > 0 && H > 0 && W > 0 && Cout > 0 && stride >= 1);
    DH_REQUIRE((C1 % 64) == 0 && (C2 % 64) == 0 && C1 > 0 && C2 > 0 && (Ho - 1) * stride < H && (Wo - 1) * stride < W);
    DH_REQUIRE(((uintptr_t)y % 16) == 0 && ((uintptr_t)x % 16) == 0 && ((uintptr_t)w % 16) == 0 && (long long)N * Ho * Wo < (1ll << 31));
    GemmBf16Params p{};
    p.A = (const uint16_t*)y; p.lda = C1; p.W = (const uint16_t*)w; p.ldw = C1 + C2; p.bias = shift;
    p.C = out; p.ldc = Cout; p.M = N * Ho * Wo; p.N = Cout; p.K = C1 + C2; p.relu = relu;
    p.A2 = (const uint16_t*)x; p.K1 = C1; p.a2_H = H; p.a2_W = W; p.a2_C = C2; p.a2_stride = stride; p.a2_Ho = Ho; p.a2_Wo = Wo;
    dh_prof_set_tag("1x1");
    dh_prof_set_dims(p.M, Cout, p.K);
    DhProfScope prof("dh_conv2d_nhwc_bn_act", 2.0 * p.M * Cout * p.K,
                     2.0 * ((double)p.M * C1 + (double)p.M * C2 + (double)Cout * p.K + (double)p.M * Cout), stream);
    DH_DISPATCH_16(dtype, launch_gemm_bf16<T, false>(p, (hipStream_t)stream));
    DH_LAUNCH_CHECK();
}

// Vocabulary projection for beam search: logits[M,V] fp32 = A[M,K] * W[V,K]^T + bias, plus group_max[m, g] =
// max of logits[m, 64g .. 64g+63] (always the 128x128 tile: its waves own 64-column groups).
extern "C" int dh_vocab_logits(const void* A, int lda, const void* W, int ldw, const float* bias, float* logits, int ldl,
                               float* group_max, int gm_ld, int M, int V, int K, int dtype, void* stream) {
    if (!DH_IS_16BIT(dtype)) return DH_ERR_UNSUPPORTED;
    DH_REQUIRE(A && W && group_max && M > 0 && V > 0 && K > 0 && (!logits || ldl >= V) && gm_ld >= 2 * dh_cdiv(V, 128));
    DH_REQUIRE(logits || (K >= 128 && (K % 64) == 0));                  // group maxima only: the persistent kernels
    DH_REQUIRE((K % 8) == 0 && (lda % 8) == 0 && (ldw % 8) == 0 && lda >= K && ldw >= K);
    DH_REQUIRE(((uintptr_t)A % 16) == 0 && ((uintptr_t)W % 16) == 0);
    dh_prof_set_tag("vocab");
    dh_prof_set_dims(M, V, K);
    DhProfScope prof("dh_linear", 2.0 * M * V * K, 2.0 * ((double)M * K + (double)V * K) + 4.0 * M * V, stream);
    if ((!logits || ((ldl % 4) == 0 && ((uintptr_t)logits % 16) == 0)) && K >= 128 && (K % 64) == 0) {
        VocabParams v{};
        v.A = (const uint16_t*)A; v.lda = lda; v.W = (const uint16_t*)W; v.ldw = ldw; v.bias = bias;
        v.C = logits; v.ldc = ldl; v.gmax = group_max; v.gmax_ld = gm_ld; v.M = M; v.N = V; v.K = K;
        v.tiles_m = dh_cdiv(M, 128); v.tiles_n = dh_cdiv(V, 128);
        const int ntiles = v.tiles_m * v.tiles_n;
        // measured at M = 1280, V = 36,541, K = 512 (us per launch): 2-slab ring, two 8-wave workgroups per CU 84;
        // 3- / 4-slab ring with one workgroup per CU 107 / 105; 256 x 128 tiles on 16 waves 85 / 84 (2 / 3 slabs);
        // the one-tile-per-workgroup kernel below 93.  Without the 187 MB of logits stores the kernel takes 67 us.
        // (256 x 128 tiles with 64 x 64 per wave: 83 / 89 us with a 2- / 3-slab ring; 256 x 256 with 128 x 64 per wave: 90 us --
        //  at two waves per SIMD this simple schedule does not fill the matrix pipe; 128 x 128 below: 78 us)
        // (a two-group variant -- the workgroup's wave halves run the loop one barrier apart so that one half's MFMAs
        //  overlap the other half's LDS reads / LDS-DMA issue; 3-slab ring, 16 waves on 256 x 128 -- measured 75.6 us
        //  against 78 us standalone and no difference in the full step, so it is not kept)
        // (also tried, bit-exact but slower than this kernel's 78 us: 256 x 256 tiles, 128 x 64 per wave, K slabs of 32 in a
        //  4-slab ring, the two wave halves one barrier apart -- 85 us with LDS-DMA, 170 us register-staged; with 64-byte
        //  row segments every 128-byte line is fetched twice.  scratch/dma_probe shows the LDS-DMA path itself sustains
        //  110-125 GB/s per CU against the ~46 GB/s this kernel draws: DESIGN.md section 9.)
        // The 256 x 256 kernel (vocab256_kernel): with the 187 MB of fp32 logits to store both tile kernels take the same 86-89 us per
        // launch (store-bound), so logits go through the 128 x 128 one; where nothing is stored (group maxima only) the bigger tile is used.
        {   // A-stationary kernel: K = 512, row tiles of 128, groups of tiles_m workgroups per XCD (32 CUs each)
            // default since round 2: in the C2 / C3 steps 3-5 % faster than the 128 x 128 kernel below (2.43 vs 2.49 ms and 2.81 vs 2.99 ms
            // of classifier time per step, three alternating runs in one call); DH_VOCAB_AREG=0 restores the tile kernel
            const int areg = dh_opt(DH_OPT_VOCAB_AREG);
            const int tm128 = dh_cdiv(M, 128), tn128 = dh_cdiv(V, 128);
            // round 3: 256-row tiles (32 MFMAs per wave per slab instead of 16) when the rows fill them (DH_VOCAB_AREG=128: never)
            const int tm256 = dh_cdiv(M, 256);
            if (areg && areg != 128 && logits && K == 512 && M >= 512 && (M % 256) == 0 && ldl >= tn128 * 128) {
                // a launch takes `t` row tiles with (32 / t) * t >= 28 of an XCD's 32 CUs busy (t = 1-8, 10, 14-16, 28-32) and
                // 8 * (32 / t) column groups; more rows than that (batches of 1,024+ images on one GPU) go in several launches --
                // the rows of a launch only share W, which L2 serves (batch 1,024: 0.24 -> of the tile kernel otherwise)
                auto fits = [&](int t) { return t >= 1 && t <= 32 && (32 / t) * t >= 28 && tn128 >= 8 * (32 / t); };
                int plan[64], np = 0, left = tm256;
                while (left > 0 && np < 64) {
                    int t = left < 32 ? left : 32;
                    while (t > 0 && !(fits(t) && (left - t == 0 || left - t >= 2))) --t;   // (never leave a single 256-row tile: M >= 512)
                    if (t == 0) break;
                    plan[np++] = t; left -= t;
                }
                if (left == 0) {
                    int row0 = 0;
                    for (int i = 0; i < np; ++i) {
                        VocabParams c = v;
                        c.A = v.A + (size_t)row0 * lda; c.C = v.C + (size_t)row0 * ldl; c.gmax = v.gmax + (size_t)row0 * gm_ld;
                        c.M = plan[i] * 256; c.tiles_m = plan[i]; c.tiles_n = tn128;
                        DH_DISPATCH_16(dtype, hipLaunchKernelGGL((vocab_areg256_kernel<T>), dim3(256), dim3(512), 0, (hipStream_t)stream, c));
                        row0 += plan[i] * 256;
                    }
                    DH_LAUNCH_CHECK();
                }
            }
            if (areg && logits && K == 512 && tm128 <= 32 && (32 / tm128) * tm128 >= 28 && tn128 >= 8 * (32 / tm128)) {
                v.tiles_m = tm128; v.tiles_n = tn128;
                DH_DISPATCH_16(dtype, hipLaunchKernelGGL((vocab_areg_kernel<T>), dim3(256), dim3(512), 0, (hipStream_t)stream, v));
                DH_LAUNCH_CHECK();
            }
        }
        if (!logits && M >= 512) {
            v.tiles_m = dh_cdiv(M, 256); v.tiles_n = dh_cdiv(V, 256);
            const int nt = v.tiles_m * v.tiles_n;
            DH_DISPATCH_16(dtype, hipLaunchKernelGGL((vocab256_kernel<T>), dim3(nt < 256 ? nt : 256), dim3(512), 0, (hipStream_t)stream, v));
        } else {
            DH_DISPATCH_16(dtype, hipLaunchKernelGGL((vocab_logits_kernel<T, 2, 128, 128, 4, 8>), dim3(ntiles < 512 ? ntiles : 512), dim3(512), 0,
                                                     (hipStream_t)stream, v));
        }
        DH_LAUNCH_CHECK();
    }
    GemmBf16Params p{};
    p.A = (const uint16_t*)A; p.lda = lda; p.W = (const uint16_t*)W; p.ldw = ldw; p.bias = bias;
    p.C = logits; p.ldc = ldl; p.M = M; p.N = V; p.K = K; p.out_f32 = 1; p.gmax = group_max; p.gmax_ld = gm_ld;
    p.tiles_m = dh_cdiv(M, 128); p.tiles_n = dh_cdiv(V, 128);
    DH_DISPATCH_16(dtype, hipLaunchKernelGGL((gemm_bf16_kernel<T, 128, 128, 4, false, 2, 8>), dim3(p.tiles_m * p.tiles_n), dim3(512), 0,
                                             (hipStream_t)stream, p));
    DH_LAUNCH_CHECK();
}



// ---- teacher-forced scoring without materialising the logits --------------------------------------------------------
// logp[m] = log_softmax(A[m,:] W^T + bias)[targets[m]]: the classifier GEMM leaves per (row, 64-column group) the maximum
// and the sum of exp(logit - maximum) plus the target's logit (36,541 fp32 logits per row never go to memory), and one
// wave per row folds the groups: M = max_g, S = sum_g gsum_g * exp(gmax_g - M), logp = target - M - log S.
__global__ __launch_bounds__(256) void lse_combine_kernel(const float* __restrict__ gmax, const float* __restrict__ gsum, int ld,
                                                           int n_groups, const float* __restrict__ tgt, const int64_t* __restrict__ targets,
                                                           int V, float* __restrict__ logp, int rows) {
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (r >= rows) return;
    float m = -INFINITY;
    for (int g = lane; g < n_groups; g += 64) m = fmaxf(m, gmax[(size_t)r * ld + g]);
    m = wave_max(m);
    float s = 0.f;
    for (int g = lane; g < n_groups; g += 64) {
        const float gm = gmax[(size_t)r * ld + g];
        if (gm > -INFINITY) s += gsum[(size_t)r * ld + g] * expf(gm - m);
    }
    s = wave_sum(s);
    if (lane == 0) {
        const int64_t t = targets[r];
        logp[r] = (t >= 0 && t < V) ? (tgt[r] - m) - logf(s) : 0.f;
    }
}

extern "C" int dh_vocab_logprob(const void* A, int lda, const void* W, int ldw, const float* bias, const int64_t* targets,
                                float* logp, float* group_max, float* group_sum, float* target_logit, int gm_ld, int M, int V,
                                int K, int dtype, void* stream) {
    if (!DH_IS_16BIT(dtype)) return DH_ERR_UNSUPPORTED;
    DH_REQUIRE(A && W && targets && logp && group_max && group_sum && target_logit && M > 0 && V > 0 && K > 0);
    DH_REQUIRE(gm_ld >= 2 * dh_cdiv(V, 128) && (K % 64) == 0 && K >= 128 && (lda % 8) == 0 && (ldw % 8) == 0 && lda >= K && ldw >= K);
    DH_REQUIRE(((uintptr_t)A % 16) == 0 && ((uintptr_t)W % 16) == 0);
    dh_prof_set_tag("logprob");
    dh_prof_set_dims(M, V, K);
    DhProfScope prof("dh_linear", 2.0 * M * V * K, 2.0 * ((double)M * K + (double)V * K), stream);
    VocabParams v{};
    v.A = (const uint16_t*)A; v.lda = lda; v.W = (const uint16_t*)W; v.ldw = ldw; v.bias = bias;
    v.gmax = group_max; v.gmax_ld = gm_ld; v.gsum = group_sum; v.targets = targets; v.tgt_logit = target_logit;
    v.M = M; v.N = V; v.K = K; v.tiles_m = dh_cdiv(M, 128); v.tiles_n = dh_cdiv(V, 128);
    const int ntiles = v.tiles_m * v.tiles_n;
    hipStream_t s = (hipStream_t)stream;
    // 256 x 256 tiles (vocab256_kernel) once there are enough rows to fill them: without the logits stores the classifier is
    // MFMA-bound and the bigger tile pays (teacher-forced scoring of 9,000 captions: 36.5 -> 29.1 ms per pass)
    if (M >= 512) {
        v.tiles_m = dh_cdiv(M, 256); v.tiles_n = dh_cdiv(V, 256);
        const int nt = v.tiles_m * v.tiles_n;
        DH_DISPATCH_16(dtype, hipLaunchKernelGGL((vocab256_kernel<T, true>), dim3(nt < 256 ? nt : 256), dim3(512), 0, s, v));
    } else {
        DH_DISPATCH_16(dtype, hipLaunchKernelGGL((vocab_logits_kernel<T, 2, 128, 128, 4, 8, true>), dim3(ntiles < 512 ? ntiles : 512), dim3(512), 0, s, v));
    }
    hipLaunchKernelGGL(lse_combine_kernel, dim3(dh_cdiv(M, 4)), dim3(256), 0, s, group_max, group_sum, gm_ld, 2 * dh_cdiv(V, 128),
                       target_logit, targets, V, logp, M);
    DH_LAUNCH_CHECK();
}
