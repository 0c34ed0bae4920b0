// dh_linear_ln for the DECODE shapes of the 16-bit Transformer chain with the WEIGHTS STATIONARY IN REGISTERS
// (transformers.py:97 fc_q/k/v, :127 fc_o, :162-163 fc_1 / fc_2, applied to the rows of ONE position: 1,280 rows at 256 images x beam 5).
//
// The tile kernel (gemm_bf16.hip, 64 x 64 tiles, both operands through an LDS ring) runs these launches at 0.03-0.09 of any roofline:
// 160 workgroups on 256 CUs for the N = 512 projections, one barrier + one counted wait per 64-k slab (32 of them at K = 2,048) with two
// MFMAs per wave in between.  Here -- the partition lstm_wreg.hip proved on the LSTM gate GEMM:
//   * ONE round of <= 256 workgroups: a workgroup owns BN = 16 x NW output columns (one 16-column MFMA tile per wave) x RL activation
//     rows; (N, M) = (2048, 1280) -> 16 x 16, (1536, 1280) -> 12 x 20, (512, 1280) -> 8 x 32 workgroups;
//   * a wave keeps its 16 weight rows x all K as MFMA fragments in registers, loaded straight from L2 out of the fragment-packed
//     weights (dh_pack_mfma_fragments: coalesced 1 KB loads) -- K = 512: 16 fragments; K = 2,048: 64 fragments (256 registers of the
//     512 a one-wave-per-SIMD workgroup owns);
//   * only the [RL x K] activation block crosses LDS (LDS-DMA, whole block resident: <= 160 KB), ONE wait + ONE barrier, then
//     TM x K / 32 MFMAs per wave with nothing but LDS fragment reads in between;
//   * workgroups are mapped onto the 8 XCDs as compact sub-grids (xn column groups x 8 / xn row groups), so an XCD's L2 fetches the
//     fewest distinct weight / activation blocks.
// Results are BIT-IDENTICAL to dh_linear_ln's tile kernels: the same MFMA operand contents and k order per output (32-k steps
// ascending), the same epilogue arithmetic (deferred-LayerNorm fold on the accumulators, LayerNorm of the residual rows, rounding,
// per-(row, 64-column tile) statistics of the rounded values with the same 8 x 8 summation tree) -- tests/test_bf16_gpu.py.
#include "common.h"
#include "prof.h"

namespace {
struct LwParams {
    const uint16_t* A; int lda;
    const uint4* wp;                                   // fragment-packed weights [K / 32][N / 16][64] x 16 bytes
    const float* bias;
    const uint16_t* res; int ldres;
    uint16_t* C; int ldc;
    int M, N, relu;
    int tiles_m, tiles_n, xn;                          // xn: XCD column groups (0 = linear block order)
    const float2* a_stats; int a_nt; float a_eps; const float* a_colsum;
    const float2* r_stats; int r_nt; float r_eps; const float* r_gamma; const float* r_beta;
    float2* o_stats;
};

// global -> LDS, 16 bytes per lane: wave-uniform base (SGPR pair) + per-lane byte offset
__device__ __forceinline__ void lw_dma16(const void* base, unsigned off, void* lds_dst) {
    const unsigned m = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(dh_lptr_t)lds_dst);
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" :: "s"(m), "v"(off), "s"(base) : "memory");
}

// NW waves (16 output columns each), RL activation rows in LDS, KQ = K / 512, LNX: 0 = (deferred LayerNorm on the A rows |
// plain) + optional ReLU; 1 = residual (optionally pre-LayerNorm) + statistics of the output rows
// One output block (column block cb, row block rb) of the GEMM (`lds`: NSLAB * SLABB bytes; the block's threads must all call it)
template <typename OT, int NW, int RL, int KQ, int LNX>
__device__ __forceinline__ void lw_item(const LwParams& p, const int cb, const int rb, unsigned char* lds) {
    constexpr int NT = 64 * NW, BN = 16 * NW, TM = (RL + 15) / 16, RG = RL / 8, NSLAB = 8 * KQ, SLABB = RL * 128;
    constexpr int KF = 16 * KQ;                         // 32-k fragments per wave
    constexpr int WIN_SLABS = 65536 / SLABB, WIN = WIN_SLABS * SLABB, NWIN = (NSLAB + WIN_SLABS - 1) / WIN_SLABS;
    constexpr int PF = 3;                              // LDS fragment reads this many MFMAs ahead
    constexpr int CHUNKS = BN / 8, SLOTS = BN / 4, EP_IT = (RL * CHUNKS + NT - 1) / NT;
    static_assert(RL % 8 == 0 && RL <= NT && NSLAB * SLABB <= 163840 && RL * BN * 4 + RL * 8 <= NSLAB * SLABB, "LDS budget");
    static_assert((NSLAB * RG) % NW == 0, "pieces per wave");
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, lq = lane >> 4, lr = lane >> 3, lpos = lane & 7;
    const int m0 = rb * RL, n0 = cb * BN;

    // ---- epilogue operands: requested BEFORE the LDS-DMA transfers (ordinary loads the compiler counts; vmcnt retires in order) -------
    const float4 b4 = *reinterpret_cast<const float4*>(p.bias + n0 + 16 * wave + 4 * lq);
    // deferred LayerNorm of the A rows: thread r < RL fetches the statistics partials of block row r (64 contiguous bytes per row:
    // 4 coalesced instructions per wave) and leaves (mean, rstd) in LDS for the lanes whose accumulators hold that row -- fetched per
    // accumulator lane instead, they are 4 TM scattered loads per wave, as many vector-memory instructions as the operands themselves
    float4 a_raw[4];
    float4 cs4 = make_float4(0.f, 0.f, 0.f, 0.f);
    const bool a_ln = LNX == 0 && p.a_stats != nullptr;
    if (LNX == 0) {
        if (a_ln) {
            if (tid < RL) ln_load(p.a_stats + (size_t)min(m0 + tid, p.M - 1) * p.a_nt, p.a_nt, a_raw);
            cs4 = *reinterpret_cast<const float4*>(p.a_colsum + n0 + 16 * wave + 4 * lq);
        }
    }
    uint4 rq[EP_IT];
    float4 r_raw[EP_IT][4], rg[EP_IT][2], rb4[EP_IT][2];
    const bool r_ln = LNX == 1 && p.r_stats != nullptr;
    if (LNX == 1) {
#pragma unroll
        for (int it = 0; it < EP_IT; ++it) {
            const int c = min(tid + it * NT, RL * CHUNKS - 1), row = c / CHUNKS, ch = c - row * CHUNKS;
            const int m = min(m0 + row, p.M - 1), n = n0 + ch * 8;
            rq[it] = *reinterpret_cast<const uint4*>(p.res + (size_t)m * p.ldres + n);
            if (r_ln) {
                ln_load(p.r_stats + (size_t)m * p.r_nt, p.r_nt, r_raw[it]);
                rg[it][0] = *reinterpret_cast<const float4*>(p.r_gamma + n); rg[it][1] = *reinterpret_cast<const float4*>(p.r_gamma + n + 4);
                rb4[it][0] = *reinterpret_cast<const float4*>(p.r_beta + n); rb4[it][1] = *reinterpret_cast<const float4*>(p.r_beta + n + 4);
            }
        }
    }
    // ---- the activation block: slab s = k 64 s .. + 63 of all RL rows, piece = 8 rows x 128 bytes; wave w stages slabs w, w + NW, ... ----
    {
        unsigned ro[RG];                               // byte offsets of this lane's source chunk in the rows of each 8-row group
        const unsigned swz = (unsigned)((lpos ^ lr) << 4);     // source chunk of LDS slot lpos in a row with (row & 7) == lr
#pragma unroll
        for (int g = 0; g < RG; ++g) ro[g] = (unsigned)min(m0 + g * 8 + lr, p.M - 1) * (unsigned)p.lda * 2u + swz;
        constexpr int SPW = (NSLAB + NW - 1) / NW;     // slabs per wave
#pragma unroll
        for (int sl = 0; sl < SPW; ++sl) {
            const int s = wave + NW * sl;
            if (s < NSLAB) {
                unsigned char* dst = lds + s * SLABB;
#pragma unroll
                for (int g = 0; g < RG; ++g) lw_dma16(p.A, ro[g] + 128u * s, dst + g * 1024);
            }
        }
    }
    // ---- this wave's 16 weight rows x all K: KF fragments of 1 KB, straight into registers ----------------------------------------------
    uint4 wf[KF];
    {
        const uint4* wsrc = p.wp + ((size_t)(n0 / 16 + wave)) * 64 + lane;
        const size_t fstep = (size_t)(p.N / 16) * 64;
#pragma unroll
        for (int f = 0; f < KF; ++f) wf[f] = wsrc[(size_t)f * fstep];
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's LDS-DMA pieces, fragments and operands have landed
    // statistics -> mean / rstd now (frees the raw partials' registers before the MFMA loop)
    float a_mu[TM], a_rs[TM], r_mu[EP_IT], r_rs[EP_IT];
    float2 my_stat = make_float2(0.f, 1.f);            // (mean, rstd) of block row `tid`: handed to the accumulator lanes in the epilogue
    if (a_ln && tid < RL) ln_math(a_raw, p.a_nt, p.a_eps, my_stat.x, my_stat.y);
    if (r_ln) {
#pragma unroll
        for (int it = 0; it < EP_IT; ++it) ln_math(r_raw[it], p.r_nt, p.r_eps, r_mu[it], r_rs[it]);
    }
    __syncthreads();

    // ---- TM row tiles x KF k-steps; fragment reads PF steps ahead of their MFMAs ---------------------------------------------------------
    dh_f32x4 acc[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) acc[i] = dh_f32x4{0.f, 0.f, 0.f, 0.f};
    // LDS read bases: (k half) x (64 KB window) [x last-tile variant]; row 16 i + l15 has (row & 7) == (l15 & 7).  With RL % 16 == 8 the
    // upper half of the last tile does not exist: those lanes re-read the lower half's rows (same row & 7), their outputs are dropped
    constexpr bool PARTIAL = (RL % 16) != 0;
    unsigned rd_base[2][NWIN], rd_last[2][NWIN];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int wdw = 0; wdw < NWIN; ++wdw) {
            rd_base[kk][wdw] = (unsigned)(l15 * 128 + (((kk * 4 + lq) ^ (l15 & 7)) << 4) + wdw * WIN);
            asm volatile("" : "+v"(rd_base[kk][wdw]));
            rd_last[kk][wdw] = PARTIAL ? (unsigned)((l15 & 7) * 128 + (((kk * 4 + lq) ^ (l15 & 7)) << 4) + wdw * WIN) : rd_base[kk][wdw];
            if (PARTIAL) asm volatile("" : "+v"(rd_last[kk][wdw]));
        }
    uint4 fa[PF + 1];
    auto rd = [&](int t) {                             // t = TM f + i: fragment step f = 2 s + kk, row tile i
        const int f = t / TM, i = t - f * TM, s = f >> 1, kk = f & 1;
        const int off = s * SLABB + i * 2048, wdw = off / WIN;
        const unsigned base = (PARTIAL && i == TM - 1) ? rd_last[kk][wdw] : rd_base[kk][wdw];
        fa[t % (PF + 1)] = *reinterpret_cast<const uint4*>(lds + base + (off - wdw * WIN));
    };
#pragma unroll
    for (int t = 0; t < PF; ++t) rd(t);
#pragma unroll
    for (int t = 0; t < KF * TM; ++t) {
        const int f = t / TM, i = t - f * TM;
        if (t + PF < KF * TM) rd(t + PF);
        acc[i] = Op16<OT>::mfma(wf[f], fa[t % (PF + 1)], acc[i]);
        __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();                                   // every wave is done reading the block: LDS is free for the epilogue

    // ---- epilogue: acc[i][r] = C[m0 + 16 i + l15][n0 + 16 wave + 4 lq + r], staged as fp32 rows (XOR-swizzled 16-byte slots) --------------
    float* ep = reinterpret_cast<float*>(lds);
    if (a_ln) {
        // the block rows' (mean, rstd) through the LDS the operands have left (behind the staging tile): the activation block fills the
        // kernel's whole allocation, which keeps TWO 80-row workgroups per CU possible (2 x 80 KB)
        float2* row_stat = reinterpret_cast<float2*>(lds + RL * BN * 4);
        if (tid < RL) row_stat[tid] = my_stat;
        __syncthreads();
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const float2 ms = row_stat[min(16 * i + l15, RL - 1)];
            a_mu[i] = ms.x; a_rs[i] = ms.y;
        }
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int row = 16 * i + l15, slot = 4 * wave + lq;
        if (PARTIAL && row >= RL) continue;
        float4 v;
        if (a_ln) {                                    // rstd * (acc - mu * colsum) + bias'
            v.x = fmaf(a_rs[i], fmaf(-a_mu[i], cs4.x, acc[i][0]), b4.x);
            v.y = fmaf(a_rs[i], fmaf(-a_mu[i], cs4.y, acc[i][1]), b4.y);
            v.z = fmaf(a_rs[i], fmaf(-a_mu[i], cs4.z, acc[i][2]), b4.z);
            v.w = fmaf(a_rs[i], fmaf(-a_mu[i], cs4.w, acc[i][3]), b4.w);
        } else {
            v.x = acc[i][0] + b4.x; v.y = acc[i][1] + b4.y; v.z = acc[i][2] + b4.z; v.w = acc[i][3] + b4.w;
        }
        *reinterpret_cast<float4*>(ep + row * BN + ((slot ^ (row & (SLOTS - 1))) << 2)) = v;
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < EP_IT; ++it) {
        const int c0 = tid + it * NT;
        const int c = min(c0, RL * CHUNKS - 1), row = c / CHUNKS, ch = c - row * CHUNKS;
        const int m = m0 + row, n = n0 + ch * 8;
        const int sw = row & (SLOTS - 1);
        const float4 lo = *reinterpret_cast<const float4*>(ep + row * BN + (((2 * ch) ^ sw) << 2));
        const float4 hi = *reinterpret_cast<const float4*>(ep + row * BN + (((2 * ch + 1) ^ sw) << 2));
        float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
        const bool ok = c0 < RL * CHUNKS && m < p.M;
        if (LNX == 0) {
            if (p.relu) {
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = fmaxf(v[u], 0.f);
            }
            if (ok) store16(reinterpret_cast<OT*>(p.C + (size_t)m * p.ldc + n), v);
        } else {
            const uint32_t w4[4] = {rq[it].x, rq[it].y, rq[it].z, rq[it].w};
            float rr[8];
#pragma unroll
            for (int u = 0; u < 4; ++u) Op16<OT>::unpack2(w4[u], rr[2 * u], rr[2 * u + 1]);
            if (r_ln) {
                const float g8[8] = {rg[it][0].x, rg[it][0].y, rg[it][0].z, rg[it][0].w, rg[it][1].x, rg[it][1].y, rg[it][1].z, rg[it][1].w};
                const float b8[8] = {rb4[it][0].x, rb4[it][0].y, rb4[it][0].z, rb4[it][0].w, rb4[it][1].x, rb4[it][1].y, rb4[it][1].z, rb4[it][1].w};
#pragma unroll
                for (int u = 0; u < 8; ++u) rr[u] = fmaf((rr[u] - r_mu[it]) * r_rs[it], g8[u], b8[u]);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] += rr[u];
            if (p.relu) {
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = fmaxf(v[u], 0.f);
            }
            // statistics of the ROUNDED values; 8 consecutive lanes = the 8 chunks of one (row, 64-column tile): every lane takes part
            float s1 = 0.f;
#pragma unroll
            for (int u = 0; u < 8; ++u) { v[u] = Op16<OT>::to_f32(Op16<OT>::from_f32(v[u])); s1 += v[u]; }
            const float mean = sum8(s1) * (1.0f / 64.0f);
            float s2 = 0.f;
#pragma unroll
            for (int u = 0; u < 8; ++u) { const float d = v[u] - mean; s2 = fmaf(d, d, s2); }
            s2 = sum8(s2);
            if (ok) {
                store16(reinterpret_cast<OT*>(p.C + (size_t)m * p.ldc + n), v);
                if ((ch & 7) == 0) p.o_stats[(size_t)m * (p.N / 64) + (n >> 6)] = make_float2(mean, s2);
            }
        }
    }
}

template <typename OT, int NW, int RL, int KQ, int LNX>
__global__ __launch_bounds__(64 * NW, NW == 8 ? 2 : 1) void linear_wreg_kernel(LwParams p) {
    constexpr int NSLAB = 8 * KQ, SLABB = RL * 128;
    __shared__ __attribute__((aligned(16))) unsigned char lds[NSLAB * SLABB];
    int cb, rb;
    if (p.xn) {                                        // XCD x = blockIdx % 8 owns column group x % xn, row group x / xn
        const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
        const int cpg = p.tiles_n / p.xn, rpg = p.tiles_m / (8 / p.xn);
        cb = (xcd % p.xn) * cpg + idx % cpg; rb = (xcd / p.xn) * rpg + idx / cpg;
    } else {
        cb = blockIdx.x % p.tiles_n; rb = blockIdx.x / p.tiles_n;
    }
    lw_item<OT, NW, RL, KQ, LNX>(p, cb, rb, lds);
}

// XCD column groups: the divisor xn of 8 (tiles_n % xn == 0, tiles_m % (8 / xn) == 0) with the fewest operand bytes per XCD; 0 = none fits
int pick_xn(int tiles_m, int tiles_n, double w_block_bytes, double a_block_bytes) {
    int best = 0; double best_bytes = 0.0;
    for (int xn = 1; xn <= 8; xn *= 2) {
        const int xm = 8 / xn;
        if (tiles_n % xn || tiles_m % xm) continue;
        const double bytes = (tiles_n / xn) * w_block_bytes + (tiles_m / xm) * a_block_bytes;
        if (!best || bytes < best_bytes) { best = xn; best_bytes = bytes; }
    }
    return best;
}
}  // namespace

// 1 when dh_linear_ln_wreg takes the shape: K = 512 (any N % 128 == 0 without residual / statistics, N % 64 == 0 with them) or
// K = 2,048 with residual + statistics (the position-wise feed-forward's second layer)
extern "C" int dh_linear_ln_wreg_supported(int N, int K, int with_residual_stats) {
    if (with_residual_stats) return (K == 512 || K == 2048) && (N % 64) == 0;
    return K == 512 && (N % 64) == 0;
}

// Fraction of the resident workgroup slots the launch keeps busy when it needs more than one residency round (1.0 when everything
// is co-resident): the callers prefer the tile kernels below ~0.85 (e.g. 3,000 rows: fc_2 = 600 workgroups at one per CU = 3 rounds
// for 2.34 rounds of work).  Slots per CU: K = 2,048 one (160 KB LDS); the 4-wave 40 KB forms three; the 8-wave forms two.
extern "C" double dh_linear_ln_wreg_occupancy(int M, int N, int K, int with_residual_stats) {
    if (!dh_linear_ln_wreg_supported(N, K, with_residual_stats) || M <= 0) return 0.0;
    long long wgs, cap;
    if (with_residual_stats || (N % 128) != 0 || (long long)dh_cdiv(M, 40) * (N / 64) <= 256) {
        wgs = (long long)dh_cdiv(M, 40) * (N / 64); cap = K == 2048 ? 256 : 768;
    } else {
        const bool rl64 = dh_cdiv(M, 64) * (N / 128) <= 256;
        wgs = (long long)dh_cdiv(M, rl64 ? 64 : 80) * (N / 128); cap = 512;
    }
    if (wgs <= cap) return 1.0;
    const long long rounds = (wgs + cap - 1) / cap;
    return (double)wgs / (double)(rounds * cap);
}

// dh_linear_ln with `w_packed` = dh_pack_mfma_fragments(W [N, K]) in place of W; same arguments, restrictions as above:
//   * ln->a_stats (or no LayerNorm at all), optional ReLU, no residual, no output statistics; or
//   * residual (+ optional ln->r_stats) AND ln->o_stats.
extern "C" int dh_linear_ln_wreg(const void* A, int lda, const void* w_packed, const float* bias, const void* residual, int ldres,
                                 void* C, int ldc, int M, int N, int K, int relu, const dh_ln_fold_t* ln, int dtype, void* stream) {
    if (!DH_IS_16BIT(dtype)) return DH_ERR_UNSUPPORTED;
    DH_REQUIRE(A && w_packed && bias && C && ln && M > 0 && N > 0 && (lda % 8) == 0 && lda >= K && ldc >= N && (ldc % 8) == 0);
    DH_REQUIRE(((uintptr_t)A % 16) == 0 && ((uintptr_t)w_packed % 16) == 0 && ((uintptr_t)C % 16) == 0 && ((uintptr_t)bias % 16) == 0);
    DH_REQUIRE((unsigned long long)M * (unsigned)lda * 2ull < (1ull << 32));
    const bool lnx = residual != nullptr;
    DH_REQUIRE(dh_linear_ln_wreg_supported(N, K, lnx));
    if (lnx) {
        DH_REQUIRE(ln->o_stats && !ln->a_stats && ldres >= N && (ldres % 8) == 0 && ((uintptr_t)residual % 16) == 0 && ((uintptr_t)ln->o_stats % 8) == 0);
        DH_REQUIRE(!ln->r_stats || (ln->r_gamma && ln->r_beta && ln->r_tiles >= 2 && ln->r_tiles <= 8 && (ln->r_tiles % 2) == 0 && ln->r_tiles * 64 == N &&
                                    ((uintptr_t)ln->r_stats % 16) == 0 && ((uintptr_t)ln->r_gamma % 16) == 0 && ((uintptr_t)ln->r_beta % 16) == 0));
    } else {
        DH_REQUIRE(!ln->o_stats && !ln->r_stats);
        DH_REQUIRE(!ln->a_stats || (ln->a_colsum && ln->a_tiles == 8 && ((uintptr_t)ln->a_stats % 16) == 0 && ((uintptr_t)ln->a_colsum % 16) == 0));
    }
    LwParams p{};
    p.A = (const uint16_t*)A; p.lda = lda; p.wp = (const uint4*)w_packed; p.bias = bias; p.res = (const uint16_t*)residual; p.ldres = ldres;
    p.C = (uint16_t*)C; p.ldc = ldc; p.M = M; p.N = N; p.relu = relu;
    p.a_stats = (const float2*)ln->a_stats; p.a_nt = ln->a_tiles; p.a_eps = ln->a_eps; p.a_colsum = ln->a_colsum;
    p.r_stats = (const float2*)ln->r_stats; p.r_nt = ln->r_tiles; p.r_eps = ln->r_eps; p.r_gamma = ln->r_gamma; p.r_beta = ln->r_beta;
    p.o_stats = (float2*)ln->o_stats;
    dh_prof_set_dims(M, N, K);
    DhProfScope prof("dh_linear", 2.0 * M * N * K, 2.0 * ((double)M * K + (double)N * K + (double)M * N * (residual ? 2 : 1)), stream);
    hipStream_t s = (hipStream_t)stream;
    if (lnx) {
        p.tiles_n = N / 64; p.tiles_m = dh_cdiv(M, 40);
        p.xn = pick_xn(p.tiles_m, p.tiles_n, 64.0 * K * 2, 40.0 * K * 2);
        DH_DISPATCH_16(dtype, {
            if (K == 512) hipLaunchKernelGGL((linear_wreg_kernel<T, 4, 40, 1, 1>), dim3(p.tiles_m * p.tiles_n), dim3(256), 0, s, p);
            else hipLaunchKernelGGL((linear_wreg_kernel<T, 4, 40, 4, 1>), dim3(p.tiles_m * p.tiles_n), dim3(256), 0, s, p);
        });
        DH_LAUNCH_CHECK();
    }
    if ((N % 128) != 0 || (long long)dh_cdiv(M, 40) * (N / 64) <= 256) {
        // narrow outputs (the cross-attention query projection, N = D = 512): 64-column x 40-row blocks, 4 waves, as the residual form
        p.tiles_n = N / 64; p.tiles_m = dh_cdiv(M, 40);
        p.xn = pick_xn(p.tiles_m, p.tiles_n, 64.0 * K * 2, 40.0 * K * 2);
        DH_DISPATCH_16(dtype, hipLaunchKernelGGL((linear_wreg_kernel<T, 4, 40, 1, 0>), dim3(p.tiles_m * p.tiles_n), dim3(256), 0, s, p));
        DH_LAUNCH_CHECK();
    }
    p.tiles_n = N / 128;
    // 64-row blocks when they still fit ONE round of 256 workgroups (more of them = more CUs busy), else 80-row blocks
    const bool rl64 = dh_cdiv(M, 64) * p.tiles_n <= 256;
    p.tiles_m = dh_cdiv(M, rl64 ? 64 : 80);
    p.xn = pick_xn(p.tiles_m, p.tiles_n, 128.0 * K * 2, (rl64 ? 64.0 : 80.0) * K * 2);
    const dim3 grid(p.tiles_m * p.tiles_n);
    DH_DISPATCH_16(dtype, {
        if (rl64) hipLaunchKernelGGL((linear_wreg_kernel<T, 8, 64, 1, 0>), grid, dim3(512), 0, s, p);
        else hipLaunchKernelGGL((linear_wreg_kernel<T, 8, 80, 1, 0>), grid, dim3(512), 0, s, p);
    });
    DH_LAUNCH_CHECK();
}
