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
    // L2 prefetch for the NEXT launch of the stream (dh_linear_ln_wreg_prefetch): workgroups n_work .. gridDim.x - 1 only do this
    const unsigned char* pf_base[2]; unsigned pf_stride, pf_bytes[2]; int pf_groups, pf_tpg, pf_tiles, pf_part, pf_parts, n_work;
};

// global -> LDS, 16 bytes per lane: wave-uniform base (SGPR pair) + per-lane byte offset
__device__ __forceinline__ void lw_dma16(const void* base, unsigned off, void* lds_dst) {
    const unsigned m = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(dh_lptr_t)lds_dst);
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" :: "s"(m), "v"(off), "s"(base) : "memory");
}

// NW waves (16 output columns each), RL activation rows in LDS, KQ = K / 512, LNX: 0 = (deferred LayerNorm on the A rows |
// plain) + optional ReLU; 1 = residual (optionally pre-LayerNorm) + statistics of the output rows
// One output block (column block cb, row block rb) of the GEMM: the whole body of linear_wreg_kernel, callable from the persistent
// chain kernel below as well (`lds`: NSLAB * SLABB bytes, free on entry; the block's threads must all call it)
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

// Prefetch workgroups (blockIdx >= n_work; dh_linear_ln_wreg_prefetch): pull the operands the NEXT kernel of the stream will read
// into the L2 of the XCD that kernel's workgroups will run on, while this (latency-bound, 5 us, a few MB of traffic) GEMM runs.  The
// operands are two arrays of tiles (stride pf_stride, pf_bytes[a] used bytes per tile); the consumer's workgroup g reads tiles
// g * pf_tpg .. + pf_tpg - 1 and runs on XCD g % 8 (workgroups go round-robin over the XCDs), as prefetch workgroup j does on XCD
// (n_work + j) % 8 = j % 8.  Read-only lines survive a kernel boundary in L2 (tools/probe/launch_floor_probe.hip).  The loads are
// LDS-DMA transfers into a scratch kilobyte per wave: no registers, nothing waits for them but the end of the wave.  A wrong placement
// guess costs time only -- nothing reads what lands in LDS.
template <int NT>
__device__ __forceinline__ void lw_prefetch(const LwParams& p, unsigned char* lds) {
    const int j = (int)blockIdx.x - p.n_work, n_pf = (int)gridDim.x - p.n_work;
    const int xcd = j & 7, slot = j >> 3, per_xcd = max(1, n_pf >> 3);
    const int tid = threadIdx.x;
    unsigned char* dst = lds + (tid >> 6) * 1024;
    // (part k of n: this launch takes every n-th of an XCD's groups -- the tiles can be spread over several launches in front of the consumer)
    for (int g = xcd + 8 * (p.pf_part + p.pf_parts * slot); g < p.pf_groups; g += 8 * p.pf_parts * per_xcd) {
        for (int t = g * p.pf_tpg; t < min((g + 1) * p.pf_tpg, p.pf_tiles); ++t) {
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                const unsigned char* src = p.pf_base[a] + (size_t)t * p.pf_stride;
                const unsigned bytes = p.pf_bytes[a];
                for (unsigned off = 0; off < bytes; off += NT * 16u)         // (uniform bounds; lanes past the end repeat the last chunk)
                    dh_lds_dma16(src + min(off + (unsigned)tid * 16u, bytes - 16u), dst);
            }
        }
    }
}

template <typename OT, int NW, int RL, int KQ, int LNX>
__global__ __launch_bounds__(64 * NW, NW == 8 ? 2 : 1) void linear_wreg_kernel(LwParams p) {
    constexpr int NSLAB = 8 * KQ, SLABB = RL * 128;
    __shared__ __attribute__((aligned(16))) unsigned char lds[NSLAB * SLABB];
    if (NW == 4 && KQ == 1 && p.pf_groups && (int)blockIdx.x >= p.n_work) { lw_prefetch<64 * NW>(p, lds); return; }
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

// ---- several dependent GEMMs of one decode position in ONE launch (round 5) ------------------------------------------------------------
// The deferred-LayerNorm chain of a decoder layer ends in up to four GEMMs in a row -- (enc_)fc_o -> fc_1 -> fc_2 -> the NEXT layer's
// fc_q|k|v -- each consuming whole rows of its predecessor: four launches at 5 - 11 us, of which ~2.5 us are the boundary and ~1.6 us
// the cold first touch of rows the previous launch wrote back (tools/probe/launch_floor_probe.hip).  Here they are phases of one
// persistent launch, 256 workgroups x 4 waves, every phase made of the SAME output blocks lw_item<.., 4, 40, KQ, LNX> computes for the
// stand-alone launches (64 columns x 40 rows; per-output arithmetic does not depend on the tiling: results bit-identical):
//   * the rows are cut into 8 groups of consecutive 40-row blocks, group g is worked on ONLY by workgroups that find themselves on
//     XCD g (s_getreg HW_REG_XCC_ID -- the hardware id, not blockIdx % 8): a phase's outputs are then read back through the SAME L2
//     they were written to (plain stores keep the line there), so no agent-scope release / acquire (an L2 write-back and an L1
//     invalidate per workgroup and phase: 24 us per step in the probe) is needed -- 1.1 us per hand-over in the probe.  No line is
//     re-written inside the launch, so no L1 can hold a stale copy; L1s are clean at launch;
//   * inside a group the blocks of a phase are claimed dynamically (one counter per (group, phase)), so ANY number >= 1 of workgroups
//     on an XCD completes its group; a workgroup enters phase p + 1 when the group's `done` counter of phase p is full -- every claimed
//     block belongs to a running workgroup, so the wait cannot deadlock whatever the dispatcher does;
//   * an XCD that received no workgroup at all (never observed) leaves its group untouched: the LAST workgroup to finish (global
//     counter) runs such groups alone, then zeroes the counters for the next launch.  Every spin is bounded (error word on timeout).
// MEASURED (tools/chain_bench.py): correct and bit-identical at every size, and the hand-over protocol costs what the
// probe said (1.7 - 2 us per phase, polling included) -- but the launch is SLOWER than the four launches it replaces: 54 against 33 us at
// 1,280 rows, 34 against 21 us at 160 rows (C3 step 25.9 against 21.1 ms, C5's 38-template shard 16.9 against 13.6 ms).  A block inside
// the chain takes as long as the stand-alone kernel INCLUDING its launch (its own load -> MFMA -> store chain of latencies is the cost,
// not the boundary), and phases with several blocks per workgroup (fc_1: 4, qkv: 3 at 1,280 rows) run them back to back without overlap
// where the stand-alone 8-wave kernels own 128 x 80 blocks.  Opt-in (option "decode_chain_fusion"); what it needs to win: the next
// block's operands (weights, A rows) requested while the current block computes.
struct ChainParams {
    LwParams ph[4];
    int form[4];                                       // 0: <4,40,1,0>, 1: <4,40,1,1>, 2: <4,40,4,1>
    int inv[4];                                        // phase reads a buffer an EARLIER phase of this launch re-wrote after a still earlier one
                                                       // read it (this CU's L1 may hold the old line): agent-scope acquire in front of it
    int n_ph, n_rb, rpg;                               // phases, 40-row blocks, blocks per group
    unsigned* sync;                                    // [0,8) arrive, [8,40) claim[g][p], [40,72) done[g][p], 72 finished, 73 error
};

__device__ __forceinline__ unsigned chain_load(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

template <typename OT>
__device__ __forceinline__ void chain_blocks(const ChainParams& P, const int ph, const int rb0, const int first, const int last, unsigned char* lds) {
    const LwParams& p = P.ph[ph];
    const int tn = p.tiles_n;
    for (int i = first; i < last; ++i) {                // (scalar bounds: a uniform loop)
        const int cb = i % tn, rb = rb0 + i / tn;
        if (P.form[ph] == 0) lw_item<OT, 4, 40, 1, 0>(p, cb, rb, lds);
        else if (P.form[ph] == 1) lw_item<OT, 4, 40, 1, 1>(p, cb, rb, lds);
        else lw_item<OT, 4, 40, 4, 1>(p, cb, rb, lds);
        __syncthreads();                                 // LDS is free for the next block
    }
}

// Group g of the rows, worked on by the workgroups of one XCD.  The blocks of a phase are claimed in BATCHES of ceil(blocks / 32) (32 =
// the workgroups an XCD is expected to receive), and the batches of ALL phases are claimed up front in one round trip: an agent-scope
// atomic is a fabric round trip of 1 - 2 us, and the first version (one claim + one `done` increment per block, each waited for)
// spent 27 us per layer in them.  If the `done` counter of a phase does not fill within a short time -- fewer workgroups on this XCD
// than expected -- the waiting workgroups claim further batches, so any number >= 1 of workgroups still completes the group.
template <typename OT>
__device__ __forceinline__ void chain_group(const ChainParams& P, const int g, unsigned char* lds, const int expect) {
    int* mail = reinterpret_cast<int*>(lds);             // mailbox in the (free between blocks) operand area: the K = 2,048 form uses all 160 KB
    const int rb0 = g * P.rpg, nrb = min(P.rpg, P.n_rb - rb0);
    if (nrb <= 0) return;
    unsigned* claim = P.sync + 8 + 4 * g;
    unsigned* done = P.sync + 40 + 4 * g;
    int base[4], kb[4], total[4];
#pragma unroll
    for (int ph = 0; ph < 4; ++ph) {
        total[ph] = ph < P.n_ph ? nrb * P.ph[ph].tiles_n : 0;
        kb[ph] = max(1, (total[ph] + expect - 1) / expect);
    }
    if (threadIdx.x == 0) {
        unsigned got[4];
#pragma unroll
        for (int ph = 0; ph < 4; ++ph)
            got[ph] = ph < P.n_ph ? __hip_atomic_fetch_add(&claim[ph], (unsigned)kb[ph], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
#pragma unroll
        for (int ph = 0; ph < 4; ++ph) mail[ph] = (int)got[ph];
    }
    __syncthreads();
#pragma unroll
    for (int ph = 0; ph < 4; ++ph) base[ph] = __builtin_amdgcn_readfirstlane(mail[ph]);
    __syncthreads();
    for (int ph = 0; ph < P.n_ph; ++ph) {
        int first = base[ph], lastb = min(first + kb[ph], total[ph]);
        int stalls = 0;
        for (;;) {
            if (first < lastb) {
                chain_blocks<OT>(P, ph, rb0, first, lastb, lds);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's stores have reached the XCD's L2
                __syncthreads();                                 // ... all four waves'
                if (threadIdx.x == 0) __hip_atomic_fetch_add(&done[ph], (unsigned)(lastb - first), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (ph + 1 == P.n_ph) break;                 // nobody in this launch reads the last phase's rows
            // the group's rows of this phase are complete before anybody reads them
            if (threadIdx.x == 0) {
                int spin = 0;
                while (chain_load(&done[ph]) < (unsigned)total[ph] && ++spin < (1 << 12)) __builtin_amdgcn_s_sleep(2);
                mail[0] = spin >= (1 << 12);
            }
            __syncthreads();
            const int stalled = __builtin_amdgcn_readfirstlane(mail[0]);
            __syncthreads();
            if (!stalled) break;
            // not complete after ~ 1 ms: workgroups may be missing on this XCD -- take another batch (if any is left) and wait again
            if (threadIdx.x == 0) mail[0] = (int)__hip_atomic_fetch_add(&claim[ph], (unsigned)kb[ph], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __syncthreads();
            first = __builtin_amdgcn_readfirstlane(mail[0]);
            __syncthreads();
            lastb = min(first + kb[ph], total[ph]);
            if (first >= lastb && ++stalls > 512) {      // ~ 0.5 s without progress: give up loudly instead of hanging
                if (threadIdx.x == 0) __hip_atomic_fetch_or(P.sync + 73, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
            }
        }
        if (ph + 1 < P.n_ph && P.inv[ph + 1]) {          // this CU's L1 may hold lines the phase re-wrote
            if (threadIdx.x < 64) asm volatile("buffer_inv sc1\n\ts_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
    }
}

template <typename OT>
__global__ __launch_bounds__(256, 1) void decode_gemm_chain_kernel(ChainParams P) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[163840];
    int& s_item = *reinterpret_cast<int*>(lds);
    const int xcc = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 7;       // HW_REG_XCC_ID[3:0]: the XCD this workgroup runs on
    if (threadIdx.x == 0) __hip_atomic_fetch_add(P.sync + xcc, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    chain_group<OT>(P, xcc, lds, (int)gridDim.x / 8);
    __syncthreads();
    if (threadIdx.x == 0) s_item = (int)__hip_atomic_fetch_add(P.sync + 72, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const bool last = __builtin_amdgcn_readfirstlane(s_item) == (int)gridDim.x - 1;
    __syncthreads();
    if (!last) return;
    for (int x = 0; x < 8; ++x)                          // (every workgroup has arrived by now: the counts are final)
        if (chain_load(P.sync + x) == 0) chain_group<OT>(P, x, lds, 1);
    __syncthreads();
    if (threadIdx.x < 73) P.sync[threadIdx.x] = 0u;     // ready for the next launch (word 73, the error flag, is sticky)
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
    return dh_linear_ln_wreg_prefetch(A, lda, w_packed, bias, residual, ldres, C, ldc, M, N, K, relu, ln, nullptr, 0, dtype, stream);
}

// dh_linear_ln_wreg + `n_pf` extra workgroups that pull `pf`'s tiles (the operands of the NEXT kernel on the stream) into L2 while the
// GEMM runs -- only in the 4-wave K = 512 forms (64-column x 40-row blocks: the residual form, and without residual N % 128 != 0 or
// <= 256 blocks -- fc_o and the cross-attention's fc_q); other forms ignore `pf`.  Same results as dh_linear_ln_wreg (the extra workgroups write nothing).
extern "C" int dh_linear_ln_wreg_prefetch(const void* A, int lda, const void* w_packed, const float* bias, const void* residual, int ldres,
                                          void* C, int ldc, int M, int N, int K, int relu, const dh_ln_fold_t* ln,
                                          const dh_l2_prefetch_t* pf, int n_pf, int dtype, void* stream) {
    if (!DH_IS_16BIT(dtype)) return DH_ERR_UNSUPPORTED;
    DH_REQUIRE(!pf || n_pf <= 0 || (pf->base[0] && pf->base[1] && pf->n_tiles > 0 && pf->tiles_per_group > 0 && pf->tile_bytes[0] >= 16 &&
                                    pf->tile_bytes[1] >= 16 && (pf->tile_bytes[0] % 16) == 0 && (pf->tile_bytes[1] % 16) == 0 &&
                                    pf->tile_bytes[0] <= pf->tile_stride && pf->tile_bytes[1] <= pf->tile_stride && (pf->tile_stride % 16) == 0 &&
                                    pf->parts >= 1 && pf->part >= 0 && pf->part < pf->parts &&
                                    ((uintptr_t)pf->base[0] % 16) == 0 && ((uintptr_t)pf->base[1] % 16) == 0 && n_pf <= 4096));
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
    auto with_prefetch = [&]() -> int {                  // extra workgroups of a 4-wave K = 512 launch
        p.n_work = p.tiles_m * p.tiles_n;
        if (!pf || n_pf <= 0 || (p.n_work % 8) != 0) return 0;       // (a grid that is no multiple of 8 would shift the prefetchers' XCDs)
        p.pf_base[0] = (const unsigned char*)pf->base[0]; p.pf_base[1] = (const unsigned char*)pf->base[1];
        p.pf_stride = pf->tile_stride; p.pf_bytes[0] = pf->tile_bytes[0]; p.pf_bytes[1] = pf->tile_bytes[1];
        p.pf_tiles = pf->n_tiles; p.pf_tpg = pf->tiles_per_group; p.pf_groups = dh_cdiv(pf->n_tiles, pf->tiles_per_group);
        p.pf_part = pf->part; p.pf_parts = pf->parts;
        return (n_pf + 7) / 8 * 8;
    };
    if (lnx) {
        p.tiles_n = N / 64; p.tiles_m = dh_cdiv(M, 40);
        p.xn = pick_xn(p.tiles_m, p.tiles_n, 64.0 * K * 2, 40.0 * K * 2);
        DH_DISPATCH_16(dtype, {
            if (K == 512) hipLaunchKernelGGL((linear_wreg_kernel<T, 4, 40, 1, 1>), dim3(p.tiles_m * p.tiles_n + with_prefetch()), dim3(256), 0, s, p);
            else hipLaunchKernelGGL((linear_wreg_kernel<T, 4, 40, 4, 1>), dim3(p.tiles_m * p.tiles_n), dim3(256), 0, s, p);
        });
        DH_LAUNCH_CHECK();
    }
    if ((N % 128) != 0 || (long long)dh_cdiv(M, 40) * (N / 64) <= 256) {
        // narrow outputs (the cross-attention query projection, N = D = 512): 64-column x 40-row blocks, 4 waves, as the residual form
        p.tiles_n = N / 64; p.tiles_m = dh_cdiv(M, 40);
        p.xn = pick_xn(p.tiles_m, p.tiles_n, 64.0 * K * 2, 40.0 * K * 2);
        DH_DISPATCH_16(dtype, hipLaunchKernelGGL((linear_wreg_kernel<T, 4, 40, 1, 0>), dim3(p.tiles_m * p.tiles_n + with_prefetch()), dim3(256), 0, s, p));
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

// One launch for up to four dependent GEMMs of a decode position (decode_gemm_chain_kernel above).  Step i: C_i = epilogue(A_i W_i^T) in
// one of dh_linear_ln_wreg's two forms, on w_packed = dh_pack_mfma_fragments(W_i); a later step may read what an earlier one wrote
// (whole rows).  Same results as the same steps through dh_linear_ln_wreg / dh_linear_ln, bit for bit.  `sync`: 74 uint32 of device
// memory, zero before the FIRST launch that uses them (the kernel leaves them zero), private to the stream; sync[73] != 0 afterwards =
// a bounded wait timed out (never observed; results then undefined).
extern "C" int dh_decode_gemm_chain_supported(int N, int K, int with_residual_stats) {
    if (with_residual_stats) return (K == 512 || K == 2048) && (N % 64) == 0;
    return K == 512 && (N % 64) == 0;
}

extern "C" int dh_decode_gemm_chain(const dh_chain_step_t* steps, int n_steps, int M, uint32_t* sync, int dtype, void* stream) {
    if (!DH_IS_16BIT(dtype)) return DH_ERR_UNSUPPORTED;
    DH_REQUIRE(steps && n_steps >= 1 && n_steps <= 4 && M > 0 && sync && ((uintptr_t)sync % 4) == 0);
    ChainParams P{};
    P.n_ph = n_steps; P.n_rb = dh_cdiv(M, 40); P.rpg = dh_cdiv(P.n_rb, 8); P.sync = sync;
    double flops = 0.0, bytes = 0.0;
    for (int i = 0; i < n_steps; ++i) {
        const dh_chain_step_t& st = steps[i];
        const dh_ln_fold_t* ln = &st.ln;
        const bool lnx = st.residual != nullptr;
        DH_REQUIRE(st.A && st.w_packed && st.bias && st.C && st.N > 0 && (st.lda % 8) == 0 && st.lda >= st.K && st.ldc >= st.N && (st.ldc % 8) == 0);
        DH_REQUIRE(((uintptr_t)st.A % 16) == 0 && ((uintptr_t)st.w_packed % 16) == 0 && ((uintptr_t)st.C % 16) == 0 && ((uintptr_t)st.bias % 16) == 0);
        DH_REQUIRE((unsigned long long)M * (unsigned)st.lda * 2ull < (1ull << 32) && dh_decode_gemm_chain_supported(st.N, st.K, lnx));
        if (lnx) {
            DH_REQUIRE(ln->o_stats && !ln->a_stats && st.ldres >= st.N && (st.ldres % 8) == 0 && ((uintptr_t)st.residual % 16) == 0 && ((uintptr_t)ln->o_stats % 8) == 0);
            DH_REQUIRE(!ln->r_stats || (ln->r_gamma && ln->r_beta && ln->r_tiles >= 2 && ln->r_tiles <= 8 && (ln->r_tiles % 2) == 0 && ln->r_tiles * 64 == st.N &&
                                        ((uintptr_t)ln->r_stats % 16) == 0 && ((uintptr_t)ln->r_gamma % 16) == 0 && ((uintptr_t)ln->r_beta % 16) == 0));
        } else {
            DH_REQUIRE(!ln->o_stats && !ln->r_stats);
            DH_REQUIRE(!ln->a_stats || (ln->a_colsum && ln->a_tiles == 8 && ((uintptr_t)ln->a_stats % 16) == 0 && ((uintptr_t)ln->a_colsum % 16) == 0));
        }
        LwParams& p = P.ph[i];
        p.A = (const uint16_t*)st.A; p.lda = st.lda; p.wp = (const uint4*)st.w_packed; p.bias = st.bias; p.res = (const uint16_t*)st.residual; p.ldres = st.ldres;
        p.C = (uint16_t*)st.C; p.ldc = st.ldc; p.M = M; p.N = st.N; p.relu = st.relu;
        p.a_stats = (const float2*)ln->a_stats; p.a_nt = ln->a_tiles; p.a_eps = ln->a_eps; p.a_colsum = ln->a_colsum;
        p.r_stats = (const float2*)ln->r_stats; p.r_nt = ln->r_tiles; p.r_eps = ln->r_eps; p.r_gamma = ln->r_gamma; p.r_beta = ln->r_beta;
        p.o_stats = (float2*)ln->o_stats;
        p.tiles_n = st.N / 64; p.tiles_m = P.n_rb; p.xn = 0;
        P.form[i] = lnx ? (st.K == 2048 ? 2 : 1) : 0;
        flops += 2.0 * M * st.N * st.K;
        bytes += 2.0 * ((double)M * st.K + (double)st.N * st.K + (double)M * st.N * (lnx ? 2 : 1));
        // does this step read a buffer that an earlier step of the launch wrote AFTER a still earlier step had read it?
        const void* in[4] = {st.A, st.residual, ln->a_stats, ln->r_stats};
        for (int b = 0; b < 4 && !P.inv[i]; ++b) {
            if (!in[b]) continue;
            for (int q = 1; q < i && !P.inv[i]; ++q) {
                if (in[b] != steps[q].C && in[b] != (const void*)steps[q].ln.o_stats) continue;
                for (int r = 0; r < q; ++r)
                    if (in[b] == steps[r].A || in[b] == steps[r].residual || in[b] == (const void*)steps[r].ln.a_stats ||
                        in[b] == (const void*)steps[r].ln.r_stats) { P.inv[i] = 1; break; }
            }
        }
    }
    dh_prof_set_tag("chain");
    dh_prof_set_dims(M, n_steps, 0);
    DhProfScope prof("dh_linear", flops, bytes, stream);
    DH_DISPATCH_16(dtype, hipLaunchKernelGGL((decode_gemm_chain_kernel<T>), dim3(256), dim3(256), 0, (hipStream_t)stream, P));
    DH_LAUNCH_CHECK();
}
