// One LSTM layer time step with the gate weights STATIONARY IN REGISTERS (16-bit paths; nn.LSTM of rnn_models.py:23-24, one step of
// :80 / :108) -- the same arithmetic as lstm_fused.hip (gates = [x | h_prev[parent]] [W_ih | W_hh]^T + b on the matrix cores, cell
// update in the epilogue), re-tiled for the decode shape (1,280 beam rows x 2,048 gate columns x K <= 1,024):
//   * the 64 x 64 tile kernel stages BOTH operands through LDS for each of its 640 tiles (147 MB of LDS-DMA per launch, 2.5 rounds of
//     workgroups) and runs at the piece rate of the LDS-DMA path: 17-20 us for 4 GFLOP;
//   * here a workgroup owns 128 gate rows (32 hidden units) x 80 activation rows, 16 x 16 workgroups = ONE round of 256.  Each wave keeps
//     its 16 gate rows x all K as MFMA fragments in registers (K / 32 x 4 <= 128 VGPRs), loaded straight from L2 out of the
//     fragment-packed weights (dh_pack_mfma_fragments: coalesced 1 KB loads); only the [80 rows x K] activation block goes through LDS
//     (LDS-DMA with the beam-parent / token gather in the source address; <= 160 KB: the WHOLE block is resident, no ring) -- 42 MB of
//     LDS-DMA per launch instead of 147 MB, and every byte of it is requested in the first microsecond;
//   * ONE wait + ONE barrier, then 5 x K / 32 MFMAs per wave with nothing but LDS fragment reads in between; the accumulator quad
//     is (i, f, g, o) of one hidden unit for one row (gate-interleaved weight rows), so the cell update runs in registers.
// All LDS-DMA transfers are issued before the first ordinary load whose completion the compiler counts (vmcnt retires in order, so
// its counted waits then cover the transfers too) and none after.
// Results: the same MFMA chain per output (k ascending) and the same epilogue as lstm_layer_fused_kernel -- bit-identical.
// MEASURED (round 3, 1,280 rows, Hh 512): 14.5 / 15.3 us per launch (K = 768 / 1,024; rocprofv3 kernel time) against 17.7 us for the
// tile kernel; C2 step 7.74 vs 7.92 ms.  In-kernel phase stamps (s_memtime, K = 1,024, cycles): row-index loads 2.5 k, issuing 5 operand
// loads + 20 LDS-DMA pieces per wave 5.8 k, issuing 32 fragment loads 2.9 k, last byte landed +1.8 k, slowest wave at the barrier +5.8 k,
// 160 MFMAs + 160 fragment reads per wave 5.0 k, cell update + stores 4.7 k: the kernel is bound by the rate at which a CU's eight waves
// can issue 1 KB vector-memory instructions (416 of them per CU: ~30 cycles each, the same ~33 B/clk per CU every LDS-DMA GEMM in this
// library runs at), not by MFMA or LDS; a 16 x 16 partition of [2,048 x K] x [K x 1,280] is the minimum of that byte count (416 KB per
// CU; the tile kernel moves 575 KB per CU).
#include "common.h"
#include "prof.h"

__device__ uint4 lw_zero_page[4];

namespace {
struct LstmWregParams {
    const uint16_t* x_rows; int ldx, x_div;
    const uint16_t* emb; const int32_t* tokens; int tok_ld, tok_pos;
    const uint16_t* h_prev; const float* c_prev; const int32_t* hparent;
    uint16_t* h_next; float* c_next;
    uint16_t* h_out; int ld_out;
    const uint4* wp; const float* bias;          // fragment-packed gate-interleaved weights [K / 32][4 Hh / 16][64], bias [4 Hh]
    int rows, row_mult, E, Hh, tiles_m, tiles_n;
};

__device__ __forceinline__ float lw_sigmoid(float x) { return __frcp_rn(1.0f + __expf(-x)); }
__device__ __forceinline__ float lw_tanh(float x) {
    const float e = __expf(-2.0f * fabsf(x));
    const float t = (1.0f - e) * __frcp_rn(1.0f + e);
    return copysignf(t, x);
}

// global -> LDS, 16 bytes per lane: wave-uniform base (SGPR pair) + per-lane byte offset
__device__ __forceinline__ void lw_dma16(const void* base, unsigned off, void* lds_dst) {
    const unsigned m = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(dh_lptr_t)lds_dst);
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" :: "s"(m), "v"(off), "s"(base) : "memory");
}

template <typename OT, int NSLAB>
__global__ __launch_bounds__(512, 1) void lstm_wreg_kernel(LstmWregParams p) {
    constexpr int BM = 80, TM = 5, RG = BM / 8, KF = 2 * NSLAB, SLABB = BM * 128, PF = 3;
    __shared__ __attribute__((aligned(16))) unsigned char lds[NSLAB * SLABB];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, lq = lane >> 4, lr = lane >> 3, lpos = lane & 7;
    // the tiles_m workgroups of a gate-column block share blockIdx % 8 (one XCD's L2 under round-robin placement: speed only)
    int cb, rb;
    if ((p.tiles_n & 7) == 0) {
        const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
        cb = xcd * (p.tiles_n >> 3) + idx / p.tiles_m; rb = idx % p.tiles_m;
    } else {
        cb = blockIdx.x % p.tiles_n; rb = blockIdx.x / p.tiles_n;
    }
    const int m0 = rb * BM, n0 = cb * 128;
    const int N = 4 * p.Hh;

    // ---- row indices first (ordinary loads, consumed before any LDS-DMA is issued) ----------------------------------------------------
    unsigned xo[RG], ho[RG];                              // element offsets of the loader rows' x part and (parent's) h part
#pragma unroll
    for (int g = 0; g < RG; ++g) {
        const int m = min(m0 + g * 8 + lr, p.rows - 1);   // rows past the end: clamped (finite garbage in never-stored outputs)
        const int rl = m * p.row_mult;
        xo[g] = p.tokens ? (unsigned)p.tokens[(size_t)rl * p.tok_ld + p.tok_pos] * (unsigned)p.E : (unsigned)(m / p.x_div) * (unsigned)p.ldx;
        ho[g] = (unsigned)(p.hparent ? p.hparent[rl] : rl) * (unsigned)p.Hh;
    }
    int hp_e[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int m = min(m0 + 16 * i + l15, p.rows - 1);
        hp_e[i] = p.hparent ? p.hparent[m * p.row_mult] : m * p.row_mult;
    }
    // ---- epilogue operands (bias quad of this lane's hidden unit, the parents' cell state): requested BEFORE the LDS-DMA transfers -- their
    //      addresses depend on the index loads above, and a counted wait for those placed behind the transfers would wait for the transfers
    const int u = (n0 + 16 * wave) / 4 + lq;              // hidden unit of this lane's accumulator quads
    const float4 b4 = *reinterpret_cast<const float4*>(p.bias + n0 + 16 * wave + 4 * lq);
    float c0[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) c0[i] = p.c_prev ? p.c_prev[(size_t)hp_e[i] * p.Hh + u] : 0.f;
    // ---- the activation block: slab s = k 64 s .. (wholly in the x part or in the h part: E % 64 == 0), piece = 8 rows x 128 bytes --------
    {
        const uint16_t* xbase = p.tokens ? p.emb : p.x_rows;
        const unsigned swz8 = (unsigned)((lpos ^ lr) << 3);   // source chunk of LDS slot lpos in a row with (row & 7) == lr
#pragma unroll
        for (int sl = 0; sl < (NSLAB + 7) / 8; ++sl) {
            const int s = wave + 8 * sl;
            if (s < NSLAB) {
                const int k = 64 * s;
                unsigned char* dst = lds + s * SLABB;
                if (k < p.E) {
#pragma unroll
                    for (int g = 0; g < RG; ++g) lw_dma16(xbase, (xo[g] + k + swz8) * 2u, dst + g * 1024);
                } else if (p.h_prev) {
#pragma unroll
                    for (int g = 0; g < RG; ++g) lw_dma16(p.h_prev, (ho[g] + (k - p.E) + swz8) * 2u, dst + g * 1024);
                } else {
#pragma unroll
                    for (int g = 0; g < RG; ++g) dh_lds_dma16(lw_zero_page, dst + g * 1024);
                }
            }
        }
    }
    // ---- this wave's 16 gate rows x all K: KF fragments of 1 KB, straight into registers ---------------------------------------------------
    uint4 wf[KF];
    {
        const uint4* wsrc = p.wp + ((size_t)(n0 / 16 + wave)) * 64 + lane;
        const size_t fstep = (size_t)(N / 16) * 64;
#pragma unroll
        for (int f = 0; f < KF; ++f) wf[f] = wsrc[(size_t)f * fstep];
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's LDS-DMA pieces, fragments and operands have landed
    __syncthreads();

    // ---- 5 row tiles x KF k-steps; fragment reads PF tiles ahead of their MFMAs -----------------------------------------------------------------
    dh_f32x4 acc[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) acc[i] = dh_f32x4{0.f, 0.f, 0.f, 0.f};
    // LDS read bases: (k half) x (64 KB window); row 16 i + l15 has (row & 7) == (l15 & 7)
    unsigned rd_base[2][3];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int wdw = 0; wdw < 3; ++wdw) {
            rd_base[kk][wdw] = (unsigned)(l15 * 128 + (((kk * 4 + lq) ^ (l15 & 7)) << 4) + wdw * 61440);
            asm volatile("" : "+v"(rd_base[kk][wdw]));
        }
    uint4 fa[PF + 1];
    auto rd = [&](int t) {                                // t = 5 f + i: fragment step f = 2 s + kk, row tile i
        const int f = t / TM, i = t - f * TM, s = f >> 1, kk = f & 1;
        const int off = s * SLABB + i * 2048, wdw = off / 61440;       // 6 slabs per 61,440-byte window: offsets < 64 KB
        fa[t % (PF + 1)] = *reinterpret_cast<const uint4*>(lds + rd_base[kk][wdw] + (off - wdw * 61440));
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
    // ---- cell update in registers: acc[i] = (i, f, g, o) pre-activations of unit u for row m0 + 16 i + l15 -------------------------------------
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int m = m0 + 16 * i + l15;
        if (m >= p.rows) continue;
        const size_t rl = (size_t)m * p.row_mult;
        const float gi = acc[i][0] + b4.x, gf = acc[i][1] + b4.y, gg = acc[i][2] + b4.z, go = acc[i][3] + b4.w;
        const float c1 = lw_sigmoid(gf) * c0[i] + lw_sigmoid(gi) * lw_tanh(gg);
        const float h1 = lw_sigmoid(go) * lw_tanh(c1);
        const uint16_t hb = Op16<OT>::from_f32(h1);
        p.c_next[rl * p.Hh + u] = c1;
        p.h_next[rl * p.Hh + u] = hb;
        p.h_out[(size_t)m * p.ld_out + u] = hb;
    }
}
}  // namespace

extern "C" int dh_lstm_layer_wreg_supported(int E, int Hh) {
    const int K = E + Hh;
    return (E % 64) == 0 && (Hh % 32) == 0 && (K == 768 || K == 1024);
}

extern "C" int dh_lstm_layer_wreg(const void* x_rows, int ldx, int x_div, const void* emb, const int32_t* tokens, int tok_ld,
                                  int tok_pos, const void* h_prev, const float* c_prev, const int32_t* hparent, void* h_next,
                                  float* c_next, void* h_out, int ld_out, const void* w_packed, const float* b_il, int rows,
                                  int row_mult, int E, int Hh, int dtype, void* stream) {
    if (!DH_IS_16BIT(dtype)) return DH_ERR_UNSUPPORTED;
    DH_REQUIRE((x_rows || (emb && tokens)) && h_next && c_next && h_out && w_packed && b_il && dh_lstm_layer_wreg_supported(E, Hh));
    DH_REQUIRE(rows > 0 && row_mult > 0 && x_div > 0 && (ldx % 8) == 0 && ld_out >= Hh);
    DH_REQUIRE((h_prev == nullptr) == (c_prev == nullptr) && h_prev != h_next && c_prev != c_next);
    DH_REQUIRE(((uintptr_t)w_packed % 16) == 0 && ((uintptr_t)b_il % 16) == 0 && ((uintptr_t)x_rows % 16) == 0 &&
               ((uintptr_t)emb % 16) == 0 && ((uintptr_t)h_prev % 16) == 0);
    LstmWregParams p{};
    p.x_rows = (const uint16_t*)x_rows; p.ldx = ldx; p.x_div = x_div;
    p.emb = (const uint16_t*)emb; p.tokens = tokens; p.tok_ld = tok_ld; p.tok_pos = tok_pos;
    p.h_prev = (const uint16_t*)h_prev; p.c_prev = c_prev; p.hparent = hparent;
    p.h_next = (uint16_t*)h_next; p.c_next = c_next; p.h_out = (uint16_t*)h_out; p.ld_out = ld_out;
    p.wp = (const uint4*)w_packed; p.bias = b_il; p.rows = rows; p.row_mult = row_mult; p.E = E; p.Hh = Hh;
    p.tiles_m = dh_cdiv(rows, 80); p.tiles_n = (4 * Hh) / 128;
    const double K = E + Hh;
    DhProfScope prof("dh_lstm_layer_fused", 2.0 * rows * 4 * Hh * K, 2.0 * (rows * K + 4.0 * Hh * K) + 12.0 * rows * Hh, stream);
    const dim3 grid(p.tiles_m * p.tiles_n);
    hipStream_t s = (hipStream_t)stream;
    DH_DISPATCH_16(dtype, {
        if (E + Hh == 768) hipLaunchKernelGGL((lstm_wreg_kernel<T, 12>), grid, dim3(512), 0, s, p);
        else hipLaunchKernelGGL((lstm_wreg_kernel<T, 16>), grid, dim3(512), 0, s, p);
    });
    DH_LAUNCH_CHECK();
}
