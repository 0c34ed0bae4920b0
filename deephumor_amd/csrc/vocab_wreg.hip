// The classifier GEMM of a beam-search step, logits[M, V] = A[M, 512] W[V, 512]^T + b with the group maxima the samplers use
// (decoder.classifier of rnn_models.py:45 / transformers.py:489 inside generate(); dh_vocab_logits), with the WEIGHTS STREAMED FROM
// L2 INTO REGISTERS and the activation rows resident in LDS -- the structure of conv_s4.hip / conv_s3.hip.
//
// vocab_areg256_kernel (vocab_areg.h) keeps the activation fragments in registers and streams W through an 8-slab LDS ring: per 16 KB
// slab a wave pays a counted wait, a barrier, two LDS-DMA pieces and 16 ds_read_b128 for 32 MFMAs, every wave reads the WHOLE slab from
// LDS, and a wave's epilogue (bias, group maxima, 187 MB of fp32 logits per step) overlaps nothing but its SIMD partner, which is in its
// own epilogue at the same time: 65 us per step at 1,280 rows, MFMA pipe busy 0.29.  Here:
//   * a workgroup (4 waves, one per SIMD, one workgroup per CU) owns ONE block of 80 activation rows for its whole life: 80 x 512 x 2 B
//     = 80 KB in LDS (LDS-DMA, 128-byte rows per 64-k plane, XOR swizzle), staged once;
//   * the vocabulary is walked in chunks of 256 columns; a wave owns 64 of them (4 column tiles) and loads their weight fragments
//     straight from L2 -- the padded weights are stored once per plan in MFMA fragment order (dh_pack_mfma_fragments: one coalesced
//     1 KB load per 16 columns x 32 k) -- fifteen k-steps ahead in a ring of 16 (256 registers of the 512 a one-wave-per-SIMD workgroup owns): no LDS ring, no LDS-DMA in the loop, NO BARRIER;
//   * a k-step = 5 ds_read_b128 (the block's five row tiles) feeding 20 MFMAs: one LDS fragment read per FOUR MFMAs;
//   * the four waves run unsynchronised: one wave's epilogue (bias add, group maxima, full-line fp32 stores) runs under the other
//     SIMDs' MFMAs, and the next chunk's weight fragments are already in flight;
//   * the 16 row-block workgroups that walk the same chunks sit on ONE XCD (blockIdx % 8): a chunk's 256 KB of weights is fetched into
//     that L2 once.
// MEASURED (1,280 x 36,541 x 512, per launch): back-to-back 58-61 us against 64-67 us for vocab_areg256_kernel (group maxima only: 46 / 50);
// with the caches flushed between launches (320 MB read-modify-write) 88-91 against 72-75 us (92-99 before the chunk-major weight layout
// and the next-chunk touch below; plain instead of non-temporal stores: 97): what the cold launch pays is not found yet -- not the
// TLB reach of the weights, not the logits stores, not L2 misses of the ring loads.
// In the LSTM decode chain (C2) the weights survive in the Infinity Cache from one position to the next: classifier 1.92-1.96 against
// 2.11-2.16 ms per step, step 7.16-7.25 against 7.23-7.39 ms (three alternating runs) -- the default there; in the Transformer chain (~1 GB of KV cache per position in between) the
// step takes the same time with either kernel: opt-in (DH_VOCAB_WREG_TRANSFORMER=1).  A ring of 8 lost in the chain (2.39 vs 2.11 ms of
// classifier time per C2 step); plain instead of non-temporal logits stores lost there too (the 187 MB evict the weights).
// Same MFMA chain per output as vocab_areg256_kernel / vocab_logits_kernel (weights = A operand, k ascending, one accumulator per
// output), same bias add: BIT-IDENTICAL logits and group maxima, padding columns included (weight rows / bias entries past V are copies
// of row V - 1, a group that starts past V gets -inf).
#include <stdlib.h>
#include "common.h"
#include "prof.h"
#include "options.h"

namespace {
struct VwParams {
    const uint16_t* A; int lda;
    const uint4* wp;                                      // fragments of the padded weights [Vpad][512], chunk-major: [((chunk * 16 + s) * 16 + tile) * 64 + lane]
    const float* bias;                                    // padded [Vpad]
    float* C; int ldc; float* gmax; int gmax_ld;
    int M, V, NT, nrb, nchunk, nt;
    int nrb_real;                                         // row blocks that exist (ceil(M / 80)); nrb = that rounded up to a power of two
};

__device__ __forceinline__ float4 vw_swap1(float4 v) {    // lane ^ 1
    float4 r;
    r.x = dpp_get<0xB1>(v.x); r.y = dpp_get<0xB1>(v.y); r.z = dpp_get<0xB1>(v.z); r.w = dpp_get<0xB1>(v.w);
    return r;
}
// one 32-column half of a 16-row accumulator block as FULL 128-byte lines (gemm_bf16.hip, store_half_full_lines: same instruction
// stream): lane pairs (l15, l15 ^ 1) swap one quad, an instruction then covers 8 rows x 128 bytes
__device__ __forceinline__ void vw_store_half(float* r_even, size_t ldc, float4 va, float4 vb, bool odd, bool nt, bool ok0, bool ok1) {
#define VW_SEL4(c, a, b) make_float4((c) ? (a).x : (b).x, (c) ? (a).y : (b).y, (c) ? (a).z : (b).z, (c) ? (a).w : (b).w)
    const float4 own = VW_SEL4(odd, vb, va);
    const float4 rcv = vw_swap1(VW_SEL4(odd, va, vb));
    // non-temporal: 187 MB of logits per step stream THROUGH the L2 that holds the weight chunks the other row blocks are about to read
    typedef float vw_f4 __attribute__((ext_vector_type(4)));
    const float4 lo = VW_SEL4(odd, rcv, own), hi = VW_SEL4(odd, own, rcv);
    if (nt) {
        if (ok0) __builtin_nontemporal_store(vw_f4{lo.x, lo.y, lo.z, lo.w}, reinterpret_cast<vw_f4*>(r_even));
        if (ok1) __builtin_nontemporal_store(vw_f4{hi.x, hi.y, hi.z, hi.w}, reinterpret_cast<vw_f4*>(r_even + ldc));
    } else {
        if (ok0) *reinterpret_cast<float4*>(r_even) = lo;
        if (ok1) *reinterpret_cast<float4*>(r_even + ldc) = hi;
    }
#undef VW_SEL4
}

template <typename OT, bool PREFETCH>
__global__ __launch_bounds__(256, 1) void vocab_wreg_kernel(VwParams p) {
    constexpr int RB = 80, TM = 5, TN = 4, KS = 16, CB = 8, RING = 16, PF = 3;
    constexpr int PLANE = RB * 128;                       // 10 KB per 64-k plane
    __shared__ __attribute__((aligned(16))) unsigned char lds[CB * PLANE];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, lq = lane >> 4, lr = lane >> 3, lpos = lane & 7;

    // workgroup -> (row block, chunk walker): the nrb row blocks of a walker share blockIdx % 8
    const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3, gpx = 32 / p.nrb;
    if (local >= gpx * p.nrb) return;
    const int rb = local % p.nrb, cg = xcd * gpx + local / p.nrb, ncg = 8 * gpx;
    if (cg >= p.nchunk || rb >= p.nrb_real) return;       // (row counts that are no power-of-two number of blocks: the padding blocks idle)
    const int m0 = rb * RB;

    // ---- weight fragments of the first RING - 1 k-steps of the first chunk (plain loads: the compiler counts them) -----------------------
    // chunk-major fragments: a chunk's 16 k-steps x 16 tiles are 256 KB contiguous (k-step-major over the whole vocabulary, every k-step
    // of a chunk sat 2.3 MB from the next: 16 translations per chunk, and +25 us per launch with a cold TLB)
    constexpr size_t sstep = 16 * 64;                     // uint4 elements between k-steps
    const uint4* wbase = p.wp + (size_t)(wave * TN) * 64 + lane;                  // chunk c, step s, tile j: wbase[((c * 16 + s) * 16 + j) * 64]
    uint4 wq[RING][TN];
    {
        const uint4* w0 = wbase + (size_t)cg * 16 * sstep;
#pragma unroll
        for (int s = 0; s < RING - 1; ++s)
#pragma unroll
            for (int j = 0; j < TN; ++j) wq[s][j] = w0[(size_t)s * sstep + j * 64];
    }
    // ---- the row block: 8 planes x 10 groups of 8 rows, 20 pieces per wave; rows past M (last block of a row count that is no multiple
    //      of 80) re-read row M - 1, their outputs are not stored ----------------------------------------------------------------------------
#pragma unroll
    for (int u = 0; u < CB * (RB / 8) / 4; ++u) {
        const int pc = wave * (CB * (RB / 8) / 4) + u, cb = pc / (RB / 8), g = pc - cb * (RB / 8);
        dh_lds_dma16(p.A + (size_t)min(m0 + g * 8 + lr, p.M - 1) * p.lda + cb * 64 + ((lpos ^ lr) << 3), lds + cb * PLANE + g * 1024);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's pieces (and its first fragments) have landed
    __syncthreads();

    // LDS read bases per (k half, plane half): row 16 i + l15 has (row & 7) == (l15 & 7); offsets below stay inside the ds_read field
    unsigned rd_base[2][2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            rd_base[kk][hf] = (unsigned)(l15 * 128 + (((kk * 4 + lq) ^ (l15 & 7)) << 4) + hf * 4 * PLANE);
            asm volatile("" : "+v"(rd_base[kk][hf]));
        }

    const uint4* wnext = wbase + (size_t)cg * 16 * sstep + (size_t)(RING - 1) * sstep;      // fragments of step s + RING - 1

#pragma unroll 1
    for (int c = cg; c < p.nchunk; c += ncg) {
        const bool more = c + ncg < p.nchunk;
        const int n0 = c * 256 + wave * 64;               // this wave's 64 columns
        // ONE of the walker's row-block workgroups (a different one per chunk) touches the NEXT chunk's weight lines now, a whole chunk
        // time ahead: when the 37 MB of weights are not in the Infinity Cache (another 200+ MB moved since the last position) the 16
        // lock-stepped workgroups otherwise all miss on the same lines with only 60 KB per wave in flight.  64 lanes x 8 loads of 4 bytes
        // = the wave's 512 lines (16 k-steps x 4 KB); the values are not used.
        uint32_t pfv[8];
        const bool pf = PREFETCH && more && rb == (((c - cg) / ncg) & (p.nrb - 1));
        if (pf) {
            const unsigned char* nb = reinterpret_cast<const unsigned char*>(wbase - lane + (size_t)(c + ncg) * 16 * sstep);
#pragma unroll
            for (int e = 0; e < 8; ++e)
                pfv[e] = *reinterpret_cast<const uint32_t*>(nb + (size_t)(2 * e + (lane >> 5)) * sstep * 16 + (lane & 31) * 128);
        }
        float4 b4[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j) b4[j] = p.bias ? *reinterpret_cast<const float4*>(p.bias + n0 + 16 * j + 4 * lq) : make_float4(0.f, 0.f, 0.f, 0.f);
        dh_f32x4 acc[TM][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = dh_f32x4{0.f, 0.f, 0.f, 0.f};
        uint4 fa[PF + 1];
        auto rd = [&](int t) {                            // t = TM s + i: k-step s (plane s / 2, half s % 2), row tile i
            const int s = t / TM, i = t - s * TM;
            fa[t % (PF + 1)] = *reinterpret_cast<const uint4*>(lds + rd_base[s & 1][s >> 3] + (((s >> 1) & 3) * PLANE + i * 2048));
        };
#pragma unroll
        for (int t = 0; t < PF; ++t) rd(t);
#pragma unroll
        for (int t = 0; t < KS * TM; ++t) {
            const int s = t / TM, i = t - s * TM;
            if (i == 0) {                                 // the weight fragments RING - 1 k-steps ahead: this chunk's, or the next chunk's first
                if (s + RING - 1 == KS) wnext = wbase + (size_t)(c + ncg) * 16 * sstep;
                if (s + RING - 1 < KS || more) {
#pragma unroll
                    for (int j = 0; j < TN; ++j) wq[(s + RING - 1) % RING][j] = wnext[j * 64];
                    wnext += sstep;
                }
            }
            if (t + PF < KS * TM) rd(t + PF);
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = Op16<OT>::mfma(wq[s % RING][j], fa[t % (PF + 1)], acc[i][j]);
            __builtin_amdgcn_sched_barrier(0);
        }
        // ---- epilogue: acc[i][j][r] = row m0 + 16 i + l15, column n0 + 16 j + 4 lq + r.  (The epilogue of chunk c riding on the k-steps of
        //      chunk c + 1 -- two accumulator sets, one half row tile per k-step -- was measured: 62.6 against 58.2 us; the second
        //      accumulator set costs AGPR <-> VGPR moves in the loop that the spread-out stores do not buy back.) ------------------------------
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int m = m0 + 16 * i + l15, m_even = m0 + 16 * i + (l15 & ~1);
            float* r_even = p.C + (size_t)m_even * p.ldc + n0 + ((l15 & 1) ? 16 : 0) + 4 * lq;
            float mxv = -INFINITY;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                float4 va, vb;
                va.x = acc[i][2 * h][0] + b4[2 * h].x; va.y = acc[i][2 * h][1] + b4[2 * h].y;
                va.z = acc[i][2 * h][2] + b4[2 * h].z; va.w = acc[i][2 * h][3] + b4[2 * h].w;
                vb.x = acc[i][2 * h + 1][0] + b4[2 * h + 1].x; vb.y = acc[i][2 * h + 1][1] + b4[2 * h + 1].y;
                vb.z = acc[i][2 * h + 1][2] + b4[2 * h + 1].z; vb.w = acc[i][2 * h + 1][3] + b4[2 * h + 1].w;
                mxv = fmaxf(fmaxf(mxv, fmaxf(fmaxf(va.x, va.y), fmaxf(va.z, va.w))), fmaxf(fmaxf(vb.x, vb.y), fmaxf(vb.z, vb.w)));
                if (p.C) vw_store_half(r_even + 32 * h, p.ldc, va, vb, l15 & 1, p.nt != 0, m_even < p.M, m_even + 1 < p.M);
            }
            mxv = fmaxf(mxv, __shfl_xor(mxv, 16, 64));
            mxv = fmaxf(mxv, __shfl_xor(mxv, 32, 64));
            if (p.gmax && lq == 0 && m < p.M) p.gmax[(size_t)m * p.gmax_ld + n0 / 64] = n0 < p.V ? mxv : -INFINITY;      // -inf for a group that starts past V
        }
        if (pf) {
#pragma unroll
            for (int e = 0; e < 8; ++e) asm volatile("" :: "v"(pfv[e]));
        }
    }
}
}  // namespace

// 1 when dh_vocab_logits_wreg takes the shape: K = 512; M = 80 x {1, 2, 4, 8, 16, 32} rows, or (round 5: the small-shard regime) ANY
// M <= 640 -- the row blocks are then padded to a power of two with idle workgroups and the last block's missing rows are masked; a
// logits row stride and a group-maxima stride that cover the vocabulary padded to whole 256-column chunks
extern "C" int dh_vocab_logits_wreg_supported(int M, int V, int K, int ldl, int gm_ld) {
    if (K != 512 || M <= 0 || V <= 0) return 0;
    const int nrb = dh_cdiv(M, 80), vpad = dh_cdiv(V, 256) * 256;
    const bool pow2_blocks = (M % 80) == 0 && (nrb == 1 || nrb == 2 || nrb == 4 || nrb == 8 || nrb == 16 || nrb == 32);
    return (pow2_blocks || M <= 640) && (ldl == 0 || (ldl >= vpad && (ldl % 4) == 0)) && gm_ld >= vpad / 64;
}

// logits [M, ldl] fp32 (may be NULL: group maxima only) and group_max [M, gm_ld] as dh_vocab_logits writes them, from
// w_packed = dh_pack_mfma_fragments(W padded to Vpad = ceil(V / 256) * 256 rows with copies of row V - 1), re-ordered chunk-major
// ([k-step][Vpad / 16 tiles] -> [Vpad / 256 chunks][k-step][16 tiles]; deephumor_amd.hip.pack_vocab_weights), and bias_padded [Vpad]
// (padded likewise; NULL = no bias).  Bit-identical to dh_vocab_logits on the columns [0, Vpad).
extern "C" int dh_vocab_logits_wreg(const void* A, int lda, const void* w_packed, const float* bias_padded, float* logits, int ldl,
                                    float* group_max, int gm_ld, int M, int V, int K, int dtype, void* stream) {
    if (!DH_IS_16BIT(dtype)) return DH_ERR_UNSUPPORTED;
    DH_REQUIRE(A && w_packed && group_max && dh_vocab_logits_wreg_supported(M, V, K, logits ? ldl : 0, gm_ld));
    DH_REQUIRE(lda >= K && (lda % 8) == 0 && ((uintptr_t)A % 16) == 0 && ((uintptr_t)w_packed % 16) == 0 && ((uintptr_t)bias_padded % 16) == 0 &&
               ((uintptr_t)logits % 16) == 0);
    VwParams p{};
    p.A = (const uint16_t*)A; p.lda = lda; p.wp = (const uint4*)w_packed; p.bias = bias_padded; p.C = logits; p.ldc = ldl;
    p.gmax = group_max; p.gmax_ld = gm_ld; p.M = M; p.V = V; p.nchunk = dh_cdiv(V, 256); p.NT = p.nchunk * 16;
    p.nrb_real = dh_cdiv(M, 80);
    p.nrb = 1;
    while (p.nrb < p.nrb_real) p.nrb *= 2;
    dh_prof_set_tag("vocab");
    dh_prof_set_dims(M, V, K);
    DhProfScope prof("dh_linear", 2.0 * M * V * K, 2.0 * ((double)M * K + (double)V * K) + 4.0 * M * V, stream);
    hipStream_t s = (hipStream_t)stream;
    p.nt = 1;                                            // non-temporal logits stores (plain ones evict the weights: measured slower)
    DH_DISPATCH_16(dtype, hipLaunchKernelGGL((vocab_wreg_kernel<T, true>), dim3(256), dim3(256), 0, s, p));
    DH_LAUNCH_CHECK();
}
