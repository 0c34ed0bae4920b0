// The decoder LAYERS of one decode position of the 16-bit Transformer chain as ONE persistent launch (round 6).
// (DecoderLayer.forward, transformers.py:343-377, applied to the rows of one position inside generate's loop :547-573.)
//
// The launch chain of runtime.hip runs a layer as 8 dependent launches -- fc_q|k|v, self-attention, fc_o, fc_q, cross-attention,
// enc fc_o, fc_1, fc_2 -- 48 per position at ~8 us each whatever they compute (a boundary + the cold first
// touch of the predecessor's rows + the kernel's own fill / drain).  Round 5's four-GEMM persistent chain kept that structure (32
// workgroups of an XCD per phase, every phase waiting for the whole group) and lost.  Here the work is cut the other way:
//   * a CLUSTER of 8 workgroups (all on one XCD: blockIdx b -> XCD b % 8, cluster (b % 8) * 4 + b / 64, member (b / 8) % 8) owns 40 rows
//     (8 images x beam 5) through ALL layers; member w owns HEAD w and COLUMN BLOCK w (64 columns) of every 512-wide tensor;
//   * what a member needs from the others is exactly six full-row hand-overs per layer, each a barrier among 8 workgroups through one
//     counter in their XCD's L2 (plain stores + s_waitcnt + one relaxed atomic; L1 invalidate behind it; no L2 write-back):
//       [q_w | k_w | v_w = LN(x) Wqkv_w ; self-attention of head w over the 40 rows]   -> att[:, w]        B1
//       [o_w = LN(x)_w + att Wo_w + bo, statistics]                                      -> o[:, w], st1    B2
//       [q_w = LN(o) Wq_w ; cross-attention of head w over the 8 images' patches]        -> att[:, w]        B3
//       [y2_w = LN(o)_w + att Weo_w + beo, statistics]                                   -> y2[:, w], st2    B4
//       [ff[:, 4w..4w+3] = relu(LN(y2) W1 + b1)]  (4 column blocks of 2,048)             -> ff                B5
//       [x_w = LN(y2)_w + ff W2_w + b2, statistics]  (K = 2,048 in four chunks)          -> x[:, w], st0      B6
//     -- the QKV projection feeds the self-attention and fc_q feeds the cross-attention INSIDE a workgroup (head w's columns are
//     column block w), so two of the chain's eight seams need no hand-over at all;
//   * every GEMM block is linear_wreg.hip's 64-column x 40-row block (weights stationary in registers, the [40 x 512] activation
//     block whole in LDS by LDS-DMA), same MFMA operand contents, k order and epilogue arithmetic: BIT-IDENTICAL to the chain
//     (tests/test_bf16_gpu.py: every intermediate buffer and whole decodes); the next block's weight fragments are requested while
//     the current block computes (two register sets), fc_2's K chunks are double-buffered in LDS;
//   * the attention phases run the arithmetic of attn_decode_reg_kernel / attn_cross_mfma_kernel (attn_items.h) with the loads of
//     several (row, head) items of a wave in flight together (4 waves per CU have to hide what 40 waves per CU hide in the
//     stand-alone launches).
// Placement: correctness does NOT depend on the dispatcher: every member publishes the XCD it runs on (HW_REG_XCC_ID); a cluster whose
// members disagree (never observed) adds an agent-scope release (L2 write-back) to its barriers.  All 256 workgroups must become
// resident (256 CUs, one workgroup each); waits are bounded (error word, checked by the host with the beam error word).
#include <stdlib.h>
#include "common.h"
#include "prof.h"
#include "attn_items.h"

namespace {
constexpr int RL = 40, BN = 64, NT = 256, NW = 4, TM = 3, RG = RL / 8, SLABB = RL * 128, ABYTES = 8 * SLABB;   // 40,960 bytes per [40 x 512] block
constexpr int KF = 16, PF = 3, CHUNKS = BN / 8, SLOTS = BN / 4, EP_IT = (RL * CHUNKS + NT - 1) / NT;
constexpr int EP_BYTES = RL * BN * 4 + RL * 8;          // fp32 staging tile + the block rows' (mean, rstd)
constexpr int LDS_BYTES = 2 * ABYTES + EP_BYTES;
constexpr int SPIN_LIMIT = 1 << 20;

struct DlLayer {                                        // one decoder layer (device-resident table, built once per run)
    const uint4 *wqkv_pk, *wo_pk, *wq_pk, *weo_pk, *w1_pk, *w2_pk;
    const float *bqkv, *bo, *bq, *beo, *b1, *b2;        // (bqkv / bq / b1: with the folded LayerNorm's beta term where one is folded)
    const float *cs_qkv, *cs_q, *cs_1;                  // column sums of the gamma-folded weights (cs_qkv: NULL in layer 0)
    const float *ln1_g, *ln1_b, *ln2_g, *ln2_b, *ln3_g, *ln3_b;
    float ln1_eps, ln2_eps, ln3_eps, sa_scale, ea_scale; int pad_;
    uint16_t *kcache, *vcache; const uint16_t *kp, *vt;
};

struct DlParams {
    const DlLayer* layers; int n_layers;
    uint16_t *x, *qkv, *att, *o, *q, *ff, *y2; float2 *st0, *st1, *st2;
    const int32_t* tokens; int tok_ld; const int32_t* src; int src_ld; const uint8_t* keymask;
    int rows, rows_per_img, row_mult, rows_total, t, S, pad_index, kp_dperm, n_rb, iters, dbg;
    unsigned* sync;                                     // [0, 32) cluster counters, [32, 64) their bases, [64, 320) XCD of every workgroup + 1, 320 error
};

struct GemmP {                                          // one GEMM phase (the LwParams of linear_wreg.hip)
    const uint16_t* A; int lda; const uint4* wp; const float* bias; const uint16_t* res; int ldres; uint16_t* C; int ldc; int M, N, relu;
    const float2* a_stats; float a_eps; const float* a_colsum;
    const float2* r_stats; float r_eps; const float* r_gamma; const float* r_beta; float2* o_stats;
};

__device__ __forceinline__ void dl_dma16(const void* base, unsigned off, void* lds_dst) {
    const unsigned m = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(dh_lptr_t)lds_dst);
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" :: "s"(m), "v"(off), "s"(base) : "memory");
}

// ---- hand-over among the 8 workgroups of a cluster -------------------------------------------------------------------------------
struct Cluster { unsigned* ctr; unsigned* err; unsigned target; int slow; };

__device__ __forceinline__ void cluster_barrier(Cluster& c) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's stores have reached the XCD's L2 (the L1 is write-through)
    __syncthreads();
    c.target += 8u;
    if (threadIdx.x == 0) {
        if (c.slow) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");      // members on different XCDs: write the L2 back
        __hip_atomic_fetch_add(c.ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int spin = 0;
        while ((int)(__hip_atomic_load(c.ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - c.target) < 0) {
            // bounded: ~1 s without the other members (fewer than 256 resident workgroups), or another cluster has already given up
            if (++spin > SPIN_LIMIT || ((spin & 1023) == 0 && __hip_atomic_load(c.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
                __hip_atomic_fetch_or(c.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
            }
            __builtin_amdgcn_s_sleep(1);
        }
    }
    __syncthreads();
    asm volatile("buffer_inv sc1" ::: "memory");       // this CU's L1 may hold lines the other members have re-written
}

// ---- one GEMM phase of a member: NU output blocks (column blocks cb0, cb0 + cbs, ...) of row block rb, K = 512 KQ -------------------
// LNX 0: (deferred LayerNorm on the A rows | plain) + optional ReLU; 1: residual (optionally pre-LayerNorm) + statistics of the output
// rows.  The arithmetic of every output is lw_item's (linear_wreg.hip), line by line.  NU > 1: the blocks share the activation block;
// KQ = 4 (NU = 1): the activation chunks alternate between the two LDS buffers.
template <typename OT, int LNX, int NU, int KQ>
__device__ __forceinline__ void gemm_phase(const GemmP& p, const int rb, const int cb0, const int cbs, unsigned char* lds) {
    static_assert(NU == 1 || KQ == 1, "either several blocks or several K chunks");
    constexpr int NS = NU * KQ;                         // steps: (block, chunk)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, lq = lane >> 4, lr = lane >> 3, lpos = lane & 7;
    const int m0 = rb * RL;
    float* const ep = reinterpret_cast<float*>(lds + 2 * ABYTES);
    float2* const row_stat = reinterpret_cast<float2*>(lds + 2 * ABYTES + RL * BN * 4);

    // ---- activation chunk c of the block rows -> LDS buffer (c & 1): slab s = k 64 s .. + 63 of all RL rows, piece = 8 rows x 128 bytes;
    //      wave w stages slabs w, w + 4 ----
    unsigned ro[RG];
    const unsigned swz = (unsigned)((lpos ^ lr) << 4);
#pragma unroll
    for (int g = 0; g < RG; ++g) ro[g] = (unsigned)min(m0 + g * 8 + lr, p.M - 1) * (unsigned)p.lda * 2u + swz;
    auto stage_a = [&](int c) {
        unsigned char* dst = lds + (c & 1) * ABYTES;
#pragma unroll
        for (int sl = 0; sl < 2; ++sl) {
            const int s = wave + NW * sl;
#pragma unroll
            for (int g = 0; g < RG; ++g) dl_dma16(p.A, ro[g] + 128u * (unsigned)(8 * c + s), dst + s * SLABB + g * 1024);
        }
    };
    // ---- this wave's 16 weight rows x 512 k of step st: 16 fragments of 1 KB straight into registers ----
    uint4 wf[2][KF];
    const size_t fstep = (size_t)(p.N / 16) * 64;
    auto load_w = [&](int st, uint4 (&dst)[KF]) {
        const int u = st / KQ, kq = st - u * KQ;
        const uint4* wsrc = p.wp + ((size_t)kq * KF * (p.N / 16) + (size_t)((cb0 + u * cbs) * 4 + wave)) * 64 + lane;
#pragma unroll
        for (int f = 0; f < KF; ++f) dst[f] = wsrc[(size_t)f * fstep];
    };

    // ---- phase start: the epilogue's row operands (ordinary loads first: vmcnt retires in order), chunk 0, the first weights ----
    float4 a_raw[4];
    const bool a_ln = LNX == 0 && p.a_stats != nullptr;
    if (LNX == 0 && a_ln && tid < RL) ln_load(p.a_stats + (size_t)min(m0 + tid, p.M - 1) * 8, 8, a_raw);
    uint4 rq[EP_IT];
    float4 r_raw[EP_IT][4], rg[EP_IT][2], rb4[EP_IT][2];
    const bool r_ln = LNX == 1 && p.r_stats != nullptr;
    if (LNX == 1) {
        const int n0 = cb0 * BN;
#pragma unroll
        for (int it = 0; it < EP_IT; ++it) {
            const int c = min(tid + it * NT, RL * CHUNKS - 1), row = c / CHUNKS, ch = c - row * CHUNKS;
            const int m = min(m0 + row, p.M - 1), n = n0 + ch * 8;
            rq[it] = *reinterpret_cast<const uint4*>(p.res + (size_t)m * p.ldres + n);
            if (r_ln) {
                ln_load(p.r_stats + (size_t)m * 8, 8, r_raw[it]);
                rg[it][0] = *reinterpret_cast<const float4*>(p.r_gamma + n); rg[it][1] = *reinterpret_cast<const float4*>(p.r_gamma + n + 4);
                rb4[it][0] = *reinterpret_cast<const float4*>(p.r_beta + n); rb4[it][1] = *reinterpret_cast<const float4*>(p.r_beta + n + 4);
            }
        }
    }
    stage_a(0);
    load_w(0, wf[0]);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float2 my_stat = make_float2(0.f, 1.f);            // (mean, rstd) of block row `tid`
    if (a_ln && tid < RL) ln_math(a_raw, 8, p.a_eps, my_stat.x, my_stat.y);
    float r_mu[EP_IT], r_rs[EP_IT];
    if (r_ln) {
#pragma unroll
        for (int it = 0; it < EP_IT; ++it) ln_math(r_raw[it], 8, p.r_eps, r_mu[it], r_rs[it]);
    }
    if (a_ln && tid < RL) row_stat[tid] = my_stat;      // (the staging area is free: nobody is in an epilogue)
    __syncthreads();
    float a_mu[TM], a_rs[TM];
    if (a_ln) {
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const float2 ms = row_stat[min(16 * i + l15, RL - 1)];
            a_mu[i] = ms.x; a_rs[i] = ms.y;
        }
    }
    // LDS read bases of the MFMA fragments: (k half) [x last-tile variant]; row 16 i + l15 has (row & 7) == (l15 & 7)
    unsigned rd_base[2], rd_last[2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        rd_base[kk] = (unsigned)(l15 * 128 + (((kk * 4 + lq) ^ (l15 & 7)) << 4));
        asm volatile("" : "+v"(rd_base[kk]));
        rd_last[kk] = (unsigned)((l15 & 7) * 128 + (((kk * 4 + lq) ^ (l15 & 7)) << 4));
        asm volatile("" : "+v"(rd_last[kk]));
    }

    dh_f32x4 acc[TM];
#pragma unroll
    for (int st = 0; st < NS; ++st) {
        const int u = st / KQ, kq = st - u * KQ;
        const int cb = cb0 + u * cbs, n0 = cb * BN;
        if (kq == 0) {
#pragma unroll
            for (int i = 0; i < TM; ++i) acc[i] = dh_f32x4{0.f, 0.f, 0.f, 0.f};
        }
        if (KQ > 1 && st > 0) {                          // chunk st of the activation rows (requested during the previous chunk) has landed
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
        // the next step's operands: requested now, used after this step's MFMAs
        if (st + 1 < NS) {
            if (KQ > 1) stage_a(st + 1);
            load_w(st + 1, wf[(st + 1) & 1]);
        }
        float4 b4 = *reinterpret_cast<const float4*>(p.bias + n0 + 16 * wave + 4 * lq);
        float4 cs4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (a_ln) cs4 = *reinterpret_cast<const float4*>(p.a_colsum + n0 + 16 * wave + 4 * lq);
        // ---- TM row tiles x KF k-steps; fragment reads PF steps ahead of their MFMAs ----
        const unsigned char* Ab = lds + ((KQ > 1 ? st : 0) & 1) * ABYTES;
        uint4 fa[PF + 1];
        auto rd = [&](int t) {                           // t = TM f + i: fragment step f = 2 s + kk, row tile i
            const int f = t / TM, i = t - f * TM, s = f >> 1, kk = f & 1;
            const unsigned base = (i == TM - 1) ? rd_last[kk] : rd_base[kk];
            fa[t % (PF + 1)] = *reinterpret_cast<const uint4*>(Ab + base + (s * SLABB + i * 2048));
        };
#pragma unroll
        for (int t = 0; t < PF; ++t) rd(t);
#pragma unroll
        for (int t = 0; t < KF * TM; ++t) {
            const int f = t / TM, i = t - f * TM;
            if (t + PF < KF * TM) rd(t + PF);
            acc[i] = Op16<OT>::mfma(wf[st & 1][f], fa[t % (PF + 1)], acc[i]);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (kq + 1 < KQ) continue;
        __syncthreads();                                 // the previous block's epilogue is done with the staging tile
        // ---- epilogue (lw_item's): acc[i][r] = C[m0 + 16 i + l15][n0 + 16 wave + 4 lq + r], staged as fp32 rows (XOR-swizzled 16-byte slots) ----
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int row = 16 * i + l15, slot = 4 * wave + lq;
            if (row >= RL) continue;
            float4 v;
            if (a_ln) {                                  // rstd * (acc - mu * colsum) + bias'
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
                    for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
                }
                if (ok) store16(reinterpret_cast<OT*>(p.C + (size_t)m * p.ldc + n), v);
            } else {
                const uint32_t w4[4] = {rq[it].x, rq[it].y, rq[it].z, rq[it].w};
                float rr[8];
#pragma unroll
                for (int e = 0; e < 4; ++e) Op16<OT>::unpack2(w4[e], rr[2 * e], rr[2 * e + 1]);
                if (r_ln) {
                    const float g8[8] = {rg[it][0].x, rg[it][0].y, rg[it][0].z, rg[it][0].w, rg[it][1].x, rg[it][1].y, rg[it][1].z, rg[it][1].w};
                    const float b8[8] = {rb4[it][0].x, rb4[it][0].y, rb4[it][0].z, rb4[it][0].w, rb4[it][1].x, rb4[it][1].y, rb4[it][1].z, rb4[it][1].w};
#pragma unroll
                    for (int e = 0; e < 8; ++e) rr[e] = fmaf((rr[e] - r_mu[it]) * r_rs[it], g8[e], b8[e]);
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] += rr[e];
                if (p.relu) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
                }
                // statistics of the ROUNDED values; 8 consecutive lanes = the 8 chunks of one (row, 64-column tile): every lane takes part
                float s1 = 0.f;
#pragma unroll
                for (int e = 0; e < 8; ++e) { v[e] = Op16<OT>::to_f32(Op16<OT>::from_f32(v[e])); s1 += v[e]; }
                const float mean = sum8(s1) * (1.0f / 64.0f);
                float s2 = 0.f;
#pragma unroll
                for (int e = 0; e < 8; ++e) { const float d = v[e] - mean; s2 = fmaf(d, d, s2); }
                s2 = sum8(s2);
                if (ok) {
                    store16(reinterpret_cast<OT*>(p.C + (size_t)m * p.ldc + n), v);
                    if ((ch & 7) == 0) p.o_stats[(size_t)m * (p.N / 64) + (n >> 6)] = make_float2(mean, s2);
                }
            }
        }
    }
    __syncthreads();                                     // every wave is done with the activation buffers and the staging tile
}

// ---- self-attention of head h over rows [m0, m0 + 40) of this position (attn_decode_reg_kernel's arithmetic per (row, head)) -----------
// wave w takes rows w, w + 4, ...; the loads of NB rows are in flight together: ancestor / token indices of all NB, then their K, V and q.
template <typename T, int NIT, int NB>
__device__ __forceinline__ void self_attn_rows(const DlParams& P, const DlLayer& L, const int m0, const int h) {
    constexpr int DH = 64, LPK = 8, KPI = 8;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int D = 512, Lk = P.t + 1, t = P.t;
    const int kg = lane / LPK, dc = lane % LPK;
    const T* qb = reinterpret_cast<const T*>(P.qkv);
    T* kc = reinterpret_cast<T*>(L.kcache);
    T* vc = reinterpret_cast<T*>(L.vcache);
    const int n_rows = min(RL, P.rows - m0);
    for (int r0 = wave; r0 < n_rows; r0 += NW * NB) {
        int phys[NB][NIT], aux[NB][NIT];
        bool rok[NB];
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const int r = r0 + b * NW;
            rok[b] = r < n_rows;
            const int rc = m0 + (rok[b] ? r : r0), rl = rc * P.row_mult;
#pragma unroll
            for (int it = 0; it < NIT; ++it) { phys[b][it] = 0; aux[b][it] = 0; }
            if (t > 0) {
#pragma unroll
                for (int it = 0; it < NIT; ++it) phys[b][it] = P.src[(size_t)rl * P.src_ld + min(it * KPI + kg, t - 1)];
                if (P.tokens) {
#pragma unroll
                    for (int it = 0; it < NIT; ++it) aux[b][it] = P.tokens[(size_t)rl * P.tok_ld + min(max(it * KPI + kg - 1, 0), t - 1)];
                }
            }
        }
        Raw8<T> kr[NB][NIT], vr[NB][NIT], qr[NB];        // (q kept packed until its row's arithmetic: 4 registers instead of 8)
        bool live[NIT];
#pragma unroll
        for (int it = 0; it < NIT; ++it) live[it] = it * KPI + kg < Lk;
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const int rc = m0 + (rok[b] ? r0 + b * NW : r0);
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int j = it * KPI + kg;
                if (live[it]) {
                    const T *kptr, *vptr;
                    if (j < t) {
                        const size_t off = ((size_t)j * P.rows_total + phys[b][it]) * D + h * DH + dc * 8;
                        kptr = kc + off; vptr = vc + off;
                    } else {
                        kptr = qb + (size_t)rc * (3 * D) + D + h * DH + dc * 8;
                        vptr = qb + (size_t)rc * (3 * D) + 2 * D + h * DH + dc * 8;
                    }
                    raw_load(kptr, kr[b][it]);
                    raw_load(vptr, vr[b][it]);
                }
            }
            raw_load(qb + (size_t)rc * (3 * D) + h * DH + dc * 8, qr[b]);
        }
        __builtin_amdgcn_sched_barrier(0);               // every load of the NB rows is issued before any of the arithmetic below
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            if (!rok[b]) continue;                       // (wave-uniform)
            const int rc = m0 + r0 + b * NW, rl = rc * P.row_mult;
            float e[NIT], qv[8];
            raw_unpack(qr[b], qv);
            float mx = -INFINITY;
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                e[it] = -INFINITY;
                if (live[it]) {
                    const int j = it * KPI + kg;
                    const bool masked = (j >= 1) && P.tokens && (aux[b][it] == P.pad_index);
                    float kk[8];
                    raw_unpack(kr[b][it], kk);
                    float a = 0.f;
#pragma unroll
                    for (int u = 0; u < 8; ++u) a = fmaf(kk[u], qv[u], a);
                    a = sum8(a);                          // DPP fold over the row's 8 chunk lanes
                    e[it] = masked ? -1e8f : SmMath<T>::div(a, L.sa_scale);
                }
                mx = fmaxf(mx, e[it]);
            }
            mx = wave_max(mx);
            float sum = 0.f;
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                e[it] = live[it] ? SmMath<T>::exp(e[it] - mx) : 0.f;
                sum += e[it];
            }
            sum = wave_sum(sum) / (float)LPK;
            float o8[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) o8[u] = 0.f;
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                if (live[it]) {
                    float vv[8];
                    raw_unpack(vr[b][it], vv);
                    const float pj = SmMath<T>::div(e[it], sum);
#pragma unroll
                    for (int u = 0; u < 8; ++u) o8[u] = fmaf(pj, vv[u], o8[u]);
                }
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) o8[u] = key_slots_sum8(o8[u]);
            if (kg == 0) {
                store8(reinterpret_cast<T*>(P.att) + (size_t)rc * D + h * DH + dc * 8, o8);
                copy8(kc + ((size_t)t * P.rows_total + rl) * D + h * DH + dc * 8, qb + (size_t)rc * (3 * D) + D + h * DH + dc * 8);
                copy8(vc + ((size_t)t * P.rows_total + rl) * D + h * DH + dc * 8, qb + (size_t)rc * (3 * D) + 2 * D + h * DH + dc * 8);
            }
        }
    }
}

// ---- cross-attention of head h for the images whose rows lie in [m0, m0 + 40) (attn_cross_mfma_kernel's arithmetic per (image, head)) ----
// wave w takes images w, w + 4, ... of the block, two at a time (both items' K / V^T / q fragments requested up front).
template <typename T>
__device__ __forceinline__ void cross_attn_images(const DlParams& P, const DlLayer& L, const int m0, const int h) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l15 = lane & 15, lq = lane >> 4;
    const int rpi = P.rows_per_img, S = P.S, D = 512, H = 8;
    const int img0 = m0 / rpi, n_img = (min(RL, P.rows - m0) + rpi - 1) / rpi;
    for (int i0 = wave; i0 < n_img; i0 += 2 * NW) {
        uint4 kf[2][4][2], vf[2][4][2], qf[2][2];
        uint64_t kbits[2];
        bool iok[2];
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            iok[b] = i0 + b * NW < n_img;
            const int img = img0 + (iok[b] ? i0 + b * NW : i0);
            const uint16_t* kb = L.kp + ((size_t)img * H + h) * 4096;
            const uint16_t* vb = L.vt + ((size_t)img * H + h) * 4096;
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    kf[b][j][kk] = *reinterpret_cast<const uint4*>(kb + min(16 * j + l15, S - 1) * 64 + 32 * kk + 8 * lq);
                    vf[b][j][kk] = *reinterpret_cast<const uint4*>(vb + (16 * j + l15) * 64 + 32 * kk + 8 * lq);
                }
            const bool live = l15 < rpi;
            const size_t qrow = (size_t)img * rpi + (live ? l15 : 0);
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const uint16_t* qp = P.q + qrow * D + h * 64;
                uint4 tq;
                if (P.kp_dperm) {
                    const uint2 lo = *reinterpret_cast<const uint2*>(qp + 32 * kk + 4 * lq), hi = *reinterpret_cast<const uint2*>(qp + 32 * kk + 16 + 4 * lq);
                    tq = make_uint4(lo.x, lo.y, hi.x, hi.y);
                } else {
                    tq = *reinterpret_cast<const uint4*>(qp + 32 * kk + 8 * lq);
                }
                qf[b][kk] = live ? tq : make_uint4(0u, 0u, 0u, 0u);
            }
            kbits[b] = __ballot(P.keymask[img * S + min(lane, S - 1)] != 0);
        }
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            if (!iok[b]) continue;                       // (wave-uniform)
            const int img = img0 + i0 + b * NW;
            const bool live = l15 < rpi;
            const size_t qrow = (size_t)img * rpi + (live ? l15 : 0);
            uint16_t* orow = P.att + qrow * D + h * 64;
            cross_core<T>(kf[b], vf[b], qf[b], kbits[b], S, L.ea_scale, live, orow, lq);
        }
    }
}

// (debug, DH_DL_DEBUG & 2: workgroup 0 leaves s_memrealtime stamps (100 MHz) after every phase of layer 0 in sync[400 ...])
#define DL_STAMP(k) do { if ((P.dbg & 2) && b == 0 && l == 0 && tid == 0) P.sync[400 + (k)] = (unsigned)__builtin_amdgcn_s_memrealtime(); } while (0)

template <typename OT>
__global__ __launch_bounds__(NT, 1) void decode_layers_kernel(DlParams P) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[LDS_BYTES];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int cl = (b & 7) * 4 + (b >> 6), w = (b >> 3) & 7;        // cluster, member = head = column block
    // ---- where do the members of this cluster run?  (correctness does not depend on the answer, the barrier's cost does) ----
    Cluster C{P.sync + cl, P.sync + 320, 0u, 0};
    {
        const unsigned xcc = (unsigned)__builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 15u;      // HW_REG_XCC_ID[3:0]
        if (tid == 0) __hip_atomic_store(P.sync + 64 + b, xcc + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        C.target = __hip_atomic_load(P.sync + 32 + cl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // the counter's value when this launch began
        C.slow = 1;                                      // (the first barrier hands the XCD ids over: full fence)
        cluster_barrier(C);
        int same = 1;
        for (int mb = 0; mb < 8; ++mb) {
            const unsigned other = __hip_atomic_load(P.sync + 64 + (cl & 3) * 64 + mb * 8 + (cl >> 2), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            same &= (other == xcc + 1u);
        }
        C.slow = !same;
    }
    const unsigned base0 = C.target - 8u;
    for (int iter = 0; iter < P.iters; ++iter) {
        const int rb = cl + 32 * iter;
        const bool work = rb < P.n_rb;                   // (a cluster without a row block still takes part in nothing: its barriers are its own)
        const int m0 = rb * RL;
        for (int l = 0; l < P.n_layers; ++l) {
            const DlLayer& L = P.layers[l];
            const DlLayer& Lp = P.layers[l > 0 ? l - 1 : 0];
            GemmP g{};
            g.M = P.rows;
            DL_STAMP(0);
            if (work) {
                // 1. q_w | k_w | v_w = LN3_prev(X) Wqkv^T + b (column blocks w, 8 + w, 16 + w), then head w's self-attention
                g.A = P.x; g.lda = 512; g.wp = L.wqkv_pk; g.bias = L.bqkv; g.C = P.qkv; g.ldc = 1536; g.N = 1536; g.relu = 0;
                g.a_stats = l > 0 ? P.st0 : nullptr; g.a_eps = Lp.ln3_eps; g.a_colsum = L.cs_qkv;
                gemm_phase<OT, 0, 3, 1>(g, rb, w, 8, lds);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                asm volatile("buffer_inv sc1" ::: "memory");   // q / k / v of this position: written by this workgroup's other waves
                DL_STAMP(1);
                if (P.t + 1 <= 16 && !(P.dbg & 1)) self_attn_rows<OT, 2, 5>(P, L, m0, w);
                else self_attn_rows<OT, 5, 4>(P, L, m0, w);
            }
            DL_STAMP(2);
            cluster_barrier(C);
            DL_STAMP(3);
            if (work) {
                // 2. Y1_w = LN3_prev(X)_w + att Wo^T + bo, statistics -> st1
                g = GemmP{}; g.M = P.rows;
                g.A = P.att; g.lda = 512; g.wp = L.wo_pk; g.bias = L.bo; g.res = P.x; g.ldres = 512; g.C = P.o; g.ldc = 512; g.N = 512;
                if (l > 0) { g.r_stats = P.st0; g.r_eps = Lp.ln3_eps; g.r_gamma = Lp.ln3_g; g.r_beta = Lp.ln3_b; }
                g.o_stats = P.st1;
                gemm_phase<OT, 1, 1, 1>(g, rb, w, 0, lds);
            }
            DL_STAMP(4);
            cluster_barrier(C);
            DL_STAMP(5);
            if (work) {
                // 3. q_w = LN1(Y1) Wq^T + bq, then head w's attention over the images' patches
                g = GemmP{}; g.M = P.rows;
                g.A = P.o; g.lda = 512; g.wp = L.wq_pk; g.bias = L.bq; g.C = P.q; g.ldc = 512; g.N = 512;
                g.a_stats = P.st1; g.a_eps = L.ln1_eps; g.a_colsum = L.cs_q;
                gemm_phase<OT, 0, 1, 1>(g, rb, w, 0, lds);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                asm volatile("buffer_inv sc1" ::: "memory");
                DL_STAMP(14);
                cross_attn_images<OT>(P, L, m0, w);
            }
            DL_STAMP(6);
            cluster_barrier(C);
            DL_STAMP(7);
            if (work) {
                // 4. Y2_w = LN1(Y1)_w + att Weo^T + beo, statistics -> st2
                g = GemmP{}; g.M = P.rows;
                g.A = P.att; g.lda = 512; g.wp = L.weo_pk; g.bias = L.beo; g.res = P.o; g.ldres = 512; g.C = P.y2; g.ldc = 512; g.N = 512;
                g.r_stats = P.st1; g.r_eps = L.ln1_eps; g.r_gamma = L.ln1_g; g.r_beta = L.ln1_b; g.o_stats = P.st2;
                gemm_phase<OT, 1, 1, 1>(g, rb, w, 0, lds);
            }
            DL_STAMP(8);
            cluster_barrier(C);
            DL_STAMP(9);
            if (work) {
                // 5. ff[:, 4w .. 4w + 3] = relu(LN2(Y2) W1^T + b1)
                g = GemmP{}; g.M = P.rows;
                g.A = P.y2; g.lda = 512; g.wp = L.w1_pk; g.bias = L.b1; g.C = P.ff; g.ldc = 2048; g.N = 2048; g.relu = 1;
                g.a_stats = P.st2; g.a_eps = L.ln2_eps; g.a_colsum = L.cs_1;
                gemm_phase<OT, 0, 4, 1>(g, rb, 4 * w, 1, lds);
            }
            DL_STAMP(10);
            cluster_barrier(C);
            DL_STAMP(11);
            if (work) {
                // 6. X_w = LN2(Y2)_w + ff W2^T + b2, statistics -> st0  (LN3 of this layer now pending on X)
                g = GemmP{}; g.M = P.rows;
                g.A = P.ff; g.lda = 2048; g.wp = L.w2_pk; g.bias = L.b2; g.res = P.y2; g.ldres = 512; g.C = P.x; g.ldc = 512; g.N = 512;
                g.r_stats = P.st2; g.r_eps = L.ln2_eps; g.r_gamma = L.ln2_g; g.r_beta = L.ln2_b; g.o_stats = P.st0;
                gemm_phase<OT, 1, 1, 4>(g, rb, w, 0, lds);
            }
            DL_STAMP(12);
            cluster_barrier(C);
            DL_STAMP(13);
        }
    }
    // the counter's value for the next launch on this stream (every member has read the old one long ago)
    if (w == 0 && tid == 0) __hip_atomic_store(P.sync + 32 + cl, C.target, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    (void)base0;
}
__global__ void set_layer_kernel(DlLayer* table, int l, DlLayer v) { table[l] = v; }
}  // namespace

// Bytes of the device-resident layer table of dh_decode_layers (one entry per decoder layer).
extern "C" int dh_decode_layers_table_bytes(int n_layers) { return n_layers > 0 ? n_layers * (int)sizeof(DlLayer) : 0; }

// 1 when the model / position shape is one the persistent layer kernel takes: 16-bit, encoder attention on packed K / V^T tiles, D = 512 =
// 8 heads x 64, feed-forward width 2,048, every layer with fragment-packed and LayerNorm-folded weights, 40 % rows_per_img == 0,
// history of at most 40 positions, S <= 64.
extern "C" int dh_decode_layers_supported(const dh_tr_model_t* m, int rows_per_img, int t) {
    if (!m || !m->layers || !DH_IS_16BIT(m->dtype) || !m->cross || m->D != 512 || m->n_heads != 8 || m->pf_dim != 2048 || m->S <= 0 || m->S > 64) return 0;
    if (rows_per_img <= 0 || rows_per_img > 16 || (RL % rows_per_img) != 0 || t < 0 || t + 1 > 40 || !m->keymask) return 0;
    for (int l = 0; l < m->n_layers; ++l) {
        const dh_tr_layer_t& L = m->layers[l];
        if (!(L.wqkv_pk && L.wo_pk && L.wq_pk && L.weo_pk && L.w1_pk && L.w2_pk && L.kp && L.vt && L.bq_f && L.b1_f && L.cs_q && L.cs_1)) return 0;
        if (l > 0 && !(L.bqkv_f && L.cs_qkv)) return 0;
    }
    return 1;
}

// Fills the device-resident layer table (`table`: dh_decode_layers_table_bytes(n_layers) bytes of device memory) from the model
// description -- once per run, stream-ordered, one tiny launch per layer.
extern "C" int dh_decode_layers_table(const dh_tr_model_t* m, void* table, void* stream) {
    DH_REQUIRE(m && table && dh_decode_layers_supported(m, 1, 0) && m->n_layers <= 64);
    for (int l = 0; l < m->n_layers; ++l) {
        const dh_tr_layer_t& L = m->layers[l];
        DlLayer d{};
        d.wqkv_pk = (const uint4*)L.wqkv_pk; d.wo_pk = (const uint4*)L.wo_pk; d.wq_pk = (const uint4*)L.wq_pk; d.weo_pk = (const uint4*)L.weo_pk;
        d.w1_pk = (const uint4*)L.w1_pk; d.w2_pk = (const uint4*)L.w2_pk;
        d.bqkv = l > 0 ? L.bqkv_f : L.bqkv; d.bo = L.bo; d.bq = L.bq_f; d.beo = L.beo; d.b1 = L.b1_f; d.b2 = L.b2;
        d.cs_qkv = l > 0 ? L.cs_qkv : nullptr; d.cs_q = L.cs_q; d.cs_1 = L.cs_1;
        d.ln1_g = L.ln1_g; d.ln1_b = L.ln1_b; d.ln2_g = L.ln2_g; d.ln2_b = L.ln2_b; d.ln3_g = L.ln3_g; d.ln3_b = L.ln3_b;
        d.ln1_eps = L.ln1_eps; d.ln2_eps = L.ln2_eps; d.ln3_eps = L.ln3_eps; d.sa_scale = L.sa_scale; d.ea_scale = L.ea_scale;
        d.kcache = (uint16_t*)L.kcache; d.vcache = (uint16_t*)L.vcache; d.kp = (const uint16_t*)L.kp; d.vt = (const uint16_t*)L.vt;
        // (the entry travels as a kernel ARGUMENT: no host buffer has to outlive the call, and the launch can be captured in a hipGraph)
        hipLaunchKernelGGL(set_layer_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, (DlLayer*)table, l, d);
    }
    DH_LAUNCH_CHECK();
}

// All decoder layers of one decode position in ONE launch: reads sc->x (the embedded rows), leaves sc->x / sc->st0 as the launch chain
// does (the last layer's pre-LayerNorm rows + statistics), appends this position's K / V to the caches.  `table`: filled by
// dh_decode_layers_table for THIS model / run; `sync`: 321 uint32 of device memory private to the stream, zero before the first use
// (sync[320] != 0 afterwards = a bounded wait timed out: the results are undefined and the words must be zeroed again).
extern "C" int dh_decode_layers(const dh_tr_model_t* m, const dh_tr_scratch_t* sc, const void* table, const int32_t* tokens, int tok_ld,
                                const int32_t* src, int src_ld, int n_img, int rows_per_img, int row_mult, int rows_total, int t,
                                uint32_t* sync, void* stream) {
    DH_REQUIRE(m && sc && table && sync && n_img > 0 && dh_decode_layers_supported(m, rows_per_img, t));
    DH_REQUIRE(sc->x && sc->qkv && sc->att && sc->o && sc->q && sc->ff && sc->y2 && sc->st0 && sc->st1 && sc->st2 && (t == 0 || src));
    DlParams P{};
    P.layers = (const DlLayer*)table; P.n_layers = m->n_layers;
    P.x = (uint16_t*)sc->x; P.qkv = (uint16_t*)sc->qkv; P.att = (uint16_t*)sc->att; P.o = (uint16_t*)sc->o; P.q = (uint16_t*)sc->q;
    P.ff = (uint16_t*)sc->ff; P.y2 = (uint16_t*)sc->y2; P.st0 = (float2*)sc->st0; P.st1 = (float2*)sc->st1; P.st2 = (float2*)sc->st2;
    P.tokens = m->pad_index >= 0 ? tokens : nullptr; P.tok_ld = tok_ld; P.src = src; P.src_ld = src_ld; P.keymask = m->keymask;
    P.rows = n_img * rows_per_img; P.rows_per_img = rows_per_img; P.row_mult = row_mult; P.rows_total = rows_total; P.t = t; P.S = m->S;
    P.pad_index = m->pad_index; P.kp_dperm = m->layers[0].kp_dperm; P.n_rb = dh_cdiv(P.rows, RL); P.iters = dh_cdiv(P.n_rb, 32);
    P.sync = sync;
    { const char* e = getenv("DH_DL_DEBUG"); P.dbg = e ? atoi(e) : 0; }
    dh_prof_set_tag("layers");
    dh_prof_set_dims(P.rows, m->n_layers, t);
    DhProfScope prof("dh_decode_layers", 0.0, 0.0, stream);
    DH_DISPATCH_16(m->dtype, hipLaunchKernelGGL((decode_layers_kernel<T>), dim3(256), dim3(NT), 0, (hipStream_t)stream, P));
    DH_LAUNCH_CHECK();
}
