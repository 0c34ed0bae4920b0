// ---- A-stationary persistent classifier kernel (included by gemm_bf16.hip) ------------------------------------------------------
// The classifier GEMM logits[M, V] = A[M, 512] W[V, 512]^T + b has a tiny activation operand (1.3 MB at 1,280 beam rows) and a
// 37 MB weight operand, yet the tile kernels stream BOTH through LDS for every tile: at 128 x 128 tiles that is 32 KB of LDS-DMA
// per 512 MFMA cycles, one slab of look-ahead, and the loop runs at the L2 -> LDS rate (0.25 of the MFMA peak).  Here the
// activation operand never touches LDS:
//   * a workgroup (8 waves, one per CU) owns ONE 128-row tile of A for its whole life; every wave keeps the MFMA fragments of
//     its 32 rows x all 512 k in REGISTERS (2 row tiles x 16 k-steps x 4 VGPRs = 128), loaded once straight from global memory;
//   * only W streams: [128 vocabulary rows][64 k] slabs of 16 KB through an 8-slab LDS ring -- seven slabs (3.5 tile-slab times)
//     in flight instead of one, half the LDS-DMA bytes and a third of the LDS reads per MFMA;
//   * a tile = 8 slabs = exactly one trip around the ring, so every buffer index and every A-fragment index is static;
//   * stores ride on the deep ring: loads, stores and LDS-DMA retire in issue order, and the seven slabs requested BEFORE a
//     tile's epilogue are waited for first -- the epilogue's stores get seven slab times to drain before a wait covers them.
//     vmcnt bookkeeping: `issued` counts this wave's vector-memory instructions, mark[b] remembers the count right after the
//     transfer into ring buffer b; allowed outstanding at the wait for that buffer = issued - mark[b].
//   * the tiles_m workgroups that share a W panel (same vocabulary rows, different row tiles) sit on the same XCD and walk the
//     same sequence of panels, so a panel is fetched into that XCD's L2 once.
// Same results as vocab_logits_kernel bit for bit (same MFMA chain per output: k ascending, one accumulator per output).
// MEASURED (round 2, 1280 x 36541 x 512): 91 us with the logits / 75 us group maxima only, against 83 / 63 us for the default
// 128 x 128 kernel on the same box -- opt-in (DH_VOCAB_AREG=1), not the default.  The STAMP variant (DH_VOCAB_AREG=2,
// tools/areg_stamps.py) shows why: of ~1,900 cycles per 16 KB slab the wave spends 575 waiting for the slab's transfer although
// it was requested seven slabs earlier, 480 at the barrier (the other waves' waits), 400 issuing 2 LDS-DMA pieces + 8 LDS reads,
// 300 in its 16 MFMAs; the figures do not change when every workgroup streams the SAME panel (all L2 hits), and with MFMAs and
// LDS reads removed the transfer skeleton alone still takes ~1,270 cycles per slab: the LDS-DMA path itself delivers ~13-17 bytes
// per cycle per CU in this access pattern (8 rows x 128 B per piece), whatever the look-ahead -- the rate every LDS-DMA GEMM of
// this library runs at (32 pieces per 512 MFMA cycles at 128 x 128 tiles -> 0.25 of the MFMA peak, 64 per 2,048 at 256 x 256 -> 0.5).
#pragma once

__device__ unsigned long long dh_areg_stamps[256 * 8];             // developer instrumentation (STAMP variant only)

template <typename OT, bool STAMP = false, int DBG = 0>   // DBG (instrumentation): 1 = no MFMAs, 2 = no LDS reads either
__global__ __launch_bounds__(512, 1) void vocab_areg_kernel(VocabParams p) {
    unsigned long long st_acc[6] = {0, 0, 0, 0, 0, 0}, st_t = 0;
#define DH_STAMP(k) do { if constexpr (STAMP) { const unsigned long long now_ = clock64(); st_acc[k] += now_ - st_t; st_t = now_; } } while (0)
    constexpr int BM = 128, BN = 128, NW = 8, NS = 8, KS = 16;      // K = 512: 16 k-steps of 32, 8 slabs of 64
    constexpr int SLAB = BN * 128;                                  // 16 KB
    constexpr int TM = 2, TN = 4;                                   // wave tile 32 x 64
    constexpr int G = 2;                                            // LDS-DMA pieces per wave per slab
    __shared__ __attribute__((aligned(16))) unsigned char lds[NS * SLAB + NW * 256];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    unsigned char* bias_lds = lds + NS * SLAB + wave * 256;
    const int wm0 = (wave & 3) * 32, wn0 = (wave >> 2) * 64;
    const int lr = lane >> 3, lpos = lane & 7, swz = lpos ^ lr;
    const int l15 = lane & 15, lq = lane >> 4;

    // workgroup -> (row tile, panel group): the groups of one XCD (= blockIdx % 8) are made of that XCD's workgroups only
    const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3, per_xcd = (int)gridDim.x >> 3;
    const int gpx = per_xcd / p.tiles_m;                            // groups per XCD (host: >= 1)
    if (local >= gpx * p.tiles_m) return;
    const int tm = local % p.tiles_m, grp = xcd * gpx + local / p.tiles_m, ngrp = 8 * gpx;
    const int my_tiles = grp < p.tiles_n ? (p.tiles_n - grp + ngrp - 1) / ngrp : 0;
    if (my_tiles == 0) return;
    const int m0 = tm * BM;

    // ---- this wave's A fragments: rows m0 + wm0 + 16 i + l15, k = 32 ks + 8 lq .. + 7 -------------------------------------------------
    uint4 afr[KS][TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const uint16_t* arow = p.A + (size_t)min(m0 + wm0 + 16 * i + l15, p.M - 1) * p.lda + 8 * lq;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) afr[ks][i] = *reinterpret_cast<const uint4*>(arow + 32 * ks);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                // from here on only the counted operations below are in flight

    // ---- W loader: 7 slabs ahead; rows past V are clamped (finite garbage in never-stored outputs) --------------------------------------
    unsigned b_off[G];
    const unsigned char* b_base = reinterpret_cast<const unsigned char*>(p.W);
    int ld_it = 0, ld_s = 0;
    auto set_load_tile = [&](int it) {
        const int tn = (STAMP && p.tgt_logit) ? 0 : grp + it * ngrp;      // (instrumentation: every workgroup streams panel 0 -> all L2 hits)
#pragma unroll
        for (int i = 0; i < G; ++i)
            b_off[i] = (STAMP && p.gsum) ? (unsigned)min(tn * BN + (wave * G + i) * 8 + lr, p.N - 1) * 128u + swz * 16   // (timing probe: slab-major W)
                                         : (unsigned)min(tn * BN + (wave * G + i) * 8 + lr, p.N - 1) * (unsigned)(p.ldw * 2) + swz * 16;
    };
    int issued = 0;
    int mark[NS];
#pragma unroll
    for (int b = 0; b < NS; ++b) mark[b] = 0;
    auto stage_into = [&](int buf) {                                // buf is a compile-time constant at every call site
        unsigned char* slab = lds + buf * SLAB;
        const unsigned kb = (STAMP && p.gsum) ? (unsigned)ld_s * (unsigned)p.N * 128u : (unsigned)ld_s * 128u;
#pragma unroll
        for (int i = 0; i < G; ++i) dh_lds_dma16_s(b_base + kb, b_off[i], slab + (wave * G + i) * 1024);
        issued += G;
        mark[buf] = issued;
        if (++ld_s == NS) { ld_s = 0; if (++ld_it < my_tiles) set_load_tile(ld_it); }
    };
    set_load_tile(0);
#pragma unroll
    for (int u = 0; u < NS - 1; ++u) stage_into(u);                 // slabs 0 .. 6 of the first tile

    if constexpr (STAMP) st_t = clock64();
    for (int it = 0; it < my_tiles; ++it) {
        const int tn = grp + it * ngrp, n0 = tn * BN;
        const bool full = m0 + BM <= p.M && n0 + BN <= p.N;
        dh_f32x4 acc[TN][TM];
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int i = 0; i < TM; ++i) acc[j][i] = dh_f32x4{0.f, 0.f, 0.f, 0.f};
        int bias_mark = 0;
#pragma unroll
        for (int t = 0; t < NS; ++t) {                              // slab t of the tile lives in ring buffer t
            DH_STAMP(4);
            wait_vmcnt_any(issued - mark[t]);
            DH_STAMP(0);
            __builtin_amdgcn_s_barrier();                           // slab t complete for every wave; buffer (t + 7) % 8 fully consumed
            DH_STAMP(1);
            const unsigned char* sb = lds + t * SLAB;
            uint4 fw[2][TN];
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int rr = wn0 + j * 16 + l15;
                    if constexpr (DBG < 2) fw[kk][j] = *reinterpret_cast<const uint4*>(sb + rr * 128 + (((kk * 4 + lq) ^ (rr & 7)) << 4));
                    else fw[kk][j] = make_uint4(rr, t, kk, j);
                }
            __builtin_amdgcn_sched_barrier(0);
            // 7 slabs ahead: slab t + 7 (of this tile or the next one).  (Measured and not kept: waves 0-3 issuing their pieces before
            // their MFMAs and their SIMD partners 4-7 after them -- 86 instead of 75 us.)
            if (ld_it < my_tiles) stage_into((t + NS - 1) % NS);
            if (t == 0) {
                const int n = n0 + wn0 + lane;
                dh_lds_dma4(p.bias + (p.bias && n < p.N ? n : 0), bias_lds);
                issued += 1;
                bias_mark = issued;
            }
            DH_STAMP(2);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            DH_STAMP(3);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int i = 0; i < TM; ++i) {
                        if constexpr (DBG == 0) acc[j][i] = Op16<OT>::mfma(fw[kk][j], afr[2 * t + kk][i], acc[j][i]);
                        else if (kk == 0 && i == 0) acc[j][i][0] += __uint_as_float(fw[0][j].x ^ fw[1][j].y ^ afr[2 * t][0].x);
                    }
        }
        DH_STAMP(4);
        // ---- epilogue from registers -----------------------------------------------------------------------------------------------
        wait_vmcnt_any(issued - bias_mark);                         // the tile's bias strip has landed
        float4 b4[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j)
            b4[j] = p.bias ? *reinterpret_cast<const float4*>(bias_lds + (16 * j + 4 * lq) * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
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
                    if (p.C) { store_half_full_lines(r_even + 32 * h, p.ldc, va, vb, l15 & 1); issued += 2; }
                }
                mxv = fmaxf(mxv, __shfl_xor(mxv, 16, 64));
                mxv = fmaxf(mxv, __shfl_xor(mxv, 32, 64));
                if (p.gmax) {
                    if (lq == 0) p.gmax[(size_t)m * p.gmax_ld + (n0 + wn0) / 64] = mxv;
                    issued += 1;
                }
            }
        } else {
            // edge tile: element-wise, store count not uniform -> drain everything and restart the bookkeeping
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
                const int gidx = (n0 + wn0) / 64;
                if (lq == 0 && m < p.M && p.gmax && gidx < p.gmax_ld) p.gmax[(size_t)m * p.gmax_ld + gidx] = mxv;   // -inf for a group past V
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            issued = 0;
#pragma unroll
            for (int b = 0; b < NS; ++b) mark[b] = 0;
        }
    }
    if constexpr (STAMP) {
        DH_STAMP(5);
        if (lane == 0 && wave == 0)
            for (int k = 0; k < 6; ++k) dh_areg_stamps[blockIdx.x * 8 + k] = st_acc[k];
        if (lane == 0 && wave == 0) dh_areg_stamps[blockIdx.x * 8 + 6] = (unsigned long long)my_tiles;
    }
#undef DH_STAMP
}
