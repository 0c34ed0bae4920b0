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
//     vmcnt bookkeeping: at the wait for slab g at least two operations per slab staged after it are younger (bias strip and
//     stores only add to that), so an immediate count of 2 * (slabs staged after g) is always sufficient.
//   * the tiles_m workgroups that share a W panel (same vocabulary rows, different row tiles) sit on the same XCD and walk the
//     same sequence of panels, so a panel is fetched into that XCD's L2 once.
// Same results as vocab_logits_kernel bit for bit (same MFMA chain per output: k ascending, one accumulator per output).
// MEASURED (round 2, 1280 x 36541 x 512): stand-alone 76.8-79.5 us with the logits against 78.5-83.4 us for the 128 x 128 tile kernel
// on the same boxes (66 vs 63 us group maxima only); in the C2 / C3 steps 2.43 vs 2.49 ms and 2.81 vs 2.99 ms of classifier time
// per step (three alternating runs in one call) -- the default for K = 512 with the logits (DH_VOCAB_AREG=0: the tile kernel).
// Without any epilogue the launch takes 55 us; the group-maxima epilogue adds 11 us and the logits stores another 13: every
// vector-memory instruction (LDS-DMA piece or store) costs its wave ~100 cycles of issue next to running MFMAs, and nothing
// overlaps a wave's epilogue but its SIMD partner, which is in its own epilogue at the same time.
// History of this kernel: with `issued - mark` run-time counts in front of every s_waitcnt (a switch = a tree of taken scalar
// branches) it took 91 / 75.5 us; immediates in steady state gave 79.6 / 66.6; two slabs per barrier changed nothing (78.9 / 66.2);
// SIMD partners issuing their transfers at opposite ends of the MFMA block made it worse (86).  Phase stamps of the first version
// (s_memtime around every phase, ~1,900 cycles per 16 KB slab): 575 in the vmcnt wait although the slab had been requested seven
// slabs earlier, 480 at the barrier, 400 issuing 2 LDS-DMA pieces + 8 LDS reads, 300 in the 16 MFMAs; unchanged when every
// workgroup streamed the SAME panel (all L2 hits) or slab-major weights (1 KB contiguous pieces): not a memory-side limit.
#pragma once

template <typename OT>
__global__ __launch_bounds__(512, 1) void vocab_areg_kernel(VocabParams p) {
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
    const int m0 = tm * BM, total = my_tiles * NS;

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
        const int tn = grp + it * ngrp;
#pragma unroll
        for (int i = 0; i < G; ++i)
            b_off[i] = (unsigned)min(tn * BN + (wave * G + i) * 8 + lr, p.N - 1) * (unsigned)(p.ldw * 2) + swz * 16;
    };
    auto stage_into = [&](int buf) {                                // buf is a compile-time constant at every call site
        unsigned char* slab = lds + buf * SLAB;
        const unsigned kb = (unsigned)ld_s * 128u;
#pragma unroll
        for (int i = 0; i < G; ++i) dh_lds_dma16_s(b_base + kb, b_off[i], slab + (wave * G + i) * 1024);
        if (++ld_s == NS) { ld_s = 0; if (++ld_it < my_tiles) set_load_tile(ld_it); }
    };
    set_load_tile(0);
#pragma unroll
    for (int u = 0; u < NS - 2; ++u) stage_into(u);                 // slabs 0 .. 5 of the first tile (three pairs)

    for (int it = 0; it < my_tiles; ++it) {
        const int tn = grp + it * ngrp, n0 = tn * BN;
        const bool full = m0 + BM <= p.M && n0 + BN <= p.N;
        dh_f32x4 acc[TN][TM];
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int i = 0; i < TM; ++i) acc[j][i] = dh_f32x4{0.f, 0.f, 0.f, 0.f};
        // Two slabs per barrier (a pair = ring buffers 2 s, 2 s + 1): one wait + one barrier per 32 MFMAs.  Operations younger than
        // the pair's second slab at the wait: 2 per slab staged after it -- the next two pairs, min(4, slabs left) slabs (the pair
        // three steps ahead is staged below, after the wait); the bias strip and epilogue stores only add to that, so an immediate
        // count is always sufficient (a run-time switch in front of s_waitcnt costs a bare ring 45 % of its rate: ta_probe.hip).
#pragma unroll
        for (int st = 0; st < NS / 2; ++st) {
            const int g1 = it * NS + 2 * st + 1;                    // stream index of the pair's second slab
            wait_vmcnt_hot<2 * 4>(2 * min(4, total - 1 - g1));
            __builtin_amdgcn_s_barrier();                           // the pair complete for every wave; the previous pair's buffers consumed
            uint4 fwa[2][TN], fwb[2][TN];
            auto read_slab = [&](uint4 (&fw)[2][TN], int buf) {
                const unsigned char* sb = lds + buf * SLAB;
#pragma unroll
                for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        const int rr = wn0 + j * 16 + l15;
                        fw[kk][j] = *reinterpret_cast<const uint4*>(sb + rr * 128 + (((kk * 4 + lq) ^ (rr & 7)) << 4));
                    }
            };
            auto mfma_slab = [&](const uint4 (&fw)[2][TN], int t) {
#pragma unroll
                for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
#pragma unroll
                        for (int i = 0; i < TM; ++i) acc[j][i] = Op16<OT>::mfma(fw[kk][j], afr[2 * t + kk][i], acc[j][i]);
            };
            read_slab(fwa, 2 * st);
            __builtin_amdgcn_sched_barrier(0);
            // the pair three steps ahead goes into the buffers of the pair consumed in the previous step
            if (ld_it < my_tiles) stage_into((2 * st + NS - 2) % NS);
            if (ld_it < my_tiles) stage_into((2 * st + NS - 1) % NS);
            if (st == 0) {
                const int n = n0 + wn0 + lane;
                dh_lds_dma4(p.bias ? p.bias + (n < p.N ? n : 0) : reinterpret_cast<const float*>(dh_zero_page), bias_lds);   // bias == NULL: the zero page, never address 0
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            read_slab(fwb, 2 * st + 1);                             // the second slab's fragments load under the first slab's MFMAs
            mfma_slab(fwa, 2 * st);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            mfma_slab(fwb, 2 * st + 1);
        }
        // ---- epilogue from registers -----------------------------------------------------------------------------------------------
        // the tile's bias strip (issued in step 0; the three later steps stage six slabs = 12 operations after it, unless this is the
        // last tile) has landed
        if (it + 1 < my_tiles) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
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
                    if (p.C) store_half_full_lines(r_even + 32 * h, p.ldc, va, vb, l15 & 1);
                }
                mxv = fmaxf(mxv, __shfl_xor(mxv, 16, 64));
                mxv = fmaxf(mxv, __shfl_xor(mxv, 32, 64));
                if (p.gmax && lq == 0) p.gmax[(size_t)m * p.gmax_ld + (n0 + wn0) / 64] = mxv;
            }
        } else {
            // edge tile: element-wise
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
        }
    }
}

// ---- A-stationary, 256-row tiles (round 3) ------------------------------------------------------------------------------------------
// Same structure, but the eight waves own eight DIFFERENT 32-row strips (a 256-row tile, no row duplicated in registers) and each
// wave computes its strip against all 128 columns of the panel.  Per [128][64] weight slab a wave now issues 32 MFMAs for its two
// LDS-DMA pieces and one barrier (128-row kernel above: 16), so the fixed per-slab costs that bound that loop -- ~100 cycles of issue per
// LDS-DMA piece next to running MFMAs, the counted wait, the barrier -- are spread over twice the matrix work; a panel is also
// fetched from L2 by half as many workgroups.  Price: every wave reads the WHOLE slab from LDS (128 KB of LDS reads per slab per CU
// for 1,024 MFMA cycles = 128 B/clk, half the LDS rate) and the register budget is exact: 128 A fragments + 64 accumulators +
// a ring of 4 weight-fragment registers (one ds_read_b128 per two MFMAs, issued three steps ahead of them).
// No edge path (it cost registers the loop does not have): the host selects this kernel only for M % 256 == 0 and a logits row stride
// that covers whole panels (ldc >= tiles_n * 128: the decoders pad it), so every tile is stored in full.  Weight rows and bias
// entries past V are clamped to row V - 1: the padding columns hold copies of logit[V - 1], which leave the boundary group's
// maximum unchanged, and a group that starts past V gets -inf.
// Bit-identical to vocab_areg_kernel / vocab_logits_kernel (same MFMA chain per output).
// lane id without a live register: recomputed (2 VALU) wherever lane-derived values are needed outside the main loop
__device__ __forceinline__ int dh_lane_now() {
    int l;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
    return l;
}

template <typename OT>
__global__ __launch_bounds__(512, 1) void vocab_areg256_kernel(VocabParams p) {
    constexpr int BM = 256, BN = 128, NW = 8, NS = 8, KS = 16;
    constexpr int SLAB = BN * 128;                                  // 16 KB
    constexpr int TM = 2, TN = 8, PF = 3;                           // wave tile 32 x 128; LDS fragment reads run PF steps ahead
    constexpr int G = 2;
    __shared__ __attribute__((aligned(16))) unsigned char lds[NS * SLAB + NW * 512];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    unsigned char* bias_lds = lds + NS * SLAB + wave * 512;
    const int wm0 = wave * 32;
    const int lr = lane >> 3, lpos = lane & 7, swz = lpos ^ lr;
    const int l15 = lane & 15, lq = lane >> 4;

    const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3, per_xcd = (int)gridDim.x >> 3;
    const int gpx = per_xcd / p.tiles_m;
    if (local >= gpx * p.tiles_m) return;
    const int tm = local % p.tiles_m, grp = xcd * gpx + local / p.tiles_m, ngrp = 8 * gpx;
    const int my_tiles = grp < p.tiles_n ? (p.tiles_n - grp + ngrp - 1) / ngrp : 0;
    if (my_tiles == 0) return;
    const int m0 = tm * BM, total = my_tiles * NS;

    uint4 afr[KS][TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const uint16_t* arow = p.A + (size_t)min(m0 + wm0 + 16 * i + l15, p.M - 1) * p.lda + 8 * lq;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) afr[ks][i] = *reinterpret_cast<const uint4*>(arow + 32 * ks);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    unsigned b_off[G];
    const unsigned char* b_base = reinterpret_cast<const unsigned char*>(p.W);
    int ld_it = 0, ld_s = 0;
    auto set_load_tile = [&](int it) {                              // once per tile: lane-derived terms are recomputed, not kept live
        const int tn = grp + it * ngrp, ln = dh_lane_now();
        const unsigned r8 = (unsigned)ln >> 3, sw16 = (((unsigned)ln & 7u) ^ r8) << 4;
#pragma unroll
        for (int i = 0; i < G; ++i)
            b_off[i] = (unsigned)min(tn * BN + (wave * G + i) * 8 + (int)r8, p.N - 1) * (unsigned)(p.ldw * 2) + sw16;
    };
    auto stage_into = [&](int buf) {
        unsigned char* slab = lds + buf * SLAB;
        const unsigned kb = (unsigned)ld_s * 128u;
#pragma unroll
        for (int i = 0; i < G; ++i) dh_lds_dma16_s(b_base + kb, b_off[i], slab + (wave * G + i) * 1024);
        if (++ld_s == NS) { ld_s = 0; if (++ld_it < my_tiles) set_load_tile(ld_it); }
    };
    set_load_tile(0);
#pragma unroll
    for (int u = 0; u < NS - 2; ++u) stage_into(u);

    // LDS read bases (opaque to the compiler: otherwise it materialises one address register per (slab, column tile) beyond the
    // 64 KB reach of the ds_read offset field -- 32 of them -- and spills A fragments for it)
    unsigned rd_base[2][2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            rd_base[kk][hf] = (unsigned)(l15 * 128 + (((kk * 4 + lq) ^ (l15 & 7)) << 4) + hf * 4 * SLAB);
            asm volatile("" : "+v"(rd_base[kk][hf]));
        }

    for (int it = 0; it < my_tiles; ++it) {
        const int tn = grp + it * ngrp, n0 = tn * BN;
        dh_f32x4 acc[TN][TM];
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int i = 0; i < TM; ++i) acc[j][i] = dh_f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int st = 0; st < NS / 2; ++st) {
            const int g1 = it * NS + 2 * st + 1;
            wait_vmcnt_hot<2 * 4>(2 * min(4, total - 1 - g1));
            __builtin_amdgcn_s_barrier();
            // the pair's 32 fragment steps: q = 16 * slab + 8 * kk + column tile; ONE ds_read_b128 (16 vocabulary rows x 32 k) feeds two
            // MFMAs (the wave's two row tiles).  The reads run PF steps ahead of their MFMAs in a ring of PF + 1 fragment registers
            // (counted lgkmcnt waits, generated by the compiler from this program order): ~PF x 32 MFMA cycles of cover for the LDS
            // latency out of the wave's own stream -- a two-deep chunk pipeline with the same 16 registers covered only 64.
            // address = per-lane base of (k half, ring half) + a compile-time offset < 64 KB (the ds_read offset field): row
            // rr = 16 jt + l15 has rr & 7 == l15 & 7, so the XOR swizzle depends on the lane and kk only
            uint4 fw[PF + 1];
            auto read_q = [&](int q) {
                const int slab = 2 * st + (q >> 4), kk = (q >> 3) & 1, jt = q & 7;
                fw[q % (PF + 1)] = *reinterpret_cast<const uint4*>(lds + rd_base[kk][slab >> 2] + ((slab & 3) * SLAB + jt * 16 * 128));
            };
            auto mfma_q = [&](int q) {
                const int t = 2 * st + (q >> 4), kk = (q >> 3) & 1, jt = q & 7;
#pragma unroll
                for (int i = 0; i < TM; ++i) acc[jt][i] = Op16<OT>::mfma(fw[q % (PF + 1)], afr[2 * t + kk][i], acc[jt][i]);
            };
#pragma unroll
            for (int q = 0; q < PF; ++q) read_q(q);
            __builtin_amdgcn_sched_barrier(0);
            if (ld_it < my_tiles) stage_into((2 * st + NS - 2) % NS);
            if (ld_it < my_tiles) stage_into((2 * st + NS - 1) % NS);
            if (st == 0) {
                // the strip's LDS address is re-derived from the (scalar) wave number here: kept live across the tile loop it ended
                // up in a spilled VGPR, and the reload's vmcnt(0) drained the ring once per tile
                int w2 = wave;
                asm volatile("" : "+s"(w2));
                unsigned char* strip = lds + NS * SLAB + w2 * 512;
                const int ln = dh_lane_now();
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int n = n0 + 64 * u + ln;
                    dh_lds_dma4(p.bias ? p.bias + min(n, p.N - 1) : reinterpret_cast<const float*>(dh_zero_page), strip + 256 * u);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < 32; ++q) {
                if (q + PF < 32) read_q(q + PF);
                mfma_q(q);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (it + 1 < my_tiles) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        // the register budget of the loop is exact (see above): keep the compiler from hoisting the epilogue's per-lane row addresses
        // out of the tile loop (it spilled them to scratch) -- they are re-derived from an opaque copy of the lane id per tile
        const int lane_e = dh_lane_now();
        const int l15 = lane_e & 15, lq = lane_e >> 4;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int m = m0 + wm0 + 16 * i + l15;
            float* r_even = p.C + (size_t)(m0 + wm0 + 16 * i + (l15 & ~1)) * p.ldc + n0 + ((l15 & 1) ? 16 : 0) + 4 * lq;
#pragma unroll
            for (int gq = 0; gq < 2; ++gq) {                     // the two 64-column groups of the panel
                float mxv = -INFINITY;
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) {
                    const int h = 2 * gq + hh;
                    const float4 ba = p.bias ? *reinterpret_cast<const float4*>(bias_lds + (16 * (2 * h) + 4 * lq) * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
                    const float4 bb = p.bias ? *reinterpret_cast<const float4*>(bias_lds + (16 * (2 * h + 1) + 4 * lq) * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
                    float4 va, vb;
                    va.x = acc[2 * h][i][0] + ba.x; va.y = acc[2 * h][i][1] + ba.y; va.z = acc[2 * h][i][2] + ba.z; va.w = acc[2 * h][i][3] + ba.w;
                    vb.x = acc[2 * h + 1][i][0] + bb.x; vb.y = acc[2 * h + 1][i][1] + bb.y;
                    vb.z = acc[2 * h + 1][i][2] + bb.z; vb.w = acc[2 * h + 1][i][3] + bb.w;
                    mxv = fmaxf(fmaxf(mxv, fmaxf(fmaxf(va.x, va.y), fmaxf(va.z, va.w))), fmaxf(fmaxf(vb.x, vb.y), fmaxf(vb.z, vb.w)));
                    if (p.C) store_half_full_lines(r_even + 32 * h, p.ldc, va, vb, l15 & 1);
                    __builtin_amdgcn_sched_barrier(0);          // one 32-column half at a time: its accumulators die here
                }
                mxv = fmaxf(mxv, __shfl_xor(mxv, 16, 64));
                mxv = fmaxf(mxv, __shfl_xor(mxv, 32, 64));
                if (p.gmax && lq == 0 && n0 / 64 + gq < p.gmax_ld)                                     // -inf for a group that starts past V
                    p.gmax[(size_t)m * p.gmax_ld + n0 / 64 + gq] = n0 + 64 * gq < p.N ? mxv : -INFINITY;
            }
        }
    }
}
