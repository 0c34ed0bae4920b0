// ---- persistent classifier kernel with DEFERRED stores (included by gemm_bf16.hip) -----------------------------------------------
// The classifier is co-bound by its main loop and by 187 MB of fp32 logits per launch, and the two do not overlap inside a
// workgroup: loads, stores and LDS-DMA retire in issue order, so the wait for the next tile's first slabs also waits for every
// store of the tile before it.  Here a wave keeps TWO accumulator sets.  While tile i+1 accumulates into one, tile i's finished
// set (bias already added) is drained a few store instructions per K slab: the stores are spread over the whole main loop of the
// next tile, each slab wait only has to cover the handful of stores issued since that slab's transfer was requested, and the
// HBM write stream runs under the MFMAs instead of between them.  vmcnt bookkeeping is a running count of issued vector-memory
// operations and the value it had right after each in-flight slab's LDS-DMA (wave-uniform integers): allowed outstanding =
// issued - mark(slab).  Wave tile 64 x 64 (one 64-column group per wave, as the group maxima need); tile = (64 WAVES_M) x (64 WAVES_N).
#pragma once

template <typename OT, int WAVES_M, int WAVES_N, int NS, int OCC>
__global__ __launch_bounds__(64 * WAVES_M * WAVES_N, OCC) void vocab_dacc_kernel(VocabParams p) {
    constexpr int NW = WAVES_M * WAVES_N, BM = 64 * WAVES_M, BN = 64 * WAVES_N;
    constexpr int A_BYTES = BM * 128, SLAB = A_BYTES + BN * 128;
    constexpr int TM = 4, TN = 4, UNITS = 2 * TM;                   // drain unit = (row tile i, 32-column half h): 2 store instructions
    constexpr int IA = BM / (8 * NW), IB = BN / (8 * NW), G = IA + IB;
    static_assert(IA >= 1 && IB >= 1, "every wave stages pieces of both operands");
    __shared__ __attribute__((aligned(16))) unsigned char lds[NS * SLAB + NW * 256];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    unsigned char* bias_lds = lds + NS * SLAB + wave * 256;
    const int wm0 = (wave % WAVES_M) * 64, wn0 = (wave / WAVES_M) * 64;
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
    const int nslab = p.K / 64, total = my_tiles * nslab;           // host: K % 64 == 0, K / 64 > NS
    const int units_per_slab = (UNITS + nslab - 1) / nslab;

    // ---- loader (NS - 1 slabs ahead of the MFMAs); rows past M / V are clamped (finite garbage in never-stored outputs) ----------
    unsigned a_off[IA], b_off[IB];
    const unsigned char* a_base = reinterpret_cast<const unsigned char*>(p.A);
    const unsigned char* b_base = reinterpret_cast<const unsigned char*>(p.W);
    int ld_it = 0, ld_s = 0, ld_g = 0;
    auto set_load_tile = [&](int it) {
        const int tile = first + idx + it * nbx, tm = tile % p.tiles_m, tn = tile / p.tiles_m;
#pragma unroll
        for (int i = 0; i < IA; ++i)
            a_off[i] = (unsigned)min(tm * BM + (wave * IA + i) * 8 + lr, p.M - 1) * (unsigned)(p.lda * 2) + swz * 16;
#pragma unroll
        for (int i = 0; i < IB; ++i)
            b_off[i] = (unsigned)min(tn * BN + (wave * IB + i) * 8 + lr, p.N - 1) * (unsigned)(p.ldw * 2) + swz * 16;
    };
    int issued = 0;                                                 // vector-memory operations issued by this wave so far
    int mark[NS - 1];                                               // `issued` right after the DMA of slabs g, g+1, .. g+NS-2
    auto stage_next = [&]() {
        unsigned char* slab = lds + (ld_g % NS) * SLAB;
        const unsigned kb = (unsigned)ld_s * 128u;
#pragma unroll
        for (int i = 0; i < IA; ++i) dh_lds_dma16_s(a_base + kb, a_off[i], slab + (wave * IA + i) * 1024);
#pragma unroll
        for (int i = 0; i < IB; ++i) dh_lds_dma16_s(b_base + kb, b_off[i], slab + A_BYTES + (wave * IB + i) * 1024);
        issued += G;
        ++ld_g;
        if (++ld_s == nslab) { ld_s = 0; if (++ld_it < my_tiles) set_load_tile(ld_it); }
    };
    set_load_tile(0);
#pragma unroll
    for (int u = 0; u < NS - 1; ++u) { stage_next(); mark[u] = issued; }

    dh_f32x4 acc0[TN][TM], acc1[TN][TM];
    int dr_next = UNITS, dr_m0 = 0, dr_n0 = 0;                      // pending drain: next unit, tile origin
    float dr_mx = -INFINITY;

    // one drain unit of the finished set `acc`: row tile i = U / 2, 32-column half h = U % 2 -> two full-line store instructions,
    // plus, with the second half, the row tile's group maximum
    auto drain_unit = [&](dh_f32x4 (&acc)[TN][TM], auto UC) {
        constexpr int U = decltype(UC)::value, i = U >> 1, h = U & 1;
        const int m = dr_m0 + wm0 + 16 * i + l15;
        float4 va, vb;
        va.x = acc[2 * h][i][0]; va.y = acc[2 * h][i][1]; va.z = acc[2 * h][i][2]; va.w = acc[2 * h][i][3];
        vb.x = acc[2 * h + 1][i][0]; vb.y = acc[2 * h + 1][i][1]; vb.z = acc[2 * h + 1][i][2]; vb.w = acc[2 * h + 1][i][3];
        const float mx = fmaxf(fmaxf(fmaxf(va.x, va.y), fmaxf(va.z, va.w)), fmaxf(fmaxf(vb.x, vb.y), fmaxf(vb.z, vb.w)));
        dr_mx = h == 0 ? mx : fmaxf(dr_mx, mx);
        if (p.C) {
            float* r_even = p.C + (size_t)(dr_m0 + wm0 + 16 * i + (l15 & ~1)) * p.ldc + dr_n0 + wn0 + ((l15 & 1) ? 16 : 0) + 4 * lq + 32 * h;
            store_half_full_lines(r_even, p.ldc, va, vb, l15 & 1);
            issued += 2;
        }
        if (h == 1 && p.gmax) {
            float gm = fmaxf(dr_mx, __shfl_xor(dr_mx, 16, 64));
            gm = fmaxf(gm, __shfl_xor(gm, 32, 64));
            if (lq == 0) p.gmax[(size_t)m * p.gmax_ld + (dr_n0 + wn0) / 64] = gm;
            issued += 1;
        }
    };
    auto drain_some = [&](dh_f32x4 (&acc)[TN][TM], int n) {
        for (int k = 0; k < n && dr_next < UNITS; ++k, ++dr_next) {
            switch (dr_next) {
                case 0: drain_unit(acc, std::integral_constant<int, 0>{}); break;
                case 1: drain_unit(acc, std::integral_constant<int, 1>{}); break;
                case 2: drain_unit(acc, std::integral_constant<int, 2>{}); break;
                case 3: drain_unit(acc, std::integral_constant<int, 3>{}); break;
                case 4: drain_unit(acc, std::integral_constant<int, 4>{}); break;
                case 5: drain_unit(acc, std::integral_constant<int, 5>{}); break;
                case 6: drain_unit(acc, std::integral_constant<int, 6>{}); break;
                default: drain_unit(acc, std::integral_constant<int, 7>{}); break;
            }
        }
    };

    int g = 0;
    // tile `it` accumulates into accC while the previous tile's finished set accD drains
    auto run_tile = [&](dh_f32x4 (&accC)[TN][TM], dh_f32x4 (&accD)[TN][TM], int it) {
        const int tile = first + idx + it * nbx, tm = tile % p.tiles_m, tn = tile / p.tiles_m;
        const int m0 = tm * BM, n0 = tn * BN;
        const bool full = m0 + BM <= p.M && n0 + BN <= p.N;
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int i = 0; i < TM; ++i) accC[j][i] = dh_f32x4{0.f, 0.f, 0.f, 0.f};
        int bias_mark = 0;
        for (int t = 0; t < nslab; ++t, ++g) {
            wait_vmcnt_any(issued - mark[0]);             // slab g has landed; younger: later slabs, drained stores, the bias strip
            __builtin_amdgcn_s_barrier();
            const unsigned char* sa = lds + (g % NS) * SLAB;
            const unsigned char* sb = sa + A_BYTES;
            // fragments per 32-deep k half (32 registers instead of 64: with two accumulator sets the wave is at the 256-register
            // limit -- the all-fragments-first schedule of vocab_logits_kernel spills inside this loop and runs at half the speed)
            uint4 fa[TM], fw[TN];
            auto read_frags = [&](int kk) {
                const int c = kk * 4 + lq;
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    const int rr = wm0 + i * 16 + l15;
                    fa[i] = *reinterpret_cast<const uint4*>(sa + rr * 128 + ((c ^ (rr & 7)) << 4));
                }
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int rr = wn0 + j * 16 + l15;
                    fw[j] = *reinterpret_cast<const uint4*>(sb + rr * 128 + ((c ^ (rr & 7)) << 4));
                }
            };
            read_frags(0);
            __builtin_amdgcn_sched_barrier(0);
            // issue side of the slab, inside the LDS-read latency window: next slab's DMA, the bias strip, drained stores
#pragma unroll
            for (int u = 0; u + 1 < NS - 1; ++u) mark[u] = mark[u + 1];
            if (ld_g < total) stage_next();
            mark[NS - 2] = issued;
            if (t == 0) {
                const int n = n0 + wn0 + lane;
                dh_lds_dma4(p.bias ? p.bias + (n < p.N ? n : 0) : reinterpret_cast<const float*>(dh_zero_page), bias_lds);   // bias == NULL: the zero page, never address 0
                issued += 1;
                bias_mark = issued;
            }
            drain_some(accD, units_per_slab);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int i = 0; i < TM; ++i) accC[j][i] = Op16<OT>::mfma(fw[j], fa[i], accC[j][i]);
            __builtin_amdgcn_sched_barrier(0);
            read_frags(1);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int i = 0; i < TM; ++i) accC[j][i] = Op16<OT>::mfma(fw[j], fa[i], accC[j][i]);
        }
        // the tile's bias strip has landed once everything issued up to it has
        wait_vmcnt_any(issued - bias_mark);
        float4 b4[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j)
            b4[j] = p.bias ? *reinterpret_cast<const float4*>(bias_lds + (16 * j + 4 * lq) * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                accC[j][i][0] += b4[j].x; accC[j][i][1] += b4[j].y; accC[j][i][2] += b4[j].z; accC[j][i][3] += b4[j].w;
            }
        if (full) {                                       // drained during the next tile (or after the loop)
            dr_next = 0; dr_m0 = m0; dr_n0 = n0;
            return;
        }
        // edge tile: immediate, element-wise (its store count is not uniform: drain the queue afterwards)
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int m = m0 + wm0 + 16 * i + l15;
            float mxv = -INFINITY;
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) {
                    const int n = n0 + wn0 + 16 * j + 4 * lq + rr;
                    if (n < p.N) {
                        const float v = accC[j][i][rr];
                        mxv = fmaxf(mxv, v);
                        if (m < p.M && p.C) p.C[(size_t)m * p.ldc + n] = v;
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
        for (int u = 0; u < NS - 1; ++u) mark[u] = 0;
    };

    for (int it = 0; it < my_tiles; it += 2) {
        run_tile(acc0, acc1, it);
        if (it + 1 < my_tiles) run_tile(acc1, acc0, it + 1);
    }
    if (my_tiles & 1) drain_some(acc0, UNITS); else drain_some(acc1, UNITS);
}
