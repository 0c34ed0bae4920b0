// 1x1 convolution + BatchNorm + ReLU of the ResNet-50 bottlenecks' `conv1` (torchvision Bottleneck.conv1 / bn1 / relu; reference
// encoders.py:37-38, :56) for the K >= 512 layers, with the WEIGHTS STATIONARY IN REGISTERS and the pixels streamed:
//   y[M, N] = relu((x[M, K] W[N, K]^T) * scale[n] + shift[n]),   M = images x pixels (12,544 ... 200,704), N = 128 ... 512, K = 512 / 1,024.
//
// The implicit-GEMM tile kernel (gemm_bf16.hip, 128 x 128 tiles, both operands through a two-slab LDS ring) runs these layers at
// 0.26-0.50 of their roofline ({50176 x 256 x 1024}: 48.5 us against 16 us of HBM time): one barrier + one wait per 64-k slab with
// 8-16 MFMAs per wave in between, and the weight panel staged again for every 128-pixel tile.  Here (the partition of
// linear_wreg.hip / lstm_wreg.hip, made persistent over the pixels):
//   * a workgroup (8 waves, one per CU) owns ONE 128-column block of the output channels for its whole life: a wave keeps its 16 weight
//     rows x all K as MFMA fragments in registers (64 or 128 VGPRs), loaded once from the fragment-packed weights;
//   * the pixels stream through TWO LDS buffers of RB rows x all K (64 KB each) by LDS-DMA: the transfer of block i + 1 is issued
//     behind block i's barrier (every wave has then left block i - 1, whose buffer it takes) and is in flight during block i's MFMAs;
//   * per block: ONE barrier, TM x K / 32 MFMAs per wave fed by ds_read_b128 only, then BatchNorm + ReLU on the accumulators and
//     8-byte stores straight from them;
//   * vmcnt bookkeeping with immediates: vector-memory operations retire in order, so when block i is waited for exactly the TM
//     stores of block i - 1 may still be outstanding;
//   * the N / 128 workgroups that stream the same pixels sit on ONE XCD (blockIdx % 8) and walk them in the same order: a pixel block
//     is fetched into that L2 once.
// Same MFMA operand contents and k order per output as the tile kernel, same epilogue arithmetic: BIT-IDENTICAL (tests/test_bf16_gpu.py).
#include <stdlib.h>
#include <type_traits>
#include "common.h"
#include "prof.h"

namespace {
struct C1Params {
    const uint16_t* x; const uint4* wp; const float* scale; const float* shift; uint16_t* y;
    const uint16_t* res;                                  // RES instances: residual rows [M, N]
    const uint4* w1p; const float* scale1; const float* shift1; uint16_t* y1n;     // F1 instances: next conv1 (fragments [8][4][64]), [M, 64]
    int M, N, relu, nb_n, wg_per_n, nblk;
    // second source of the virtual operand [x | x2 at the strided pixels] (dual form; x2 == nullptr otherwise): k slabs ns1 .. come from
    // x2 [Nimg, H2, W2, C2]; row r = (n, oy, ox) of the Ho x Wo output grid reads pixel (n, oy * stride, ox * stride)
    const uint16_t* x2; int ns1, howo, wo, h2w2, w2, stride2, c2;
    unsigned magic_howo, magic_wo;                        // floor(2^32 / d) + 1: r / d == umulhi(r, magic) while r * d < 2^32
};

__device__ __forceinline__ void c1_dma16(const void* base, unsigned off, void* lds_dst) {
    const unsigned m = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(dh_lptr_t)lds_dst);
    // (the base is wave-uniform by construction; said explicitly because an if-converted source select otherwise lands in VGPRs)
    const uint64_t b = ((uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)((uintptr_t)base >> 32)) << 32) |
                       (uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)base);
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" :: "s"(m), "v"(off), "s"(b) : "memory");
}

// KF = K / 32 fragments per column tile, RB rows per block (RB x K x 2 B = 64 KB), TN column tiles per wave (NW = 8 / TN waves).
// TN = 2: every pixel fragment read from LDS feeds two MFMAs -- with one column tile per wave all eight waves read the WHOLE pixel block
// (8 x 64 KB per block at 128 B/clk = twice the block's MFMA time).  MEASURED: no gain -- TN = 2 with 4 waves 42 / 56 / 91 / 75 us, with 8
// waves and 256 channels per workgroup 79.5 us for {200704 x 256 x 512}, against 36 / 53 / 76 / 59 us for TN = 1 with 8 waves: these layers
// run at 3.6-4.9 TB/s of HBM traffic, not at the LDS read rate.  The dispatch uses TN = 1.
// F1 = 64 (the K = 128 dual instance, all 256 output channels in the workgroup): the NEXT bottleneck's conv1 + bn1 + relu (256 -> 64) on
// the block's rounded outputs, which also go into an LDS operand tile [4 k blocks][RB rows][128 B] (conv_s1.hip does the same behind
// the stage-1 tail): wave w = column tile w % 4 of the 64 channels x row tiles 4 (w / 4) .. + 3, its 8 weight fragments stationary
template <typename OT, int KF, int RB, int TN, int NW, bool RES = false, int F1 = 0>      // TN = 1 or 2; RES: + residual [M, N] before the ReLU
__global__ __launch_bounds__(64 * NW, 1) void conv1x1_wreg_kernel(C1Params p) {
    constexpr int K = 32 * KF, NSLAB = K / 64, SLABB = RB * 128, BUF = NSLAB * SLABB, TM = RB / 16, RG = RB / 8;
    constexpr int NT = 64 * NW, BN = 16 * NW * TN;         // output channels per workgroup
    constexpr int PPW = NSLAB * RG / NW;                  // LDS-DMA pieces per wave per block
    constexpr int PF = 3;
    static_assert(BUF <= 65536 && (NSLAB * RG) % NW == 0, "two buffers of <= 64 KB (one ds_read offset window each)");
    static_assert(F1 == 0 || (F1 == 64 && TN == 2 && NW == 8 && RB == 128), "the fused next conv1: the K = 128 dual instance only");
    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * BUF + (TN > 1 ? BN * 8 : 0) + (F1 ? RB * 512 : 0)];   // + scale | shift (TN > 1) + out tile (F1)
    unsigned char* const otile = lds + 2 * BUF + (TN > 1 ? BN * 8 : 0);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, lq = lane >> 4, lr = lane >> 3, lpos = lane & 7;
    // workgroup -> (column block, member j of that block's pixel walkers); the nb_n column blocks of member j share blockIdx % 8
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    const int nb = idx % p.nb_n, j = (idx / p.nb_n) * 8 + xcd;
    const int n0 = nb * BN;

    // ---- this wave's 16 weight rows x all K + its columns' BatchNorm scale / shift ------------------------------------------------------
    uint4 wf[TN][KF];
    float4 sc4[TN], sh4[TN];
#pragma unroll
    for (int c = 0; c < TN; ++c) {
        const uint4* wsrc = p.wp + ((size_t)(n0 / 16 + wave * TN + c)) * 64 + lane;
        const size_t fstep = (size_t)(p.N / 16) * 64;
#pragma unroll
        for (int f = 0; f < KF; ++f) wf[c][f] = wsrc[(size_t)f * fstep];
        sc4[c] = p.scale ? *reinterpret_cast<const float4*>(p.scale + n0 + 16 * (wave * TN + c) + 4 * lq) : make_float4(1.f, 1.f, 1.f, 1.f);
        sh4[c] = *reinterpret_cast<const float4*>(p.shift + n0 + 16 * (wave * TN + c) + 4 * lq);
    }
    // two column tiles per wave: the 16 epilogue constants live in LDS (read back per block) -- the K = 768 instance has no registers for them
    float4* const bn_lds = reinterpret_cast<float4*>(lds + 2 * BUF) + (wave * TN) * 8 + lq;      // [tile][scale | shift][quad]
    if constexpr (TN > 1) {
        if (l15 == 0) { bn_lds[0] = sc4[0]; bn_lds[4] = sh4[0]; bn_lds[8] = sc4[TN - 1]; bn_lds[12] = sh4[TN - 1]; }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    uint4 w1f[F1 ? 8 : 1];
    float4 sc1 = make_float4(0.f, 0.f, 0.f, 0.f), sh1 = sc1;
    if constexpr (F1 > 0) {
#pragma unroll
        for (int g = 0; g < 8; ++g) w1f[g] = p.w1p[(size_t)(g * (F1 / 16) + (wave & 3)) * 64 + lane];
        sc1 = *reinterpret_cast<const float4*>(p.scale1 + 16 * (wave & 3) + 4 * lq);
        sh1 = *reinterpret_cast<const float4*>(p.shift1 + 16 * (wave & 3) + 4 * lq);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // from here on only the counted operations below are in flight

    // ---- pixel-block loader: block b = rows b RB .. + RB - 1 (rows past M re-read row M - 1) -----
    const unsigned swz = (unsigned)((lpos ^ lr) << 4);
    auto stage = [&](int b, int buf) {
        unsigned char* dst0 = lds + buf * BUF;
#pragma unroll
        for (int i = 0; i < PPW; ++i) {                   // piece = (8-row group g, 64-column slab s), dealt round-robin to the waves
            const int pc = wave + NW * i, s = pc % NSLAB, g = pc / NSLAB;
            const unsigned r = (unsigned)min(b * RB + g * 8 + lr, p.M - 1);
            if (s < p.ns1) {
                c1_dma16(p.x, r * (unsigned)(p.ns1 * 128) + swz + 128u * s, dst0 + s * SLABB + g * 1024);
            } else {                                      // (wave-uniform branch: s depends on the wave and the unrolled i only)
                const unsigned n = __umulhi(r, p.magic_howo), rem = r - n * (unsigned)p.howo;
                const unsigned oy = __umulhi(rem, p.magic_wo), ox = rem - oy * (unsigned)p.wo;
                const unsigned r2 = n * (unsigned)p.h2w2 + (oy * (unsigned)p.w2 + ox) * (unsigned)p.stride2;
                c1_dma16(p.x2, r2 * (unsigned)(p.c2 * 2) + swz + 128u * (s - p.ns1), dst0 + s * SLABB + g * 1024);
            }
        }
    };
    // LDS read bases per (k half, buffer): row 16 i + l15 has (row & 7) == (l15 & 7); a buffer is one 64 KB ds_read window
    unsigned rd_base[2][2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int bf = 0; bf < 2; ++bf) {
            rd_base[kk][bf] = (unsigned)(l15 * 128 + (((kk * 4 + lq) ^ (l15 & 7)) << 4) + bf * BUF);
            asm volatile("" : "+v"(rd_base[kk][bf]));
        }

    const int first = j, step = p.wg_per_n;
    int nmine = first < p.nblk ? (p.nblk - first + step - 1) / step : 0;
    if (nmine == 0) return;
    stage(first, 0);
    if (nmine > 1) stage(first + step, 1);
    uint16_t* const ybase = p.y + n0 + 16 * wave * TN + 4 * lq;
    auto block = [&](auto BUFC, const int i) {
        constexpr int buf = decltype(BUFC)::value;
        const int b = first + i * step;
        // block i has landed: the only younger operations are the transfer of block i + 1 when it was issued in the prologue (i = 0),
        // else the TN TM stores of block i - 1 (the transfer of block i was issued before them)
        if (i == 0) {
            if (nmine > 1) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(PPW) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(%0)" :: "n"((RES ? 2 : 1) * TN * TM + (F1 ? 4 : 0)) : "memory");   // (RES: + block i - 1's residual loads; F1: + its 4 y1_next stores)
        }
        __builtin_amdgcn_s_barrier();                     // every wave's pieces of block i landed AND every wave has left block i - 1
        if (i >= 1 && i + 1 < nmine) stage(b + step, buf ^ 1);   // block i + 1 into block i - 1's buffer: in flight during this block
        // the block's residual quads in accumulator layout, requested in front of the MFMAs (plain loads behind the transfer just issued:
        // the compiler's own counts for them can only be stricter than needed, the immediate above counts them)
        uint2 rq[TN][TM];
        if constexpr (RES) {
#pragma unroll
            for (int c = 0; c < TN; ++c)
#pragma unroll
                for (int t = 0; t < TM; ++t)
                    rq[c][t] = *reinterpret_cast<const uint2*>(p.res + (n0 + 16 * wave * TN + 4 * lq) + (size_t)min(b * RB + 16 * t + l15, p.M - 1) * p.N + 16 * c);
            asm volatile("" ::: "memory");
        }
        dh_f32x4 acc[TN][TM];
#pragma unroll
        for (int c = 0; c < TN; ++c)
#pragma unroll
            for (int t = 0; t < TM; ++t) acc[c][t] = dh_f32x4{0.f, 0.f, 0.f, 0.f};
        uint4 fa[PF + 1];
        auto rd = [&](int q) {                            // q = TM f + t: fragment step f = 2 s + kk, row tile t
            const int f = q / TM, t = q - f * TM, s = f >> 1, kk = f & 1;
            fa[q % (PF + 1)] = *reinterpret_cast<const uint4*>(lds + rd_base[kk][buf] + (s * SLABB + t * 2048));
        };
#pragma unroll
        for (int q = 0; q < PF; ++q) rd(q);
#pragma unroll
        for (int q = 0; q < KF * TM; ++q) {
            const int f = q / TM, t = q - f * TM;
            if (q + PF < KF * TM) rd(q + PF);
#pragma unroll
            for (int c = 0; c < TN; ++c) acc[c][t] = Op16<OT>::mfma(wf[c][f], fa[q % (PF + 1)], acc[c][t]);
            __builtin_amdgcn_sched_barrier(0);
        }
        // BatchNorm + ReLU on the accumulators, stored straight from them: acc[c][t][r] = pixel b RB + 16 t + l15, channel
        // n0 + 16 (wave TN + c) + 4 lq + r -- 8 bytes per lane (the staged 16-byte form cost a second barrier per block).  The store is
        // unconditional (the waits above count it): lanes past M re-store pixel M - 1's own values (rows past M were loaded as copies of it)
#pragma unroll
        for (int c = 0; c < TN; ++c)
#pragma unroll
            for (int t = 0; t < TM; ++t) {
                float4 sc = sc4[c], sh = sh4[c];
                if constexpr (TN > 1) { sc = bn_lds[c * 8]; sh = bn_lds[c * 8 + 4]; }
                float v0 = fmaf(acc[c][t][0], sc.x, sh.x), v1 = fmaf(acc[c][t][1], sc.y, sh.y);
                float v2 = fmaf(acc[c][t][2], sc.z, sh.z), v3 = fmaf(acc[c][t][3], sc.w, sh.w);
                if constexpr (RES) {
                    float r0, r1, r2, r3;
                    Op16<OT>::unpack2(rq[c][t].x, r0, r1); Op16<OT>::unpack2(rq[c][t].y, r2, r3);
                    v0 += r0; v1 += r1; v2 += r2; v3 += r3;
                }
                if (p.relu) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); v2 = fmaxf(v2, 0.f); v3 = fmaxf(v3, 0.f); }
                uint2 pk;
                pk.x = (uint32_t)Op16<OT>::from_f32(v0) | ((uint32_t)Op16<OT>::from_f32(v1) << 16);
                pk.y = (uint32_t)Op16<OT>::from_f32(v2) | ((uint32_t)Op16<OT>::from_f32(v3) << 16);
                const int m = min(b * RB + 16 * t + l15, p.M - 1);
                *reinterpret_cast<uint2*>(ybase + (size_t)m * p.N + 16 * c) = pk;
                if constexpr (F1 > 0) {                  // the same 4 channels into the operand tile: k block, 16-byte chunk, half
                    const int ch0 = 16 * (wave * TN + c) + 4 * lq, r = 16 * t + l15;
                    *reinterpret_cast<uint2*>(otile + (ch0 >> 6) * (RB * 128) + r * 128 + ((((ch0 & 63) >> 3) ^ (r & 7)) << 4) + (lq & 1) * 8) = pk;
                }
            }
        if constexpr (F1 > 0) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                 // the block's 256 channels of every row are in the tile
            const int ct = wave & 3, rg = wave >> 2;
            dh_f32x4 acc1[4];
#pragma unroll
            for (int tt = 0; tt < 4; ++tt) acc1[tt] = dh_f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int g = 0; g < 8; ++g)                   // k = 32 g ..: k block g / 2, half g % 2
#pragma unroll
                for (int tt = 0; tt < 4; ++tt) {
                    const uint4 fb = *reinterpret_cast<const uint4*>(otile + (g >> 1) * (RB * 128) + (16 * (4 * rg + tt) + l15) * 128 +
                                                                    ((((g & 1) * 4 + lq) ^ (l15 & 7)) << 4));
                    acc1[tt] = Op16<OT>::mfma(w1f[g], fb, acc1[tt]);
                }
#pragma unroll
            for (int tt = 0; tt < 4; ++tt) {
                const float v0 = fmaxf(fmaf(acc1[tt][0], sc1.x, sh1.x), 0.f), v1 = fmaxf(fmaf(acc1[tt][1], sc1.y, sh1.y), 0.f);
                const float v2 = fmaxf(fmaf(acc1[tt][2], sc1.z, sh1.z), 0.f), v3 = fmaxf(fmaf(acc1[tt][3], sc1.w, sh1.w), 0.f);
                uint2 pk;
                pk.x = (uint32_t)Op16<OT>::from_f32(v0) | ((uint32_t)Op16<OT>::from_f32(v1) << 16);
                pk.y = (uint32_t)Op16<OT>::from_f32(v2) | ((uint32_t)Op16<OT>::from_f32(v3) << 16);
                const int m = min(b * RB + 16 * (4 * rg + tt) + l15, p.M - 1);
                *reinterpret_cast<uint2*>(p.y1n + (size_t)m * F1 + 16 * ct + 4 * lq) = pk;
            }
        }
    };
    for (int i = 0; i < nmine; i += 2) {
        block(std::integral_constant<int, 0>{}, i);
        if (i + 1 < nmine) block(std::integral_constant<int, 1>{}, i + 1);
    }
}
}  // namespace

// 1 when dh_conv1x1_wreg_nhwc takes the layer: Cin 256, 512 or 1,024, Cout a multiple of 128 with Cout / 128 in {1, 2, 4, 8, 16}, enough pixels
// to give every CU several 64 KB blocks
extern "C" int dh_conv1x1_wreg_supported(long long M, int Cin, int Cout) {
    const int nb = Cout / 128;
    return (Cin == 256 || Cin == 512 || Cin == 1024) && (Cout % 128) == 0 && (nb == 1 || nb == 2 || nb == 4 || nb == 8 || nb == 16) &&
           M >= 8192 && M * Cin * 2 < (1ll << 32);
}

// y [M, Cout] = relu?((x [M, Cin] w^T) * scale + shift), channels-last rows (a 1x1 stride-1 convolution + BatchNorm [+ ReLU] without
// residual, or for Cin = 512 + residual [M, Cout] before the ReLU: conv3 of the stage-4 bottlenecks); w_packed =
// dh_pack_mfma_fragments(w [Cout, Cin]).  Bit-identical to dh_conv2d_nhwc_bn_act(KS = 1).
extern "C" int dh_conv1x1_wreg_nhwc(const void* x, const void* w_packed, const float* scale, const float* shift, const void* residual, void* y,
                                    long long M, int Cin, int Cout, int relu, int dtype, void* stream) {
    if (!DH_IS_16BIT(dtype)) return DH_ERR_UNSUPPORTED;
    DH_REQUIRE(x && w_packed && scale && shift && y && dh_conv1x1_wreg_supported(M, Cin, Cout) && (!residual || Cin == 512));
    DH_REQUIRE(((uintptr_t)residual % 16) == 0);
    DH_REQUIRE(((uintptr_t)x % 16) == 0 && ((uintptr_t)w_packed % 16) == 0 && ((uintptr_t)y % 16) == 0 && ((uintptr_t)scale % 16) == 0 &&
               ((uintptr_t)shift % 16) == 0);
    C1Params p{};
    p.x = (const uint16_t*)x; p.wp = (const uint4*)w_packed; p.scale = scale; p.shift = shift; p.y = (uint16_t*)y; p.res = (const uint16_t*)residual;
    p.M = (int)M; p.N = Cout; p.relu = relu; p.nb_n = Cout / 128; p.wg_per_n = 256 / p.nb_n; p.ns1 = Cin / 64;
    const int rb = 65536 / (2 * Cin);                    // rows per 64 KB block: 128 / 64 / 32
    p.nblk = dh_cdiv(M, rb);
    dh_prof_set_tag("1x1");
    dh_prof_set_dims((int)M, Cout, Cin);
    DhProfScope prof("dh_conv2d_nhwc_bn_act", 2.0 * M * Cout * Cin, 2.0 * ((double)M * Cin + (double)Cout * Cin + (double)M * Cout * (residual ? 2 : 1)), stream);
    hipStream_t s = (hipStream_t)stream;
    DH_DISPATCH_16(dtype, {
        if (Cin == 256) hipLaunchKernelGGL((conv1x1_wreg_kernel<T, 8, 128, 1, 8>), dim3(256), dim3(512), 0, s, p);
        else if (Cin == 512 && residual) hipLaunchKernelGGL((conv1x1_wreg_kernel<T, 16, 64, 1, 8, true>), dim3(256), dim3(512), 0, s, p);
        else if (Cin == 512) hipLaunchKernelGGL((conv1x1_wreg_kernel<T, 16, 64, 1, 8>), dim3(256), dim3(512), 0, s, p);
        else hipLaunchKernelGGL((conv1x1_wreg_kernel<T, 32, 32, 1, 8>), dim3(256), dim3(512), 0, s, p);
    });
    DH_LAUNCH_CHECK();
}

// The end of a stage's FIRST bottleneck, relu(bn3(conv3(y)) + bn_d(downsample(x))), in the same streaming form: one GEMM over the virtual
// operand [y | x at the strided pixels] with the BatchNorm scales folded into the weights (dh_conv1x1_dual_nhwc, gemm_bf16.hip), for the
// two HBM-bound instances: stage 1 (C1 = C2 = 64, Cout = 256: 616 MB per 256 images) and stage 2 (C1 = 128, C2 = 256, Cout = 512).
// A workgroup owns 256 output channels (two column tiles per wave).
extern "C" int dh_conv1x1_dual_wreg_supported(int N, int Ho, int Wo, int H, int W, int C1, int C2, int Cout) {
    const int K = C1 + C2, nb = Cout / 256;
    const long long M = (long long)N * Ho * Wo;
    // the pixel map divides by Ho Wo and Wo with 32-bit magic numbers (exact while M Ho Wo < 2^32); byte offsets are 32-bit
    return N > 0 && Ho > 0 && Wo > 0 && H > 0 && W > 0 && (K == 128 || K == 384 || K == 768) && (C1 % 64) == 0 && (C2 % 64) == 0 &&
           (Cout % 256) == 0 && (nb == 1 || nb == 2 || nb == 4 || nb == 8) && M >= 8192 && M * Ho * Wo < (1ll << 32) &&
           M * C1 * 2 < (1ll << 32) && (long long)N * H * W * C2 * 2 < (1ll << 32);
}

// out [N, Ho, Wo, Cout] = relu?([y | x at (oy * stride, ox * stride)] w^T + shift); w_packed = dh_pack_mfma_fragments(w [Cout, C1 + C2]).
// Bit-identical to dh_conv1x1_dual_nhwc.
extern "C" int dh_conv1x1_dual_wreg_nhwc(const void* y, const void* x, const void* w_packed, const float* shift, void* out, int N, int Ho,
                                         int Wo, int C1, int H, int W, int C2, int stride, int Cout, int relu, const void* w1_packed,
                                         const float* scale1, const float* shift1, void* y1_next, int N1, int dtype, void* stream) {
    if (!DH_IS_16BIT(dtype)) return DH_ERR_UNSUPPORTED;
    const long long M = (long long)N * Ho * Wo;
    DH_REQUIRE(y && x && w_packed && shift && out && N > 0 && Ho > 0 && Wo > 0 && H > 0 && W > 0 && stride >= 1 &&
               dh_conv1x1_dual_wreg_supported(N, Ho, Wo, H, W, C1, C2, Cout));
    DH_REQUIRE((Ho - 1) * stride < H && (Wo - 1) * stride < W);
    // the NEXT bottleneck's conv1 + bn1 + relu (Cout -> N1) in the same launch: the K = 128, Cout = 256, N1 = 64 instance (stage 1)
    DH_REQUIRE(!w1_packed || (scale1 && shift1 && y1_next && N1 == 64 && C1 + C2 == 128 && Cout == 256 && relu == 1 && ((uintptr_t)w1_packed % 16) == 0 &&
                              ((uintptr_t)scale1 % 16) == 0 && ((uintptr_t)shift1 % 16) == 0 && ((uintptr_t)y1_next % 16) == 0));
    DH_REQUIRE(((uintptr_t)y % 16) == 0 && ((uintptr_t)x % 16) == 0 && ((uintptr_t)w_packed % 16) == 0 && ((uintptr_t)out % 16) == 0 &&
               ((uintptr_t)shift % 16) == 0);
    C1Params p{};
    p.x = (const uint16_t*)y; p.wp = (const uint4*)w_packed; p.scale = nullptr; p.shift = shift; p.y = (uint16_t*)out;
    p.M = (int)M; p.N = Cout; p.relu = relu; p.nb_n = Cout / 256; p.wg_per_n = 256 / p.nb_n; p.ns1 = C1 / 64;
    p.w1p = (const uint4*)w1_packed; p.scale1 = scale1; p.shift1 = shift1; p.y1n = (uint16_t*)y1_next;
    p.x2 = (const uint16_t*)x; p.howo = Ho * Wo; p.wo = Wo; p.h2w2 = H * W; p.w2 = W; p.stride2 = stride; p.c2 = C2;
    p.magic_howo = (unsigned)((1ull << 32) / (unsigned)p.howo + 1); p.magic_wo = (unsigned)((1ull << 32) / (unsigned)Wo + 1);
    const int K = C1 + C2;
    p.nblk = dh_cdiv(M, K == 128 ? 128 : K == 384 ? 64 : 32);
    dh_prof_set_tag("1x1");
    dh_prof_set_dims((int)M, Cout, K);
    DhProfScope prof("dh_conv2d_nhwc_bn_act", 2.0 * M * Cout * K,
                     2.0 * ((double)M * C1 + (double)M * C2 + (double)Cout * K + (double)M * Cout), stream);
    hipStream_t s = (hipStream_t)stream;
    DH_DISPATCH_16(dtype, {
        if (K == 128 && w1_packed) hipLaunchKernelGGL((conv1x1_wreg_kernel<T, 4, 128, 2, 8, false, 64>), dim3(256), dim3(512), 0, s, p);
        else if (K == 128) hipLaunchKernelGGL((conv1x1_wreg_kernel<T, 4, 128, 2, 8>), dim3(256), dim3(512), 0, s, p);
        else if (K == 768) hipLaunchKernelGGL((conv1x1_wreg_kernel<T, 24, 32, 2, 8>), dim3(256), dim3(512), 0, s, p);
        else hipLaunchKernelGGL((conv1x1_wreg_kernel<T, 12, 64, 2, 8>), dim3(256), dim3(512), 0, s, p);
    });
    DH_LAUNCH_CHECK();
}
