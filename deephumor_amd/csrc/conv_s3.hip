// The tail of a STAGE-3 ResNet-50 bottleneck (14 x 14 pixels, C = 256 -> 4 C = 1024) in one launch, 16-bit channels-last:
//     out = relu(bn3(conv3_1x1(relu(bn2(conv2_3x3(y1))))) + residual)
// (torchvision Bottleneck.forward from conv2 on; reference encoders.py:37-38,56 -- layer3.1 .. layer3.5).
//
// As two implicit GEMMs (gemm_bf16.hip) these layers stream BOTH operands through LDS -- 9.2 KB of LDS-DMA per output pixel for the
// 3x3, 32 pieces per 512 MFMA cycles -- and run at 0.25-0.34 of the MFMA peak (93 + 69 us per block at 256 images).  The stage-1/2
// form (conv3x3.hip: patch in LDS, weight slabs through an LDS ring) does not fit: one image's patch is 128 KB at 256 channels and
// leaves no room for a ring.  Here NOTHING streams through LDS:
//   * ONE IMAGE per workgroup (8 waves, one workgroup per CU): its 16 x 16 x 256 input patch (zero halo included) is brought into LDS
//     once (LDS-DMA, 128-byte pixel rows per 64-channel plane, XOR-swizzled) and serves all nine taps; the MFMA's activation operand is
//     formed at ds_read time (lane = (pixel, k quarter) reads 8 channels of patch pixel (r + kh, c + kw));
//   * the WEIGHTS go from L2 straight into registers: they are pre-packed in MFMA fragment order (dh_pack_mfma_fragments: one
//     coalesced 1 KB load per 16 output channels x 32 k), every wave owns 32 output channels and loads only its own fragments, three
//     k-steps ahead.  No ring, no LDS-DMA pieces in the loop, no barrier: the eight waves run the whole 3x3 unsynchronised;
//   * a wave computes its 32 channels for all 196 pixels (13 x 2 MFMA tiles, 104 accumulator registers): 26 MFMAs per two 1 KB
//     weight loads and 13 ds_read_b128;
//   * the 16-bit conv2 tile is staged over the dead patch in the GEMM operand format and is the activation operand of the 1x1
//     expansion (4 chunks of 256 output channels, same loop); epilogue wave-local through a 2 KB strip: BatchNorm, residual add,
//     ReLU, one rounding, 16-byte stores.
// Numerics: the same MFMA chain per output as the implicit GEMM (k ascending over (tap, channel)), fp32 BatchNorm on the accumulators,
// one rounding of y2 and one of the output -- bit-identical to dh_conv2d_nhwc_bn_act (3x3) + dh_conv2d_nhwc_bn_act (1x1, residual).
#include "common.h"
#include "prof.h"

__device__ uint4 dh_s3_zero_page[4];

namespace {
struct S3Params {
    const uint16_t* x;                                   // y1 [N,14,14,256]
    const uint4* w2p; const uint4* w3p;                  // fragment-packed [72][16][64] and [8][64][64] uint4
    const float* scale2; const float* shift2; const float* scale3; const float* shift3;
    const uint16_t* res; uint16_t* out;                  // [N,14,14,1024]
    uint16_t* y2;                                        // unfused form: [N,14,14,256]
};

template <typename OT, bool FUSE>
__global__ __launch_bounds__(512, 1) void conv_s3_kernel(S3Params p) {
    constexpr int C = 256, CB = 4, HW = 14, PITCH = 16, NPX = HW * HW, TM = 13, TN = 2;
    constexpr int PLANE = PITCH * PITCH * 128, PATCH = CB * PLANE;           // 32 KB per 64-channel plane, 128 KB
    constexpr int YPLANE = TM * 16 * 128;                                    // y2 staging: [k block][208 pixels][128 B]
    constexpr int NSTEP2 = 9 * CB * 2;                                       // k32 steps of the 3x3
    constexpr int PF = 3;                                                    // LDS fragment reads run PF pixel tiles ahead of their MFMAs
    __shared__ __attribute__((aligned(16))) unsigned char lds[PATCH];
    static_assert(CB * YPLANE + 8 * 2048 <= PATCH, "y2 tile + the per-wave fp32 strips live in the dead patch");

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, lq = lane >> 4, lr = lane >> 3, lpos = lane & 7;
    const int n = blockIdx.x;
    const unsigned char* zero = reinterpret_cast<const unsigned char*>(dh_s3_zero_page);

    // ---- weight fragments of the first three k-steps (plain loads: the compiler counts them) -------------------------------------
    const uint4* w2 = p.w2p + (size_t)(2 * wave) * 64 + lane;               // step s, tile j: w2[(s * 16 + j) * 64]
    uint4 wq[4][TN];
#pragma unroll
    for (int s = 0; s < 3; ++s)
#pragma unroll
        for (int j = 0; j < TN; ++j) wq[s][j] = w2[(size_t)(s * 16 + j) * 64];

    // ---- the patch: rows -1 .. 14, columns -1 .. 14, 4 planes of 64 channels; piece = 8 patch pixels x 128 bytes -------------------
    {
        const uint16_t* img = p.x + (size_t)n * NPX * C;
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int pc = wave * 16 + u, cb = pc >> 5, pp = (pc & 31) * 8 + lr;
            const int gy = (pp >> 4) - 1, gx = (pp & 15) - 1;
            const bool ok = (unsigned)gy < (unsigned)HW && (unsigned)gx < (unsigned)HW;
            const void* src = ok ? (const void*)(img + (gy * HW + gx) * C + cb * 64 + ((lpos ^ (pp & 7)) << 3)) : (const void*)zero;
            dh_lds_dma16(src, lds + pc * 1024);
        }
    }
    dh_f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = dh_f32x4{0.f, 0.f, 0.f, 0.f};
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's patch pieces (and its first fragments) have landed
    __syncthreads();

    // ---- conv2: nine taps x 4 channel blocks x 2 k-halves; no barrier, no LDS traffic but the 13 fragment reads per step --------------
    const uint4* wnext = w2 + (size_t)3 * 16 * 64;       // fragments of step s + 3
    int kh = 0, kw = 0;
#pragma unroll 1
    for (int tap = 0; tap < 9; ++tap) {
        const int tapoff = kh * PITCH + kw;
        unsigned a0[TM];                                  // byte address of (pixel, k quarter lq) in plane 0, k half 0
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            // patch pixel of this tap under the lane's pixel of m-tile i (re-derived per tap: 13 registers the loop does not have;
            // q / 14 == (q * 4682) >> 16 for q < 208; pixels past 195 of the last tile are clamped and never stored)
            const int q = min(16 * i + l15, NPX - 1), r = (q * 4682) >> 16;
            const int pp = r * PITCH + (q - r * HW) + tapoff;
            a0[i] = (unsigned)(pp * 128 + ((lq ^ (pp & 7)) << 4));
        }
        // (opaque: the compiler otherwise re-derives every address in front of every read instead of keeping the 13 registers)
#pragma unroll
        for (int i = 0; i < TM; ++i) asm volatile("" : "+v"(a0[i]));
        // The tap's 8 k-steps x 13 pixel tiles as ONE software pipeline: fragment t = 13 u + i is read PF tiles ahead of its two
        // MFMAs into a ring of PF + 1 registers (left to itself the compiler reads every fragment into the SAME register right in
        // front of its MFMAs -- an exposed LDS round trip per tile)
        uint4 fa[PF + 1];
        auto rd = [&](int t) {
            const int u = t / TM, i = t - u * TM;
            fa[t % (PF + 1)] = *reinterpret_cast<const uint4*>(lds + (u >> 1) * PLANE + (a0[i] ^ ((u & 1) << 6)));
        };
#pragma unroll
        for (int t = 0; t < PF; ++t) rd(t);
#pragma unroll
        for (int t = 0; t < 8 * TM; ++t) {
            const int u = t / TM, i = t - u * TM;
            if (i == 0 && tap * 8 + u + 3 < NSTEP2) {     // the weight fragments three k-steps ahead
#pragma unroll
                for (int j = 0; j < TN; ++j) wq[(u + 3) & 3][j] = wnext[(size_t)j * 64];
                wnext += 16 * 64;
            }
            if (t + PF < 8 * TM) rd(t + PF);
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = Op16<OT>::mfma(wq[u & 3][j], fa[t % (PF + 1)], acc[i][j]);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (++kw == 3) { kw = 0; ++kh; }
    }

    // ---- y2 = relu(bn2(conv2)) as 16-bit ---------------------------------------------------------------------------------------------
    const int co = 32 * wave;                             // this wave's output channels of conv2: co + 16 j + 4 lq + r
    float4 sc[TN], sh[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        sc[j] = *reinterpret_cast<const float4*>(p.scale2 + co + 16 * j + 4 * lq);
        sh[j] = *reinterpret_cast<const float4*>(p.shift2 + co + 16 * j + 4 * lq);
    }
    const uint4* w3 = p.w3p + (size_t)(2 * wave) * 64 + lane;                // step g, chunk c, tile j: w3[(g * 64 + 16 c + j) * 64]
    if constexpr (FUSE) {
#pragma unroll
        for (int s = 0; s < 3; ++s)
#pragma unroll
            for (int j = 0; j < TN; ++j) wq[s][j] = w3[(size_t)(s * 64 + j) * 64];
    }
    __syncthreads();                                      // every wave is done with the patch: it becomes the y2 tile
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int q = 16 * i + l15;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const float v0 = fmaxf(fmaf(acc[i][j][0], sc[j].x, sh[j].x), 0.f), v1 = fmaxf(fmaf(acc[i][j][1], sc[j].y, sh[j].y), 0.f);
            const float v2 = fmaxf(fmaf(acc[i][j][2], sc[j].z, sh[j].z), 0.f), v3 = fmaxf(fmaf(acc[i][j][3], sc[j].w, sh[j].w), 0.f);
            uint2 o;
            o.x = (uint32_t)Op16<OT>::from_f32(v0) | ((uint32_t)Op16<OT>::from_f32(v1) << 16);
            o.y = (uint32_t)Op16<OT>::from_f32(v2) | ((uint32_t)Op16<OT>::from_f32(v3) << 16);
            // GEMM operand format [k block][pixel][128 B], 16-byte chunks XOR-swizzled by the pixel index
            const int nn = co + 16 * j + 4 * lq, kb = nn >> 6, ch = (nn & 63) >> 3;
            *reinterpret_cast<uint2*>(lds + kb * YPLANE + q * 128 + ((ch ^ (q & 7)) << 4) + (lq & 1) * 8) = o;
        }
    }
    __syncthreads();

    if constexpr (!FUSE) {
        // the image's 196 x 256 outputs are one contiguous block of the channels-last tensor: 16-byte chunks, lane-contiguous
        uint16_t* out = p.y2 + (size_t)n * NPX * C;
        for (int idx = tid; idx < NPX * (C / 8); idx += 512) {
            const int q = idx >> 5, c8 = idx & 31;
            *reinterpret_cast<uint4*>(out + (size_t)idx * 8) =
                *reinterpret_cast<const uint4*>(lds + (c8 >> 3) * YPLANE + q * 128 + (((c8 & 7) ^ (q & 7)) << 4));
        }
        return;
    } else {
        // ---- conv3 (1x1, 256 -> 1024): 4 chunks of 256 output channels, this wave's 32 of each; 8 k-steps per chunk ----------------------
        unsigned char* const strip = lds + CB * YPLANE + wave * 2048;
        const int epx = lane >> 2, ec4 = lane & 3;        // epilogue lane role: pixel of the tile, 8-channel group
        const size_t pix0 = (size_t)n * NPX;
        const unsigned a3 = (unsigned)(l15 * 128 + ((lq ^ (l15 & 7)) << 4));      // (16 i + l15) & 7 == l15 & 7
#pragma unroll 1
        for (int c = 0; c < 4; ++c) {
            // this lane's residual chunks: the first RQ tiles requested before the MFMAs, tile i + RQ from tile i's epilogue
            constexpr int RQ = 4;
            uint4 rq[RQ];
            const int cbase = c * 256 + co + 8 * ec4;
            const uint16_t* resp = p.res + pix0 * 1024 + cbase;
#pragma unroll
            for (int i = 0; i < RQ; ++i) rq[i] = *reinterpret_cast<const uint4*>(resp + (size_t)min(16 * i + epx, NPX - 1) * 1024);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = dh_f32x4{0.f, 0.f, 0.f, 0.f};
            {
                uint4 fa[PF + 1];
                auto rd = [&](int t) {
                    const int u = t / TM, i = t - u * TM;
                    fa[t % (PF + 1)] = *reinterpret_cast<const uint4*>(lds + (u >> 1) * YPLANE + ((a3 ^ ((u & 1) << 6)) + i * 2048));
                };
#pragma unroll
                for (int t = 0; t < PF; ++t) rd(t);
#pragma unroll
                for (int t = 0; t < 8 * TM; ++t) {
                    const int u = t / TM, i = t - u * TM;
                    // the stream of weight fragments is [step g][chunk c]: three steps ahead = step u + 3 of this chunk or step u - 5
                    // of the next one
                    if (i == 0 && c * 8 + u + 3 < 32) {
                        const int un = (u + 3) & 7;
                        const uint4* src = w3 + ((size_t)un * 64 + 16 * (u + 3 >= 8 ? c + 1 : c)) * 64;
#pragma unroll
                        for (int j = 0; j < TN; ++j) wq[(u + 3) & 3][j] = src[(size_t)j * 64];
                    }
                    if (t + PF < 8 * TM) rd(t + PF);
#pragma unroll
                    for (int j = 0; j < TN; ++j) acc[i][j] = Op16<OT>::mfma(wq[u & 3][j], fa[t % (PF + 1)], acc[i][j]);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            // this lane's 8 channels' BatchNorm constants: requested only now (16 registers the MFMA loop does not have; L2 hits)
            asm volatile("" ::: "memory");
            const float4 s3a = *reinterpret_cast<const float4*>(p.scale3 + cbase), s3b = *reinterpret_cast<const float4*>(p.scale3 + cbase + 4);
            const float4 h3a = *reinterpret_cast<const float4*>(p.shift3 + cbase), h3b = *reinterpret_cast<const float4*>(p.shift3 + cbase + 4);
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                // accumulator layout -> strip: lane (pixel l15, quarter lq) holds channels 16 j + 4 lq .. + 3 = 16-byte chunk 4 j + lq
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    *reinterpret_cast<float4*>(strip + l15 * 128 + (((4 * j + lq) ^ (l15 & 7)) << 4)) =
                        make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                const float4 lo = *reinterpret_cast<const float4*>(strip + epx * 128 + (((2 * ec4) ^ (epx & 7)) << 4));
                const float4 hi = *reinterpret_cast<const float4*>(strip + epx * 128 + (((2 * ec4 + 1) ^ (epx & 7)) << 4));
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                float v[8] = {fmaf(lo.x, s3a.x, h3a.x), fmaf(lo.y, s3a.y, h3a.y), fmaf(lo.z, s3a.z, h3a.z), fmaf(lo.w, s3a.w, h3a.w),
                              fmaf(hi.x, s3b.x, h3b.x), fmaf(hi.y, s3b.y, h3b.y), fmaf(hi.z, s3b.z, h3b.z), fmaf(hi.w, s3b.w, h3b.w)};
                const uint32_t w4[4] = {rq[i % RQ].x, rq[i % RQ].y, rq[i % RQ].z, rq[i % RQ].w};
                if (i + RQ < TM) rq[i % RQ] = *reinterpret_cast<const uint4*>(resp + (size_t)min(16 * (i + RQ) + epx, NPX - 1) * 1024);
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    float lo16, hi16;
                    Op16<OT>::unpack2(w4[u], lo16, hi16);
                    v[2 * u] = fmaxf(v[2 * u] + lo16, 0.f); v[2 * u + 1] = fmaxf(v[2 * u + 1] + hi16, 0.f);
                }
                const int q = 16 * i + epx;
                if (q < NPX) store16(reinterpret_cast<OT*>(p.out) + (pix0 + q) * 1024 + cbase, v);
            }
        }
    }
}

// w [R][K] (16-bit, row-major) -> MFMA A-operand fragments: out[(s * (R / 16) + rt) * 64 + lane] (16 bytes) = the 8 consecutive k
// values 32 s + 8 (lane >> 4) .. of row 16 rt + (lane & 15): one coalesced 1 KB load per (16 rows x 32 k) fragment
__global__ __launch_bounds__(256) void pack_fragments_kernel(const uint16_t* __restrict__ w, uint4* __restrict__ out, int R, int K) {
    const size_t total = (size_t)(K / 32) * (R / 16) * 64;
    for (size_t idx = blockIdx.x * 256ull + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256ull) {
        const int lane = (int)(idx & 63);
        const size_t t = idx >> 6;
        const int rt = (int)(t % (R / 16)), s = (int)(t / (R / 16));
        out[idx] = *reinterpret_cast<const uint4*>(w + (size_t)(16 * rt + (lane & 15)) * K + 32 * s + 8 * (lane >> 4));
    }
}
}  // namespace

extern "C" int dh_pack_mfma_fragments(const void* w, void* out, int R, int K, void* stream) {
    DH_REQUIRE(w && out && R > 0 && K > 0 && (R % 16) == 0 && (K % 32) == 0 && ((uintptr_t)w % 16) == 0 && ((uintptr_t)out % 16) == 0);
    DhProfScope prof("dh_pack_mfma_fragments", 0.0, 4.0 * R * K, stream);
    const size_t total = (size_t)(K / 32) * (R / 16) * 64;
    const int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(pack_fragments_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)w, (uint4*)out, R, K);
    DH_LAUNCH_CHECK();
}

extern "C" int dh_bottleneck_tail_s3_supported(int H, int W, int C) { return H == 14 && W == 14 && C == 256; }

extern "C" int dh_bottleneck_tail_s3_nhwc(const void* y1, const void* w2_packed, const float* scale2, const float* shift2,
                                          const void* w3_packed, const float* scale3, const float* shift3, const void* residual,
                                          void* out, int N, int H, int W, int C, int dtype, void* stream) {
    if (!DH_IS_16BIT(dtype)) return DH_ERR_UNSUPPORTED;
    DH_REQUIRE(y1 && w2_packed && scale2 && shift2 && out && N > 0 && dh_bottleneck_tail_s3_supported(H, W, C));
    DH_REQUIRE((w3_packed && scale3 && shift3 && residual) || (!w3_packed && !residual));
    DH_REQUIRE(((uintptr_t)y1 % 16) == 0 && ((uintptr_t)w2_packed % 16) == 0 && ((uintptr_t)w3_packed % 16) == 0 &&
               ((uintptr_t)residual % 16) == 0 && ((uintptr_t)out % 16) == 0 && ((uintptr_t)scale2 % 16) == 0 &&
               ((uintptr_t)shift2 % 16) == 0 && ((uintptr_t)scale3 % 16) == 0 && ((uintptr_t)shift3 % 16) == 0);
    S3Params p{};
    p.x = (const uint16_t*)y1; p.w2p = (const uint4*)w2_packed; p.w3p = (const uint4*)w3_packed;
    p.scale2 = scale2; p.shift2 = shift2; p.scale3 = scale3; p.shift3 = shift3;
    p.res = (const uint16_t*)residual; p.out = (uint16_t*)out; p.y2 = (uint16_t*)out;
    const double px = (double)N * H * W;
    hipStream_t s = (hipStream_t)stream;
    if (w3_packed) {
        dh_prof_set_tag("3x3+1x1");
        dh_prof_set_dims(N * H * W, 4 * C, 9 * C + C / 4);
        DhProfScope prof("dh_conv2d_nhwc_bn_act", 2.0 * px * C * 9.0 * C + 2.0 * px * 4.0 * C * C,
                         2.0 * (px * C + 9.0 * C * C + 4.0 * C * C + 2.0 * px * 4 * C), stream);
        DH_DISPATCH_16(dtype, hipLaunchKernelGGL((conv_s3_kernel<T, true>), dim3(N), dim3(512), 0, s, p));
        DH_LAUNCH_CHECK();
    }
    dh_prof_set_tag("3x3");
    dh_prof_set_dims(N * H * W, C, 9 * C);
    DhProfScope prof("dh_conv2d_nhwc_bn_act", 2.0 * px * C * 9.0 * C, 2.0 * (px * C + 9.0 * C * C + px * C), stream);
    DH_DISPATCH_16(dtype, hipLaunchKernelGGL((conv_s3_kernel<T, false>), dim3(N), dim3(512), 0, s, p));
    DH_LAUNCH_CHECK();
}
