// The tail of a STAGE-1 ResNet-50 bottleneck (56 x 56 pixels, C = 64 -> 4 C = 256) in one launch, 16-bit channels-last, OPTIONALLY WITH THE
// NEXT BLOCK'S conv1 (1x1, 256 -> N1 = 64, + bn1 + relu) behind it:
//     out = relu(bn3(conv3_1x1(relu(bn2(conv2_3x3(y1))))) + residual)           y1_next = relu(bn1'(conv1'_1x1(out)))
// (torchvision Bottleneck.forward from conv2 on, and the next Bottleneck's conv1 / bn1 / relu; reference encoders.py:37-38,56 --
// layer1.1, layer1.2 and the conv1 of layer1.2 / layer2.0).
//
// Stage 1 is HBM-bound: its 256-channel tensors are 411 MB at 256 images, the tail kernel (conv3x3.hip) moves 925 MB in 193 us and the
// conv1 launch behind it reads the 411 MB it has just written back for another 108-124 us.  Here the output tile never has to be read
// again for that conv1: the structure of conv_s2.hip (patch in LDS, weights from L2 into registers, no ring) leaves LDS for it --
//   * a workgroup (4 waves = 2 pixel halves x 2 channel halves) owns FOUR output rows of one image (224 pixels = 14 row tiles): its
//     (4 + 2) x (56 + 2) x 64 input patch (45 KB, zero halo) comes into LDS once; two workgroups per CU;
//   * conv2 (18 k-steps) and conv3 (4 chunks of 64 output channels x 2 k-steps) as in conv_s2.hip: a wave computes 7 row tiles x 32
//     channels, weight fragments three k-steps ahead in registers; wave-local residual / ReLU epilogue through a 2 KB strip;
//   * FUSED conv1': the rounded 16-bit output chunk (224 pixels x 64 channels, 28 KB) is also written into a second LDS tile in the GEMM
//     operand format; behind a barrier every wave runs the chunk's two k-steps of the next conv1 for its 7 row tiles x N1 / 2
//     channels into accumulators that live across the four chunks (k ascending over the 256 channels: the chain of the stand-alone
//     conv1); after the last chunk BatchNorm + ReLU, 8-byte stores of y1_next.
// Bit-identical to dh_bottleneck_tail_nhwc followed by dh_conv2d_nhwc_bn_act / dh_conv1x1_wreg_nhwc (1x1).
#include "common.h"
#include "prof.h"

__device__ uint4 dh_s1_zero_page[4];

namespace {
struct S1Params {
    const uint16_t* x;                                   // y1 [N,56,56,64]
    const uint4* w2p; const uint4* w3p;                  // fragment-packed [18][4][64] and [2][16][64] uint4
    const float* scale2; const float* shift2; const float* scale3; const float* shift3;
    const uint16_t* res; uint16_t* out;                  // [N,56,56,256]
    const uint4* w1p; const float* scale1; const float* shift1; uint16_t* y1n;      // next conv1: [8][N1/16][64] uint4, [N,56,56,N1]
};

// N1 = 0: the tail only
template <typename OT, int N1>
__global__ __launch_bounds__(256, 2) void conv_s1_kernel(S1Params p) {
    constexpr int C = 64, HW = 56, TR = 4, PITCH = HW + 2, NPX = TR * HW, TMA = NPX / 16, TM = TMA / 2, TN = 2, NW = 4;
    constexpr int NPP = (TR + 2) * PITCH, PP_ROWS = (NPP + 7) / 8 * 8, NPIECE = PP_ROWS / 8;          // 348 patch pixels in 352 slots
    constexpr int PATCH = PP_ROWS * 128;                                     // 44 KB (one 64-channel plane)
    constexpr int YPLANE = NPX * 128;                                        // y2 / out-chunk tile: [224 pixels][128 B]
    constexpr int NSTEP2 = 9 * 2, NT2 = C / 16, NT3 = 4 * C / 16;            // k32 steps of the 3x3 (2 per tap); row tiles of the packed weights
    constexpr int TN1 = N1 / 32, NT1 = N1 / 16;                              // next conv1: column tiles per wave (N1 / 2 channels), row tiles
    constexpr int PF = 3;
    constexpr bool FUSE1 = N1 > 0;
    __shared__ __attribute__((aligned(16))) unsigned char lds[PATCH + (FUSE1 ? YPLANE : 0)];
    static_assert(YPLANE + NW * 2048 <= PATCH, "y2 tile + the per-wave fp32 strips live in the dead patch");
    unsigned char* const otile = lds + PATCH;            // FUSE1: the rounded output chunk, operand of the next conv1

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave & 1, wn = wave >> 1;              // pixel half (row tiles 7 wm ..), channel half
    const int l15 = lane & 15, lq = lane >> 4, lr = lane >> 3, lpos = lane & 7;
    const int n = blockIdx.x / (HW / TR), y0 = (blockIdx.x - n * (HW / TR)) * TR;
    const unsigned char* zero = reinterpret_cast<const unsigned char*>(dh_s1_zero_page);

    // ---- weight fragments of the first three k-steps ---------------------------------------------------------------------------------------
    const uint4* w2 = p.w2p + (size_t)(TN * wn) * 64 + lane;                // step s, tile j: w2[(s * NT2 + j) * 64]
    uint4 wq[4][TN];
#pragma unroll
    for (int s = 0; s < 3; ++s)
#pragma unroll
        for (int j = 0; j < TN; ++j) wq[s][j] = w2[(size_t)(s * NT2 + j) * 64];

    // ---- the patch: image rows y0 - 1 .. y0 + 4, columns -1 .. 56; piece = 8 patch pixels x 128 bytes ---------------------------------------
    {
        const uint16_t* img = p.x + (size_t)n * HW * HW * C;
        for (int pc = wave; pc < NPIECE; pc += NW) {
            const int pp = pc * 8 + lr;
            const int pr = (pp * 1130) >> 16, pcx = pp - pr * PITCH;       // pp / 58 for pp < 352
            const int gy = y0 - 1 + pr, gx = pcx - 1;
            const bool ok = pp < NPP && (unsigned)gy < (unsigned)HW && (unsigned)gx < (unsigned)HW;
            const void* src = ok ? (const void*)(img + (gy * HW + gx) * C + ((lpos ^ (pp & 7)) << 3)) : (const void*)zero;
            dh_lds_dma16(src, lds + pc * 1024);
        }
    }
    dh_f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = dh_f32x4{0.f, 0.f, 0.f, 0.f};
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    // patch pixel of tap (0, 0) under this lane's pixel of every row tile: q = 16 (7 wm + i) + l15 = (row q / 56, column q % 56)
    int pp0[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int q = 16 * (TM * wm + i) + l15, r = (q * 1171) >> 16;      // q / 56 for q < 224
        pp0[i] = r * PITCH + (q - r * HW);
    }

    // ---- conv2: nine taps x 2 k-halves; no barrier ------------------------------------------------------------------------------------------
    const uint4* wnext = w2 + (size_t)3 * NT2 * 64;
    int kh = 0, kw = 0;
    uint4 fa[PF + 1];
#pragma unroll 1
    for (int tp = 0; tp < 9; tp += 2) {                   // two taps per trip (4 k-steps: one turn of the weight ring); the tenth is skipped
        unsigned a0[2][TM];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int tapoff = kh * PITCH + kw;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int pp = pp0[i] + tapoff;
                a0[h][i] = (unsigned)(pp * 128 + ((lq ^ (pp & 7)) << 4));
                asm volatile("" : "+v"(a0[h][i]));
            }
            if (++kw == 3) { kw = 0; ++kh; }
        }
        const int nst = tp + 1 < 9 ? 4 : 2;               // k-steps of this trip
        auto rd = [&](int t) {
            const int u = t / TM, i = t - u * TM;
            fa[t % (PF + 1)] = *reinterpret_cast<const uint4*>(lds + (a0[u >> 1][i] ^ ((u & 1) << 6)));
        };
#pragma unroll
        for (int t = 0; t < PF; ++t) rd(t);
#pragma unroll
        for (int t = 0; t < 4 * TM; ++t) {
            const int u = t / TM, i = t - u * TM;
            if (u < nst) {
                if (i == 0 && tp * 2 + u + 3 < NSTEP2) {
#pragma unroll
                    for (int j = 0; j < TN; ++j) wq[(u + 3) & 3][j] = wnext[(size_t)j * 64];
                    wnext += NT2 * 64;
                }
                if (t + PF < nst * TM) rd(t + PF);
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = Op16<OT>::mfma(wq[u & 3][j], fa[t % (PF + 1)], acc[i][j]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }

    // ---- y2 = relu(bn2(conv2)) as 16-bit, over the dead patch in the GEMM operand format ------------------------------------------------------
    const int co = 32 * wn;                               // this wave's output channels of conv2: co + 16 j + 4 lq + r
    {
        float4 sc[TN], sh[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            sc[j] = *reinterpret_cast<const float4*>(p.scale2 + co + 16 * j + 4 * lq);
            sh[j] = *reinterpret_cast<const float4*>(p.shift2 + co + 16 * j + 4 * lq);
        }
        __syncthreads();                                  // every wave is done with the patch: it becomes the y2 tile
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int q = 16 * (TM * wm + i) + l15;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const float v0 = fmaxf(fmaf(acc[i][j][0], sc[j].x, sh[j].x), 0.f), v1 = fmaxf(fmaf(acc[i][j][1], sc[j].y, sh[j].y), 0.f);
                const float v2 = fmaxf(fmaf(acc[i][j][2], sc[j].z, sh[j].z), 0.f), v3 = fmaxf(fmaf(acc[i][j][3], sc[j].w, sh[j].w), 0.f);
                uint2 o;
                o.x = (uint32_t)Op16<OT>::from_f32(v0) | ((uint32_t)Op16<OT>::from_f32(v1) << 16);
                o.y = (uint32_t)Op16<OT>::from_f32(v2) | ((uint32_t)Op16<OT>::from_f32(v3) << 16);
                const int ch = (co + 16 * j + 4 * lq) >> 3;
                *reinterpret_cast<uint2*>(lds + q * 128 + ((ch ^ (q & 7)) << 4) + (lq & 1) * 8) = o;
            }
        }
    }
    // conv3's weights: stream [chunk c][step g], 8 steps in all; w3[(g * NT3 + 4 c + j) * 64]
    const uint4* w3 = p.w3p + (size_t)(TN * wn) * 64 + lane;
#pragma unroll
    for (int s = 0; s < 3; ++s)
#pragma unroll
        for (int j = 0; j < TN; ++j) wq[s][j] = w3[(size_t)((s & 1) * NT3 + 4 * (s >> 1) + j) * 64];
    __syncthreads();

    // ---- conv3 (1x1, 64 -> 256): 4 chunks of 64 output channels, this wave's 32 of each for its 7 row tiles; 2 k-steps per chunk -------------
    unsigned char* const strip = lds + YPLANE + wave * 2048;
    const int epx = lane >> 2, ec4 = lane & 3;            // epilogue lane role: pixel of the tile, 8-channel group
    const size_t pix0 = ((size_t)n * HW + y0) * HW + 16 * TM * wm;          // this wave's first pixel
    const unsigned a3 = (unsigned)((16 * TM * wm + l15) * 128 + ((lq ^ (l15 & 7)) << 4));            // (16 t + l15) & 7 == l15 & 7
    dh_f32x4 acc1[TM][FUSE1 ? TN1 : 1];
    const uint4* w1 = p.w1p + (size_t)(TN1 * wn) * 64 + lane;               // step g (0 .. 7), tile j: w1[(g * NT1 + j) * 64]
    if constexpr (FUSE1) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN1; ++j) acc1[i][j] = dh_f32x4{0.f, 0.f, 0.f, 0.f};
    }
    // (the chunk loop is unrolled: the weight ring's slot of step 2 c + u is static)
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        constexpr int RQ = 4;
        uint4 rq[RQ];
        const int cbase = c * C + co + 8 * ec4;
        const uint16_t* resp = p.res + pix0 * (4 * C) + cbase;
#pragma unroll
        for (int i = 0; i < RQ; ++i) rq[i] = *reinterpret_cast<const uint4*>(resp + (size_t)(16 * i + epx) * (4 * C));
        uint4 w1q[2][FUSE1 ? TN1 : 1];                    // the chunk's two k-steps of the next conv1's weights, requested before the MFMAs
        if constexpr (FUSE1) {
#pragma unroll
            for (int g = 0; g < 2; ++g)
#pragma unroll
                for (int j = 0; j < TN1; ++j) w1q[g][j] = w1[(size_t)((2 * c + g) * NT1 + j) * 64];
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = dh_f32x4{0.f, 0.f, 0.f, 0.f};
        {
            auto rd = [&](int t) {
                const int u = t / TM, i = t - u * TM;
                fa[t % (PF + 1)] = *reinterpret_cast<const uint4*>(lds + ((a3 ^ ((u & 1) << 6)) + i * 2048));
            };
#pragma unroll
            for (int t = 0; t < PF; ++t) rd(t);
#pragma unroll
            for (int t = 0; t < 2 * TM; ++t) {
                const int u = t / TM, i = t - u * TM;
                // three steps ahead in the stream [chunk][step]: step s = 2 c + u + 3
                if (i == 0 && 2 * c + u + 3 < 8) {
                    const int s = 2 * c + u + 3;
                    const uint4* src = w3 + ((size_t)(s & 1) * NT3 + 4 * (s >> 1)) * 64;
#pragma unroll
                    for (int j = 0; j < TN; ++j) wq[s & 3][j] = src[(size_t)j * 64];
                }
                if (t + PF < 2 * TM) rd(t + PF);
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = Op16<OT>::mfma(wq[(2 * c + u) & 3][j], fa[t % (PF + 1)], acc[i][j]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        asm volatile("" ::: "memory");
        const float4 s3a = *reinterpret_cast<const float4*>(p.scale3 + cbase), s3b = *reinterpret_cast<const float4*>(p.scale3 + cbase + 4);
        const float4 h3a = *reinterpret_cast<const float4*>(p.shift3 + cbase), h3b = *reinterpret_cast<const float4*>(p.shift3 + cbase + 4);
        if constexpr (FUSE1) { if (c > 0) __syncthreads(); }      // every wave has finished the previous chunk's conv1' reads of the out tile
#pragma unroll
        for (int i = 0; i < TM; ++i) {
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
            if (i + RQ < TM) rq[i % RQ] = *reinterpret_cast<const uint4*>(resp + (size_t)(16 * (i + RQ) + epx) * (4 * C));
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                float lo16, hi16;
                Op16<OT>::unpack2(w4[u], lo16, hi16);
                v[2 * u] = fmaxf(v[2 * u] + lo16, 0.f); v[2 * u + 1] = fmaxf(v[2 * u + 1] + hi16, 0.f);
            }
            uint4 o16;
            o16.x = (uint32_t)Op16<OT>::from_f32(v[0]) | ((uint32_t)Op16<OT>::from_f32(v[1]) << 16);
            o16.y = (uint32_t)Op16<OT>::from_f32(v[2]) | ((uint32_t)Op16<OT>::from_f32(v[3]) << 16);
            o16.z = (uint32_t)Op16<OT>::from_f32(v[4]) | ((uint32_t)Op16<OT>::from_f32(v[5]) << 16);
            o16.w = (uint32_t)Op16<OT>::from_f32(v[6]) | ((uint32_t)Op16<OT>::from_f32(v[7]) << 16);
            *reinterpret_cast<uint4*>(p.out + (pix0 + 16 * i + epx) * (4 * C) + cbase) = o16;
            if constexpr (FUSE1) {
                // the same 8 channels (chunk-local 32 wn + 8 ec4 ..) of pixel q into the operand tile: 16-byte chunk 4 wn + ec4 of row q
                const int q = 16 * (TM * wm + i) + epx;
                *reinterpret_cast<uint4*>(otile + q * 128 + (((4 * wn + ec4) ^ (q & 7)) << 4)) = o16;
            }
        }
        if constexpr (FUSE1) {
            __syncthreads();                              // the chunk's 64 channels of every pixel are in the tile
            // ---- next conv1: k = 64 c .. 64 c + 63 (two k-steps), this wave's 7 row tiles x N1 / 2 channels ---------------------------------
            uint4 fb[PF + 1];
            auto rd1 = [&](int t) {
                const int u = t / TM, i = t - u * TM;
                fb[t % (PF + 1)] = *reinterpret_cast<const uint4*>(otile + ((a3 ^ ((u & 1) << 6)) + i * 2048));
            };
#pragma unroll
            for (int t = 0; t < PF; ++t) rd1(t);
#pragma unroll
            for (int t = 0; t < 2 * TM; ++t) {
                const int u = t / TM, i = t - u * TM;
                if (t + PF < 2 * TM) rd1(t + PF);
#pragma unroll
                for (int j = 0; j < TN1; ++j) acc1[i][j] = Op16<OT>::mfma(w1q[u][j], fb[t % (PF + 1)], acc1[i][j]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    if constexpr (FUSE1) {
        // ---- y1_next = relu(bn1'(conv1')): acc1[i][j][r] = pixel 16 (7 wm + i) + l15, channel (N1 / 2) wn + 16 j + 4 lq + r -------------------
        const int c1 = (N1 / 2) * wn + 4 * lq;
        uint16_t* const o1 = p.y1n + (pix0 + l15) * N1 + c1;
#pragma unroll
        for (int j = 0; j < TN1; ++j) {
            const float4 sc = *reinterpret_cast<const float4*>(p.scale1 + c1 + 16 * j), sh = *reinterpret_cast<const float4*>(p.shift1 + c1 + 16 * j);
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const float v0 = fmaxf(fmaf(acc1[i][j][0], sc.x, sh.x), 0.f), v1 = fmaxf(fmaf(acc1[i][j][1], sc.y, sh.y), 0.f);
                const float v2 = fmaxf(fmaf(acc1[i][j][2], sc.z, sh.z), 0.f), v3 = fmaxf(fmaf(acc1[i][j][3], sc.w, sh.w), 0.f);
                uint2 o;
                o.x = (uint32_t)Op16<OT>::from_f32(v0) | ((uint32_t)Op16<OT>::from_f32(v1) << 16);
                o.y = (uint32_t)Op16<OT>::from_f32(v2) | ((uint32_t)Op16<OT>::from_f32(v3) << 16);
                *reinterpret_cast<uint2*>(o1 + (size_t)(16 * i) * N1 + 16 * j) = o;
            }
        }
    }
}
}  // namespace

extern "C" int dh_bottleneck_tail_s1_supported(int H, int W, int C, int N1) {
    return H == 56 && W == 56 && C == 64 && (N1 == 0 || N1 == 64);      // (N1 = 128: 112 more accumulator registers than two workgroups per CU have)
}

// out [N,56,56,256] = relu(bn3(conv3(relu(bn2(conv2(y1))))) + residual) and, with w1_packed != NULL, y1_next [N,56,56,N1] =
// relu(conv1'(out) * scale1 + shift1) (the next bottleneck's conv1 + bn1 + relu; N1 = 64).  w2_packed / w3_packed / w1_packed =
// dh_pack_mfma_fragments of w2 [64][3*3*64], w3 [256][64], w1' [N1][256].  Bit-identical to dh_bottleneck_tail_nhwc (+ the 1x1 launch).
extern "C" int dh_bottleneck_tail_s1_nhwc(const void* y1, const void* w2_packed, const float* scale2, const float* shift2,
                                          const void* w3_packed, const float* scale3, const float* shift3, const void* residual, void* out,
                                          const void* w1_packed, const float* scale1, const float* shift1, void* y1_next, int N1, int N,
                                          int H, int W, int C, int dtype, void* stream) {
    if (!DH_IS_16BIT(dtype)) return DH_ERR_UNSUPPORTED;
    DH_REQUIRE(y1 && w2_packed && scale2 && shift2 && w3_packed && scale3 && shift3 && residual && out && N > 0 &&
               dh_bottleneck_tail_s1_supported(H, W, C, w1_packed ? N1 : 0) && (long long)N * (H / 4) < (1ll << 31));
    DH_REQUIRE(!w1_packed || (scale1 && shift1 && y1_next && N1 > 0));
    DH_REQUIRE(((uintptr_t)y1 % 16) == 0 && ((uintptr_t)w2_packed % 16) == 0 && ((uintptr_t)w3_packed % 16) == 0 &&
               ((uintptr_t)residual % 16) == 0 && ((uintptr_t)out % 16) == 0 && ((uintptr_t)scale2 % 16) == 0 &&
               ((uintptr_t)shift2 % 16) == 0 && ((uintptr_t)scale3 % 16) == 0 && ((uintptr_t)shift3 % 16) == 0 &&
               ((uintptr_t)w1_packed % 16) == 0 && ((uintptr_t)scale1 % 16) == 0 && ((uintptr_t)shift1 % 16) == 0 && ((uintptr_t)y1_next % 16) == 0);
    S1Params p{};
    p.x = (const uint16_t*)y1; p.w2p = (const uint4*)w2_packed; p.w3p = (const uint4*)w3_packed;
    p.scale2 = scale2; p.shift2 = shift2; p.scale3 = scale3; p.shift3 = shift3;
    p.res = (const uint16_t*)residual; p.out = (uint16_t*)out;
    p.w1p = (const uint4*)w1_packed; p.scale1 = scale1; p.shift1 = shift1; p.y1n = (uint16_t*)y1_next;
    const double px = (double)N * H * W;
    const int n1 = w1_packed ? N1 : 0;
    hipStream_t s = (hipStream_t)stream;
    dh_prof_set_tag(n1 ? "3x3+1x1+1x1" : "3x3+1x1");
    dh_prof_set_dims(N * H * W, 4 * C, 9 * C + C / 4 + n1);
    DhProfScope prof("dh_conv2d_nhwc_bn_act", 2.0 * px * C * 9.0 * C + 2.0 * px * 4.0 * C * C + 2.0 * px * 4.0 * C * n1,
                     2.0 * (px * C + 9.0 * C * C + 4.0 * C * C + 2.0 * px * 4 * C + px * n1 + 4.0 * C * n1), stream);
    const dim3 grid(N * (H / 4));
    DH_DISPATCH_16(dtype, {
        if (n1 == 0) hipLaunchKernelGGL((conv_s1_kernel<T, 0>), grid, dim3(256), 0, s, p);
        else hipLaunchKernelGGL((conv_s1_kernel<T, 64>), grid, dim3(256), 0, s, p);
    });
    DH_LAUNCH_CHECK();
}
