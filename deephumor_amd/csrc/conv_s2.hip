// The tail of a STAGE-2 ResNet-50 bottleneck (28 x 28 pixels, C = 128 -> 4 C = 512) in one launch, 16-bit channels-last:
//     out = relu(bn3(conv3_1x1(relu(bn2(conv2_3x3(y1))))) + residual)
// (torchvision Bottleneck.forward from conv2 on; reference encoders.py:37-38,56 -- layer2.1 .. layer2.3).
//
// conv3x3_direct_kernel<FUSE> (conv3x3.hip) streams the weights of this stage through a TWO-slab LDS ring (all the LDS two 4-wave
// workgroups per CU leave next to the 46 KB patch): a counted wait and a barrier per 16 KB slab with 28 MFMAs per wave in between;
// measured back-to-back at 256 images: 169 us per launch, 123 us with neither the residual reads nor the output writes -- the 1x1
// phase (two slabs per 128-channel chunk) runs at 0.4 PF.  Here, in the structure of conv_s3.hip, NOTHING streams through LDS:
//   * a workgroup (4 waves) owns FOUR output rows of one image: its (4 + 2) x (28 + 2) x 128 input patch (46 KB, zero halo) is brought
//     into LDS once (LDS-DMA, 128-byte pixel rows per 64-channel plane, XOR-swizzled) and serves all nine taps; THREE workgroups per CU
//     (no ring to pay for): one loads its patch or stores while the others compute;
//   * the weights go from L2 straight into registers in MFMA fragment order (dh_pack_mfma_fragments), every wave owns 32 output channels
//     and loads only its own fragments, three k-steps ahead: no LDS-DMA pieces in the loop, NO BARRIER in the 3x3 or between the
//     chunks of the 1x1;
//   * a wave computes its 32 channels for all 112 pixels (7 x 2 MFMA tiles): 14 MFMAs per two 1 KB weight loads and 7 ds_read_b128;
//   * the 16-bit conv2 tile is staged over the dead patch in the GEMM operand format and is the activation operand of the 1x1 expansion
//     (4 chunks of 128 output channels, same loop); epilogue wave-local through a 2 KB strip: BatchNorm, residual add, ReLU, one
//     rounding, 16-byte stores.
// Numerics: the same MFMA chain per output as the implicit GEMM (k ascending over (tap, channel)), fp32 BatchNorm on the accumulators,
// one rounding of y2 and one of the output -- bit-identical to dh_bottleneck_tail_nhwc and to the two implicit GEMMs.
#include "common.h"
#include "prof.h"

__device__ uint4 dh_s2_zero_page[4];

namespace {
struct S2Params {
    const uint16_t* x;                                   // y1 [N,28,28,128]
    const uint4* w2p; const uint4* w3p;                  // fragment-packed [36][8][64] and [4][32][64] uint4
    const float* scale2; const float* shift2; const float* scale3; const float* shift3;
    const uint16_t* res; uint16_t* out;                  // [N,28,28,512]
    const uint4* w1p; const float* scale1; const float* shift1; uint16_t* y1n;      // next conv1 (FUSE1): fragments [16][8][64], [N,28,28,128]
};

// FUSE1: + the NEXT bottleneck's conv1 + bn1 + relu (512 -> 128) on the rounded output chunks, through a second LDS operand tile (as
// conv_s1.hip): two workgroups per CU instead of three, two barriers per chunk
template <typename OT, bool FUSE1>
__global__ __launch_bounds__(256, FUSE1 ? 2 : 3) void conv_s2_kernel(S2Params p) {
    constexpr int C = 128, CB = 2, HW = 28, TR = 4, PITCH = HW + 2, NPX = TR * HW, TM = 7, TN = 2, NW = 4;
    constexpr int NPP = (TR + 2) * PITCH, PP_ROWS = (NPP + 7) / 8 * 8, NPIECE = CB * PP_ROWS / 8;      // 180 patch pixels in 184 slots
    constexpr int PLANE = PP_ROWS * 128, PATCH = CB * PLANE;                 // 23 KB per 64-channel plane
    constexpr int YPLANE = NPX * 128;                                        // y2 staging: [k block][112 pixels][128 B]
    constexpr int NSTEP2 = 9 * CB * 2, SPT = CB * 2;                         // k32 steps of the 3x3; steps per tap
    constexpr int NT2 = C / 16, NT3 = 4 * C / 16, KS3 = C / 32;              // row tiles of the packed weights; k-steps of the 1x1
    constexpr int PF = 3;
    constexpr int N1 = 128, TN1 = 2, NT1 = N1 / 16;      // next conv1: this wave's 32 of its 128 output channels
    __shared__ __attribute__((aligned(16))) unsigned char lds[PATCH + (FUSE1 ? CB * YPLANE : 0)];
    unsigned char* const otile = lds + PATCH;            // FUSE1: the rounded output chunk [k block][112 pixels][128 B]
    static_assert(NPX == TM * 16 && CB * YPLANE + NW * 2048 <= PATCH, "y2 tile + the per-wave fp32 strips live in the dead patch");

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, lq = lane >> 4, lr = lane >> 3, lpos = lane & 7;
    const int n = blockIdx.x / (HW / TR), y0 = (blockIdx.x - n * (HW / TR)) * TR;
    const unsigned char* zero = reinterpret_cast<const unsigned char*>(dh_s2_zero_page);

    // ---- weight fragments of the first three k-steps (plain loads: the compiler counts them) ---------------------------------------------
    const uint4* w2 = p.w2p + (size_t)(TN * wave) * 64 + lane;              // step s, tile j: w2[(s * NT2 + j) * 64]
    uint4 wq[4][TN];
#pragma unroll
    for (int s = 0; s < 3; ++s)
#pragma unroll
        for (int j = 0; j < TN; ++j) wq[s][j] = w2[(size_t)(s * NT2 + j) * 64];

    // ---- the patch: image rows y0 - 1 .. y0 + 4, columns -1 .. 28, 2 planes of 64 channels; piece = 8 patch pixels x 128 bytes ------------
    {
        const uint16_t* img = p.x + (size_t)n * HW * HW * C;
        for (int pc = wave; pc < NPIECE; pc += NW) {
            const int cb = pc / (PP_ROWS / 8), pp = (pc - cb * (PP_ROWS / 8)) * 8 + lr;
            const int pr = (pp * 2185) >> 16, pcx = pp - pr * PITCH;       // pp / 30 for pp < 184
            const int gy = y0 - 1 + pr, gx = pcx - 1;
            const bool ok = pp < NPP && (unsigned)gy < (unsigned)HW && (unsigned)gx < (unsigned)HW;
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

    // patch pixel of tap (0, 0) under this lane's pixel of every row tile: q = 16 i + l15 = (row q / 28, column q % 28)
    int pp0[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int q = 16 * i + l15, r = (q * 2341) >> 16;                  // q / 28 for q < 112
        pp0[i] = r * PITCH + (q - r * HW);
    }

    // ---- conv2: nine taps x 2 channel planes x 2 k-halves; no barrier, no LDS traffic but the 7 fragment reads per step ------------------
    const uint4* wnext = w2 + (size_t)3 * NT2 * 64;      // fragments of step s + 3
    int kh = 0, kw = 0;
#pragma unroll 1
    for (int tap = 0; tap < 9; ++tap) {
        const int tapoff = kh * PITCH + kw;
        unsigned a0[TM];                                  // byte address of (pixel, k quarter lq) in plane 0, k half 0
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int pp = pp0[i] + tapoff;
            a0[i] = (unsigned)(pp * 128 + ((lq ^ (pp & 7)) << 4));
        }
#pragma unroll
        for (int i = 0; i < TM; ++i) asm volatile("" : "+v"(a0[i]));
        uint4 fa[PF + 1];
        auto rd = [&](int t) {
            const int u = t / TM, i = t - u * TM;
            fa[t % (PF + 1)] = *reinterpret_cast<const uint4*>(lds + (u >> 1) * PLANE + (a0[i] ^ ((u & 1) << 6)));
        };
#pragma unroll
        for (int t = 0; t < PF; ++t) rd(t);
#pragma unroll
        for (int t = 0; t < SPT * TM; ++t) {
            const int u = t / TM, i = t - u * TM;
            if (i == 0 && tap * SPT + u + 3 < NSTEP2) {   // the weight fragments three k-steps ahead (ring of 4: SPT == 4)
#pragma unroll
                for (int j = 0; j < TN; ++j) wq[(u + 3) & 3][j] = wnext[(size_t)j * 64];
                wnext += NT2 * 64;
            }
            if (t + PF < SPT * TM) rd(t + PF);
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = Op16<OT>::mfma(wq[u & 3][j], fa[t % (PF + 1)], acc[i][j]);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (++kw == 3) { kw = 0; ++kh; }
    }

    // ---- y2 = relu(bn2(conv2)) as 16-bit, over the dead patch in the GEMM operand format ------------------------------------------------------
    const int co = 32 * wave;                             // this wave's output channels of conv2: co + 16 j + 4 lq + r
    float4 sc[TN], sh[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        sc[j] = *reinterpret_cast<const float4*>(p.scale2 + co + 16 * j + 4 * lq);
        sh[j] = *reinterpret_cast<const float4*>(p.shift2 + co + 16 * j + 4 * lq);
    }
    const uint4* w3 = p.w3p + (size_t)(TN * wave) * 64 + lane;              // step g, chunk c, tile j: w3[(g * NT3 + 8 c + j) * 64]
#pragma unroll
    for (int s = 0; s < 3; ++s)
#pragma unroll
        for (int j = 0; j < TN; ++j) wq[s][j] = w3[(size_t)(s * NT3 + j) * 64];
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
            const int nn = co + 16 * j + 4 * lq, kb = nn >> 6, ch = (nn & 63) >> 3;
            *reinterpret_cast<uint2*>(lds + kb * YPLANE + q * 128 + ((ch ^ (q & 7)) << 4) + (lq & 1) * 8) = o;
        }
    }
    __syncthreads();

    // ---- conv3 (1x1, 128 -> 512): 4 chunks of 128 output channels, this wave's 32 of each; 4 k-steps per chunk; no barrier ------------------
    unsigned char* const strip = lds + CB * YPLANE + wave * 2048;
    const int epx = lane >> 2, ec4 = lane & 3;            // epilogue lane role: pixel of the tile, 8-channel group
    const size_t pix0 = ((size_t)n * HW + y0) * HW;
    const unsigned a3 = (unsigned)(l15 * 128 + ((lq ^ (l15 & 7)) << 4));      // (16 i + l15) & 7 == l15 & 7
    dh_f32x4 acc1[TM][FUSE1 ? TN1 : 1];
    const uint4* w1 = p.w1p + (size_t)(TN1 * wave) * 64 + lane;              // step g (0 .. 15), tile j: w1[(g * NT1 + j) * 64]
    if constexpr (FUSE1) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN1; ++j) acc1[i][j] = dh_f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll 1
    for (int c = 0; c < 4; ++c) {
        // this lane's residual chunks: the first RQ tiles requested before the MFMAs, tile i + RQ from tile i's epilogue
        constexpr int RQ = 4;
        uint4 rq[RQ];
        const int cbase = c * C + co + 8 * ec4;
        const uint16_t* resp = p.res + pix0 * (4 * C) + cbase;
#pragma unroll
        for (int i = 0; i < RQ; ++i) rq[i] = *reinterpret_cast<const uint4*>(resp + (size_t)(16 * i + epx) * (4 * C));
        uint4 w1q[KS3][FUSE1 ? TN1 : 1];                 // the chunk's four k-steps of the next conv1's weights, requested before the MFMAs
        if constexpr (FUSE1) {
#pragma unroll
            for (int g = 0; g < KS3; ++g)
#pragma unroll
                for (int j = 0; j < TN1; ++j) w1q[g][j] = w1[(size_t)((KS3 * c + g) * NT1 + j) * 64];
        }
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
            for (int t = 0; t < KS3 * TM; ++t) {
                const int u = t / TM, i = t - u * TM;
                // the stream of weight fragments is [chunk c][step g]: three steps ahead = step u + 3 of this chunk or step u - 1 of the next
                if (i == 0 && c * KS3 + u + 3 < 4 * KS3) {
                    const int un = (u + 3) & (KS3 - 1);
                    const uint4* src = w3 + ((size_t)un * NT3 + 8 * (u + 3 >= KS3 ? c + 1 : c)) * 64;
#pragma unroll
                    for (int j = 0; j < TN; ++j) wq[(u + 3) & 3][j] = src[(size_t)j * 64];
                }
                if (t + PF < KS3 * TM) rd(t + PF);
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = Op16<OT>::mfma(wq[u & 3][j], fa[t % (PF + 1)], acc[i][j]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        asm volatile("" ::: "memory");
        const float4 s3a = *reinterpret_cast<const float4*>(p.scale3 + cbase), s3b = *reinterpret_cast<const float4*>(p.scale3 + cbase + 4);
        const float4 h3a = *reinterpret_cast<const float4*>(p.shift3 + cbase), h3b = *reinterpret_cast<const float4*>(p.shift3 + cbase + 4);
        if constexpr (FUSE1) { if (c > 0) __syncthreads(); }      // every wave has finished the previous chunk's conv1' reads of the out tile
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
                // the same 8 channels (chunk-local 32 wave + 8 ec4 ..) of pixel q into the operand tile: k block wave / 2, chunk 4 (wave % 2) + ec4
                const int q = 16 * i + epx;
                *reinterpret_cast<uint4*>(otile + (wave >> 1) * YPLANE + q * 128 + (((4 * (wave & 1) + ec4) ^ (q & 7)) << 4)) = o16;
            }
        }
        if constexpr (FUSE1) {
            __syncthreads();                              // the chunk's 128 channels of every pixel are in the tile
            // ---- next conv1: k = 128 c .. 128 c + 127 (four k-steps), this wave's 32 channels x 7 row tiles -----------------------------------
            uint4 fb[PF + 1];
            auto rd1 = [&](int t) {
                const int u = t / TM, i = t - u * TM;
                fb[t % (PF + 1)] = *reinterpret_cast<const uint4*>(otile + (u >> 1) * YPLANE + ((a3 ^ ((u & 1) << 6)) + i * 2048));
            };
#pragma unroll
            for (int t = 0; t < PF; ++t) rd1(t);
#pragma unroll
            for (int t = 0; t < KS3 * TM; ++t) {
                const int u = t / TM, i = t - u * TM;
                if (t + PF < KS3 * TM) rd1(t + PF);
#pragma unroll
                for (int j = 0; j < TN1; ++j) acc1[i][j] = Op16<OT>::mfma(w1q[u][j], fb[t % (PF + 1)], acc1[i][j]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    if constexpr (FUSE1) {
        // ---- y1_next = relu(bn1'(conv1')): acc1[i][j][r] = pixel 16 i + l15, channel 32 wave + 16 j + 4 lq + r ---------------------------------
        const int c1 = 32 * wave + 4 * lq;
        uint16_t* const o1 = p.y1n + (pix0 + l15) * N1 + c1;
#pragma unroll
        for (int j = 0; j < TN1; ++j) {
            const float4 sc1 = *reinterpret_cast<const float4*>(p.scale1 + c1 + 16 * j), sh1 = *reinterpret_cast<const float4*>(p.shift1 + c1 + 16 * j);
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const float v0 = fmaxf(fmaf(acc1[i][j][0], sc1.x, sh1.x), 0.f), v1 = fmaxf(fmaf(acc1[i][j][1], sc1.y, sh1.y), 0.f);
                const float v2 = fmaxf(fmaf(acc1[i][j][2], sc1.z, sh1.z), 0.f), v3 = fmaxf(fmaf(acc1[i][j][3], sc1.w, sh1.w), 0.f);
                uint2 o;
                o.x = (uint32_t)Op16<OT>::from_f32(v0) | ((uint32_t)Op16<OT>::from_f32(v1) << 16);
                o.y = (uint32_t)Op16<OT>::from_f32(v2) | ((uint32_t)Op16<OT>::from_f32(v3) << 16);
                *reinterpret_cast<uint2*>(o1 + (size_t)(16 * i) * N1 + 16 * j) = o;
            }
        }
    }
}
}  // namespace

extern "C" int dh_bottleneck_tail_s2_supported(int H, int W, int C) { return H == 28 && W == 28 && C == 128; }

// out [N,28,28,512] = relu(bn3(conv3(relu(bn2(conv2(y1))))) + residual); w2_packed = dh_pack_mfma_fragments(w2 [128][3*3*128]),
// w3_packed = dh_pack_mfma_fragments(w3 [512][128]).  Bit-identical to dh_bottleneck_tail_nhwc.
extern "C" int dh_bottleneck_tail_s2_nhwc(const void* y1, const void* w2_packed, const float* scale2, const float* shift2,
                                          const void* w3_packed, const float* scale3, const float* shift3, const void* residual,
                                          void* out, const void* w1_packed, const float* scale1, const float* shift1, void* y1_next, int N1,
                                          int N, int H, int W, int C, int dtype, void* stream) {
    if (!DH_IS_16BIT(dtype)) return DH_ERR_UNSUPPORTED;
    DH_REQUIRE(y1 && w2_packed && scale2 && shift2 && w3_packed && scale3 && shift3 && residual && out && N > 0 &&
               dh_bottleneck_tail_s2_supported(H, W, C) && (long long)N * (H / 4) < (1ll << 31));
    // the NEXT bottleneck's conv1 + bn1 + relu (512 -> 128) in the same launch
    DH_REQUIRE(!w1_packed || (scale1 && shift1 && y1_next && N1 == 128 && ((uintptr_t)w1_packed % 16) == 0 && ((uintptr_t)scale1 % 16) == 0 &&
                              ((uintptr_t)shift1 % 16) == 0 && ((uintptr_t)y1_next % 16) == 0));
    DH_REQUIRE(((uintptr_t)y1 % 16) == 0 && ((uintptr_t)w2_packed % 16) == 0 && ((uintptr_t)w3_packed % 16) == 0 &&
               ((uintptr_t)residual % 16) == 0 && ((uintptr_t)out % 16) == 0 && ((uintptr_t)scale2 % 16) == 0 &&
               ((uintptr_t)shift2 % 16) == 0 && ((uintptr_t)scale3 % 16) == 0 && ((uintptr_t)shift3 % 16) == 0);
    S2Params p{};
    p.x = (const uint16_t*)y1; p.w2p = (const uint4*)w2_packed; p.w3p = (const uint4*)w3_packed;
    p.scale2 = scale2; p.shift2 = shift2; p.scale3 = scale3; p.shift3 = shift3;
    p.res = (const uint16_t*)residual; p.out = (uint16_t*)out;
    p.w1p = (const uint4*)w1_packed; p.scale1 = scale1; p.shift1 = shift1; p.y1n = (uint16_t*)y1_next;
    const double px = (double)N * H * W;
    const int n1 = w1_packed ? N1 : 0;
    dh_prof_set_tag(n1 ? "3x3+1x1+1x1" : "3x3+1x1");
    dh_prof_set_dims(N * H * W, 4 * C, 9 * C + C / 4 + n1);
    DhProfScope prof("dh_conv2d_nhwc_bn_act", 2.0 * px * C * 9.0 * C + 2.0 * px * 4.0 * C * C + 2.0 * px * 4.0 * C * n1,
                     2.0 * (px * C + 9.0 * C * C + 4.0 * C * C + 2.0 * px * 4 * C + px * n1 + 4.0 * C * n1), stream);
    hipStream_t s = (hipStream_t)stream;
    DH_DISPATCH_16(dtype, {
        if (n1) hipLaunchKernelGGL((conv_s2_kernel<T, true>), dim3(N * (H / 4)), dim3(256), 0, s, p);
        else hipLaunchKernelGGL((conv_s2_kernel<T, false>), dim3(N * (H / 4)), dim3(256), 0, s, p);
    });
    DH_LAUNCH_CHECK();
}
