// The 3x3 convolution + BatchNorm + ReLU of the STAGE-4 ResNet-50 bottlenecks (7 x 7 pixels, 512 -> 512 channels, stride 1), 16-bit
// channels-last:  y2 = relu(bn2(conv2_3x3(y1)))   (torchvision Bottleneck.conv2 / bn2 / relu of layer4.1 and layer4.2; reference
// encoders.py:37-38, :56).
//
// As an implicit GEMM {12544 x 512 x 4608} (gemm_bf16.hip, 128 x 128 tiles) the layer streams both operands through LDS -- 64 B of
// L2 -> LDS traffic per 4,096 flop, which is the per-CU ingest rate, not the matrix pipe -- and gives 392 tiles to 256 CUs (two rounds, the
// second 53 % full): 92 us, 0.26 of the MFMA peak.  The stage-3 form (conv_s3.hip: one image per workgroup, all channels) has 4.7 MB of
// weights per workgroup here.  This kernel:
//   * a workgroup = (a PAIR of images, one HALF of the output channels): 128 x 2 = 256 workgroups for 256 images, ONE round; the
//     workgroups of one half sit on the same XCDs (blockIdx % 8 parity), so an L2 holds 2.4 MB of weights;
//   * the pair's 98 pixels x 512 channels (98 KB) come into LDS once (LDS-DMA, 128-byte pixel rows per 64-channel plane, XOR swizzle) and
//     serve all nine taps WITHOUT a halo: the tap's neighbour is addressed at ds_read time, a neighbour outside the image is the zero
//     pixel (index 98);
//   * the WEIGHTS go from L2 straight into registers in MFMA fragment order (dh_pack_mfma_fragments): 4 waves (one per SIMD), each owns
//     64 output channels (4 column tiles) x all 98 pixels (7 row tiles): a pixel fragment read from LDS feeds FOUR MFMAs (conv_s3's two
//     leave the LDS port as busy as the matrix pipe), 28 MFMAs per 4 KB of weights, weight fragments seven k-steps ahead in a ring of 8;
//   * no barrier in the loop; BatchNorm + ReLU on the accumulators, 8-byte stores.
// Numerics: the same MFMA chain per output as the implicit GEMM (k ascending over (tap, channel)), fp32 BatchNorm on the accumulators, one
// rounding -- bit-identical to dh_conv2d_nhwc_bn_act (3x3, stride 1, pad 1).
// MEASURED (256 images, back-to-back launches): 59 us on random data, 49 us on zeros (the chip holds a lower clock under random-data MFMA
// load) against 81 us for the tile kernel; with every weight load aimed at one L1-resident 4 KB 54 us, without the LDS reads 57 us:
// neither L2 nor LDS bounds it -- 64.5 k MFMA cycles per wave are 27 us at 2.4 GHz, the rest is the exposed patch load (all 256
// workgroups at once), the epilogue and the clock.
#include "common.h"
#include "prof.h"

__device__ uint4 dh_s4_zero_page[4];

namespace {
struct S4Params {
    const uint16_t* x; const uint4* wp; const float* scale; const float* shift; uint16_t* y; int N;
};

template <typename OT>
__global__ __launch_bounds__(256, 1) void conv_s4_kernel(S4Params p) {
    constexpr int C = 512, CB = 8, HW = 7, NPX = 49, ROWS = 2 * NPX, ZP = ROWS, TM = 7, TN = 4, NG = 13;
    constexpr int PLANE = NG * 8 * 128;                   // 104 pixel slots (98 pixels, the zero pixel, 5 spare) x 128 B per 64-channel plane
    constexpr int RING = 8, PF = 3;
    __shared__ __attribute__((aligned(16))) unsigned char lds[CB * PLANE];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, lq = lane >> 4, lr = lane >> 3, lpos = lane & 7;
    const int xcd = blockIdx.x & 7, half = xcd & 1, pair = (blockIdx.x >> 3) * 4 + (xcd >> 1);
    if (2 * pair >= p.N) return;
    const int nvalid = min(2, p.N - 2 * pair) * NPX;      // 98, or 49 for the odd last image
    const unsigned char* zero = reinterpret_cast<const unsigned char*>(dh_s4_zero_page);

    // ---- weight fragments of the first RING - 1 k-steps (plain loads: the compiler counts them) ---------------------------------------
    const uint4* w = p.wp + (size_t)(half * 16 + wave * TN) * 64 + lane;     // step s, tile j: w[(s * 32 + j) * 64]
    uint4 wq[RING][TN];
#pragma unroll
    for (int s = 0; s < RING - 1; ++s)
#pragma unroll
        for (int j = 0; j < TN; ++j) wq[s][j] = w[(size_t)(s * 32 + j) * 64];

    // ---- the pair's pixels: 8 planes x 13 groups of 8 pixel rows; slots past the pair's pixels (the zero pixel among them) read zeros ----
    {
        const uint16_t* img = p.x + (size_t)pair * ROWS * C;
#pragma unroll
        for (int u = 0; u < CB * NG / 4; ++u) {
            const int pc = wave * (CB * NG / 4) + u, cb = pc / NG, g = pc - cb * NG, pp = g * 8 + lr;
            const void* src = pp < nvalid ? (const void*)(img + pp * C + cb * 64 + ((lpos ^ (pp & 7)) << 3)) : (const void*)zero;
            dh_lds_dma16(src, lds + cb * PLANE + g * 1024);
        }
    }
    dh_f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = dh_f32x4{0.f, 0.f, 0.f, 0.f};
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's pieces (and its first fragments) have landed
    __syncthreads();

    // this lane's pixel of every row tile: image, row, column (q >= nvalid: the zero pixel for every tap)
    int prow[TM], pcol[TM], pbase[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int q = 16 * i + l15, im = q >= NPX ? 1 : 0, px = q - NPX * im, r = (px * 37) >> 8;
        prow[i] = q < nvalid ? r : -4; pcol[i] = px - HW * r; pbase[i] = NPX * im;
    }

    // ---- nine taps x 8 channel planes x 2 k-halves = 144 k-steps of 7 x 4 MFMAs; no barrier ------------------------------------------------
    const uint4* wnext = w + (size_t)(RING - 1) * 32 * 64;                   // fragments of step s + RING - 1
    int kh = 0, kw = 0;
#pragma unroll 1
    for (int tap = 0; tap < 9; ++tap) {
        unsigned a0[TM];                                  // byte address of (neighbour pixel, k quarter lq) in plane 0, k half 0
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int rr = prow[i] + kh - 1, cc = pcol[i] + kw - 1;
            const bool ok = (unsigned)rr < (unsigned)HW && (unsigned)cc < (unsigned)HW;
            const int pp = ok ? pbase[i] + rr * HW + cc : ZP;
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
        for (int t = 0; t < 16 * TM; ++t) {
            const int u = t / TM, i = t - u * TM;
            if (i == 0 && tap * 16 + u + RING - 1 < 9 * 16) {                // the weight fragments RING - 1 k-steps ahead
#pragma unroll
                for (int j = 0; j < TN; ++j) wq[(u + RING - 1) % RING][j] = wnext[(size_t)j * 64];
                wnext += 32 * 64;
            }
            if (t + PF < 16 * TM) rd(t + PF);
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = Op16<OT>::mfma(wq[u % RING][j], fa[t % (PF + 1)], acc[i][j]);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (++kw == 3) { kw = 0; ++kh; }
    }

    // ---- y2 = relu(bn2(conv2)): acc[i][j][r] = pixel 16 i + l15, channel co + 16 j + 4 lq + r ----------------------------------------------
    const int co = half * 256 + wave * 64 + 4 * lq;
    uint16_t* const out = p.y + (size_t)pair * ROWS * C + co;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const float4 sc = *reinterpret_cast<const float4*>(p.scale + co + 16 * j), sh = *reinterpret_cast<const float4*>(p.shift + co + 16 * j);
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const float v0 = fmaxf(fmaf(acc[i][j][0], sc.x, sh.x), 0.f), v1 = fmaxf(fmaf(acc[i][j][1], sc.y, sh.y), 0.f);
            const float v2 = fmaxf(fmaf(acc[i][j][2], sc.z, sh.z), 0.f), v3 = fmaxf(fmaf(acc[i][j][3], sc.w, sh.w), 0.f);
            uint2 o;
            o.x = (uint32_t)Op16<OT>::from_f32(v0) | ((uint32_t)Op16<OT>::from_f32(v1) << 16);
            o.y = (uint32_t)Op16<OT>::from_f32(v2) | ((uint32_t)Op16<OT>::from_f32(v3) << 16);
            const int q = 16 * i + l15;
            if (q < nvalid) *reinterpret_cast<uint2*>(out + (size_t)q * C + 16 * j) = o;
        }
    }
}
}  // namespace

extern "C" int dh_conv3x3_s4_supported(int H, int W, int C) { return H == 7 && W == 7 && C == 512; }

// y [N,7,7,512] = relu(conv3x3(x [N,7,7,512], stride 1, pad 1) * scale + shift); w_packed = dh_pack_mfma_fragments(w [512][3*3*512]).
// Bit-identical to dh_conv2d_nhwc_bn_act(KS = 3, stride 1, pad 1, relu).
extern "C" int dh_conv3x3_s4_nhwc(const void* x, const void* w_packed, const float* scale, const float* shift, void* y, int N, int H,
                                  int W, int C, int dtype, void* stream) {
    if (!DH_IS_16BIT(dtype)) return DH_ERR_UNSUPPORTED;
    DH_REQUIRE(x && w_packed && scale && shift && y && N > 0 && dh_conv3x3_s4_supported(H, W, C));
    DH_REQUIRE(((uintptr_t)x % 16) == 0 && ((uintptr_t)w_packed % 16) == 0 && ((uintptr_t)y % 16) == 0 && ((uintptr_t)scale % 16) == 0 &&
               ((uintptr_t)shift % 16) == 0);
    S4Params p{};
    p.x = (const uint16_t*)x; p.wp = (const uint4*)w_packed; p.scale = scale; p.shift = shift; p.y = (uint16_t*)y; p.N = N;
    const double px = (double)N * H * W;
    dh_prof_set_tag("3x3");
    dh_prof_set_dims(N * H * W, C, 9 * C);
    DhProfScope prof("dh_conv2d_nhwc_bn_act", 2.0 * px * C * 9.0 * C, 2.0 * (px * C + 9.0 * C * C + px * C), stream);
    const int npairs = (N + 1) / 2, grid = dh_cdiv(npairs, 4) * 8;
    hipStream_t s = (hipStream_t)stream;
    DH_DISPATCH_16(dtype, hipLaunchKernelGGL((conv_s4_kernel<T>), dim3(grid), dim3(256), 0, s, p));
    DH_LAUNCH_CHECK();
}
