// 3x3 / stride 1 / pad 1 convolution + BatchNorm + ReLU as a DIRECT convolution on the matrix cores, channels-last 16-bit
// (the conv2 of the ResNet-50 bottlenecks of stages 1 and 2: 56 x 56 x 64 and 28 x 28 x 128; torchvision Bottleneck.conv2 / bn2
// / relu behind reference encoders.py:37-38,56).
//
// As an implicit GEMM (gemm_bf16.hip) these layers re-stream the activation from L2 into LDS once per filter tap -- 9 x the
// tensor -- and run at the L2 -> LDS ingest rate (~45 GB/s per CU), 0.22-0.25 of the MFMA peak.  Here a workgroup owns FOUR
// output rows of one image and all output channels:
//   * the (4 + 2) x (W + 2) input pixels under them are brought into LDS ONCE (LDS-DMA, 128-byte pixel rows per 64-channel
//     block, XOR-swizzled 16-byte chunks, halo pixels from a zero page) and serve all nine taps: the MFMA's activation operand
//     is formed at ds_read time -- lane (pixel, k-quarter) reads the 16 bytes of 8 channels of patch pixel (r + kh, c + kw);
//   * only the weights stream: one [Cout][64 k] slab per (tap, 64-channel block) through an LDS ring, 8-16 KB per slab
//     against 40-48 KB per slab of the implicit GEMM: 3.5 x less ingest per output pixel;
//   * 4 waves, each 7 x 2 MFMA tiles (112 pixels x 32 channels); BatchNorm + ReLU on the accumulators, the 16-bit tile is staged
//     through the (dead) patch and leaves as one contiguous 28 KB block of the channels-last output.
// Two workgroups per CU (78 KB of LDS each): one computes while the other loads its patch or stores.
#include "common.h"
#include "prof.h"

__device__ uint4 dh_c3_zero_page[4];                     // zero-initialised: source of halo / padding chunks

namespace {
struct C3Params {
    const uint16_t* x; const uint16_t* w;                // x [N,H,W,Cin], w [Cout][3][3][Cin]
    const float* scale; const float* shift;
    uint16_t* y;                                         // [N,H,W,Cout]
    int N, H, tiles_per_img;
};

#define DH_C3_VMCNT(x) case x: asm volatile("s_waitcnt vmcnt(" #x ")" ::: "memory"); break;
__device__ __forceinline__ void c3_wait_vmcnt(int n) {
    switch (n) {
        DH_C3_VMCNT(1) DH_C3_VMCNT(2) DH_C3_VMCNT(3) DH_C3_VMCNT(4) DH_C3_VMCNT(5) DH_C3_VMCNT(6) DH_C3_VMCNT(7) DH_C3_VMCNT(8)
        DH_C3_VMCNT(9) DH_C3_VMCNT(10) DH_C3_VMCNT(11) DH_C3_VMCNT(12)
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
}

// CB = Cin / 64; NT = Cout (64 or 128); waves WAVES_M x WAVES_N, each 7 x 2 MFMA tiles; the image width is 28 * WAVES_M
template <typename OT, int CB, int NT, int WAVES_M, int WAVES_N, int NS>
__global__ __launch_bounds__(256, 2) void conv3x3_direct_kernel(C3Params p) {
    constexpr int CIN = 64 * CB, TM = 7, TN = 2, TR = 4;
    constexpr int P = 16 * TM * WAVES_M;                 // output pixels per workgroup = TR full rows
    constexpr int WD = P / TR, PITCH = WD + 2, NPP = (TR + 2) * PITCH;
    constexpr int PP_ROWS = (NPP + 7) / 8 * 8, PLANE = PP_ROWS * 128, PATCH_BYTES = CB * PLANE;
    constexpr int NPIECE = CB * PP_ROWS / 8;
    constexpr int SLAB = NT * 128, NSLAB = 9 * CB, G = NT / 32;       // weight pieces per wave per slab
    static_assert(WAVES_M * WAVES_N == 4 && WAVES_N * TN * 16 == NT, "wave layout");
    static_assert(P * NT * 2 <= PATCH_BYTES, "the output tile is staged through the patch");
    static_assert((NS - 2) * G <= 12, "vmcnt cases");
    __shared__ __attribute__((aligned(16))) unsigned char lds[PATCH_BYTES + NS * SLAB];
    unsigned char* const ring = lds + PATCH_BYTES;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, lq = lane >> 4, lr = lane >> 3, lpos = lane & 7;
    const int n = blockIdx.x / p.tiles_per_img, y0 = (blockIdx.x - n * p.tiles_per_img) * TR;
    const int wm = wave % WAVES_M, wn0 = (wave / WAVES_M) * (TN * 16);
    const unsigned char* zero = reinterpret_cast<const unsigned char*>(dh_c3_zero_page);

    // ---- the input patch: rows y0 - 1 .. y0 + TR, columns -1 .. WD, all channels; piece = 8 patch pixels x 128 bytes ----------
    {
        const uint16_t* img = p.x + (size_t)n * p.H * WD * CIN;
        for (int pc = wave; pc < NPIECE; pc += 4) {
            const int cb = pc / (PP_ROWS / 8), pp = (pc - cb * (PP_ROWS / 8)) * 8 + lr;
            const int pr = pp / PITCH, pcx = pp - pr * PITCH;
            const int gy = y0 - 1 + pr, gx = pcx - 1;
            const bool ok = pp < NPP && (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)WD;
            const void* src = ok ? (const void*)(img + ((size_t)gy * WD + gx) * CIN + cb * 64 + ((lpos ^ (pp & 7)) << 3)) : (const void*)zero;
            dh_lds_dma16(src, lds + pc * 1024);
        }
    }
    // ---- weight slabs: slab t = (tap t / CB, channel block t % CB) = k 64 t .. 64 t + 63 of the [Cout][9 Cin] matrix -------------
    const uint16_t* w_run[G];
#pragma unroll
    for (int i = 0; i < G; ++i) {
        const int row = (wave * G + i) * 8 + lr;
        w_run[i] = p.w + (size_t)row * (9 * CIN) + ((lpos ^ (row & 7)) << 3);
    }
    auto stage_w = [&](int buf) {
        unsigned char* slab = ring + __builtin_amdgcn_readfirstlane(buf) * SLAB;
#pragma unroll
        for (int i = 0; i < G; ++i) {
            dh_lds_dma16(w_run[i], slab + (wave * G + i) * 1024);
            w_run[i] += 64;
        }
    };
#pragma unroll
    for (int u = 0; u < NS - 1; ++u) stage_w(u);

    float4 sc[TN], sh[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        sc[j] = *reinterpret_cast<const float4*>(p.scale + wn0 + 16 * j + 4 * lq);
        sh[j] = *reinterpret_cast<const float4*>(p.shift + wn0 + 16 * j + 4 * lq);
    }
    int pp0[TM];                                          // patch pixel of tap (0, 0) for this lane's pixel of m-tile i
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int q = (wm * TM + i) * 16 + l15, r = q / WD, c = q - r * WD;
        pp0[i] = r * PITCH + c;
    }
    dh_f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = dh_f32x4{0.f, 0.f, 0.f, 0.f};

#pragma unroll
    for (int t = 0; t < NSLAB; ++t) {
        c3_wait_vmcnt((NSLAB - 1 - t < NS - 2 ? NSLAB - 1 - t : NS - 2) * G);   // slab t (and, at t = 0, the patch) has landed
        __builtin_amdgcn_s_barrier();
        if (t + NS - 1 < NSLAB) stage_w((t + NS - 1) % NS);
        const int tap = t / CB, cb = t % CB, tapoff = (tap / 3) * PITCH + tap % 3;
        const unsigned char* sa = lds + cb * PLANE;
        const unsigned char* sb = ring + (t % NS) * SLAB;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int g = kk * 4 + lq;
            uint4 fw[TN], fa[TM];
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int row = wn0 + 16 * j + l15;
                fw[j] = *reinterpret_cast<const uint4*>(sb + row * 128 + ((g ^ (row & 7)) << 4));
            }
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int pp = pp0[i] + tapoff;
                fa[i] = *reinterpret_cast<const uint4*>(sa + pp * 128 + ((g ^ (pp & 7)) << 4));
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = Op16<OT>::mfma(fw[j], fa[i], acc[i][j]);
        }
    }
    __syncthreads();                                      // every wave is done with the patch: it becomes the output staging tile

    // ---- BatchNorm + ReLU, 16-bit, staged as [pixel][NT] rows (16-byte chunks XOR-swizzled by the pixel index) ------------------
    constexpr int ROWB = NT * 2, CHUNKS = NT / 8;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int q = (wm * TM + i) * 16 + l15;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const float v0 = fmaxf(fmaf(acc[i][j][0], sc[j].x, sh[j].x), 0.f), v1 = fmaxf(fmaf(acc[i][j][1], sc[j].y, sh[j].y), 0.f);
            const float v2 = fmaxf(fmaf(acc[i][j][2], sc[j].z, sh[j].z), 0.f), v3 = fmaxf(fmaf(acc[i][j][3], sc[j].w, sh[j].w), 0.f);
            uint2 o;
            o.x = (uint32_t)Op16<OT>::from_f32(v0) | ((uint32_t)Op16<OT>::from_f32(v1) << 16);
            o.y = (uint32_t)Op16<OT>::from_f32(v2) | ((uint32_t)Op16<OT>::from_f32(v3) << 16);
            const int ch = (wn0 + 16 * j + 4 * lq) >> 3;
            *reinterpret_cast<uint2*>(lds + q * ROWB + ((ch ^ (q & 7)) << 4) + (lq & 1) * 8) = o;
        }
    }
    __syncthreads();
    // the TR output rows of the tile are one contiguous block of the channels-last tensor
    uint16_t* out = p.y + ((size_t)n * p.H + y0) * WD * NT;
#pragma unroll
    for (int it = 0; it < P * CHUNKS / 256; ++it) {
        const int idx = tid + 256 * it, q = idx / CHUNKS, ch = idx - q * CHUNKS;
        *reinterpret_cast<uint4*>(out + (size_t)idx * 8) = *reinterpret_cast<const uint4*>(lds + q * ROWB + ((ch ^ (q & 7)) << 4));
    }
}
}  // namespace

// nonzero when dh_conv3x3_direct_nhwc supports the shape (the caller falls back to dh_conv2d_nhwc_bn_act otherwise)
extern "C" int dh_conv3x3_direct_supported(int H, int W, int Cin, int Cout) {
    return H > 0 && (H % 4) == 0 && ((Cin == 64 && Cout == 64 && W == 56) || (Cin == 128 && Cout == 128 && W == 28));
}

extern "C" int dh_conv3x3_direct_nhwc(const void* x, const void* w, const float* scale, const float* shift, void* y, int N, int H,
                                      int W, int Cin, int Cout, int relu, int dtype, void* stream) {
    if (!DH_IS_16BIT(dtype)) return DH_ERR_UNSUPPORTED;
    DH_REQUIRE(x && w && scale && shift && y && N > 0 && relu == 1 && dh_conv3x3_direct_supported(H, W, Cin, Cout));
    DH_REQUIRE(((uintptr_t)x % 16) == 0 && ((uintptr_t)w % 16) == 0 && ((uintptr_t)y % 16) == 0 && ((uintptr_t)scale % 16) == 0 &&
               ((uintptr_t)shift % 16) == 0 && (long long)N * (H / 4) < (1ll << 31));
    C3Params p{};
    p.x = (const uint16_t*)x; p.w = (const uint16_t*)w; p.scale = scale; p.shift = shift; p.y = (uint16_t*)y;
    p.N = N; p.H = H; p.tiles_per_img = H / 4;
    dh_prof_set_tag("3x3");
    dh_prof_set_dims(N * H * W, Cout, 9 * Cin);
    DhProfScope prof("dh_conv2d_nhwc_bn_act", 2.0 * N * H * W * Cout * 9.0 * Cin,
                     2.0 * ((double)N * H * W * Cin + (double)Cout * 9 * Cin + (double)N * H * W * Cout), stream);
    const dim3 grid(N * p.tiles_per_img);
    hipStream_t s = (hipStream_t)stream;
    DH_DISPATCH_16(dtype, {
        if (Cin == 64) hipLaunchKernelGGL((conv3x3_direct_kernel<T, 1, 64, 2, 2, 4>), grid, dim3(256), 0, s, p);
        else hipLaunchKernelGGL((conv3x3_direct_kernel<T, 2, 128, 1, 4, 2>), grid, dim3(256), 0, s, p);
    });
    DH_LAUNCH_CHECK();
}
