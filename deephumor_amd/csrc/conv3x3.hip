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
    uint16_t* y;                                         // [N,H,W,Cout] (unfused form)
    int N, H, tiles_per_img;
    // fused bottleneck tail (FUSE): out = relu(bn3(conv3_1x1(relu(bn2(conv2_3x3(x))))) + res), conv3: Cout -> 4 Cout
    const uint16_t* w3; const float* scale3; const float* shift3;      // w3 [4 Cout][Cout]
    const uint16_t* res; uint16_t* out;                  // [N,H,W,4 Cout]
};

#define DH_C3_VMCNT(x) case x: asm volatile("s_waitcnt vmcnt(" #x ")" ::: "memory"); break;
__device__ __forceinline__ void c3_wait_vmcnt(int n) {
    switch (n) {
        DH_C3_VMCNT(1) DH_C3_VMCNT(2) DH_C3_VMCNT(3) DH_C3_VMCNT(4) DH_C3_VMCNT(5) DH_C3_VMCNT(6) DH_C3_VMCNT(7) DH_C3_VMCNT(8)
        DH_C3_VMCNT(9) DH_C3_VMCNT(10) DH_C3_VMCNT(11) DH_C3_VMCNT(12)
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
}

__device__ __forceinline__ void c3_wait_vmcnt_any(int n) {
    switch (n) {
        DH_C3_VMCNT(1) DH_C3_VMCNT(2) DH_C3_VMCNT(3) DH_C3_VMCNT(4) DH_C3_VMCNT(5) DH_C3_VMCNT(6) DH_C3_VMCNT(7) DH_C3_VMCNT(8) DH_C3_VMCNT(9) DH_C3_VMCNT(10) DH_C3_VMCNT(11) DH_C3_VMCNT(12) DH_C3_VMCNT(13) DH_C3_VMCNT(14) DH_C3_VMCNT(15) DH_C3_VMCNT(16) DH_C3_VMCNT(17) DH_C3_VMCNT(18) DH_C3_VMCNT(19) DH_C3_VMCNT(20) DH_C3_VMCNT(21) DH_C3_VMCNT(22) DH_C3_VMCNT(23) DH_C3_VMCNT(24) DH_C3_VMCNT(25) DH_C3_VMCNT(26) DH_C3_VMCNT(27) DH_C3_VMCNT(28) DH_C3_VMCNT(29) DH_C3_VMCNT(30) DH_C3_VMCNT(31) DH_C3_VMCNT(32) DH_C3_VMCNT(33) DH_C3_VMCNT(34) DH_C3_VMCNT(35) DH_C3_VMCNT(36) DH_C3_VMCNT(37) DH_C3_VMCNT(38) DH_C3_VMCNT(39) DH_C3_VMCNT(40) DH_C3_VMCNT(41) DH_C3_VMCNT(42) DH_C3_VMCNT(43) DH_C3_VMCNT(44) DH_C3_VMCNT(45) DH_C3_VMCNT(46) DH_C3_VMCNT(47) DH_C3_VMCNT(48) DH_C3_VMCNT(49) DH_C3_VMCNT(50) DH_C3_VMCNT(51) DH_C3_VMCNT(52) DH_C3_VMCNT(53) DH_C3_VMCNT(54) DH_C3_VMCNT(55) DH_C3_VMCNT(56) DH_C3_VMCNT(57) DH_C3_VMCNT(58) DH_C3_VMCNT(59) DH_C3_VMCNT(60) DH_C3_VMCNT(61) DH_C3_VMCNT(62) DH_C3_VMCNT(63)
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
}

// CB = Cin / 64; NT = Cout (64 or 128); waves WAVES_M x WAVES_N, each 7 x 2 MFMA tiles; the image width is 28 * WAVES_M
// FUSE: the 1x1 expansion (conv3 + bn3 + residual + ReLU) of the bottleneck runs in the same launch on the LDS-resident y2 tile
template <typename OT, int CB, int NT, int WAVES_M, int WAVES_N, int NS, bool FUSE = false>
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
    constexpr int NSLAB3 = FUSE ? 4 * CB : 0, NSTREAM = NSLAB + NSLAB3;   // conv3: 4 chunks of NT output channels x CB k-slabs
    static_assert(!FUSE || P * CIN * 2 + 4 * 2048 <= PATCH_BYTES, "y2 tile + the four per-wave fp32 strips live in the dead patch");
    __shared__ __attribute__((aligned(16))) unsigned char lds[PATCH_BYTES + NS * SLAB];
    unsigned char* const ring = lds + PATCH_BYTES;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, lq = lane >> 4, lr = lane >> 3, lpos = lane & 7;
    const int n = blockIdx.x / p.tiles_per_img, y0 = (blockIdx.x - n * p.tiles_per_img) * TR;
    const int wm = wave % WAVES_M, wn0 = (wave / WAVES_M) * (TN * 16);
    const unsigned char* zero = reinterpret_cast<const unsigned char*>(dh_c3_zero_page);

    // per-channel BatchNorm constants of conv2: requested first, so that no compiler-generated load sits between the LDS-DMA
    // transfers whose completion the loop counts with s_waitcnt vmcnt
    float4 sc[TN], sh[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        sc[j] = *reinterpret_cast<const float4*>(p.scale + wn0 + 16 * j + 4 * lq);
        sh[j] = *reinterpret_cast<const float4*>(p.shift + wn0 + 16 * j + 4 * lq);
    }
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
    // ---- weight slabs: stream index s < NSLAB: conv2's slab (tap s / CB, channel block s % CB) = k 64 s .. of [Cout][9 Cin];
    //      s >= NSLAB (FUSE): conv3's slab (chunk (s - NSLAB) / CB of NT output channels, k block (s - NSLAB) % CB) of [4 NT][CIN]
    const uint16_t* w_run[G];
#pragma unroll
    for (int i = 0; i < G; ++i) {
        const int row = (wave * G + i) * 8 + lr;
        w_run[i] = p.w + (size_t)row * (9 * CIN) + ((lpos ^ (row & 7)) << 3);
    }
    int st = 0;                                          // stream index of the next slab to stage
    auto stage_w = [&]() {
        unsigned char* slab = ring + __builtin_amdgcn_readfirstlane(st % NS) * SLAB;
        if (FUSE && st == NSLAB) {                        // switch to the 1x1 weights: chunk 0, k block 0
#pragma unroll
            for (int i = 0; i < G; ++i) {
                const int row = (wave * G + i) * 8 + lr;
                w_run[i] = p.w3 + (size_t)row * CIN + ((lpos ^ (row & 7)) << 3);
            }
        }
        // conv2: the next 64 k of the same rows; conv3: the next k block of the chunk, after its last one the next chunk's rows
        const int step = (FUSE && st >= NSLAB && (st - NSLAB) % CB == CB - 1) ? NT * CIN - (CB - 1) * 64 : 64;
#pragma unroll
        for (int i = 0; i < G; ++i) {
            dh_lds_dma16(w_run[i], slab + (wave * G + i) * 1024);
            w_run[i] += step;
        }
        ++st;
    };
#pragma unroll
    for (int u = 0; u < NS - 1; ++u) stage_w();

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

    // ---- conv2: nine taps x CB channel blocks; only LDS-DMA transfers are outstanding here, NS - 2 slabs younger than slab t ------
    int kh = 0, kw = 0, cbk = 0;                          // wave-uniform tap / channel-block counters
#pragma unroll 1
    for (int t = 0; t < NSLAB; ++t) {
        // slab t (and, at t = 0, the patch) has landed.  Steady state: an immediate count; the run-time switch (a tree of taken
        // scalar branches) only for the last slabs of the stream
        if (__builtin_expect(NSTREAM - 1 - t >= NS - 2, 1)) asm volatile("s_waitcnt vmcnt(%0)" :: "n"((NS - 2) * G) : "memory");
        else c3_wait_vmcnt((NSTREAM - 1 - t) * G);
        __builtin_amdgcn_s_barrier();
        if (t + NS - 1 < NSTREAM) stage_w();
        const int tapoff = kh * PITCH + kw;
        const unsigned char* sa = lds + cbk * PLANE;
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
        if (++cbk == CB) { cbk = 0; if (++kw == 3) { kw = 0; ++kh; } }
    }
    __syncthreads();                                      // every wave is done with the patch: it becomes the output staging tile

    if constexpr (FUSE) {
        // ---- y2 = relu(bn2(conv2)) as 16-bit, staged over the patch in the GEMM slab format [k block][pixel][128 B], swizzled ------
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
                const int nn = wn0 + 16 * j + 4 * lq, kb = nn >> 6, ch = (nn & 63) >> 3;
                *reinterpret_cast<uint2*>(lds + kb * (P * 128) + q * 128 + ((ch ^ (q & 7)) << 4) + (lq & 1) * 8) = o;
            }
        }
        // ---- conv3 (1x1, CIN -> 4 NT) in 4 chunks of NT output channels; epilogue wave-local: one MFMA tile pair (16 pixels x 32
        //      channels, fp32) through this wave's 2 KB strip, then lane = (pixel, 8 channels): residual add, ReLU, one rounding,
        //      16-byte stores (64 contiguous bytes per pixel; the neighbouring wave writes the other half of the line).
        //      vmcnt bookkeeping from here on counts every vector-memory instruction this wave issues (`issued`) and remembers the
        //      count right after each in-flight slab's transfer (`mark`): allowed outstanding at the wait for slab s = issued - mark(s).
        //      The chunk loop is unrolled, so all of it folds to constants.
        constexpr int C3OUT = 4 * NT;
        unsigned char* const strip = lds + P * CIN * 2 + wave * 2048;
        const int epx = lane >> 2, ec4 = lane & 3;            // epilogue lane role: pixel of the tile, 8-channel group
        const size_t pix0 = ((size_t)n * p.H + y0) * WD;      // first pixel of the tile in the [N*H*W] pixel index
        int issued = (NS - 1) * G;                            // the slabs NSLAB .. NSLAB + NS - 2 staged during conv2's last iterations
        int mark[NS];
#pragma unroll
        for (int u = 0; u < NS - 1; ++u) mark[(NSLAB + u) % NS] = (u + 1) * G;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            // this lane's residual chunks of the chunk's 7 MFMA tiles + its 8 channels' BatchNorm constants, requested before the MFMAs
            uint4 rq[TM];
            const int cbase = c * NT + wn0 + 8 * ec4;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int q = (wm * TM + i) * 16 + epx;
                rq[i] = *reinterpret_cast<const uint4*>(p.res + (pix0 + q) * C3OUT + cbase);
            }
            const float4 s3a = *reinterpret_cast<const float4*>(p.scale3 + cbase), s3b = *reinterpret_cast<const float4*>(p.scale3 + cbase + 4);
            const float4 h3a = *reinterpret_cast<const float4*>(p.shift3 + cbase), h3b = *reinterpret_cast<const float4*>(p.shift3 + cbase + 4);
            asm volatile("" ::: "memory");                    // the 11 loads above are issued here, not sunk below the waits
            issued += TM + 4;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = dh_f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int cb = 0; cb < CB; ++cb) {
                const int t = NSLAB + c * CB + cb;
                c3_wait_vmcnt_any(issued - mark[t % NS]);
                __syncthreads();                          // slab t complete for every wave (and the y2 staging writes visible)
                if (t + NS - 1 < NSTREAM) { stage_w(); issued += G; mark[(t + NS - 1) % NS] = issued; }
                const unsigned char* sa = lds + cb * (P * 128);
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
                        const int q = (wm * TM + i) * 16 + l15;
                        fa[i] = *reinterpret_cast<const uint4*>(sa + q * 128 + ((g ^ (q & 7)) << 4));
                    }
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j) acc[i][j] = Op16<OT>::mfma(fw[j], fa[i], acc[i][j]);
                }
            }
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
                const uint32_t w4[4] = {rq[i].x, rq[i].y, rq[i].z, rq[i].w};
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    float lo16, hi16;
                    Op16<OT>::unpack2(w4[u], lo16, hi16);
                    v[2 * u] = fmaxf(v[2 * u] + lo16, 0.f); v[2 * u + 1] = fmaxf(v[2 * u + 1] + hi16, 0.f);
                }
                const int q = (wm * TM + i) * 16 + epx;
                store16(reinterpret_cast<OT*>(p.out) + (pix0 + q) * C3OUT + cbase, v);
            }
            asm volatile("" ::: "memory");
            issued += TM;                                     // the chunk's TM store instructions
        }
        return;
    }

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

// The tail of a ResNet bottleneck in ONE launch (16-bit dtypes, channels-last): out = relu(bn3(conv3(relu(bn2(conv2(y1))))) +
// residual) with conv2 3x3 / stride 1 / pad 1 (C -> C) and conv3 1x1 (C -> 4 C) -- torchvision Bottleneck.forward from conv2 on
// (reference encoders.py:37-38,56).  conv2 runs as in dh_conv3x3_direct_nhwc; its output tile (4 rows x W pixels x C channels)
// never leaves LDS: it is the activation operand of the 1x1 expansion, whose [4C][C] weights stream through the same LDS ring.
// Saves the write and the read-back of the conv2 output and one launch per block.  Same shapes as dh_conv3x3_direct_nhwc
// (C = 64, W = 56 or C = 128, W = 28; H % 4 == 0).  w2 [C,3,3,C], w3 [4C,C], residual / out [N,H,W,4C].
extern "C" int dh_bottleneck_tail_nhwc(const void* y1, const void* w2, const float* scale2, const float* shift2, const void* w3,
                                       const float* scale3, const float* shift3, const void* residual, void* out, int N, int H,
                                       int W, int C, int dtype, void* stream) {
    if (!DH_IS_16BIT(dtype)) return DH_ERR_UNSUPPORTED;
    DH_REQUIRE(y1 && w2 && scale2 && shift2 && w3 && scale3 && shift3 && residual && out && N > 0 && dh_conv3x3_direct_supported(H, W, C, C));
    DH_REQUIRE(((uintptr_t)y1 % 16) == 0 && ((uintptr_t)w2 % 16) == 0 && ((uintptr_t)w3 % 16) == 0 && ((uintptr_t)residual % 16) == 0 &&
               ((uintptr_t)out % 16) == 0 && ((uintptr_t)scale2 % 16) == 0 && ((uintptr_t)shift2 % 16) == 0 &&
               ((uintptr_t)scale3 % 16) == 0 && ((uintptr_t)shift3 % 16) == 0 && (long long)N * (H / 4) < (1ll << 31));
    C3Params p{};
    p.x = (const uint16_t*)y1; p.w = (const uint16_t*)w2; p.scale = scale2; p.shift = shift2;
    p.w3 = (const uint16_t*)w3; p.scale3 = scale3; p.shift3 = shift3; p.res = (const uint16_t*)residual; p.out = (uint16_t*)out;
    p.N = N; p.H = H; p.tiles_per_img = H / 4;
    dh_prof_set_tag("3x3+1x1");
    dh_prof_set_dims(N * H * W, 4 * C, 9 * C + C / 4);
    const double px = (double)N * H * W;
    DhProfScope prof("dh_conv2d_nhwc_bn_act", 2.0 * px * C * 9.0 * C + 2.0 * px * 4.0 * C * C,
                     2.0 * (px * C + 9.0 * C * C + 4.0 * C * C + 2.0 * px * 4 * C), stream);
    const dim3 grid(N * p.tiles_per_img);
    hipStream_t s = (hipStream_t)stream;
    DH_DISPATCH_16(dtype, {
        if (C == 64) hipLaunchKernelGGL((conv3x3_direct_kernel<T, 1, 64, 2, 2, 4, true>), grid, dim3(256), 0, s, p);
        else hipLaunchKernelGGL((conv3x3_direct_kernel<T, 2, 128, 1, 4, 2, true>), grid, dim3(256), 0, s, p);
    });
    DH_LAUNCH_CHECK();
}
