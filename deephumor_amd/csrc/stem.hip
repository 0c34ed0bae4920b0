// The ResNet stem as a DIRECT convolution on the matrix cores: conv 7x7 / stride 2 / pad 3 (3 -> 64 channels) + BatchNorm +
// ReLU + MaxPool2d(3, 2, 1) in one launch (torchvision resnet children conv1, bn1, relu, maxpool; reference encoders.py:37-38).
//
// Why not the implicit GEMM of gemm_bf16.hip (dh_conv2d_nhwc_bn_relu_maxpool, kept as the general form): as an im2col GEMM the
// stem moves 49 taps x 8 padded channels = 784 B per output pixel from L2 into LDS (2.5 GB per 256 images for 205 MB of unique
// input) and multiplies a K of 392 of which 147 are real -- 494 us per 256 images, 0.13 of the MFMA peak.  Here a workgroup
// loads the 35 x 36 input pixels under a 15 x 15 patch of convolution outputs ONCE into LDS (4 channels per pixel = 8 bytes:
// R, G, B, 0) and forms the MFMA operands at ds_read time:
//   * one v_mfma_f32_16x16x32 = one kernel ROW kh: k = (kw slot 0..7) x (4 channels); lane (pixel cx, quarter lq) reads the 16
//     bytes of input pixels x = 2 cx + 2 lq, 2 cx + 2 lq + 1 of input row 2 cy + kh -- contiguous and 16-byte aligned because the
//     stride is 2 -- so K is 7 x 32 = 224 (kw slot 7 and channel 3 carry zero weights) instead of 392;
//   * the 16 rows of an MFMA are the 15 pixels of one convolution row (lane 15 repeats pixel 14, never stored);
//   * the 28 weight fragments (7 kh x 4 tiles of 16 output channels) sit in LDS in fragment order (lane-linear reads);
//   * BatchNorm on the accumulators, the 15 x 15 x 64 tile is staged in LDS in the output type (rounding is monotonic, so it
//     commutes with the maximum), ReLU + pooling to 7 x 7 as ONE packed signed 16-bit maximum with +0 over the nine taps,
//     stored channels-last: the un-pooled activation never exists.
// The workgroup is persistent over patches (grid = 2 per CU): the global loads of the next patch are in flight during the
// MFMAs of the current one, and the pooling of a patch overlaps the next patch's MFMAs in the other waves / the co-resident
// workgroup.  The caller's image is read as it is: fp32 NCHW (x_fmt 0, no packing launch in front) or the 16-bit channels-last
// [N,H,W,8] tensor dh_normalize_pack_u8 / dh_pack_nchw_to_nhwc8 produce (x_fmt 1).
#include "common.h"
#include "prof.h"

namespace {
constexpr int ST_PW = 36, ST_PH = 35;                    // input patch: 35 rows x 36 pixels x 8 bytes
constexpr int ST_NPIX = ST_PW * ST_PH;                   // 1260
constexpr int ST_PATCH_BYTES = ST_NPIX * 8;              // 10,080
constexpr int ST_E_BYTES = 225 * 128;                    // 15 x 15 convolution pixels x 64 channels, 16-bit
constexpr int ST_W_BYTES = 28 * 64 * 16;                 // 7 kh x 4 channel tiles x 64 lanes x 16 bytes
constexpr int ST_LOAD_IT = (ST_NPIX + 255) / 256;        // 5

struct StemParams {
    const void* x;
    const uint16_t* w;                                   // [64][7][8][4]
    const float* scale; const float* shift;
    uint16_t* y;                                         // [N, Hp, Wp, 64]
    int N, H, W, Ho, Wo, Hp, Wp, pbh, pbw, npatch;
};

template <typename OT, int FMT>
__global__ __launch_bounds__(256, 2) void stem_direct_kernel(StemParams p) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * ST_PATCH_BYTES + ST_E_BYTES + ST_W_BYTES];
    unsigned char* const E = lds + 2 * ST_PATCH_BYTES;
    unsigned char* const Wl = E + ST_E_BYTES;
    int pidx = blockIdx.x;
    if (pidx >= p.npatch) return;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, lq = lane >> 4;

    // weights -> LDS in fragment order: fragment (kh, j), lane (l15, lq) = output channel 16 j + l15, kw slots 2 lq, 2 lq + 1
    for (int f = tid; f < 28 * 64; f += 256) {
        const int ln = f & 63, fr = f >> 6, kh = fr >> 2, j = fr & 3;
        const int c = 16 * j + (ln & 15), q = ln >> 4;
        *reinterpret_cast<uint4*>(Wl + f * 16) = *reinterpret_cast<const uint4*>(p.w + c * 224 + kh * 32 + q * 8);
    }
    float4 sc[4], sh[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        sc[j] = *reinterpret_cast<const float4*>(p.scale + 16 * j + 4 * lq);
        sh[j] = *reinterpret_cast<const float4*>(p.shift + 16 * j + 4 * lq);
    }

    const int per = p.pbh * p.pbw;
    // ---- patch loader: element e = row * 36 + col of the 35 x 36 patch, 5 elements per thread -------------------------------
    float rf[ST_LOAD_IT][3];
    uint2 rh[ST_LOAD_IT];
    // (row, col) of this thread's elements are loop-invariant; per patch only the bounds tests and one 32-bit offset remain
    int e_row[ST_LOAD_IT], e_col[ST_LOAD_IT];
#pragma unroll
    for (int i = 0; i < ST_LOAD_IT; ++i) {
        const int e = tid + 256 * i;
        e_row[i] = e / ST_PW; e_col[i] = e - e_row[i] * ST_PW;
    }
    const unsigned plane = (unsigned)p.H * (unsigned)p.W;
    auto load_patch = [&](int pi) {
        const int n = pi / per, blk = pi - n * per, by = blk / p.pbw, bx = blk - by * p.pbw;
        const int iy0 = 28 * by - 5, ix0 = 28 * bx - 5;
        const float* b0 = reinterpret_cast<const float*>(p.x) + (size_t)n * 3 * plane;            // wave-uniform bases
        const float* b1 = b0 + plane;
        const float* b2 = b1 + plane;
        const uint16_t* h0 = reinterpret_cast<const uint16_t*>(p.x) + (size_t)n * plane * 8;
#pragma unroll
        for (int i = 0; i < ST_LOAD_IT; ++i) {
            const int gy = iy0 + e_row[i], gx = ix0 + e_col[i];
            const bool ok = (i < ST_LOAD_IT - 1 || tid + 256 * i < ST_NPIX) && (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W;
            const unsigned off = ok ? (unsigned)gy * (unsigned)p.W + (unsigned)gx : 0u;
            if constexpr (FMT == 0) {
                const float v0 = b0[off], v1 = b1[off], v2 = b2[off];
                rf[i][0] = ok ? v0 : 0.f; rf[i][1] = ok ? v1 : 0.f; rf[i][2] = ok ? v2 : 0.f;
            } else {
                const uint2 v = *reinterpret_cast<const uint2*>(h0 + (size_t)off * 8);
                rh[i].x = ok ? v.x : 0u; rh[i].y = ok ? (v.y & 0xFFFFu) : 0u;          // channel 3 is a zero-weight slot: keep it finite
            }
        }
    };
    auto write_patch = [&](unsigned char* P) {
#pragma unroll
        for (int i = 0; i < ST_LOAD_IT; ++i) {
            const int e = tid + 256 * i;
            if (e < ST_NPIX) {
                uint2 v;
                if constexpr (FMT == 0) {
                    v.x = (uint32_t)Op16<OT>::from_f32(rf[i][0]) | ((uint32_t)Op16<OT>::from_f32(rf[i][1]) << 16);
                    v.y = (uint32_t)Op16<OT>::from_f32(rf[i][2]);
                } else {
                    v = rh[i];
                }
                *reinterpret_cast<uint2*>(P + e * 8) = v;
            }
        }
    };

    load_patch(pidx);
    write_patch(lds);
    __syncthreads();

    const int cx = l15 < 15 ? l15 : 14;
    const int a_off = (2 * cx + 2 * lq) * 8;
    const int nrows = wave == 3 ? 3 : 4;                                  // convolution rows 4 wave .. 4 wave + nrows - 1 of the 15
    for (int it = 0;; ++it) {
        const unsigned char* P = lds + (it & 1) * ST_PATCH_BYTES;
        const int next = pidx + (int)gridDim.x;
        const bool has_next = next < p.npatch;
        if (has_next) load_patch(next);

        dh_f32x4 acc[4][4];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[r][j] = dh_f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kh = 0; kh < 7; ++kh) {
            uint4 wf[4], af[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) wf[j] = *reinterpret_cast<const uint4*>(Wl + ((kh * 4 + j) * 64 + lane) * 16);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 2 * (4 * wave + (r < nrows ? r : 0)) + kh;
                af[r] = *reinterpret_cast<const uint4*>(P + row * (ST_PW * 8) + a_off);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[r][j] = Op16<OT>::mfma(wf[j], af[r], acc[r][j]);
        }

        if (it > 0) __syncthreads();                      // the previous patch's pooling has finished reading E
        // BatchNorm + ReLU, staged as 16-bit: acc[r][j][u] = pixel (4 wave + r, l15), channel 16 j + 4 lq + u
        if (l15 < 15) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (r < nrows) {
                    const int pix = (4 * wave + r) * 15 + l15;
                    unsigned char* row = E + pix * 128 + (lq & 1) * 8;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        // no ReLU here: the pooling below takes the maximum with +0 (signed 16-bit compare), which is the ReLU
                        uint2 o;
                        o.x = (uint32_t)Op16<OT>::from_f32(fmaf(acc[r][j][0], sc[j].x, sh[j].x)) | ((uint32_t)Op16<OT>::from_f32(fmaf(acc[r][j][1], sc[j].y, sh[j].y)) << 16);
                        o.y = (uint32_t)Op16<OT>::from_f32(fmaf(acc[r][j][2], sc[j].z, sh[j].z)) | ((uint32_t)Op16<OT>::from_f32(fmaf(acc[r][j][3], sc[j].w, sh[j].w)) << 16);
                        *reinterpret_cast<uint2*>(row + (((2 * j + (lq >> 1)) ^ (pix & 7)) << 4)) = o;
                    }
                }
            }
        }
        if (has_next) write_patch(lds + ((it + 1) & 1) * ST_PATCH_BYTES);
        __syncthreads();                                  // E complete, the next patch complete

        // ReLU + MaxPool2d(3, 2, 1): pooled pixel (py, px) of the 7 x 7 block = max(0, patch pixels (2 py + {0,1,2}, 2 px + {0,1,2})).
        // Only the first row / column of a window can lie outside the convolution output (Ho, Wo even): it is replaced by its
        // neighbour, so the nine reads are unconditional and back to back.  The maximum is taken on the 16-bit patterns as signed
        // integers: non-negative bf16 / fp16 values order like their bit patterns and every negative value (and -0) is below +0.
        {
            typedef short dh_s16x2 __attribute__((ext_vector_type(2)));
            const int n = pidx / per, blk = pidx - n * per, by = blk / p.pbw, bx = blk - by * p.pbw;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int e = tid + 256 * i;
                const int pp = e >> 3, ch = e & 7, py = pp / 7, px = pp - py * 7;
                const int gy = 7 * by + py, gx = 7 * bx + px;
                if (e < 392 && gy < p.Hp && gx < p.Wp) {
                    uint4 q[9];
#pragma unroll
                    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                        for (int dx = 0; dx < 3; ++dx) {
                            const int cy = 2 * py + dy + ((dy == 0 && gy == 0) ? 1 : 0), cxx = 2 * px + dx + ((dx == 0 && gx == 0) ? 1 : 0);
                            const int pix = cy * 15 + cxx;
                            q[dy * 3 + dx] = *reinterpret_cast<const uint4*>(E + pix * 128 + ((ch ^ (pix & 7)) << 4));
                        }
                    uint32_t m[4] = {0u, 0u, 0u, 0u};
#pragma unroll
                    for (int t = 0; t < 9; ++t) {
                        const uint32_t w4[4] = {q[t].x, q[t].y, q[t].z, q[t].w};
#pragma unroll
                        for (int u = 0; u < 4; ++u)
                            m[u] = __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(dh_s16x2, m[u]), __builtin_bit_cast(dh_s16x2, w4[u])));
                    }
                    *reinterpret_cast<uint4*>(p.y + (((size_t)n * p.Hp + gy) * p.Wp + gx) * 64 + ch * 8) = make_uint4(m[0], m[1], m[2], m[3]);
                }
            }
        }
        if (!has_next) break;
        pidx = next;
    }
}
}  // namespace

extern "C" int dh_stem_conv7_bn_relu_maxpool(const void* x, int x_fmt, const void* w, const float* scale, const float* shift, void* y,
                                             int N, int H, int W, int dtype, void* stream) {
    if (!DH_IS_16BIT(dtype)) return DH_ERR_UNSUPPORTED;
    DH_REQUIRE(x && w && scale && shift && y && N > 0 && H >= 2 && W >= 2 && (x_fmt == 0 || x_fmt == 1));
    DH_REQUIRE(((uintptr_t)x % (x_fmt ? 16 : 4)) == 0 && ((uintptr_t)w % 16) == 0 && ((uintptr_t)y % 16) == 0 &&
               ((uintptr_t)scale % 16) == 0 && ((uintptr_t)shift % 16) == 0);
    StemParams p{};
    p.x = x; p.w = (const uint16_t*)w; p.scale = scale; p.shift = shift; p.y = (uint16_t*)y;
    p.N = N; p.H = H; p.W = W;
    p.Ho = (H + 6 - 7) / 2 + 1; p.Wo = (W + 6 - 7) / 2 + 1;
    DH_REQUIRE(p.Ho >= 2 && p.Wo >= 2 && (p.Ho % 2) == 0 && (p.Wo % 2) == 0);
    p.Hp = p.Ho / 2; p.Wp = p.Wo / 2; p.pbh = (p.Hp + 6) / 7; p.pbw = (p.Wp + 6) / 7;
    const long long npatch = (long long)N * p.pbh * p.pbw;
    DH_REQUIRE(npatch < (1ll << 31) && (long long)N * 3 * H * W < (1ll << 40));
    p.npatch = (int)npatch;
    dh_prof_set_tag("stem+pool");
    dh_prof_set_dims(N * p.Ho * p.Wo, 64, 147);
    DhProfScope prof("dh_conv2d_nhwc_bn_act", 2.0 * N * p.Ho * p.Wo * 64 * 147,
                     (double)N * H * W * (x_fmt ? 16.0 : 12.0) + 2.0 * 64 * 224 + 2.0 * N * p.Hp * p.Wp * 64, stream);
    const int grid = p.npatch < 512 ? p.npatch : 512;
    hipStream_t s = (hipStream_t)stream;
    DH_DISPATCH_16(dtype, {
        if (x_fmt == 0) hipLaunchKernelGGL((stem_direct_kernel<T, 0>), dim3(grid), dim3(256), 0, s, p);
        else hipLaunchKernelGGL((stem_direct_kernel<T, 1>), dim3(grid), dim3(256), 0, s, p);
    });
    DH_LAUNCH_CHECK();
}
