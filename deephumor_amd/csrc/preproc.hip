// Image preprocessing on device (the notebook's transform, deephumor_demo.ipynb:565-567):
//   transforms.Resize((224, 224))  ->  dh_resize_u8_hwc       Pillow's antialiased bilinear resample of 8-bit images, bit-exact
//   ToTensor + Normalize           ->  dh_normalize_u8_hwc    (conv.hip: fp32 NCHW, the parity path's input), or
//                                      dh_normalize_pack_u8   fused with the stem's input packing: u8 HWC -> normalised 16-bit
//                                                              NHWC8 that the matrix-core stem convolution reads directly
// Integer / byte kernels, HBM-bound, one thread per output element group; the filter coefficient tables (a few KB, double
// precision arithmetic as Pillow's precompute_coeffs) are built by the host (deephumor_amd/experiments/inference.py).
#include "common.h"
#include "prof.h"

#define DH_RESIZE_PRECISION_BITS 22          // Pillow: 32 - 8 - 2

// one resample pass along the axis with `n_out` outputs and stride `axis_stride` (in elements); `inner` contiguous elements per
// axis position (C for the horizontal pass, W * C for the vertical one).  out[o, i] = clip8((2^21 + sum_j k[o][j] * in[lo + j, i]) >> 22)
__global__ __launch_bounds__(256) void resize_pass_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst,
                                                           const int32_t* __restrict__ bounds, const int32_t* __restrict__ kk, int ksize,
                                                           int n_in, int n_out, int inner, size_t outer) {
    const size_t total = outer * n_out * inner;
    for (size_t idx = blockIdx.x * 256ull + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256ull) {
        const int i = (int)(idx % inner);
        const size_t r = idx / inner;
        const int o = (int)(r % n_out);
        const size_t b = r / n_out;
        const int lo = bounds[2 * o], n = bounds[2 * o + 1];
        const uint8_t* p = src + (b * n_in + lo) * inner + i;
        const int32_t* k = kk + (size_t)o * ksize;
        int acc = 1 << (DH_RESIZE_PRECISION_BITS - 1);
        for (int j = 0; j < n; ++j) acc += (int)p[(size_t)j * inner] * k[j];
        acc >>= DH_RESIZE_PRECISION_BITS;
        dst[idx] = (uint8_t)(acc < 0 ? 0 : (acc > 255 ? 255 : acc));
    }
}

extern "C" int dh_resize_u8_hwc(const uint8_t* src, uint8_t* tmp, uint8_t* dst, const int32_t* bounds_x, const int32_t* kx, int ksize_x,
                                const int32_t* bounds_y, const int32_t* ky, int ksize_y, int N, int Hin, int Win, int Hout, int Wout,
                                int C, void* stream) {
    DH_REQUIRE(src && dst && N > 0 && Hin > 0 && Win > 0 && Hout > 0 && Wout > 0 && C > 0 && C <= 8);
    DH_REQUIRE((Win == Wout || (bounds_x && kx && ksize_x > 0)) && (Hin == Hout || (bounds_y && ky && ksize_y > 0)));
    DH_REQUIRE((Win == Wout || Hin == Hout || tmp) && src != dst);
    DhProfScope prof("dh_resize_u8_hwc", 0.0, (double)N * C * ((double)Hin * Win + 2.0 * Hin * Wout + (double)Hout * Wout), stream);
    hipStream_t s = (hipStream_t)stream;
    auto grid = [](size_t total) { return (int)((total + 255) / 256 < 65536 ? (total + 255) / 256 : 65536); };
    const uint8_t* cur = src;
    int h_cur = Hin;
    // PIL/Image.py, Image.resize: `if self.size[1] > self.size[0] * 100 and size[1] < self.size[1]` -- an image more than 100 x taller
    // than wide whose height shrinks is resized VERTICALLY first (8-bit intermediate [N,Hout,Win,C]), everything else horizontally first
    if (Hin > 100 * Win && Hout < Hin && Win != Wout) {
        const size_t total = (size_t)N * Hout * Win * C;
        hipLaunchKernelGGL(resize_pass_kernel, dim3(grid(total)), dim3(256), 0, s, cur, tmp, bounds_y, ky, ksize_y, Hin, Hout, Win * C, (size_t)N);
        cur = tmp;
        h_cur = Hout;
    }
    if (Win != Wout) {            // horizontal pass, 8-bit intermediate (Pillow's order)
        uint8_t* out = h_cur != Hout ? tmp : dst;
        const size_t total = (size_t)N * h_cur * Wout * C;
        hipLaunchKernelGGL(resize_pass_kernel, dim3(grid(total)), dim3(256), 0, s, cur, out, bounds_x, kx, ksize_x, Win, Wout, C, (size_t)N * h_cur);
        cur = out;
    }
    if (h_cur != Hout) {
        const size_t total = (size_t)N * Hout * Wout * C;
        hipLaunchKernelGGL(resize_pass_kernel, dim3(grid(total)), dim3(256), 0, s, cur, dst, bounds_y, ky, ksize_y, Hin, Hout, Wout * C, (size_t)N);
    } else if (Win == Wout && Hin == Hout) {
        if (hipMemcpyAsync(dst, src, (size_t)N * Hin * Win * C, hipMemcpyDeviceToDevice, s) != hipSuccess) return DH_ERR_LAUNCH;
    }
    DH_LAUNCH_CHECK();
}

// u8 [N,H,W,C] -> 16-bit channels-last [N,H,W,8]: ((x / 255 - mean[c]) / std[c]) rounded once to the storage type, channels
// C..7 zero -- exactly dh_pack_nchw_to_nhwc8(dh_normalize_u8_hwc(x)) without the fp32 NCHW tensor in between.
template <typename T>
__global__ __launch_bounds__(256) void normalize_pack_u8_kernel(const uint8_t* __restrict__ x, const float* __restrict__ mean,
                                                                 const float* __restrict__ stdv, T* __restrict__ y, int C, size_t total) {
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < total; i += (size_t)gridDim.x * 256ull) {
        float v[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) v[c] = c < C ? ((float)x[i * C + c] / 255.0f - mean[c]) / stdv[c] : 0.f;
        store16(y + i * 8, v);
    }
}

extern "C" int dh_normalize_pack_u8(const uint8_t* x, const float* mean, const float* stdv, void* y, int N, int H, int W, int C,
                                    int dtype, void* stream) {
    DH_REQUIRE(x && mean && stdv && y && N > 0 && H > 0 && W > 0 && C > 0 && C <= 8 && ((uintptr_t)y % 16) == 0);
    DhProfScope prof("dh_normalize_pack_u8", 0.0, (double)N * H * W * (C + 16.0), stream);
    const size_t total = (size_t)N * H * W;
    const int grid = (int)((total + 255) / 256 < 32768 ? (total + 255) / 256 : 32768);
    DH_DISPATCH_16(dtype, hipLaunchKernelGGL(normalize_pack_u8_kernel<T>, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, mean, stdv,
                                             (T*)y, C, total));
    DH_LAUNCH_CHECK();
}

// fp32 -> 16-bit rows that keep a NON-ZERO value non-zero: round to nearest even, but a value that underflows to zero in the storage
// type becomes the smallest subnormal of its sign (fp16: 6e-8; bf16 shares fp32's exponent range, so this only matters for fp32
// subnormals).  For the encoder's spatial features on the fp16 path: the reference reads an encoder row with ANY exactly-zero element as
// padding (transformers.py:480), and a feature below 3e-8 would otherwise mask a real image patch (one image in ~1,000).
template <typename OT>
__global__ __launch_bounds__(256) void round16_keep_nonzero_kernel(const float* __restrict__ x, uint16_t* __restrict__ y, size_t n) {
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += (size_t)gridDim.x * 256ull) {
        const float f = x[i];
        uint16_t h = Op16<OT>::from_f32(f);
        if ((h & 0x7FFFu) == 0u && f != 0.f) h |= 1u;
        y[i] = h;
    }
}

extern "C" int dh_round16_keep_nonzero(const float* x, void* y, long long n, int dtype, void* stream) {
    if (!DH_IS_16BIT(dtype)) return DH_ERR_UNSUPPORTED;
    DH_REQUIRE(x && y && n > 0);
    DhProfScope prof("dh_round16_keep_nonzero", 0.0, 6.0 * (double)n, stream);
    const int grid = (int)((n + 255) / 256 < 65536 ? (n + 255) / 256 : 65536);
    DH_DISPATCH_16(dtype, hipLaunchKernelGGL((round16_keep_nonzero_kernel<T>), dim3(grid), dim3(256), 0, (hipStream_t)stream, x, (uint16_t*)y, (size_t)n));
    DH_LAUNCH_CHECK();
}
