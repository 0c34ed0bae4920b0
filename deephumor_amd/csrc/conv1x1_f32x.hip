// The wide 1 x 1 convolutions of the fp32 trunk on the split-operand path (option "f32_split") as a STREAMING kernel: round 6.
// (torchvision Bottleneck.conv3 / bn3 + identity + relu and layer1's downsample: encoders.py:56 -- Cin = 64 ... 256, Cout = 4 Cin.)
//
// dh_conv2d_nhwc_f32x runs these layers as 128 x 128 tiles: a workgroup lives for two to eight 32-k slabs, then reads its residual
// tile and stores 64 KB.  Measured on 256 x 56 x 56 x 64 -> 256 with residual (645 us; tools/f32x_conv1x1_bench.py, profiles/r6/f32x_conv1x1_phases.txt):
// without the stores 437 us, without the residual loads 347 us, without either 226 us -- the three phases run one after the other
// (two workgroups per CU, each waiting on its own memory round trips): 2.9 TB/s of algorithmic traffic on an 8 TB/s part.  Here:
//   * linear_f32x_wreg.hip's partition: a workgroup owns 64 output columns (4 waves x 16), a wave keeps the hi AND lo fp16 planes of
//     its 16 weight rows x K as MFMA fragments in registers (K <= 256: 64 VGPRs) -- loaded ONCE;
//   * it is PERSISTENT over 32-row blocks of the activation: block i + 1 comes into the other LDS buffer by LDS-DMA (fp32, no staging
//     registers) and the residual quads of block i into registers while block i is split in place, multiplied and stored -- the loads,
//     the MFMAs and the stores of a CU's two to eight workgroups overlap instead of taking turns;
//   * the Cout / 64 column groups of a row block run on the SAME XCD at the same time (blockIdx -> (XCD, column group, row stream)), so
//     the activation block comes from HBM once and from that XCD's L2 for the other groups.
// Arithmetic: per (k step, row tile) acc += w_hi a_hi; cor += w_hi a_lo; cor += w_lo a_hi in ascending k, then
// ((acc + 2^-11 cor) + 0) * scale + shift (+ residual) (ReLU) -- the tile kernel's sequence: BIT-IDENTICAL to dh_conv2d_nhwc_f32x.
#include "common.h"
#include "prof.h"

unsigned* dh_f32x_range_flag_of(hipStream_t s);          // gemm_f32x.hip

namespace {
constexpr float kLoScale = 2048.0f, kLoInv = 1.0f / 2048.0f, kF16Max = 65504.0f;

struct CsParams {
    const float* A;                                      // [M, K] fp32 (channels-last pixels)
    const uint4* wh; const uint4* wl;                    // fragment-packed planes [K / 32][N / 16][64] x 16 bytes
    const float* scale; const float* shift; const float* res;
    float* C;                                            // [M, N]
    int M, N, K, relu;
    int ncg, nspx, nblocks;                              // column groups, row streams per XCD, row blocks
    unsigned* range_flag;
};

__device__ __forceinline__ void split2(float a, float b, uint32_t& hi, uint32_t& lo) {
    const f16_t ha = (f16_t)a, hb = (f16_t)b;
    const f16_t la = (f16_t)((a - (float)ha) * kLoScale), lb = (f16_t)((b - (float)hb) * kLoScale);
    hi = (uint32_t)__builtin_bit_cast(uint16_t, ha) | ((uint32_t)__builtin_bit_cast(uint16_t, hb) << 16);
    lo = (uint32_t)__builtin_bit_cast(uint16_t, la) | ((uint32_t)__builtin_bit_cast(uint16_t, lb) << 16);
}

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }

// 4 waves, RL = 32 rows per block, K = 32 NS.  LDS: two buffers of NS slabs; slab s = the 32 k values 32 s .. + 31 of the 32 rows, 128
// bytes per row (fp32), 16-byte slots XOR-swizzled by the row; after the in-place split slot (2q) ^ m holds the hi plane of k piece q
// (8 values), slot (2q + 1) ^ m the lo plane (linear_f32x_wreg.hip's layout).
template <int NS>
__global__ __launch_bounds__(256) void conv1x1_f32x_stream_kernel(CsParams p) {
    constexpr int NW = 4, RL = 32, TM = 2, RG = RL / 8, SLABB = RL * 128, BUF = NS * SLABB;
    constexpr int PPW = NS * RG / NW;                    // 1 KB DMA pieces (8 rows of one slab) per wave and block
    constexpr int PIECES = RL * NS * 4, C_IT = PIECES / 256;      // 32-byte pieces split in place per block
    static_assert(NS * RG % NW == 0 && PIECES % 256 == 0, "whole pieces per wave / thread");
    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * BUF];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, lq = lane >> 4, lr = lane >> 3, lpos = lane & 7;
    // blockIdx -> XCD x (= blockIdx % 8: consecutive ids go round the XCDs), column group, row stream: the column groups of one row stream
    // sit on one XCD and walk the same row blocks
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int cg = j % p.ncg, stream = xcd * p.nspx + j / p.ncg, nstreams = 8 * p.nspx;
    const int n0 = cg * 64;

    // this wave's 16 weight rows x K, both planes, once
    uint4 wfh[NS], wfl[NS];
    {
        const size_t fstep = (size_t)(p.N / 16) * 64, base = (size_t)(n0 / 16 + wave) * 64 + lane;
#pragma unroll
        for (int f = 0; f < NS; ++f) { wfh[f] = p.wh[base + (size_t)f * fstep]; wfl[f] = p.wl[base + (size_t)f * fstep]; }
    }
    const int n = n0 + 16 * wave + 4 * lq;
    const float4 sc4 = *reinterpret_cast<const float4*>(p.scale + n), sh4 = *reinterpret_cast<const float4*>(p.shift + n);
    const unsigned rd_base = (unsigned)(l15 * 128 + (((2 * lq) ^ (l15 & 7)) << 4));
    const unsigned swz = (unsigned)((lpos ^ lr) << 4);

    auto issue = [&](int buf, int rb) {                  // the fp32 block rb -> LDS buffer buf; wave w moves pieces w PPW ...
        const int m0 = rb * RL;
#pragma unroll
        for (int q = 0; q < PPW; ++q) {
            const int pc = wave * PPW + q, s = pc / RG, g = pc - s * RG;
            const float* src = p.A + (size_t)min(m0 + g * 8 + lr, p.M - 1) * p.K + 32 * s;
            dh_lds_dma16(reinterpret_cast<const unsigned char*>(src) + swz, lds + buf * BUF + s * SLABB + g * 1024);
        }
    };

    float amax = 0.f;
    int rb = stream;
    if (rb < p.nblocks) issue(0, rb);
    for (int it = 0; rb < p.nblocks; ++it, rb += nstreams) {
        const int buf = it & 1, m0 = rb * RL;
        const bool more = rb + nstreams < p.nblocks;
        // (raw barriers + explicit LDS waits: a __syncthreads() next to the residual loads in flight would also wait for the DMA of the
        // next block, which is the overlap this kernel exists for)
        __builtin_amdgcn_s_barrier();                    // every wave is done reading the other buffer (block it - 1)
        __builtin_amdgcn_sched_barrier(0);
        if (more) issue(buf ^ 1, rb + nstreams);
        // the residual quads of THIS block: requested now, used after the MFMAs
        float4 rr[TM];
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int m = min(m0 + 16 * i + l15, p.M - 1);
            rr[i] = p.res ? *reinterpret_cast<const float4*>(p.res + (size_t)m * p.N + n) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        // block rb has landed when only what was requested after it is outstanding
        if (more) { if (p.res) wait_vm<PPW + TM>(); else wait_vm<PPW>(); }
        else { if (p.res) wait_vm<TM>(); else wait_vm<0>(); }
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        unsigned char* blk = lds + buf * BUF;
#pragma unroll
        for (int c = 0; c < C_IT; ++c) {                 // split in place: piece pc = (slab, row, q)
            const int pc = tid + c * 256;
            const int q = pc & 3, row = (pc >> 2) % RL, s = (pc >> 2) / RL;
            unsigned char* a0 = blk + s * SLABB + row * 128 + (((2 * q) ^ (row & 7)) << 4);
            unsigned char* a1 = blk + s * SLABB + row * 128 + (((2 * q + 1) ^ (row & 7)) << 4);
            const float4 x0 = *reinterpret_cast<const float4*>(a0), x1 = *reinterpret_cast<const float4*>(a1);
            amax = fmaxf(amax, fmaxf(fmaxf(fabsf(x0.x), fabsf(x0.y)), fmaxf(fabsf(x0.z), fabsf(x0.w))));
            amax = fmaxf(amax, fmaxf(fmaxf(fabsf(x1.x), fabsf(x1.y)), fmaxf(fabsf(x1.z), fabsf(x1.w))));
            uint4 hi, lo;
            split2(x0.x, x0.y, hi.x, lo.x); split2(x0.z, x0.w, hi.y, lo.y);
            split2(x1.x, x1.y, hi.z, lo.z); split2(x1.z, x1.w, hi.w, lo.w);
            *reinterpret_cast<uint4*>(a0) = hi;
            *reinterpret_cast<uint4*>(a1) = lo;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        dh_f32x4 acc[TM], cor[TM];
#pragma unroll
        for (int i = 0; i < TM; ++i) { acc[i] = dh_f32x4{0.f, 0.f, 0.f, 0.f}; cor[i] = dh_f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
        for (int f = 0; f < NS; ++f) {
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const unsigned base = rd_base + f * SLABB + i * 2048;
                const uint4 fh = *reinterpret_cast<const uint4*>(blk + base);
                const uint4 fl = *reinterpret_cast<const uint4*>(blk + (base ^ 16u));
                acc[i] = Op16<f16_t>::mfma(wfh[f], fh, acc[i]);      // hi * hi
                cor[i] = Op16<f16_t>::mfma(wfh[f], fl, cor[i]);      // hi * lo
                cor[i] = Op16<f16_t>::mfma(wfl[f], fh, cor[i]);      // lo * hi
            }
        }
        // acc[i][r] = C[m0 + 16 i + l15][n + r]
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int m = m0 + 16 * i + l15;
            if (m >= p.M) continue;
            float4 v;
            v.x = fmaf(fmaf(cor[i][0], kLoInv, acc[i][0]) + 0.f, sc4.x, sh4.x); v.y = fmaf(fmaf(cor[i][1], kLoInv, acc[i][1]) + 0.f, sc4.y, sh4.y);
            v.z = fmaf(fmaf(cor[i][2], kLoInv, acc[i][2]) + 0.f, sc4.z, sh4.z); v.w = fmaf(fmaf(cor[i][3], kLoInv, acc[i][3]) + 0.f, sc4.w, sh4.w);
            if (p.res) { v.x += rr[i].x; v.y += rr[i].y; v.z += rr[i].z; v.w += rr[i].w; }
            if (p.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
            *reinterpret_cast<float4*>(p.C + (size_t)m * p.N + n) = v;
        }
    }
    if (amax >= kF16Max) atomicOr(p.range_flag, 1u);
}
}  // namespace

// 1 when dh_conv1x1_f32x_stream takes the layer: Cin = 64, 128 or 256, Cout a multiple of 64 whose column groups divide an XCD's
// workgroups, enough row blocks to keep the persistent grid busy
extern "C" int dh_conv1x1_f32x_stream_supported(int M, int Cin, int Cout) {
    if (!(Cin == 64 || Cin == 128 || Cin == 256) || Cout < 128 || (Cout % 64) != 0) return 0;
    const int ncg = Cout / 64;
    if (!(ncg == 2 || ncg == 4 || ncg == 8 || ncg == 16)) return 0;
    const int wpc = Cin == 64 ? 8 : Cin == 128 ? 4 : 2, streams = 8 * (32 * wpc / ncg);
    return (M + 31) / 32 >= 8 * streams;                 // at least eight row blocks per persistent workgroup
}

// y [M, Cout] fp32 = act((x [M, Cin] w^T) * scale + shift (+ residual)) for a channels-last 1 x 1 / stride 1 convolution; w_packed = the
// two planes of dh_split_f32x(w [Cout, Cin]) each through dh_pack_mfma_fragments.  Bit-identical to dh_conv2d_nhwc_f32x.
extern "C" int dh_conv1x1_f32x_stream(const float* x, const void* w_packed, const float* scale, const float* shift, const float* residual,
                                      float* y, int M, int Cin, int Cout, int relu, void* stream) {
    DH_REQUIRE(x && w_packed && scale && shift && y && dh_conv1x1_f32x_stream_supported(M, Cin, Cout));
    DH_REQUIRE(((uintptr_t)x % 16) == 0 && ((uintptr_t)w_packed % 16) == 0 && ((uintptr_t)y % 16) == 0 && ((uintptr_t)scale % 16) == 0 &&
               ((uintptr_t)shift % 16) == 0 && (!residual || ((uintptr_t)residual % 16) == 0));
    CsParams p{};
    p.A = x; p.wh = (const uint4*)w_packed; p.wl = p.wh + (size_t)(Cin / 32) * (Cout / 16) * 64;
    p.scale = scale; p.shift = shift; p.res = residual; p.C = y; p.M = M; p.N = Cout; p.K = Cin; p.relu = relu;
    hipStream_t s = (hipStream_t)stream;
    p.range_flag = dh_f32x_range_flag_of(s);
    if (!p.range_flag) return DH_ERR_LAUNCH;
    // workgroups per CU by the LDS a block pair takes (16 / 32 / 64 KB): 8 / 4 / 2; 32 CUs per XCD
    const int wpc = Cin == 64 ? 8 : Cin == 128 ? 4 : 2;
    p.ncg = Cout / 64; p.nspx = 32 * wpc / p.ncg; p.nblocks = dh_cdiv(M, 32);
    if (p.nspx < 1) return DH_ERR_UNSUPPORTED;
    dh_prof_set_tag("1x1");
    dh_prof_set_dims(M, Cout, Cin);
    DhProfScope prof("dh_conv2d_nhwc_f32x", 2.0 * M * Cout * Cin, 4.0 * ((double)M * Cin + (double)Cout * Cin + (double)M * Cout * (residual ? 2 : 1)), stream);
    const dim3 grid((unsigned)(8 * p.nspx * p.ncg)), block(256);
    if (Cin == 64) hipLaunchKernelGGL(conv1x1_f32x_stream_kernel<2>, grid, block, 0, s, p);
    else if (Cin == 128) hipLaunchKernelGGL(conv1x1_f32x_stream_kernel<4>, grid, block, 0, s, p);
    else hipLaunchKernelGGL(conv1x1_f32x_stream_kernel<8>, grid, block, 0, s, p);
    DH_LAUNCH_CHECK();
}
