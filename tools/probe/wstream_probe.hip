// How fast can every CU pull the SAME 0.5-1.5 MB of fragment-packed weights out of L2 while doing MFMAs on them?  The quantity that
// prices a "row-owning" attention block (one workgroup = one image runs self-attention, fc_o, fc_q, cross-attention and enc fc_o
// back to back and streams Wo | Wq | Weo, 512 KB each): DESIGN.md section 11.
//   hipcc --offload-arch=gfx950 -O3 -o wstream_probe wstream_probe.hip && ./wstream_probe
// 256 workgroups x 8 waves; wave w streams NF fragments of 1 KB (its 64 output columns x 512 k per GEMM) through a register ring
// of depth RING, one MFMA per fragment; back-to-back launches on one stream (so the per-launch figure contains the launch gap).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));

template <int NF, int RING>
__global__ __launch_bounds__(512, 1) void stream_kernel(const uint4* __restrict__ w, float* __restrict__ out, int frag_stride) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint4* src = w + (size_t)wave * 4 * 64 + lane;           // this wave's 4 column tiles; fragment f at + (f / 4) * frag_stride + (f % 4) * 64
    uint4 ring[RING];
    f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    const uint4 b = make_uint4(0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u);
#pragma unroll
    for (int f = 0; f < RING; ++f) ring[f] = src[(size_t)(f / 4) * frag_stride + (f % 4) * 64];
#pragma unroll
    for (int f = 0; f < NF; ++f) {
        const uint4 a = ring[f % RING];
        if (f + RING < NF) ring[f % RING] = src[(size_t)((f + RING) / 4) * frag_stride + ((f + RING) % 4) * 64];
        acc[f % 4] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf8, a), __builtin_bit_cast(bf8, b), acc[f % 4], 0, 0, 0);
    }
    float s = 0.f;
    for (int c = 0; c < 4; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
    out[(size_t)blockIdx.x * 512 + threadIdx.x] = s;
}

template <int NF, int RING>
static void run(const uint4* w, float* out, int frag_stride, const char* what) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int N = 400;
    float best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(a);
        for (int i = 0; i < N; ++i) hipLaunchKernelGGL((stream_kernel<NF, RING>), dim3(256), dim3(512), 0, 0, w, out, frag_stride);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        best = ms < best ? ms : best;
    }
    printf("{\"probe\": \"%s\", \"fragments_per_wave\": %d, \"kb_per_cu\": %d, \"ring\": %d, \"us_per_launch\": %.2f}\n", what, NF, NF * 8, RING, best * 1000.f / N);
}

int main() {
    const size_t bytes = 3u << 19;                                 // 1.5 MB of weights shared by every workgroup
    uint4* w; float* out;
    hipMalloc(&w, bytes); hipMalloc(&out, 256 * 512 * 4);
    hipMemset(w, 0x3f, bytes);
    const int fs = 32 * 64;                                        // [k step][N / 16 = 32 column tiles][64 lanes]
    run<16, 8>(w, out, fs, "tiny (launch floor + 128 KB per CU)");
    run<64, 8>(w, out, fs, "one GEMM's weights");
    run<64, 16>(w, out, fs, "one GEMM's weights");
    run<64, 32>(w, out, fs, "one GEMM's weights");
    run<128, 16>(w, out, fs, "two GEMMs' weights");
    run<192, 16>(w, out, fs, "three GEMMs' weights");
    run<192, 32>(w, out, fs, "three GEMMs' weights");
    return 0;
}
