// Developer probe (not part of the product): how fast can one workgroup per CU pull GEMM operand slabs through an LDS
// ring by LDS-DMA, as gemm_bf16_kernel<64,64,...> does, for different source layouts?
//   mode 0: row-major operands (a piece = 8 rows x 128 B at stride ld)       -- what dh_linear does today
//   mode 1: slab-major operands ([K/64][rows][64]: a piece = 1 KiB contiguous)
// Shapes: M x N x K = 1280 x 512 x 2048 (ffn2) and 1280 x 512 x 512 (proj); 64 x 64 tiles; NW waves; NS-deep ring.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef const void __attribute__((address_space(1)))* gptr_t;
typedef void __attribute__((address_space(3)))* lptr_t;

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }

template <int NW, int NS, int MODE>
__global__ __launch_bounds__(64 * NW) void probe(const uint16_t* A, const uint16_t* W, int M, int N, int K, int tiles_m, int tiles_n,
                                                  float* sink) {
    constexpr int BM = 64, BN = 64, SLAB = (BM + BN) * 128;
    constexpr int IA = BM / (8 * NW) > 0 ? BM / (8 * NW) : 1, IB = IA, G = IA + IB;
    __shared__ __attribute__((aligned(16))) unsigned char lds[NS * SLAB];
    const int nblk = tiles_m * tiles_n;
    int bid = blockIdx.x;
    { const int q = nblk / 8, r = nblk % 8, xcd = bid % 8, idx = bid / 8; bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx; }
    const int tm = bid / tiles_n, tn = bid % tiles_n;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane >> 3, lpos = lane & 7;
    const uint16_t* a_run[IA]; const uint16_t* b_run[IB];
    int a_stp, b_stp;
#pragma unroll
    for (int i = 0; i < IA; ++i) {
        const int row = (wave * IA + i) * 8 + lr, swz = lpos ^ (row & 7);
        if (MODE == 0) { a_run[i] = A + (size_t)(tm * BM + row) * K + swz * 8; b_run[i] = W + (size_t)(tn * BN + row) * K + swz * 8; }
        else { a_run[i] = A + (size_t)(tm * BM + row) * 64 + swz * 8; b_run[i] = W + (size_t)(tn * BN + row) * 64 + swz * 8; }
    }
    a_stp = MODE == 0 ? 64 : M * 64; b_stp = MODE == 0 ? 64 : N * 64;
    const int nslab = K / 64;
    auto stage = [&](int buf) {
        unsigned char* slab = lds + buf * SLAB;
#pragma unroll
        for (int i = 0; i < IA; ++i) {
            if ((wave * IA + i) * 8 < BM) {
                __builtin_amdgcn_global_load_lds((gptr_t)a_run[i], (lptr_t)(slab + (wave * IA + i) * 1024), 16, 0, 0);
                __builtin_amdgcn_global_load_lds((gptr_t)b_run[i], (lptr_t)(slab + BM * 128 + (wave * IA + i) * 1024), 16, 0, 0);
            }
            a_run[i] += a_stp; b_run[i] += b_stp;
        }
    };
    float acc = 0.f;
#pragma unroll
    for (int u = 0; u < NS - 1; ++u) if (u < nslab) stage(u);
    for (int t = 0; t < nslab; ++t) {
        const int newer = min(NS - 2, nslab - 1 - t) * G;
        if (newer >= 12) wait_vm<12>(); else if (newer >= 8) wait_vm<8>(); else if (newer >= 6) wait_vm<6>(); else if (newer >= 4) wait_vm<4>();
        else if (newer >= 2) wait_vm<2>(); else wait_vm<0>();
        __builtin_amdgcn_s_barrier();
        if (t + NS - 1 < nslab) stage((t + NS - 1) % NS);
        const float* s = reinterpret_cast<const float*>(lds + (t % NS) * SLAB);
        acc += s[tid] + s[2048 + tid];
    }
    if (acc == 123.456f) sink[0] = acc;
}

template <int NW, int NS, int MODE>
static float run(const uint16_t* A, const uint16_t* W, int M, int N, int K, float* sink, int iters, const std::vector<uint16_t*>& As,
                 const std::vector<uint16_t*>& Ws) {
    const int tm = M / 64, tn = N / 64;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((probe<NW, NS, MODE>), dim3(tm * tn), dim3(64 * NW), 0, 0, As[i % As.size()], Ws[i % Ws.size()], M, N, K, tm, tn, sink);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL((probe<NW, NS, MODE>), dim3(tm * tn), dim3(64 * NW), 0, 0, As[i % As.size()], Ws[i % Ws.size()], M, N, K, tm, tn, sink);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms / iters * 1e3f;
}

int main() {
    const int M = 1280, N = 512;
    std::vector<uint16_t*> As, Ws;
    for (int i = 0; i < 6; ++i) { uint16_t* a; uint16_t* w; hipMalloc(&a, (size_t)M * 2048 * 2); hipMalloc(&w, (size_t)2048 * 2048 * 2);
        hipMemset(a, 0, (size_t)M * 2048 * 2); hipMemset(w, 0, (size_t)2048 * 2048 * 2); As.push_back(a); Ws.push_back(w); }
    float* sink; hipMalloc(&sink, 64);
    for (int K : {512, 2048}) {
        for (int n : {512, 2048}) {
            if (K == 2048 && n == 2048) continue;
            const double mb = (double)(M / 64) * (n / 64) * 128.0 * K * 2 / 1e6;
#define RUN(NW, NS, MODE) { float us = run<NW, NS, MODE>(As[0], Ws[0], M, n, K, sink, 50, As, Ws); \
    printf("M=%d N=%d K=%d  NW=%d NS=%d mode=%d : %7.2f us  %6.2f TB/s  (%d WGs, %.0f KB each)\n", M, n, K, NW, NS, MODE, us, mb / us / 1e6 * 1e6 / 1e6, (M / 64) * (n / 64), 128.0 * K * 2 / 1024); }
            RUN(8, 8, 0) RUN(8, 8, 1) RUN(4, 4, 0) RUN(4, 4, 1) RUN(8, 4, 0) RUN(8, 4, 1) RUN(8, 2, 0) RUN(8, 2, 1) RUN(4, 8, 1) RUN(4, 2, 1)
        }
    }
    return 0;
}
