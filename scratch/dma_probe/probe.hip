// How fast can a CU pull L2-resident data into LDS with global_load_lds_dwordx4?  Each workgroup streams its own
// window of a buffer (window << L2, re-read many times) through a ring of 1-KiB pieces; no compute.
// Parameters: waves per workgroup, pieces in flight per wave (vmcnt depth), workgroups per CU (by LDS size).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef const void __attribute__((address_space(1)))* gptr_t;
typedef void __attribute__((address_space(3)))* lptr_t;

template <int DEPTH>
__global__ __launch_bounds__(1024) void stream_kernel(const unsigned char* __restrict__ buf, size_t window, int iters, int lds_bytes,
                                                      unsigned* sink) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = blockDim.x >> 6;
    const unsigned char* base = buf + (size_t)blockIdx.x * window;
    // each wave owns DEPTH 1-KiB slots in LDS and walks the window in steps of nw KiB
    size_t off = (size_t)wave * 1024;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            const unsigned char* src = base + (off % window) + lane * 16;
            __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(lds + ((wave * DEPTH + d) * 1024) % lds_bytes), 16, 0, 0);
            off += (size_t)nw * 1024;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    if (threadIdx.x == 0 && iters < 0) sink[0] = lds[0];
}

template <int DEPTH>
static double run(int blocks, int nw, int lds_bytes, const unsigned char* buf, size_t window, int iters, unsigned* sink) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(stream_kernel<DEPTH>, dim3(blocks), dim3(64 * nw), lds_bytes, 0, buf, window, iters, lds_bytes, sink);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(stream_kernel<DEPTH>, dim3(blocks), dim3(64 * nw), lds_bytes, 0, buf, window, iters, lds_bytes, sink);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    return (double)blocks * nw * DEPTH * 1024.0 * iters / (ms * 1e-3) / 1e9;      // GB/s whole chip
}

int main() {
    const size_t window = 64 * 1024;                     // per-workgroup window: stays in L2
    const int max_blocks = 2048;
    unsigned char* buf; (void)hipMalloc(&buf, window * max_blocks); (void)hipMemset(buf, 1, window * max_blocks);
    unsigned* sink; (void)hipMalloc(&sink, 4);
    printf("%-28s %10s %12s\n", "config", "TB/s chip", "GB/s per CU");
    struct Cfg { int wg_per_cu, nw, lds; } cfgs[] = {{1, 4, 65536}, {1, 8, 65536}, {1, 16, 65536}, {2, 8, 65536}, {2, 4, 65536},
                                                     {4, 4, 32768}, {4, 8, 32768}, {2, 16, 65536}};
    for (auto c : cfgs) {
        const int blocks = 256 * c.wg_per_cu;
        const int iters = 2000;
        double g1 = run<1>(blocks, c.nw, c.lds, buf, window, iters, sink);
        double g2 = run<2>(blocks, c.nw, c.lds, buf, window, iters / 2, sink);
        double g4 = run<4>(blocks, c.nw, c.lds, buf, window, iters / 4, sink);
        double g8 = run<8>(blocks, c.nw, c.lds, buf, window, iters / 8, sink);
        char name[64]; snprintf(name, sizeof name, "%d wg/CU x %2d waves", c.wg_per_cu, c.nw);
        printf("%-20s depth1 %6.2f TB/s (%5.1f/CU)  depth2 %6.2f (%5.1f)  depth4 %6.2f (%5.1f)  depth8 %6.2f (%5.1f)\n", name,
               g1 / 1e3, g1 / 256, g2 / 1e3, g2 / 256, g4 / 1e3, g4 / 256, g8 / 1e3, g8 / 256);
    }
    return 0;
}
