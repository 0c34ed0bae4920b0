// Price of a grid-wide barrier on MI355X, measured: the quantity that decides whether a decode position can be ONE persistent
// cooperative launch (42 dependent GEMM / attention phases per Transformer decode position).
//   hipcc --offload-arch=gfx950 -O3 -o grid_barrier_probe grid_barrier_probe.hip && ./grid_barrier_probe
// One workgroup per CU (256 x 512 threads, co-resident), N back-to-back barriers on one agent-scope counter in L2 (sc1 atomics:
// the per-XCD L2s are not coherent), with and without a little work between barriers; compared with N empty kernel launches.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__device__ __forceinline__ void grid_barrier(unsigned* counter, unsigned& epoch, unsigned n_wg) {
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        const unsigned target = (epoch + 1) * n_wg;
        __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // (bounded: a workgroup that never arrives must not hang the box)
        for (int spin = 0; spin < (1 << 22) && __hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target; ++spin) __builtin_amdgcn_s_sleep(1);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    ++epoch;
    __syncthreads();
}

__global__ __launch_bounds__(512) void barrier_loop(unsigned* counter, int n, float* sink, int work) {
    unsigned epoch = 0;
    float acc = threadIdx.x;
    for (int i = 0; i < n; ++i) {
        for (int w = 0; w < work; ++w) acc = acc * 1.0001f + 0.5f;
        grid_barrier(counter, epoch, gridDim.x);
    }
    if (acc == 12345.f) sink[0] = acc;
}

__global__ void empty_kernel(float* sink) { if (threadIdx.x == 9999) sink[0] = 1.f; }

int main() {
    unsigned* counter; float* sink;
    hipMalloc(&counter, 4); hipMalloc(&sink, 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int N = 2000;
    for (int wg : {64, 128, 256}) {
        for (int work : {0, 2000}) {
            float best = 1e30f;
            for (int rep = 0; rep < 5; ++rep) {
                hipMemset(counter, 0, 4);
                hipEventRecord(a);
                hipLaunchKernelGGL(barrier_loop, dim3(wg), dim3(512), 0, 0, counter, N, sink, work);
                hipEventRecord(b); hipEventSynchronize(b);
                float ms; hipEventElapsedTime(&ms, a, b);
                best = ms < best ? ms : best;
            }
            // the work loop alone (no barrier): subtract to isolate the barrier
            printf("{\"workgroups\": %d, \"threads\": 512, \"work_iters\": %d, \"us_per_iteration\": %.3f}\n", wg, work, best * 1000.f / N);
        }
    }
    {
        float best = 1e30f;
        for (int rep = 0; rep < 5; ++rep) {
            hipEventRecord(a);
            for (int i = 0; i < N; ++i) hipLaunchKernelGGL(empty_kernel, dim3(256), dim3(512), 0, 0, sink);
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            best = ms < best ? ms : best;
        }
        printf("{\"empty_kernel_launches_back_to_back_us_each\": %.3f}\n", best * 1000.f / N);
    }
    return 0;
}
