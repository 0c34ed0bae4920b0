// Developer probe: do two dependent kernel chains on two HIP streams overlap on one MI355X?
// Kernel: `wgs` workgroups of 512 threads with 128 KB of LDS (one per CU), each spinning ~`us` microseconds.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <chrono>

__global__ __launch_bounds__(512) void spin(float* out, long long cycles) {
    __shared__ float lds[32 * 1024];
    lds[threadIdx.x] = threadIdx.x;
    const long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < cycles) { }
    if (out && lds[threadIdx.x] < 0) out[0] = 1.f;
}

static double chain(int nstreams, hipStream_t* st, int wgs, int launches, long long cycles) {
    hipDeviceSynchronize();
    auto t0 = std::chrono::high_resolution_clock::now();
    for (int i = 0; i < launches; ++i)
        for (int s = 0; s < nstreams; ++s) hipLaunchKernelGGL(spin, dim3(wgs), dim3(512), 0, st[s], nullptr, cycles);
    hipDeviceSynchronize();
    return std::chrono::duration<double, std::micro>(std::chrono::high_resolution_clock::now() - t0).count();
}

int main() {
    hipStream_t st[4];
    for (int i = 0; i < 4; ++i) hipStreamCreateWithFlags(&st[i], hipStreamNonBlocking);
    for (long long cyc : {200LL, 500LL, 1500LL}) {         // s_memtime ticks at 100 MHz: 2 / 5 / 15 us
        chain(1, st, 160, 20, cyc);
        const double a = chain(1, st, 160, 400, cyc), b = chain(2, st, 80, 400, cyc), c = chain(2, st, 160, 400, cyc), d = chain(4, st, 40, 400, cyc);
        printf("spin %4.1f us: 1 stream x 160 WG: %6.2f us/launch | 2 streams x 80 WG: %6.2f us per pair | 2 streams x 160 WG: %6.2f | 4 streams x 40 WG: %6.2f\n",
               cyc / 100.0, a / 400, b / 400, c / 400, d / 400);
    }
    // same through a captured graph with two parallel branches
    for (long long cyc : {200LL, 500LL}) {
        hipGraph_t g; hipGraphExec_t ge;
        hipStreamBeginCapture(st[0], hipStreamCaptureModeGlobal);
        hipEvent_t fork, join; hipEventCreate(&fork); hipEventCreate(&join);
        hipEventRecord(fork, st[0]); hipStreamWaitEvent(st[1], fork, 0);
        for (int i = 0; i < 200; ++i) { hipLaunchKernelGGL(spin, dim3(80), dim3(512), 0, st[0], nullptr, cyc); hipLaunchKernelGGL(spin, dim3(80), dim3(512), 0, st[1], nullptr, cyc); }
        hipEventRecord(join, st[1]); hipStreamWaitEvent(st[0], join, 0);
        hipStreamEndCapture(st[0], &g);
        hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
        hipGraphLaunch(ge, st[0]); hipStreamSynchronize(st[0]);
        auto t0 = std::chrono::high_resolution_clock::now();
        for (int r = 0; r < 5; ++r) hipGraphLaunch(ge, st[0]);
        hipStreamSynchronize(st[0]);
        const double us = std::chrono::duration<double, std::micro>(std::chrono::high_resolution_clock::now() - t0).count();
        printf("graph, 2 branches x 200 launches x 80 WG, spin %4.1f us: %6.2f us per pair\n", cyc / 100.0, us / 5 / 200);
        hipStreamBeginCapture(st[0], hipStreamCaptureModeGlobal);
        for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(spin, dim3(160), dim3(512), 0, st[0], nullptr, cyc);
        hipStreamEndCapture(st[0], &g);
        hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
        hipGraphLaunch(ge, st[0]); hipStreamSynchronize(st[0]);
        t0 = std::chrono::high_resolution_clock::now();
        for (int r = 0; r < 5; ++r) hipGraphLaunch(ge, st[0]);
        hipStreamSynchronize(st[0]);
        const double us1 = std::chrono::duration<double, std::micro>(std::chrono::high_resolution_clock::now() - t0).count();
        printf("graph, 1 chain x 200 launches x 160 WG, spin %4.1f us: %6.2f us per launch\n", cyc / 100.0, us1 / 5 / 200);
    }
    return 0;
}
