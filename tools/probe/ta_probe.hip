// Developer probe (not part of the product): how many bytes per cycle per CU do PLAIN vector loads (global_load_dwordx4 into
// registers, lanes in the MFMA fragment pattern: 16 rows x 64 contiguous bytes per wave instruction) deliver from an
// L2-resident buffer, compared with LDS-DMA pieces (global_load_lds_dwordx4, 8 rows x 128 bytes)?
//   mode 0: plain loads, every wave of a workgroup reads its own rows      mode 1: plain loads, the 8 waves read the SAME rows (L1 hits)
//   mode 2: LDS-DMA pieces into a ring (no reads)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef void __attribute__((address_space(3)))* lptr_t;
__device__ __forceinline__ void lds_dma16(const void* gsrc, void* lds_dst) {
    const unsigned m = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(lptr_t)lds_dst);
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" :: "s"(m), "v"(gsrc) : "memory");
}

template <int MODE, int UN>
__global__ __launch_bounds__(512) void probe(const uint16_t* buf, int rows, int ld, int iters, uint32_t* sink) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[64 * 1024];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l15 = lane & 15, lq = lane >> 4, lr = lane >> 3, lpos = lane & 7;
    uint4 acc = make_uint4(0, 0, 0, 0);
    const int wsel = MODE == 1 ? 0 : wave;
    int r0 = (blockIdx.x * 8 + wsel) * 16;
    for (int it = 0; it < iters; ++it) {
        if (MODE < 2) {
            uint4 v[UN];
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                const int row = (r0 + u * 16 + l15) % rows;
                v[u] = *reinterpret_cast<const uint4*>(buf + (size_t)row * ld + ((it * 4 + lq) * 8) % ld);
            }
#pragma unroll
            for (int u = 0; u < UN; ++u) { acc.x ^= v[u].x; acc.y ^= v[u].y; acc.z ^= v[u].z; acc.w ^= v[u].w; }
        } else {
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                const int row = (r0 + u * 8 + lr) % rows;
                lds_dma16(buf + (size_t)row * ld + ((it * 8 + lpos) * 8) % ld, lds + wave * 8192 + (u & 7) * 1024);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        r0 += 16 * UN;
    }
    if (acc.x == 0x12345678u) sink[0] = acc.y + acc.z + acc.w;
}

// GEMM-loop skeleton: NS-deep ring, per iteration: counted wait for the oldest slab, workgroup barrier, UN new pieces per wave
#define VMCASE(x) case x: asm volatile("s_waitcnt vmcnt(" #x ")" ::: "memory"); break;
__device__ __forceinline__ void wait_any(int n) {
    switch (n) {
        VMCASE(1) VMCASE(2) VMCASE(3) VMCASE(4) VMCASE(5) VMCASE(6) VMCASE(7) VMCASE(8) VMCASE(9) VMCASE(10) VMCASE(11) VMCASE(12)
        VMCASE(13) VMCASE(14) VMCASE(15) VMCASE(16) VMCASE(17) VMCASE(18) VMCASE(19) VMCASE(20) VMCASE(21) VMCASE(22) VMCASE(23) VMCASE(24)
        VMCASE(25) VMCASE(26) VMCASE(27) VMCASE(28) VMCASE(29) VMCASE(30) VMCASE(31) VMCASE(32) VMCASE(33) VMCASE(34) VMCASE(35) VMCASE(36)
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
}
template <int UN, int NS, bool BARRIER, int SW = 0>
__global__ __launch_bounds__(512) void ring(const uint16_t* buf, int rows, int ld, int iters, uint32_t* sink) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[128 * 1024];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int lr = lane >> 3, lpos = lane & 7;
    int r0 = (blockIdx.x * 8 + wave) * 16;
    auto stage = [&](int it) {
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int row = (r0 + u * 8 + lr) % rows;
            lds_dma16(buf + (size_t)row * ld + ((it * 8 + lpos) * 8) % ld, lds + ((it % NS) * 8 * UN + wave * UN + u) % 128 * 1024);
        }
        r0 += 8 * UN;
    };
    for (int u = 0; u < NS - 1; ++u) stage(u);
    for (int it = 0; it < iters; ++it) {
        if (SW == 0) asm volatile("s_waitcnt vmcnt(%0)" :: "n"((NS - 2) * UN) : "memory");
        else wait_any((NS - 2) * UN + (rows == 1 ? it : 0));        // runtime value: the switch stays a switch
        if (BARRIER) __builtin_amdgcn_s_barrier();
        if (SW == 2) {                                               // + LDS reads of the consumed slab, as a GEMM wave does
            const uint4 a = *reinterpret_cast<const uint4*>(lds + ((it % NS) * 16 + (lane & 15)) * 1024 % (128 * 1024) + (lane >> 4) * 16);
            const uint4 b = *reinterpret_cast<const uint4*>(lds + ((it % NS) * 16 + (lane & 15) + 3) * 1024 % (128 * 1024) + (lane >> 4) * 16);
            if ((a.x ^ b.y) == 0x7777u) sink[1] = 1;
        }
        stage(it + NS - 1);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lds[threadIdx.x] == 77 && iters < 0) sink[0] = 1;
}

template <int UN, int NS, bool BARRIER, int SW = 0>
static void run_ring(const uint16_t* buf, int rows, int ld, uint32_t* sink) {
    const int iters = 1600 / UN;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((ring<UN, NS, BARRIER, SW>), dim3(256), dim3(512), 0, 0, buf, rows, ld, iters, sink);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0, 0);
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL((ring<UN, NS, BARRIER, SW>), dim3(256), dim3(512), 0, 0, buf, rows, ld, iters, sink);
    (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double bytes = 10.0 * 256 * 8 * (iters + NS - 1) * UN * 1024.0;
    printf("ring: %d pieces per wave per step, %d-deep, barrier %d, variant %d   %7.1f GB/s per CU  %6.2f TB/s chip\n", UN, NS, (int)BARRIER, SW,
           bytes / (ms * 1e-3) / 256 / 1e9, bytes / (ms * 1e-3) / 1e12);
}

// ring + the rest of a GEMM slab step: READS ds_read_b128 of the landed slab and MFMAS v_mfma_f32_16x16x32_bf16 per wave per step
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int UN, int NS, int READS, int MFMAS, bool LGKM_WAIT>
__global__ __launch_bounds__(512) void ring2(const uint16_t* buf, int rows, int ld, int iters, uint32_t* sink) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[128 * 1024];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int lr = lane >> 3, lpos = lane & 7;
    int r0 = (blockIdx.x * 8 + wave) * 16;
    const uint16_t* src = buf + (size_t)((r0 + lr) % (rows - 512)) * ld + lpos * 8;
    auto stage = [&](int it) {
#pragma unroll
        for (int u = 0; u < UN; ++u)
            lds_dma16(src + (size_t)((it * UN + u) % 64) * 8 * ld, lds + (((it % NS) * 8 + wave) * UN + u) % 128 * 1024);
    };
    f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    uint4 fr[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) fr[i] = make_uint4(lane + i, wave, i, 1);
    for (int u = 0; u < NS - 1; ++u) stage(u);
    for (int it = 0; it < iters; ++it) {
        asm volatile("s_waitcnt vmcnt(%0)" :: "n"((NS - 2) * UN) : "memory");
        __builtin_amdgcn_s_barrier();
        const unsigned char* sb = lds + ((it % NS) * 16 * 1024) % (128 * 1024);
#pragma unroll
        for (int i = 0; i < READS; ++i)
            fr[i & 7] = *reinterpret_cast<const uint4*>(sb + (((lane & 15) + 16 * (i & 3)) * 128 + (((lane >> 4) ^ (lane & 7)) << 4) + (i >> 2) * 8192) % (16 * 1024));
        __builtin_amdgcn_sched_barrier(0);
        stage(it + NS - 1);
        if (LGKM_WAIT) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < MFMAS; ++i)
            acc[i & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fr[i & 7]), __builtin_bit_cast(bf16x8, fr[(i + 3) & 7]), acc[i & 3], 0, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3] == 123.25f) sink[0] = 1;
}

template <int UN, int NS, int READS, int MFMAS, bool LGKM_WAIT = true>
static void run_ring2(const uint16_t* buf, int rows, int ld, uint32_t* sink) {
    const int iters = 800;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((ring2<UN, NS, READS, MFMAS, LGKM_WAIT>), dim3(256), dim3(512), 0, 0, buf, rows, ld, iters, sink);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0, 0);
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL((ring2<UN, NS, READS, MFMAS, LGKM_WAIT>), dim3(256), dim3(512), 0, 0, buf, rows, ld, iters, sink);
    (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double us_step = ms / 10 * 1e3 / iters;
    printf("ring2: %d pieces + %2d LDS reads + %2d MFMAs per wave per step, %d-deep: %6.3f us per step (%4.0f cycles at 2.1 GHz; MFMA alone would need %4.0f)\n",
           UN, READS, MFMAS, NS, us_step, us_step * 2100, MFMAS * 16.0 * 2);
}

template <int MODE, int UN>
static void run(const uint16_t* buf, int rows, int ld, uint32_t* sink, const char* name) {
    const int iters = 200;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((probe<MODE, UN>), dim3(256), dim3(512), 0, 0, buf, rows, ld, iters, sink);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL((probe<MODE, UN>), dim3(256), dim3(512), 0, 0, buf, rows, ld, iters, sink);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double bytes = 10.0 * 256 * 8 * iters * UN * 1024.0;
    printf("%-44s UN=%2d  %7.1f GB/s per CU  %6.2f TB/s chip  (%.1f us per launch)\n", name, UN, bytes / (ms * 1e-3) / 256 / 1e9,
           bytes / (ms * 1e-3) / 1e12, ms / 10 * 1e3);
}

int main() {
    const int rows = 4096, ld = 512;                      // 4 MB buffer: L2 / Infinity-Cache resident
    uint16_t* buf; uint32_t* sink;
    hipMalloc(&buf, (size_t)rows * ld * 2); hipMalloc(&sink, 64);
    hipMemset(buf, 1, (size_t)rows * ld * 2);
    if (getenv("TA_RANDOM")) {          // random bf16 in (-2, 2): MFMA-dense loops clock lower on random data than on trivial operands
        std::vector<uint16_t> h((size_t)rows * ld);
        uint32_t x = 12345u;
        for (auto& v : h) { x = x * 1664525u + 1013904223u; v = (uint16_t)(((x >> 16) & 0x807Fu) | (0x7Eu << 7) | ((x >> 3) & 0x0080u)); }
        hipMemcpy(buf, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    }
    run<0, 4>(buf, rows, ld, sink, "plain dwordx4, own rows per wave");
    run<0, 8>(buf, rows, ld, sink, "plain dwordx4, own rows per wave");
    run<0, 16>(buf, rows, ld, sink, "plain dwordx4, own rows per wave");
    run<1, 8>(buf, rows, ld, sink, "plain dwordx4, 8 waves share rows (L1 hits)");
    run<1, 16>(buf, rows, ld, sink, "plain dwordx4, 8 waves share rows (L1 hits)");
    run<2, 4>(buf, rows, ld, sink, "LDS-DMA dwordx4 pieces");
    run<2, 8>(buf, rows, ld, sink, "LDS-DMA dwordx4 pieces");
    run_ring<1, 2, true>(buf, rows, ld, sink); run_ring<2, 2, true>(buf, rows, ld, sink); run_ring<4, 2, true>(buf, rows, ld, sink);
    run_ring<8, 2, true>(buf, rows, ld, sink);
    run_ring<1, 8, true>(buf, rows, ld, sink); run_ring<2, 8, true>(buf, rows, ld, sink); run_ring<4, 4, true>(buf, rows, ld, sink);
    run_ring2<2, 8, 0, 0>(buf, rows, ld, sink); run_ring2<2, 8, 8, 0>(buf, rows, ld, sink); run_ring2<2, 8, 0, 16>(buf, rows, ld, sink);
    run_ring2<2, 8, 8, 16>(buf, rows, ld, sink); run_ring2<0, 8, 8, 16>(buf, rows, ld, sink); run_ring2<0, 8, 0, 16>(buf, rows, ld, sink);
    run_ring2<4, 2, 12, 16>(buf, rows, ld, sink); run_ring2<4, 4, 12, 16>(buf, rows, ld, sink); run_ring2<2, 8, 8, 32>(buf, rows, ld, sink);
    run_ring<2, 8, true, 1>(buf, rows, ld, sink); run_ring<2, 8, true, 2>(buf, rows, ld, sink);
    run_ring<2, 8, false>(buf, rows, ld, sink); run_ring<4, 4, false>(buf, rows, ld, sink); run_ring<2, 4, true>(buf, rows, ld, sink);
    return 0;
}
