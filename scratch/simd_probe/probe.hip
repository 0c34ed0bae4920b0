// Which SIMD does wave w of a 512-thread workgroup land on?  (HW_ID: simd_id = bits 5:4 on gfx9)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int* out) {
    const int wave = threadIdx.x >> 6;
    const unsigned hw = __builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11));   // HW_REG_HW_ID, offset 0, size 32
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 16 + wave] = (int)hw;
}
int main() {
    int* d; hipMalloc(&d, 4 * 16 * sizeof(int));
    for (int nt : {512, 1024}) {
        hipLaunchKernelGGL(k, dim3(4), dim3(nt), 0, 0, d);
        int h[64]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        for (int b = 0; b < 2; ++b) {
            printf("nt=%d block %d:", nt, b);
            for (int w = 0; w < nt / 64; ++w) printf(" w%d:simd%d/wv%d", w, (h[b * 16 + w] >> 4) & 3, h[b * 16 + w] & 15);
            printf("\n");
        }
    }
    return 0;
}
