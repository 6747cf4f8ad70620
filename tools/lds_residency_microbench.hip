// How many workgroups share a CU as a function of their LDS allocation?  N one-wave workgroups run a fixed ALU chain; when they
// no longer fit side by side the elapsed time steps up by whole multiples of one chain.
// Build: hipcc -O3 --offload-arch=gfx950 tools/lds_residency_microbench.hip -o tools/lds_residency_microbench
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void __launch_bounds__(64) k_spin(unsigned* out, int iters) {
    extern __shared__ unsigned lds[];
    unsigned x = threadIdx.x + 1;
    lds[threadIdx.x] = x;
    for (int i = 0; i < iters; ++i) x = x * 1664525u + 1013904223u + lds[(x >> 8) & 63];
    out[blockIdx.x * 64 + threadIdx.x] = x;
}
int main() {
    unsigned* d; hipMalloc(&d, 4 * 64 * 8192);
    hipFuncSetAttribute((const void*)k_spin, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    for (int wgs : {256, 320, 512, 1024}) {
        printf("workgroups=%d (1 wave each):", wgs);
        for (int kb : {1, 16, 20, 24, 32, 40, 48, 56, 64, 80, 96, 128, 160}) {
            hipLaunchKernelGGL(k_spin, dim3(wgs), dim3(64), kb * 1024, 0, d, 10); hipDeviceSynchronize();
            hipEventRecord(e0); hipLaunchKernelGGL(k_spin, dim3(wgs), dim3(64), kb * 1024, 0, d, iters); hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("  %dK:%.2f", kb, ms);
        }
        printf("  ms\n");
    }
    return 0;
}
