// How fast does ONE wave run a dependent chain of Fq products when other waves share the chip?  (The verifier's tail kernels
// — Fr program, window reduction, Horner, pairing — are such chains.)  Launches N workgroups of one wave, each lane running a
// chain of K dependent Montgomery products, and reports the time per product seen by a wave.
// Build: hipcc -O3 --offload-arch=gfx950 -I halo2_verifier_amd/csrc tools/latency_microbench.hip -o tools/latency_microbench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <string>
#include "curve.hip.h"
using namespace h2v;
namespace h2v { void set_last_error(const std::string&) {} }

__global__ void __launch_bounds__(64) k_chain(Fq* io, int iters) {
    Fq x = io[blockIdx.x * 64 + threadIdx.x], y = x;
    for (int i = 0; i < iters; ++i) x = Fq::mul_inl(x, y);
    io[blockIdx.x * 64 + threadIdx.x] = x;
}
__global__ void __launch_bounds__(256) k_chain4(Fq* io, int iters) {   // four waves in ONE workgroup: same CU for sure
    Fq x = io[blockIdx.x * 256 + threadIdx.x], y = x;
    for (int i = 0; i < iters; ++i) x = Fq::mul_inl(x, y);
    io[blockIdx.x * 256 + threadIdx.x] = x;
}

int main() {
    const int iters = 4000;
    Fq* d; hipMalloc(&d, sizeof(Fq) * 64 * 8192); hipMemset(d, 1, sizeof(Fq) * 64 * 8192);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    printf("workgroups of 1 wave; ns per product as seen by one wave (chain of %d dependent products)\n", iters);
    for (int blocks : {1, 8, 64, 128, 256, 320, 384, 512, 768, 1024, 2048, 4096}) {
        hipLaunchKernelGGL(k_chain, dim3(blocks), dim3(64), 0, 0, d, 10);
        hipDeviceSynchronize();
        hipEventRecord(e0); hipLaunchKernelGGL(k_chain, dim3(blocks), dim3(64), 0, 0, d, iters); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("  waves=%5d  %7.1f ns/product  (%6.1f G products/s)\n", blocks, ms * 1e6 / iters, (double)blocks * 64 * iters / ms / 1e6);
    }
    printf("workgroups of 4 waves (one CU each)\n");
    for (int blocks : {1, 64, 256}) {
        hipEventRecord(e0); hipLaunchKernelGGL(k_chain4, dim3(blocks), dim3(256), 0, 0, d, iters); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("  workgroups=%4d  %7.1f ns/product\n", blocks, ms * 1e6 / iters);
    }
    return 0;
}
