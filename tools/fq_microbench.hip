// Micro-benchmark of the Fq Montgomery product on gfx950: latency of a dependent chain on one wave,
// and throughput at full occupancy.  Build: hipcc -O3 --offload-arch=gfx950 -I halo2_verifier_amd/csrc tools/fq_microbench.hip -o tools/fq_microbench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "curve.hip.h"
#include "pairing.hip.h"
using namespace h2v;
namespace h2v { void set_last_error(const std::string&) {} }

__global__ void k_chain_call(Fq* io, int iters) {
    Fq a = io[threadIdx.x + blockIdx.x * blockDim.x], b = a;
    for (int i = 0; i < iters; ++i) a = Fq::mul(a, b);
    io[threadIdx.x + blockIdx.x * blockDim.x] = a;
}
__global__ void k_chain_inl(Fq* io, int iters) {
    Fq a = io[threadIdx.x + blockIdx.x * blockDim.x], b = a;
    for (int i = 0; i < iters; ++i) a = Fq::mul_inl(a, b);
    io[threadIdx.x + blockIdx.x * blockDim.x] = a;
}
__global__ void k_chain_inl4(Fq* io, int iters) {  // four independent chains per lane
    Fq a = io[threadIdx.x + blockIdx.x * blockDim.x], b = a, c = a + a, d = c + a, e = d + a;
    for (int i = 0; i < iters; ++i) { a = Fq::mul_inl(a, b); c = Fq::mul_inl(c, b); d = Fq::mul_inl(d, b); e = Fq::mul_inl(e, b); }
    io[threadIdx.x + blockIdx.x * blockDim.x] = a + c + d + e;
}
__global__ void k_chain_fq2(Fq* io, int iters) {
    Fq a = io[threadIdx.x + blockIdx.x * blockDim.x];
    Fq2 x = {a, a + a}, y = {a + a + a, a};
    for (int i = 0; i < iters; ++i) x = Fq2::mul(x, y);
    io[threadIdx.x + blockIdx.x * blockDim.x] = x.c0 + x.c1;
}
__global__ void k_chain_dbl(Fq* io, int iters) {
    Fq a = io[threadIdx.x + blockIdx.x * blockDim.x];
    G1J p; p.X = Fq::from_u32(1); p.Y = Fq::from_u32(2); p.Z = Fq::one();
    for (int i = 0; i < iters; ++i) p = g1_dbl(p);
    io[threadIdx.x + blockIdx.x * blockDim.x] = p.X + a;
}
__global__ void k_chain_f12sqr(Fq* io, int iters) {
    Fq a = io[threadIdx.x + blockIdx.x * blockDim.x];
    Fq12 f = Fq12::one(); f.c0.c0.c0 = a; f.c1.c1.c1 = a + a; f.c0.c2.c0 = a;
    for (int i = 0; i < iters; ++i) f = f * f;
    io[threadIdx.x + blockIdx.x * blockDim.x] = f.c0.c0.c0 + f.c1.c2.c1;
}

template <class K> float run(K kern, Fq* d, int blocks, int threads, int iters) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, d, 4);
    hipDeviceSynchronize();
    hipEventRecord(e0); hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, d, iters); hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}
int main() {
    int n = 4096 * 256;  // >= the largest grid launched below
    std::vector<Fq> h(n);
    for (int i = 0; i < n; ++i) h[i] = Fq::from_u32(12345 + i);
    Fq* d; hipMalloc(&d, n * sizeof(Fq)); hipMemcpy(d, h.data(), n * sizeof(Fq), hipMemcpyHostToDevice);
    const int it = 2000;
    printf("one wave, dependent chain (ns per Fq mul):\n");
    printf("  call      %8.1f\n", run(k_chain_call, d, 1, 64, it) * 1e6 / it);
    printf("  inline    %8.1f\n", run(k_chain_inl, d, 1, 64, it) * 1e6 / it);
    printf("  inline x4 %8.1f (per mul, 4 independent chains)\n", run(k_chain_inl4, d, 1, 64, it) * 1e6 / it / 4);
    printf("  Fq2::mul  %8.1f (per Fq2 product = 3 Fq muls)\n", run(k_chain_fq2, d, 1, 64, it) * 1e6 / it);
    printf("  g1_dbl    %8.1f (per doubling = 7 Fq muls)\n", run(k_chain_dbl, d, 1, 64, it) * 1e6 / it);
    printf("  Fq12 sqr  %8.1f (per Fq12 product on ONE lane, tower form)\n", run(k_chain_f12sqr, d, 1, 64, 200) * 1e6 / 200);
    printf("full chip (256 CUs x 8 waves/SIMD-ish), G Fq mul/s:\n");
    for (int wpb : {64, 256}) for (int blocks : {256, 1024, 2048, 4096}) {
        float ms = run(k_chain_inl, d, blocks, wpb, it);
        printf("  inline  blocks=%5d threads=%3d  %8.2f Gmul/s\n", blocks, wpb, (double)blocks * wpb * it / ms / 1e6);
        ms = run(k_chain_call, d, blocks, wpb, it);
        printf("  call    blocks=%5d threads=%3d  %8.2f Gmul/s\n", blocks, wpb, (double)blocks * wpb * it / ms / 1e6);
    }
    return 0;
}
