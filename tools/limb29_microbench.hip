// Micro-benchmark of the Fq Montgomery product with 9 x 29-bit limbs (product scanning, 64-bit column accumulators, no carry
// chains) on gfx950: the prototype this file started as (mul29 / sqr29 / add29, kept here), the library's own Fq::mul_inl
// (csrc/bn254.hip.h adopted the prototype; the 8 x 32-bit CIOS product it replaced measured 1052 ns on a lone wave and
// 83 G products/s chip-wide with this same harness), a two-accumulator variant (no gain: recorded negative result), and the
// product rate as a function of occupancy (1, 2, 3 waves per SIMD) — the MSM kernels run at 2 waves per SIMD.
// Build: hipcc -O3 --offload-arch=gfx950 -I halo2_verifier_amd/csrc tools/limb29_microbench.hip -o tools/limb29_microbench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "curve.hip.h"
using namespace h2v;
namespace h2v { void set_last_error(const std::string&) {} }

struct F29 { uint32_t v[9]; };
__device__ __constant__ const uint32_t P29c[9] = {0x187cfd47u, 0x10460b6u, 0x1c72a34fu, 0x2d522d0u, 0x1585d978u, 0x2db40c0u, 0xa6e141u, 0xe5c2634u, 0x30644eu};
#define MASK29 0x1fffffffu
#define INV29 0x4866389u

template <int I> struct P29 { };
__device__ __forceinline__ constexpr uint32_t p29(int i) {
    constexpr uint32_t t[9] = {0x187cfd47u, 0x10460b6u, 0x1c72a34fu, 0x2d522d0u, 0x1585d978u, 0x2db40c0u, 0xa6e141u, 0xe5c2634u, 0x30644eu};
    return t[i];
}

// a, b < 8p (limbs < 2^29)  ->  a*b/R mod p, result < 2p, R = 2^261
__device__ __forceinline__ F29 mul29(const F29& a, const F29& b) {
    uint64_t acc = 0;
    uint32_t m[9];
    F29 r;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
#pragma unroll
        for (int i = 0; i <= k; ++i) acc += (uint64_t)a.v[i] * b.v[k - i];
#pragma unroll
        for (int i = 0; i < k; ++i) acc += (uint64_t)m[i] * p29(k - i);
        m[k] = ((uint32_t)acc * INV29) & MASK29;
        acc += (uint64_t)m[k] * p29(0);
        acc >>= 29;
    }
#pragma unroll
    for (int k = 9; k < 17; ++k) {
#pragma unroll
        for (int i = k - 8; i <= 8; ++i) acc += (uint64_t)a.v[i] * b.v[k - i];
#pragma unroll
        for (int i = k - 8; i <= 8; ++i) acc += (uint64_t)m[i] * p29(k - i);
        r.v[k - 9] = (uint32_t)acc & MASK29;
        acc >>= 29;
    }
    r.v[8] = (uint32_t)acc;
    return r;
}
// two accumulators per column (a*b terms / m*p terms): halves the dependent v_mad_u64_u32 chain of a lone wave
__device__ __forceinline__ F29 mul29_2acc(const F29& a, const F29& b) {
    uint64_t acc = 0;
    uint32_t m[9];
    F29 r;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        uint64_t acc2 = 0;
#pragma unroll
        for (int i = 0; i <= k; ++i) acc += (uint64_t)a.v[i] * b.v[k - i];
#pragma unroll
        for (int i = 0; i < k; ++i) acc2 += (uint64_t)m[i] * p29(k - i);
        acc += acc2;
        m[k] = ((uint32_t)acc * INV29) & MASK29;
        acc += (uint64_t)m[k] * p29(0);
        acc >>= 29;
    }
#pragma unroll
    for (int k = 9; k < 17; ++k) {
        uint64_t acc2 = 0;
#pragma unroll
        for (int i = k - 8; i <= 8; ++i) acc += (uint64_t)a.v[i] * b.v[k - i];
#pragma unroll
        for (int i = k - 8; i <= 8; ++i) acc2 += (uint64_t)m[i] * p29(k - i);
        acc += acc2;
        r.v[k - 9] = (uint32_t)acc & MASK29;
        acc >>= 29;
    }
    r.v[8] = (uint32_t)acc;
    return r;
}
__device__ __forceinline__ F29 sqr29(const F29& a) {
    uint64_t acc = 0;
    uint32_t m[9], d[9];
    F29 r;
#pragma unroll
    for (int i = 0; i < 9; ++i) d[i] = a.v[i] << 1;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
#pragma unroll
        for (int i = 0; 2 * i < k; ++i) acc += (uint64_t)d[i] * a.v[k - i];
        if (k % 2 == 0) acc += (uint64_t)a.v[k / 2] * a.v[k / 2];
#pragma unroll
        for (int i = 0; i < k; ++i) acc += (uint64_t)m[i] * p29(k - i);
        m[k] = ((uint32_t)acc * INV29) & MASK29;
        acc += (uint64_t)m[k] * p29(0);
        acc >>= 29;
    }
#pragma unroll
    for (int k = 9; k < 17; ++k) {
#pragma unroll
        for (int i = k - 8; 2 * i < k; ++i) acc += (uint64_t)d[i] * a.v[k - i];
        if (k % 2 == 0) acc += (uint64_t)a.v[k / 2] * a.v[k / 2];
#pragma unroll
        for (int i = k - 8; i <= 8; ++i) acc += (uint64_t)m[i] * p29(k - i);
        r.v[k - 9] = (uint32_t)acc & MASK29;
        acc >>= 29;
    }
    r.v[8] = (uint32_t)acc;
    return r;
}
// a + b, inputs < 2p, output < 2p (one conditional subtraction of 2p), limbs normalised
__device__ __forceinline__ F29 add29(const F29& a, const F29& b) {
    // t = a + b - 2p with signed carries; if negative take a + b
    int32_t u[9]; uint32_t t[9];
    int32_t cu = 0; uint32_t ct = 0;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        uint32_t s = a.v[i] + b.v[i] + ct;
        int32_t w = (int32_t)(a.v[i] + b.v[i]) - (int32_t)(2 * p29(i)) + cu;   // 2*p limb < 2^30
        if (i < 8) { t[i] = s & MASK29; ct = s >> 29; u[i] = w & (int32_t)MASK29; cu = w >> 29; }
        else { t[i] = s; u[i] = w; }
    }
    F29 r;
    bool neg = u[8] < 0;
#pragma unroll
    for (int i = 0; i < 9; ++i) r.v[i] = neg ? t[i] : (uint32_t)u[i];
    return r;
}

__global__ void k29_to_mont(const uint32_t* raw, F29* out) {  // raw: 9 limbs canonical
    const F29 R2 = {{0x59bac10u, 0xd1503a3u, 0x18016b8u, 0x10ab0ca8u, 0x2632639u, 0x2c0169fu, 0x169bfd53u, 0x11869d4cu, 0x2a11a6u}};
    size_t i = threadIdx.x + (size_t)blockIdx.x * blockDim.x;
    F29 x; for (int j = 0; j < 9; ++j) x.v[j] = raw[i * 9 + j];
    out[i] = mul29(x, R2);
}
__global__ void k29_from_mont(const F29* in, uint32_t* raw) {
    size_t i = threadIdx.x + (size_t)blockIdx.x * blockDim.x;
    F29 one = {{1, 0, 0, 0, 0, 0, 0, 0, 0}};
    F29 x = mul29(in[i], one);  // < 2p
    for (int j = 0; j < 9; ++j) raw[i * 9 + j] = x.v[j];
}
__global__ void k29_chain(F29* io, int iters) {
    size_t i = threadIdx.x + (size_t)blockIdx.x * blockDim.x;
    F29 a = io[i], b = a;
    for (int k = 0; k < iters; ++k) a = mul29(a, b);
    io[i] = a;
}
__global__ void k29_chain_2acc(F29* io, int iters) {
    size_t i = threadIdx.x + (size_t)blockIdx.x * blockDim.x;
    F29 a = io[i], b = a;
    for (int k = 0; k < iters; ++k) a = mul29_2acc(a, b);
    io[i] = a;
}
__global__ void k29_chain_sqr(F29* io, int iters) {
    size_t i = threadIdx.x + (size_t)blockIdx.x * blockDim.x;
    F29 a = io[i];
    for (int k = 0; k < iters; ++k) a = sqr29(a);
    io[i] = a;
}
__global__ void k29_chain_add(F29* io, int iters) {
    size_t i = threadIdx.x + (size_t)blockIdx.x * blockDim.x;
    F29 a = io[i], b = a;
    for (int k = 0; k < iters; ++k) { a = add29(a, b); b = add29(b, a); }
    io[i] = a;
}
__global__ void k29_chain4(F29* io, int iters) {
    size_t i = threadIdx.x + (size_t)blockIdx.x * blockDim.x;
    F29 a = io[i], b = a, c = add29(a, a), d = add29(c, a), e = add29(d, a);
    for (int k = 0; k < iters; ++k) { a = mul29(a, b); c = mul29(c, b); d = mul29(d, b); e = mul29(e, b); }
    io[i] = add29(add29(a, c), add29(d, e));
}
__global__ void k32_chain(Fq* io, int iters) {
    size_t i = threadIdx.x + (size_t)blockIdx.x * blockDim.x;
    Fq a = io[i], b = a;
    for (int k = 0; k < iters; ++k) a = Fq::mul_inl(a, b);
    io[i] = a;
}
__global__ void k32_chain_add(Fq* io, int iters) {
    size_t i = threadIdx.x + (size_t)blockIdx.x * blockDim.x;
    Fq a = io[i], b = a;
    for (int k = 0; k < iters; ++k) { a = a + b; b = b + a; }
    io[i] = a;
}

template <class K, class T> float run(K kern, T* d, int blocks, int threads, int iters) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0); hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, d, iters); hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}
int main() {
    const int n = 4096 * 256;
    std::vector<uint32_t> raw((size_t)n * 9, 0);
    for (int i = 0; i < n; ++i) raw[(size_t)i * 9] = 12345 + i;
    uint32_t* d_raw; F29* d29; Fq* d32;
    hipMalloc(&d_raw, raw.size() * 4); hipMalloc(&d29, (size_t)n * sizeof(F29)); hipMalloc(&d32, (size_t)n * sizeof(Fq));
    auto reset = [&]() {
        hipMemcpy(d_raw, raw.data(), raw.size() * 4, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k29_to_mont, dim3(n / 256), dim3(256), 0, 0, d_raw, d29);
        std::vector<Fq> h(n); for (int i = 0; i < n; ++i) h[i] = Fq::from_u32(12345 + i);
        hipMemcpy(d32, h.data(), (size_t)n * sizeof(Fq), hipMemcpyHostToDevice);
        hipDeviceSynchronize();
    };
    // correctness: x^(iters+1) for x = 12345, iters = 4 and 2000, both representations
    for (int it : {4, 2000}) {
        reset();
        run(k29_chain, d29, 1, 64, it);
        hipLaunchKernelGGL(k29_from_mont, dim3(1), dim3(64), 0, 0, d29, d_raw);
        uint32_t out[9]; hipMemcpy(out, d_raw, 36, hipMemcpyDeviceToHost);
        // print as a big hex number (value may be in [p, 2p): printed as is)
        unsigned __int128 lo = 0, hi = 0;  // 261 bits: print limbs instead
        printf("iters=%d limb29 result limbs (lsb first):", it);
        for (int j = 0; j < 9; ++j) printf(" %x", out[j]);
        printf("\n");
        run(k32_chain, d32, 1, 64, it);
        Fq h; hipMemcpy(&h, d32, sizeof(Fq), hipMemcpyDeviceToHost);
        uint32_t r32[8]; h.to_raw(r32);
        printf("iters=%d library result words (lsb first):", it);
        for (int j = 0; j < 8; ++j) printf(" %08x", r32[j]);
        printf("\n");
        (void)lo; (void)hi;
    }
    const int it = 2000;
    reset();
    printf("one wave, dependent chain (ns per op):\n");
    printf("  library Fq::mul_inl %8.1f\n", run(k32_chain, d32, 1, 64, it) * 1e6 / it);
    printf("  mul 9x29 %8.1f\n", run(k29_chain, d29, 1, 64, it) * 1e6 / it);
    printf("  mul 9x29 2 accumulators %8.1f\n", run(k29_chain_2acc, d29, 1, 64, it) * 1e6 / it);
    printf("  sqr 9x29 %8.1f\n", run(k29_chain_sqr, d29, 1, 64, it) * 1e6 / it);
    printf("  mul 9x29 x4 %8.1f (per mul)\n", run(k29_chain4, d29, 1, 64, it) * 1e6 / it / 4);
    printf("  library Fq add %8.1f\n", run(k32_chain_add, d32, 1, 64, it) * 1e6 / it / 2);
    printf("  add 9x29 %8.1f\n", run(k29_chain_add, d29, 1, 64, it) * 1e6 / it / 2);
    printf("low occupancy (1024 blocks x 64 threads = 1 wave per SIMD; 2048 = 2), G mul/s:\n");
    for (int blocks : {1024, 2048, 3072}) {
        reset();
        float ms1 = run(k29_chain, d29, blocks, 64, it);
        reset();
        float ms2 = run(k29_chain_2acc, d29, blocks, 64, it);
        printf("  blocks=%5d  1 acc %8.2f   2 acc %8.2f\n", blocks, (double)blocks * 64 * it / ms1 / 1e6, (double)blocks * 64 * it / ms2 / 1e6);
    }
    printf("full chip, G op/s:\n");
    for (int blocks : {1024, 4096}) {
        reset();
        { float ms2 = run(k29_chain_2acc, d29, blocks, 256, it); printf("  mul 9x29 2acc blocks=%5d  %8.2f G/s\n", blocks, (double)blocks * 256 * it / ms2 / 1e6); reset(); }
        float ms = run(k32_chain, d32, blocks, 256, it);
        printf("  library mul blocks=%5d  %8.2f G/s\n", blocks, (double)blocks * 256 * it / ms / 1e6);
        ms = run(k29_chain, d29, blocks, 256, it);
        printf("  mul 9x29 blocks=%5d  %8.2f G/s\n", blocks, (double)blocks * 256 * it / ms / 1e6);
        ms = run(k29_chain_sqr, d29, blocks, 256, it);
        printf("  sqr 9x29 blocks=%5d  %8.2f G/s\n", blocks, (double)blocks * 256 * it / ms / 1e6);
        ms = run(k32_chain_add, d32, blocks, 256, it);
        printf("  library add blocks=%5d  %8.2f G/s\n", blocks, (double)blocks * 256 * it * 2 / ms / 1e6);
        ms = run(k29_chain_add, d29, blocks, 256, it);
        printf("  add 9x29 blocks=%5d  %8.2f G/s\n", blocks, (double)blocks * 256 * it * 2 / ms / 1e6);
    }
    return 0;
}
