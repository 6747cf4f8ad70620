// Micro-benchmark of the wave-parallel Fq12 product as csrc/pairing.hip had it in round 1 (36 Fq2::mul lanes + 6 fold lanes); the
// kernels now run a one-phase product (eight lanes per output, one reduction per output: DESIGN.md §4).  Kept for the record.
// Build: hipcc -O3 --offload-arch=gfx950 -I halo2_verifier_amd/csrc tools/wmul_microbench.hip -o tools/wmul_microbench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <string>
#include "pairing.hip.h"
using namespace h2v;
namespace h2v { void set_last_error(const std::string&) {} }

struct Sh { Fq2 prod[36]; Fq2 f[6], g[6]; };

__device__ __forceinline__ void wmul_a(Sh& s, Fq2* dst, const Fq2* x, const Fq2* y, uint32_t lane) {  // as in round 1's pairing.hip
    if (lane < 36) s.prod[lane] = Fq2::mul(x[lane / 6], y[lane % 6]);
    __syncthreads();
    if (lane < 6) {
        Fq2 lo = Fq2::zero(), hi = Fq2::zero();
        for (uint32_t i = 0; i < 6; ++i) {
            if (i <= lane) lo = lo + s.prod[i * 6 + (lane - i)];
            else hi = hi + s.prod[i * 6 + (lane + 6 - i)];
        }
        dst[lane] = lo + hi.mul_xi();
    }
    __syncthreads();
}
__device__ __forceinline__ void wmul_onlymul(Sh& s, Fq2* dst, const Fq2* x, const Fq2* y, uint32_t lane) {
    if (lane < 36) s.prod[lane] = Fq2::mul(x[lane / 6], y[lane % 6]);
    __syncthreads();
    if (lane < 6) dst[lane] = s.prod[lane * 6];
    __syncthreads();
}
// reduction spread over 12 lanes (lo and hi halves separately) with a uniform 6-term loop
__device__ __forceinline__ void wmul_b(Sh& s, Fq2* dst, const Fq2* x, const Fq2* y, uint32_t lane) {
    if (lane < 36) s.prod[lane] = Fq2::mul(x[lane / 6], y[lane % 6]);
    __syncthreads();
    if (lane < 6) {
        Fq2 lo = Fq2::zero(), hi = Fq2::zero();
#pragma unroll
        for (uint32_t i = 0; i < 6; ++i) {
            uint32_t j = (lane + 6 - i) % 6;
            Fq2 v = s.prod[i * 6 + j];
            bool is_lo = i <= lane;
            Fq2 z = Fq2::zero();
            lo = lo + (is_lo ? v : z);
            hi = hi + (is_lo ? z : v);
        }
        dst[lane] = lo + hi.mul_xi();
    }
    __syncthreads();
}

// the product as csrc/pairing.hip does it now: two sum-of-two-products passes per lane, integer fold on 12 lanes
__device__ __noinline__ Fq2 fq2_mul_dot(Fq2 a, Fq2 b) {
    const Fq nb1 = b.c1.neg();
    return {Fq::dot2_inl(a.c0, b.c0, a.c1, nb1), Fq::dot2_inl(a.c0, b.c1, a.c1, b.c0)};
}
__device__ __noinline__ Fq wfold_coord(const Fq2* __restrict__ prod, uint32_t k, uint32_t c) {
    // 32-bit per-limb sums first (six residues of 29-bit limbs cannot overflow): SL = the lo terms' coordinate, SH = the hi
    // terms' coordinate, OH = the hi terms' other coordinate
    uint32_t SL[9], SH[9], OH[9];
#pragma unroll
    for (int l = 0; l < 9; ++l) { SL[l] = 0; SH[l] = 0; OH[l] = 0; }
#pragma unroll
    for (uint32_t i = 0; i < 6; ++i) {
        const bool lo = i <= k;
        const Fq2& P = prod[i * 6 + (lo ? k - i : k + 6 - i)];
        const Fq& same = c ? P.c1 : P.c0;
        const Fq& other = c ? P.c0 : P.c1;
#pragma unroll
        for (int l = 0; l < 9; ++l) {
            const uint32_t x = same.v[l], y = other.v[l];
            SL[l] += lo ? x : 0u;
            SH[l] += lo ? 0u : x;
            OH[l] += lo ? 0u : y;
        }
    }
    // SL + 9 SH + OH (imaginary) or SL + 9 SH + (n_hi * 2p - OH) (real), carried once
    const int64_t n_hi = 5 - (int64_t)k;
    Fq v;
    int64_t carry = 0;
#pragma unroll
    for (int l = 0; l < 9; ++l) {
        int64_t t = (int64_t)SL[l] + (((int64_t)SH[l]) << 3) + (int64_t)SH[l] + carry;
        t += c ? (int64_t)OH[l] : n_hi * (int64_t)(2u * FqParams::P29(l)) - (int64_t)OH[l];
        if (l < 8) { v.v[l] = (uint32_t)(t & (int64_t)H2V_LIMB_MASK); carry = t >> 29; } else v.v[l] = (uint32_t)t;   // total < 64p < 2^261
    }
    return Fq::mul_inl(v, Fq::one());
}
__device__ __forceinline__ void wmul_c(Sh& s, Fq2* dst, const Fq2* x, const Fq2* y, uint32_t lane) {
    if (lane < 36) s.prod[lane] = fq2_mul_dot(x[lane / 6], y[lane % 6]);
    __syncthreads();
    if (lane < 12) {
        const Fq r = wfold_coord(s.prod, lane >> 1, lane & 1);
        if (lane & 1) dst[lane >> 1].c1 = r; else dst[lane >> 1].c0 = r;
    }
    __syncthreads();
}
__device__ __forceinline__ void wmul_c_onlymul(Sh& s, Fq2* dst, const Fq2* x, const Fq2* y, uint32_t lane) {
    if (lane < 36) s.prod[lane] = fq2_mul_dot(x[lane / 6], y[lane % 6]);
    __syncthreads();
    if (lane < 6) dst[lane] = s.prod[lane * 6];
    __syncthreads();
}

template <int V> __global__ void __launch_bounds__(64) k(Fq* io, int iters) {
    __shared__ Sh s;
    uint32_t lane = threadIdx.x;
    if (lane < 6) { Fq a = io[lane]; s.f[lane] = {a, a + a}; s.g[lane] = {a + a + a, a}; }
    __syncthreads();
    for (int i = 0; i < iters; ++i) {
        if (V == 0) wmul_a(s, s.f, s.f, s.g, lane);
        if (V == 1) wmul_onlymul(s, s.f, s.f, s.g, lane);
        if (V == 2) wmul_b(s, s.f, s.f, s.g, lane);
        if (V == 3) wmul_c(s, s.f, s.f, s.g, lane);
        if (V == 4) wmul_c_onlymul(s, s.f, s.f, s.g, lane);
    }
    if (lane < 6) io[64 + lane] = s.f[lane].c0 + s.f[lane].c1;
}
template <int V> float run(Fq* d, int iters) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<V>, dim3(1), dim3(64), 0, 0, d, 2); hipDeviceSynchronize();
    hipEventRecord(e0); hipLaunchKernelGGL(k<V>, dim3(1), dim3(64), 0, 0, d, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}
int main() {
    Fq h[128]; for (int i = 0; i < 128; ++i) h[i] = Fq::from_u32(1000 + i);
    Fq* d; hipMalloc(&d, sizeof(h)); hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    const int it = 500;
    printf("wave product, us per op: Karatsuba + modular fold %.2f | its products only %.2f | uniform 6-term fold %.2f\n", run<0>(d, it) * 1e3 / it, run<1>(d, it) * 1e3 / it, run<2>(d, it) * 1e3 / it);
    printf("                         dot2 + integer fold (pairing.hip) %.2f | its products only %.2f\n", run<3>(d, it) * 1e3 / it, run<4>(d, it) * 1e3 / it);
    return 0;
}
