// Would batched-affine bucket additions make msm_accumulate faster?  (Round-2 review, item 3.)
//
// msm_accumulate gives every lane a chunk of N consecutive entries of the sorted list and adds them, one mixed Jacobian addition
// (g1_madd_fast, ~1770 multiply-adds) per entry.  An AFFINE addition costs 5 products + 1 squaring (~940 multiply-adds) when the
// inverse of x2 - x1 comes from a shared inversion (Montgomery's trick: one more product going up, two coming down) — but an
// inversion is ~15 000 integer instructions (safegcd), and a wave runs in lockstep: 64 lanes inverting 64 different values cost the
// wave exactly what one lane inverting one value costs.  So the saving is decided by how many INDEPENDENT additions ONE LANE has per
// inversion.  Within a bucket the additions of a chunk form a chain; independent ones come from pairing adjacent entries (N/2
// pairs, then N/4, ...).  This program measures the three ingredients at the occupancies the kernel runs at and the two candidate
// loop bodies, with the real field / group code of the library and random 72-byte gathers like the kernel's:
//   jac     N entries per lane, one mixed Jacobian addition each                         (the kernel today)
//   inv     one safegcd inversion per lane
//   pair1   "v1": the entries taken in PAIRS — pass A computes the suffix products of the N/2 denominators (x only, stored to a
//           [k][limb][lane] scratch array), ONE inversion, pass B recovers each inverse, forms P + Q in affine coordinates and adds
//           the sum to the Jacobian accumulator: N/2 affine additions + N/2 mixed additions + 1 inversion per N entries
//   tree    every addition affine: rounds of N/2, N/4, ... pair additions, one inversion per round, sums staged in the scratch array
//           (an upper bound on what more rounds can buy: no bucket bookkeeping, no divergence)
// Build: hipcc -O3 --offload-arch=gfx950 -I halo2_verifier_amd/csrc tools/affine_microbench.hip -o tools/affine_microbench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>
#include "curve.hip.h"
using namespace h2v;
namespace h2v { void set_last_error(const std::string&) {} }

#define TABLE (1u << 16)   // points gathered from: 4.7 MB, like one XCD's share of a launch's points

__device__ __forceinline__ uint32_t rnd_next(uint32_t& s) { s = s * 1664525u + 1013904223u; return (s >> 8) & (TABLE - 1); }

__global__ void __launch_bounds__(64) k_jac(const G1A* __restrict__ tab, G1J* __restrict__ out, int n) {
    const uint32_t lane = blockIdx.x * 64 + threadIdx.x;
    uint32_t s = lane * 2654435761u + 12345u;
    G1J acc = G1J::identity();
    G1A nx = tab[rnd_next(s)];
    for (int i = 0; i < n; ++i) {
        const G1A cur = nx;
        nx = tab[rnd_next(s)];
        g1_madd_fast(acc, cur);
    }
    out[lane] = acc;
}
__global__ void __launch_bounds__(64) k_inv(const G1A* __restrict__ tab, Fq* __restrict__ out, int reps) {
    const uint32_t lane = blockIdx.x * 64 + threadIdx.x;
    Fq v = tab[lane & (TABLE - 1)].x;
    for (int i = 0; i < reps; ++i) v = v.inv();
    out[lane] = v;
}
// the affine sum of two distinct points given 1 / (x2 - x1); all results ordinary representatives below 2p
__device__ __forceinline__ G1A affine_add(const G1A& p, const G1A& q, const Fq& inv_d) {
    const Fq lam = Fq::mul_inl(Fq::lazy_sub(q.y, p.y), inv_d);
    const Fq l2 = lam.sqr_inl();
    int64_t w[9];
#pragma unroll
    for (int l = 0; l < 9; ++l) w[l] = (int64_t)l2.v[l] + (int64_t)Fq::KP29(4, l) - (int64_t)p.x.v[l] - (int64_t)q.x.v[l];
    G1A r;
    r.x = Fq::from_wide(w);
    const Fq t = Fq::mul_inl(lam, Fq::lazy_sub(p.x, r.x));
#pragma unroll
    for (int l = 0; l < 9; ++l) w[l] = (int64_t)t.v[l] + (int64_t)Fq::KP29(2, l) - (int64_t)p.y.v[l];
    r.y = Fq::from_wide(w);
    return r;
}
__device__ __forceinline__ void fq_store(uint32_t* __restrict__ base, uint32_t k, uint32_t lanes, uint32_t lane, const Fq& v) {
#pragma unroll
    for (int l = 0; l < 9; ++l) base[((size_t)k * 9 + l) * lanes + lane] = v.v[l];
}
__device__ __forceinline__ Fq fq_load(const uint32_t* __restrict__ base, uint32_t k, uint32_t lanes, uint32_t lane) {
    Fq v;
#pragma unroll
    for (int l = 0; l < 9; ++l) v.v[l] = base[((size_t)k * 9 + l) * lanes + lane];
    return v;
}
__global__ void __launch_bounds__(64) k_pair1(const G1A* __restrict__ tab, G1J* __restrict__ out, uint32_t* __restrict__ scratch, int n) {
    const uint32_t lane = blockIdx.x * 64 + threadIdx.x, lanes = gridDim.x * 64;
    const int pairs = n / 2;
    uint32_t s = lane * 2654435761u + 12345u;
    // pass A (the kernel would walk its chunk backwards; the cost is the same): suffix products of the denominators, x only
    Fq run = Fq::one();
    for (int k = 0; k < pairs; ++k) {
        const Fq x1 = tab[rnd_next(s)].x, x2 = tab[rnd_next(s)].x;
        fq_store(scratch, (uint32_t)k, lanes, lane, run);
        const Fq t = Fq::mul_inl(run, Fq::lazy_sub(x2, x1));
        if (!t.is_zero()) run = t;      // (a zero denominator — equal or opposite points — leaves the chain: the kernel takes such a pair as two single entries)
    }
    Fq inv = run.inv();
    // pass B: the same entries again (the same index stream), in the order that undoes the products
    s = lane * 2654435761u + 12345u;
    uint32_t idx[2];
    G1J acc = G1J::identity();
    // (the microbenchmark stores suffix products in forward order and consumes them backward: identical work)
    for (int k = pairs - 1; k >= 0; --k) {
        idx[0] = rnd_next(s); idx[1] = rnd_next(s);
        const G1A p = tab[idx[0]], q = tab[idx[1]];
        const Fq d = Fq::lazy_sub(q.x, p.x);
        const Fq inv_d = Fq::mul_inl(inv, fq_load(scratch, (uint32_t)k, lanes, lane));
        inv = Fq::mul_inl(inv, d);
        const G1A r = affine_add(p, q, inv_d);
        g1_madd_fast(acc, r);
    }
    out[lane] = acc;
}
__global__ void __launch_bounds__(64) k_tree(const G1A* __restrict__ tab, G1A* __restrict__ out, uint32_t* __restrict__ scratch, G1A* __restrict__ stage, int n) {
    const uint32_t lane = blockIdx.x * 64 + threadIdx.x, lanes = gridDim.x * 64;
    uint32_t s = lane * 2654435761u + 12345u;
    // round 0 reads the table, later rounds the staged sums ([i][lane], 72 B apart per lane: what a real kernel would have to do as well)
    for (int m = n, round = 0; m > 1; m >>= 1, ++round) {
        const int pairs = m / 2;
        Fq run = Fq::one();
        uint32_t s2 = s;
        for (int k = 0; k < pairs; ++k) {
            const Fq x1 = round ? stage[(size_t)(2 * k) * lanes + lane].x : tab[rnd_next(s2)].x;
            const Fq x2 = round ? stage[(size_t)(2 * k + 1) * lanes + lane].x : tab[rnd_next(s2)].x;
            fq_store(scratch, (uint32_t)k, lanes, lane, run);
            const Fq t = Fq::mul_inl(run, Fq::lazy_sub(x2, x1));
            if (!t.is_zero()) run = t;
        }
        Fq inv = run.inv();
        s2 = s;
        // (consumed backward in a real kernel; here forward over a reversed product — the same operations)
        for (int k = pairs - 1; k >= 0; --k) {
            const G1A p = round ? stage[(size_t)(2 * k) * lanes + lane] : tab[rnd_next(s2)];
            const G1A q = round ? stage[(size_t)(2 * k + 1) * lanes + lane] : tab[rnd_next(s2)];
            const Fq d = Fq::lazy_sub(q.x, p.x);
            const Fq inv_d = Fq::mul_inl(inv, fq_load(scratch, (uint32_t)k, lanes, lane));
            inv = Fq::mul_inl(inv, d);
            stage[(size_t)k * lanes + lane] = affine_add(p, q, inv_d);
        }
    }
    out[lane] = stage[lane];
}

int main() {
    std::vector<G1A> h(TABLE);
    uint32_t s = 7;
    for (auto& p : h) { for (int l = 0; l < 9; ++l) { s = s * 1103515245u + 12345u; p.x.v[l] = (s >> 3) & (l == 8 ? 0x1fffffu : H2V_LIMB_MASK); s = s * 1103515245u + 12345u; p.y.v[l] = (s >> 3) & (l == 8 ? 0x1fffffu : H2V_LIMB_MASK); } }
    G1A* tab; hipMalloc(&tab, sizeof(G1A) * TABLE); hipMemcpy(tab, h.data(), sizeof(G1A) * TABLE, hipMemcpyHostToDevice);
    const uint32_t max_lanes = 1024 * 64 * 4;
    G1J* out; hipMalloc(&out, sizeof(G1J) * max_lanes);
    uint32_t* scratch; hipMalloc(&scratch, (size_t)64 * 36 * max_lanes);
    G1A* stage; hipMalloc(&stage, (size_t)128 * sizeof(G1A) * max_lanes / 2);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto timed = [&](auto launch) { launch(); hipDeviceSynchronize(); hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); return ms; };
    printf("MI355X: 1024 SIMDs.  times in ms; 'per entry' in ns of SIMD time per list entry (= ms / (entries per lane x waves per SIMD))\n");
    for (int wps : {1, 2, 3}) {
        const uint32_t blocks = 1024 * wps;
        const float t_inv = timed([&] { hipLaunchKernelGGL(k_inv, dim3(blocks), dim3(64), 0, 0, tab, (Fq*)out, 4); }) / 4;
        printf("waves per SIMD = %d (%u lanes): one inversion per lane %.4f ms\n", wps, blocks * 64, t_inv);
        for (int n : {16, 32, 48, 64, 96, 128}) {
            if ((size_t)n * blocks * 64 > (size_t)128 * max_lanes / 2) continue;
            const float tj = timed([&] { hipLaunchKernelGGL(k_jac, dim3(blocks), dim3(64), 0, 0, tab, out, n); });
            const float tp = timed([&] { hipLaunchKernelGGL(k_pair1, dim3(blocks), dim3(64), 0, 0, tab, out, scratch, n); });
            const float tt = (n & (n - 1)) == 0 ? timed([&] { hipLaunchKernelGGL(k_tree, dim3(blocks), dim3(64), 0, 0, tab, (G1A*)out, scratch, stage, n); }) : 0.f;
            printf("  entries per lane %3d:  jac %.4f  pair1 %.4f (x%.2f)  tree %.4f (x%.2f)   per entry: jac %.1f ns, pair1 %.1f ns\n", n, tj, tp, tj / tp, tt, tt > 0 ? tj / tt : 0.f,
                   tj * 1e6 / (n * wps), tp * 1e6 / (n * wps));
        }
    }
    return 0;
}
