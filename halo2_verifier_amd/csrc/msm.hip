// Pooled BN254 G1 multi-scalar multiplication for gfx950.
//
// Replaces MSMKZG::eval -> best_multiexp -> multiexp_serial (poly/kzg/msm.rs:81-86,
// arithmetic.rs:7-108), which is a serial fixed-window Pippenger (c in {1,3,4}, 256/c+1
// windows).  The result is the same group element; the schedule is chosen for the GPU:
//
//   1. msm_count    one lane per term: extract every window's digit, histogram (window,bucket)
//   2. msm_scan     exclusive prefix sum of the histogram (one workgroup)
//   3. msm_scatter  one lane per term: counting-sort term indices into per-bucket lists
//   4. msm_bucket   one lane per (window,bucket): mixed Jacobian+affine additions over its list
//   5. msm_window   one workgroup per window: sum_b (b+1) * bucket[b] by per-lane running sums
//                   over a slice of buckets, then a tree reduction through LDS
//   6. msm_final    Horner over windows (c doublings + one add per window)
//
// Steps 1-3 are a hand-written counting sort (no atomics on points, no library sort); the only
// atomics are 32-bit counters.  The order of additions inside a bucket depends on atomic
// arrival order, but the group element — hence the affine bytes — does not.
//
// Arithmetic intensity: one term = 96 B read (32 B scalar + 64 B affine base) and `windows`
// mixed additions (~11 Fq multiplications each, ~128 v_mad_u64_u32 per multiplication), i.e.
// O(10^4-10^5) integer ops per 96 bytes: the kernel is bound by 32-bit integer multiply issue,
// not by HBM (DESIGN.md "Roofline").
#include "../../include/h2v.h"
#include "batch.h"

namespace h2v {

MsmPlan msm_plan(uint32_t n) {
    MsmPlan best{n, 1, 254, 1};
    double best_cost = 1e300;
    for (uint32_t c = 2; c <= 14; ++c) {
        uint32_t w = (254 + c - 1) / c;
        uint32_t b = (1u << c) - 1;
        double cost = (double)w * ((double)n + 2.0 * b);
        if (cost < best_cost) { best_cost = cost; best = MsmPlan{n, c, w, b}; }
    }
    return best;
}

int MsmWorkspace::alloc(uint32_t max_terms) {
    release();
    cap_terms = max_terms;
    // worst case over all n <= max_terms of windows*buckets and n*windows
    size_t mb = 0, ml = 0, mw = 0;
    for (uint32_t n = 1; n <= max_terms; n = n < 16 ? n + 1 : n + n / 8) {
        MsmPlan p = msm_plan(n);
        mb = std::max(mb, (size_t)p.windows * p.buckets);
        mw = std::max(mw, (size_t)p.windows);
    }
    {
        MsmPlan p = msm_plan(max_terms);
        mb = std::max(mb, (size_t)p.windows * p.buckets);
        mw = std::max(mw, (size_t)p.windows);
    }
    ml = (size_t)max_terms * 127;  // c >= 2  =>  windows <= 127
    cap_buckets = mb; cap_list = ml;
    H2V_HIP_CHECK(hipMalloc(&counts, mb * 4));
    H2V_HIP_CHECK(hipMalloc(&offsets, mb * 4));
    H2V_HIP_CHECK(hipMalloc(&cursor, mb * 4));
    H2V_HIP_CHECK(hipMalloc(&list, ml * 4));
    H2V_HIP_CHECK(hipMalloc(&bucket_pts, mb * sizeof(G1J)));
    H2V_HIP_CHECK(hipMalloc(&window_sums, 128 * sizeof(G1J)));
    (void)mw;
    return 0;
}
void MsmWorkspace::release() {
    if (counts) hipFree(counts);
    if (offsets) hipFree(offsets);
    if (cursor) hipFree(cursor);
    if (list) hipFree(list);
    if (bucket_pts) hipFree(bucket_pts);
    if (window_sums) hipFree(window_sums);
    counts = offsets = cursor = list = nullptr; bucket_pts = window_sums = nullptr;
    cap_terms = 0;
}

__device__ __forceinline__ uint32_t msm_digit(const uint32_t* __restrict__ s, uint32_t w, uint32_t c) {
    uint32_t off = w * c, word = off >> 5, sh = off & 31;
    uint32_t v = s[word] >> sh;
    if (sh + c > 32 && word + 1 < 8) v |= s[word + 1] << (32 - sh);
    return v & ((1u << c) - 1);
}

template <bool SCATTER>
__global__ void __launch_bounds__(256) msm_count_or_scatter(const uint32_t* __restrict__ scalars, uint32_t sstride, const G1A* __restrict__ bases, uint32_t bstride, uint32_t n, MsmPlan p,
                                                            uint32_t* __restrict__ counts, const uint32_t* __restrict__ offsets,
                                                            uint32_t* __restrict__ cursor, uint32_t* __restrict__ list) {
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const uint32_t* s = scalars + (size_t)sstride * t;  // digits are read straight from L1/L2: no runtime-indexed register array
    // identity bases contribute nothing
    const uint32_t* bw = reinterpret_cast<const uint32_t*>(bases + (size_t)bstride * t);
    uint32_t any = 0;
    for (int i = 0; i < 16; ++i) any |= bw[i];
    if (!any) return;
    for (uint32_t w = 0; w < p.windows; ++w) {
        uint32_t d = msm_digit(s, w, p.c);
        if (!d) continue;
        uint32_t b = w * p.buckets + d - 1;
        if (SCATTER) {
            uint32_t pos = atomicAdd(&cursor[b], 1u);
            list[offsets[b] + pos] = t;
        } else {
            atomicAdd(&counts[b], 1u);
        }
    }
}

// exclusive scan of counts[0..nb) -> offsets; zeroes cursor.  One workgroup of 1024 lanes.
__global__ void __launch_bounds__(1024) msm_scan(const uint32_t* __restrict__ counts, uint32_t* __restrict__ offsets, uint32_t* __restrict__ cursor, uint32_t nb) {
    __shared__ uint32_t part[1024];
    uint32_t t = threadIdx.x;
    uint32_t chunk = (nb + 1023) / 1024;
    uint32_t lo = t * chunk, hi = min(nb, lo + chunk);
    uint32_t sum = 0;
    for (uint32_t i = lo; i < hi; ++i) sum += counts[i];
    part[t] = sum;
    __syncthreads();
    for (uint32_t d = 1; d < 1024; d <<= 1) {
        uint32_t v = t >= d ? part[t - d] : 0;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    uint32_t run = part[t] - sum;
    for (uint32_t i = lo; i < hi; ++i) { offsets[i] = run; run += counts[i]; cursor[i] = 0; }
}

__global__ void __launch_bounds__(64) msm_bucket(const G1A* __restrict__ bases, uint32_t bstride, const uint32_t* __restrict__ counts, const uint32_t* __restrict__ offsets,
                                                 const uint32_t* __restrict__ list, G1J* __restrict__ bucket_pts, uint32_t nb) {
    uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nb) return;
    uint32_t cnt = counts[b], off = offsets[b];
    G1J acc = G1J::identity();
    for (uint32_t i = 0; i < cnt; ++i) {
        G1A q = bases[(size_t)list[off + i] * bstride];
        acc = g1_add_affine(acc, q);
    }
    bucket_pts[b] = acc;
}

#define MSM_WIN_THREADS 128
__global__ void __launch_bounds__(MSM_WIN_THREADS) msm_window(const G1J* __restrict__ bucket_pts, G1J* __restrict__ window_sums, MsmPlan p) {
    __shared__ G1J red[MSM_WIN_THREADS];
    uint32_t w = blockIdx.x, t = threadIdx.x;
    uint32_t slice = (p.buckets + MSM_WIN_THREADS - 1) / MSM_WIN_THREADS;
    uint32_t lo = t * slice, hi = min(p.buckets, lo + slice);
    G1J run = G1J::identity(), sum = G1J::identity();
    const G1J* bp = bucket_pts + (size_t)w * p.buckets;
    for (uint32_t b = hi; b > lo; --b) {
        run = g1_add(run, bp[b - 1]);
        sum = g1_add(sum, run);
    }
    // sum = sum_{b in slice} (b - lo + 1) B_b ; the bucket's weight is (b + 1)
    if (lo < hi && lo > 0) {
        G1J scaled = G1J::identity();
        for (int i = (int)p.c - 1; i >= 0; --i) {
            scaled = g1_dbl(scaled);
            if ((lo >> i) & 1) scaled = g1_add(scaled, run);
        }
        sum = g1_add(sum, scaled);
    }
    red[t] = sum;
    __syncthreads();
    for (uint32_t d = MSM_WIN_THREADS / 2; d > 0; d >>= 1) {
        if (t < d) red[t] = g1_add(red[t], red[t + d]);
        __syncthreads();
    }
    if (t == 0) window_sums[w] = red[0];
}

__global__ void msm_final(const G1J* __restrict__ window_sums, G1J* __restrict__ out, MsmPlan p) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    G1J acc = G1J::identity();
    for (int w = (int)p.windows - 1; w >= 0; --w) {
        for (uint32_t i = 0; i < p.c; ++i) acc = g1_dbl(acc);
        acc = g1_add(acc, window_sums[w]);
    }
    *out = acc;
}

__global__ void msm_set_identity(G1J* out) {
    if (threadIdx.x == 0 && blockIdx.x == 0) *out = G1J::identity();
}

int msm_enqueue(hipStream_t s, MsmWorkspace& ws, const uint32_t* d_scalars, const G1A* d_bases, uint32_t n, G1J* d_out) {
    return msm_enqueue_strided(s, ws, d_scalars, 8, d_bases, 1, n, d_out);
}

int msm_enqueue_strided(hipStream_t s, MsmWorkspace& ws, const uint32_t* d_scalars, uint32_t sstride, const G1A* d_bases, uint32_t bstride, uint32_t n, G1J* d_out) {
    if (n == 0) {
        hipLaunchKernelGGL(msm_set_identity, dim3(1), dim3(64), 0, s, d_out);
        return 0;
    }
    if (n > ws.cap_terms) { set_last_error("msm_enqueue: n exceeds workspace capacity"); return H2V_ERR_BAD_ARGUMENT; }
    MsmPlan p = msm_plan(n);
    uint32_t nb = p.windows * p.buckets;
    if (nb > ws.cap_buckets || (size_t)n * p.windows > ws.cap_list) { set_last_error("msm_enqueue: workspace too small"); return H2V_ERR_BAD_ARGUMENT; }
    H2V_HIP_CHECK(hipMemsetAsync(ws.counts, 0, (size_t)nb * 4, s));
    uint32_t gt = (n + 255) / 256;
    hipLaunchKernelGGL(msm_count_or_scatter<false>, dim3(gt), dim3(256), 0, s, d_scalars, sstride, d_bases, bstride, n, p, ws.counts, ws.offsets, ws.cursor, ws.list);
    hipLaunchKernelGGL(msm_scan, dim3(1), dim3(1024), 0, s, ws.counts, ws.offsets, ws.cursor, nb);
    hipLaunchKernelGGL(msm_count_or_scatter<true>, dim3(gt), dim3(256), 0, s, d_scalars, sstride, d_bases, bstride, n, p, ws.counts, ws.offsets, ws.cursor, ws.list);
    hipLaunchKernelGGL(msm_bucket, dim3((nb + 63) / 64), dim3(64), 0, s, d_bases, bstride, ws.counts, ws.offsets, ws.list, ws.bucket_pts, nb);
    hipLaunchKernelGGL(msm_window, dim3(p.windows), dim3(MSM_WIN_THREADS), 0, s, ws.bucket_pts, ws.window_sums, p);
    hipLaunchKernelGGL(msm_final, dim3(1), dim3(64), 0, s, ws.window_sums, d_out, p);
    H2V_HIP_CHECK(hipGetLastError());
    return 0;
}

}  // namespace h2v
