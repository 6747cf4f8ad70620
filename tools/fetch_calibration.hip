// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on this GPU for the two access patterns of the MSM stage, with known byte
// counts (MI355X_MICROARCH.md, HBM: "calibrate on a known byte count in your own access pattern before trusting an absolute"):
//   k_stream_read    64 MiB read once, 16 B per lane, coalesced                      (the guide's case: FETCH_SIZE reads 1/2)
//   k_gather72       4 Mi random 72-byte records out of a 16 MiB table, one per lane   (msm_accumulate's base gathers)
//   k_scatter4       4 Mi 4-byte stores at random positions of a 64 MiB array         (the counting sort's list stores)
//   k_store108       1 Mi 108-byte records, one per lane, at 128-byte stride          (bucket / piece stores)
// Round 3 (the review: "the guide's x2 is for wide coalesced streaming reads; apply it only to kernels whose fetches are streaming"):
//   k_stream4        64 MiB read once, 4 B per lane, coalesced                       (msm_accumulate / msm_fixup reading the sorted list)
//   k_gather72_big   4 Mi random 72-byte records out of a 128 MiB table (far beyond every cache)
//   k_gather128      2 Mi random 128-byte aligned records out of a 128 MiB table      (msm_window / msm_fixup reading bucket slots)
//   k_store128       1 Mi 128-byte aligned records stored at RANDOM slots of a 128 MiB array (bucket / piece slots, G1JSlot)
// Run under:  rocprofv3 --pmc FETCH_SIZE -- ./fetch_calibration   and   rocprofv3 --pmc WRITE_SIZE -- ./fetch_calibration
// Build: hipcc -O3 --offload-arch=gfx950 tools/fetch_calibration.hip -o tools/fetch_calibration
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__global__ void k_stream_read(const uint4* __restrict__ in, uint4* __restrict__ out, size_t n) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    uint4 acc = make_uint4(0, 0, 0, 0);
    for (; i < n; i += (size_t)gridDim.x * blockDim.x) { uint4 v = in[i]; acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w; }
    if (acc.x == 0x12345678u) out[0] = acc;   // keep the loads alive, (almost) never store
}
struct Rec72 { uint32_t w[18]; };
__global__ void k_gather72(const Rec72* __restrict__ table, uint32_t table_len, uint32_t* __restrict__ out, uint32_t n) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t idx = (i * 2654435761u) % table_len;   // a fixed pseudo-random permutation-like walk
    Rec72 r = table[idx];
    uint32_t x = 0;
    for (int k = 0; k < 18; ++k) x ^= r.w[k];
    if (x == 0x12345678u) out[0] = x;
}
__global__ void k_scatter4(uint32_t* __restrict__ arr, uint32_t len, uint32_t n) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    arr[(uint32_t)(((uint64_t)i * 2654435761ull) % len)] = i;
}
struct alignas(128) Rec108 { uint32_t w[27]; };
__global__ void k_store108(Rec108* __restrict__ arr, uint32_t n) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Rec108 r;
    for (int k = 0; k < 27; ++k) r.w[k] = i + k;
    arr[i] = r;
}
__global__ void k_stream4(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, size_t n) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    uint32_t acc = 0;
    for (; i < n; i += (size_t)gridDim.x * blockDim.x) acc ^= in[i];
    if (acc == 0x12345678u) out[0] = acc;
}
__global__ void k_gather72_big(const Rec72* __restrict__ table, uint32_t table_len, uint32_t* __restrict__ out, uint32_t n) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t idx = (uint32_t)(((uint64_t)i * 2654435761ull + 12345ull) % table_len);
    Rec72 r = table[idx];
    uint32_t x = 0;
    for (int k = 0; k < 18; ++k) x ^= r.w[k];
    if (x == 0x12345678u) out[0] = x;
}
struct alignas(128) Rec128 { uint32_t w[32]; };
__global__ void k_gather128(const Rec128* __restrict__ table, uint32_t table_len, uint32_t* __restrict__ out, uint32_t n) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t idx = (uint32_t)(((uint64_t)i * 2654435761ull + 777ull) % table_len);
    Rec128 r = table[idx];
    uint32_t x = 0;
    for (int k = 0; k < 32; ++k) x ^= r.w[k];
    if (x == 0x12345678u) out[0] = x;
}
__global__ void k_store128(Rec128* __restrict__ arr, uint32_t len, uint32_t n) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Rec128 r;
    for (int k = 0; k < 32; ++k) r.w[k] = i + k;
    arr[(uint32_t)(((uint64_t)i * 2654435761ull) % len)] = r;
}
int main() {
    const size_t stream_bytes = 64u << 20, table_bytes = 16u << 20, arr_bytes = 64u << 20;
    const uint32_t n_gather = 4u << 20, n_scatter = 4u << 20, n_store = 1u << 20;
    uint4 *d_in, *d_out; Rec72* d_table; uint32_t *d_o32, *d_arr; Rec108* d_rec;
    hipMalloc(&d_in, stream_bytes); hipMalloc(&d_out, 64); hipMalloc(&d_table, table_bytes); hipMalloc(&d_o32, 64);
    hipMalloc(&d_arr, arr_bytes); hipMalloc(&d_rec, (size_t)n_store * sizeof(Rec108));
    hipMemset(d_in, 1, stream_bytes); hipMemset(d_table, 2, table_bytes); hipMemset(d_arr, 0, arr_bytes);
    const size_t big_bytes = 128u << 20;
    const uint32_t n_g128 = 2u << 20, n_s128 = 1u << 20;
    void* d_big; hipMalloc(&d_big, big_bytes); hipMemset(d_big, 5, big_bytes);
    hipDeviceSynchronize();
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(k_stream4, dim3(4096), dim3(256), 0, 0, (const uint32_t*)d_in, d_o32, stream_bytes / 4);
        hipLaunchKernelGGL(k_gather72_big, dim3(n_gather / 256), dim3(256), 0, 0, (const Rec72*)d_big, (uint32_t)(big_bytes / sizeof(Rec72)), d_o32, n_gather);
        hipLaunchKernelGGL(k_gather128, dim3(n_g128 / 256), dim3(256), 0, 0, (const Rec128*)d_big, (uint32_t)(big_bytes / sizeof(Rec128)), d_o32, n_g128);
        hipLaunchKernelGGL(k_store128, dim3(n_s128 / 256), dim3(256), 0, 0, (Rec128*)d_big, (uint32_t)(big_bytes / sizeof(Rec128)), n_s128);
        hipLaunchKernelGGL(k_stream_read, dim3(4096), dim3(256), 0, 0, d_in, d_out, stream_bytes / 16);
        hipLaunchKernelGGL(k_gather72, dim3(n_gather / 256), dim3(256), 0, 0, d_table, (uint32_t)(table_bytes / sizeof(Rec72)), d_o32, n_gather);
        hipLaunchKernelGGL(k_scatter4, dim3(n_scatter / 256), dim3(256), 0, 0, d_arr, (uint32_t)(arr_bytes / 4), n_scatter);
        hipLaunchKernelGGL(k_store108, dim3(n_store / 256), dim3(256), 0, 0, d_rec, n_store);
        hipDeviceSynchronize();
    }
    printf("known bytes per dispatch: stream_read %zu read; gather72 %zu requested (%u records x 72 B) from a %zu-byte table; scatter4 %zu stored (%u x 4 B); store108 %zu stored (%u x 108 B)\n",
           stream_bytes, (size_t)n_gather * 72, n_gather, table_bytes, (size_t)n_scatter * 4, n_scatter, (size_t)n_store * 108, n_store);
    printf("round 3: stream4 %zu read; gather72_big %zu requested from a %zu-byte table; gather128 %zu requested; store128 %zu stored at random 128-byte slots\n",
           stream_bytes, (size_t)n_gather * 72, big_bytes, (size_t)n_g128 * 128, (size_t)n_s128 * 128);
    return 0;
}
