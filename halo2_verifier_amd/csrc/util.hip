// Format-conversion and small group kernels at the C-ABI boundary.
#include "../../include/h2v.h"
#include "ctx.h"

namespace h2v {

__global__ void __launch_bounds__(256) k_bases_from_bytes(const uint8_t* __restrict__ in, G1A* __restrict__ out, uint32_t* __restrict__ flags, uint32_t n) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint8_t* b = in + 64 * (size_t)i;
    uint8_t tmp[64]; uint32_t any = 0;
    for (int j = 0; j < 64; ++j) { tmp[j] = b[j]; any |= tmp[j]; }
    G1A p; uint32_t bad = 0;
    if (!any) p = G1A::identity();
    else {
        if (!Fq::from_bytes(tmp, p.x) || !Fq::from_bytes(tmp + 32, p.y)) { bad = 1; p = G1A::identity(); }
        else if (!p.on_curve()) { bad = 1; p = G1A::identity(); }
    }
    out[i] = p; flags[i] = bad;
}

__global__ void __launch_bounds__(256) k_scalars_from_bytes(const uint8_t* __restrict__ in, uint32_t* __restrict__ out, uint32_t* __restrict__ flags, uint32_t n) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint8_t* b = in + 32 * (size_t)i;
    uint32_t raw[8];
    for (int j = 0; j < 8; ++j) raw[j] = (uint32_t)b[4 * j] | ((uint32_t)b[4 * j + 1] << 8) | ((uint32_t)b[4 * j + 2] << 16) | ((uint32_t)b[4 * j + 3] << 24);
    uint32_t bad = Fr::geq_p(raw) ? 1u : 0u;
    for (int j = 0; j < 8; ++j) out[8 * (size_t)i + j] = bad ? 0u : raw[j];
    flags[i] = bad;
}

__global__ void k_point_to_bytes(const G1J* __restrict__ in, uint8_t* __restrict__ out, uint32_t* __restrict__ is_identity, uint32_t n) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    G1A a = g1_to_affine(in[i]);
    uint8_t tmp[64];
    if (a.is_identity()) { for (int j = 0; j < 64; ++j) tmp[j] = 0; is_identity[i] = 1; }
    else { a.x.to_bytes(tmp); a.y.to_bytes(tmp + 32); is_identity[i] = 0; }
    for (int j = 0; j < 64; ++j) out[64 * (size_t)i + j] = tmp[j];
}

__global__ void k_fold_pairs(const G1J* __restrict__ parts, uint32_t n_parts, G1J* __restrict__ acc, uint32_t width) {
    uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;  // even: a left channel, odd: a right channel
    if (k >= width) return;
    G1J sum = G1J::identity();
    for (uint32_t i = 0; i < n_parts; ++i) sum = g1_add(sum, parts[(size_t)i * width + k]);
    acc[k] = sum;
}

__global__ void k_affine_to_jacobian(const G1A* __restrict__ in, G1J* __restrict__ out, uint32_t n) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = G1J::from_affine(in[i]);
}
int affine_to_jacobian_enqueue(hipStream_t s, const G1A* d_in, G1J* d_out, uint32_t n) {
    if (!n) return 0;
    hipLaunchKernelGGL(k_affine_to_jacobian, dim3((n + 63) / 64), dim3(64), 0, s, d_in, d_out, n);
    H2V_HIP_CHECK(hipGetLastError());
    return 0;
}

int bases_from_bytes_enqueue(hipStream_t s, const uint8_t* d_bytes, G1A* d_out, uint32_t* d_flags, uint32_t n) {
    if (!n) return 0;
    hipLaunchKernelGGL(k_bases_from_bytes, dim3((n + 255) / 256), dim3(256), 0, s, d_bytes, d_out, d_flags, n);
    H2V_HIP_CHECK(hipGetLastError());
    return 0;
}
int scalars_from_bytes_enqueue(hipStream_t s, const uint8_t* d_bytes, uint32_t* d_out, uint32_t* d_flags, uint32_t n) {
    if (!n) return 0;
    hipLaunchKernelGGL(k_scalars_from_bytes, dim3((n + 255) / 256), dim3(256), 0, s, d_bytes, d_out, d_flags, n);
    H2V_HIP_CHECK(hipGetLastError());
    return 0;
}
int point_to_bytes_enqueue(hipStream_t s, const G1J* d_in, uint8_t* d_out_xy64, uint32_t* d_is_identity, uint32_t n) {
    if (!n) return 0;
    hipLaunchKernelGGL(k_point_to_bytes, dim3((n + 63) / 64), dim3(64), 0, s, d_in, d_out_xy64, d_is_identity, n);
    H2V_HIP_CHECK(hipGetLastError());
    return 0;
}
int fold_pairs_enqueue(hipStream_t s, const G1J* d_parts, uint32_t n_parts, G1J* d_acc, uint32_t width) {
    hipLaunchKernelGGL(k_fold_pairs, dim3((width + 63) / 64), dim3(64), 0, s, d_parts, n_parts, d_acc, width);
    H2V_HIP_CHECK(hipGetLastError());
    return 0;
}

}  // namespace h2v
