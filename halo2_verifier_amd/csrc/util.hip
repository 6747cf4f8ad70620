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

// The record a rank contributes to a sharded batch (include/h2v.h H2V_ACC_RECORD_BYTES): the two accumulators of one group —
// in the pieces the launch left them in (MsmSplit), or whole (parts = 1) — plus the number of the shard's proofs that failed
// before the MSM.  A failed proof is zeroed out of its shard's accumulators, so without the count every OTHER rank's folded
// pairing would still pass.
struct AccRecord { uint32_t failed, parts, shift, reserved; G1J left[H2V_ACC_RECORD_PIECES], right[H2V_ACC_RECORD_PIECES]; };
static_assert(sizeof(AccRecord) == H2V_ACC_RECORD_BYTES && H2V_ACC_RECORD_BYTES == 16 + 2 * H2V_ACC_RECORD_PIECES * 108, "accumulator record layout");
static_assert(H2V_ACC_RECORD_PIECES >= MSM_MAX_PARTS, "a record holds every piece a launch can leave");
// acc: whole points [2g], [2g+1] (parts == 1) — or pieces: [(2g + side) * parts + j]
__global__ void __launch_bounds__(256) k_export_records(const G1J* __restrict__ acc, const G1JSlot* __restrict__ pieces, uint32_t parts, uint32_t shift,
                                                        const int* __restrict__ status, uint32_t gs, AccRecord* __restrict__ out) {
    __shared__ uint32_t failed;
    const uint32_t g = blockIdx.x, t = threadIdx.x;
    if (t == 0) failed = 0;
    __syncthreads();
    uint32_t mine = 0;
    for (uint32_t p = t; p < gs; p += 256) mine += status[(size_t)g * gs + p] != 0 ? 1u : 0u;
    if (mine) atomicAdd(&failed, mine);
    __syncthreads();
    if (t < 2 * H2V_ACC_RECORD_PIECES) {
        const uint32_t side = t / H2V_ACC_RECORD_PIECES, j = t % H2V_ACC_RECORD_PIECES;
        G1J v = G1J::identity();
        if (j < parts) v = pieces ? pieces[(size_t)(2 * g + side) * parts + j].p : acc[2 * g + side];
        (side ? out[g].right : out[g].left)[j] = v;
    }
    if (t == 0) { out[g].failed = failed; out[g].parts = parts; out[g].shift = shift; out[g].reserved = 0; }
}
// a record's accumulator put together: sum_j 2^(shift j) piece_j (only for records that do not match the fold's own split: rare)
__device__ __noinline__ G1J acc_record_whole(const G1J* pc, uint32_t parts, uint32_t shift) {
    if (parts == 0) return G1J::identity();
    G1J a = pc[parts - 1];
    for (uint32_t j = parts - 1; j-- > 0;) {
        for (uint32_t i = 0; i < shift; ++i) a = g1_add(a, a);
        a = g1_add(a, pc[j]);
    }
    return a;
}
// Fold: piece j of (group g, side) = sum over the ranks' records of their piece j — records that were cut the same way
// (parts, shift) add up piece by piece; a record cut differently is put together first and joins piece 0, whose weight is 1.
// parts == 1: the result is the whole point, written to acc[2g + side]; else to pieces[(2g + side) * parts + j] and, in the form
// the Miller lines are evaluated at (X Z, Y, Z^3), to ready[...].  fold_failed[g] = total failed proofs over all records.
__global__ void __launch_bounds__(64) k_fold_records(const AccRecord* __restrict__ recs, uint32_t n_recs, uint32_t groups, uint32_t parts, uint32_t shift,
                                                     G1J* __restrict__ acc, G1JSlot* __restrict__ pieces, G1JSlot* __restrict__ ready, uint32_t* __restrict__ fold_failed) {
    // a team of eight lanes per output point: lane r adds the records r, r + 8, ..., then a three-level butterfly — the sum over the
    // ranks is on the critical path of every sharded launch (eight ranks: 4 dependent additions instead of 8)
    const uint32_t k = (blockIdx.x * blockDim.x + threadIdx.x) >> 3, r = threadIdx.x & 7u;
    const bool live = k < 2 * groups * parts;
    const uint32_t j = live ? k % parts : 0, gs2 = live ? k / parts : 0, g = gs2 >> 1, side = gs2 & 1u;
    G1J sum = G1J::identity();
    uint32_t failed = 0;
    if (live) for (uint32_t i = r; i < n_recs; i += 8) {
        const AccRecord& rec = recs[(size_t)i * groups + g];
        const G1J* pc = side ? rec.right : rec.left;
        const bool same = rec.parts == parts && (rec.shift == shift || parts == 1) && rec.parts <= H2V_ACC_RECORD_PIECES;
        if (same) sum = g1_add(sum, pc[j]);
        else if (j == 0) sum = g1_add(sum, acc_record_whole(pc, rec.parts <= H2V_ACC_RECORD_PIECES ? rec.parts : 0u, rec.shift));
        failed += rec.failed;
    }
    for (uint32_t d = 4; d > 0; d >>= 1) {
        G1J other;
        uint32_t* dst = reinterpret_cast<uint32_t*>(&other);
        const uint32_t* src = reinterpret_cast<const uint32_t*>(&sum);
#pragma unroll
        for (uint32_t w = 0; w < sizeof(G1J) / 4; ++w) dst[w] = (uint32_t)__shfl_down((int)src[w], d, 8);
        sum = g1_add(sum, other);   // lanes r >= 8 - d add a value they do not own: only r = 0 is kept
        failed += (uint32_t)__shfl_down((int)failed, d, 8);
    }
    if (!live || r) return;
    if (parts == 1) acc[gs2] = sum;
    else {
        pieces[k] = sum;
        G1J rd; rd.X = sum.X * sum.Z; rd.Y = sum.Y; rd.Z = sum.Z.sqr() * sum.Z;
        ready[k] = rd;
    }
    if (j == 0 && side == 0) fold_failed[g] = failed;
}
int export_records_enqueue(hipStream_t s, const G1J* d_acc, const G1JSlot* d_pieces, uint32_t parts, uint32_t shift, const int* d_status, uint32_t n, uint32_t groups, void* d_out) {
    if (!d_pieces) { parts = 1; shift = 0; }
    hipLaunchKernelGGL(k_export_records, dim3(groups), dim3(256), 0, s, d_acc, d_pieces, parts, shift, d_status, n / groups, (AccRecord*)d_out);
    H2V_HIP_CHECK(hipGetLastError());
    return 0;
}
int fold_records_enqueue(hipStream_t s, const void* d_recs, uint32_t n_recs, uint32_t groups, uint32_t parts, uint32_t shift, G1J* d_acc, G1JSlot* d_pieces, G1JSlot* d_ready,
                         uint32_t* d_fold_failed) {
    if (parts <= 1 || !d_pieces) { parts = 1; shift = 0; }
    hipLaunchKernelGGL(k_fold_records, dim3((8 * 2 * groups * parts + 63) / 64), dim3(64), 0, s, (const AccRecord*)d_recs, n_recs, groups, parts, shift, d_acc, d_pieces, d_ready, d_fold_failed);
    H2V_HIP_CHECK(hipGetLastError());
    return 0;
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
// device words -> (mapped host) words by a kernel.  A hipMemcpyAsync enqueued at launch time, behind kernels that finish milliseconds later,
// may sit on an SDMA engine's in-order queue with its dependency unmet — and the host -> device copies of the NEXT launches queue up behind
// it (measured: with several launches in flight, h2v_batch_upload_launch blocked for a whole launch, 13.1 -> 8.1 M proofs/s PCIe-inclusive).
__global__ void __launch_bounds__(256) k_copy_words(const uint32_t* __restrict__ src, uint32_t* __restrict__ dst, uint32_t n_words) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_words) dst[i] = src[i];
}
int copy_words_enqueue(hipStream_t s, const void* d_src, void* d_dst, size_t n_words, size_t lds_reserve) {
    if (!n_words) return 0;
    if (lds_reserve > 64 * 1024) H2V_HIP_CHECK(hipFuncSetAttribute((const void*)k_copy_words, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_reserve));
    hipLaunchKernelGGL(k_copy_words, dim3((uint32_t)((n_words + 255) / 256)), dim3(256), lds_reserve, s, (const uint32_t*)d_src, (uint32_t*)d_dst, (uint32_t)n_words);
    H2V_HIP_CHECK(hipGetLastError());
    return 0;
}
int point_to_bytes_enqueue(hipStream_t s, const G1J* d_in, uint8_t* d_out_xy64, uint32_t* d_is_identity, uint32_t n, size_t lds_reserve) {
    if (!n) return 0;
    if (lds_reserve > 64 * 1024) H2V_HIP_CHECK(hipFuncSetAttribute((const void*)k_point_to_bytes, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_reserve));
    hipLaunchKernelGGL(k_point_to_bytes, dim3((n + 63) / 64), dim3(64), lds_reserve, s, d_in, d_out_xy64, d_is_identity, n);
    H2V_HIP_CHECK(hipGetLastError());
    return 0;
}

}  // namespace h2v
