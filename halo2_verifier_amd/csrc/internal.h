// Internal declarations shared by the translation units of libh2v_amd.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <vector>
#include "curve.hip.h"

namespace h2v {

void set_last_error(const std::string& s);
#define H2V_HIP_CHECK(expr)                                                                     \
    do {                                                                                        \
        hipError_t e_ = (expr);                                                                 \
        if (e_ != hipSuccess) {                                                                 \
            ::h2v::set_last_error(std::string(#expr) + ": " + hipGetErrorString(e_));           \
            return H2V_ERR_DEVICE;                                                              \
        }                                                                                       \
    } while (0)

// Device allocation owned by a scope: error paths that `return` from the middle of an entry point free what they took.
template <class T> struct DevBuf {
    T* p = nullptr;
    DevBuf() {}
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    ~DevBuf() { if (p) hipFree(p); }
    int alloc(size_t count) {
        if (p) { hipFree(p); p = nullptr; }
        H2V_HIP_CHECK(hipMalloc(&p, (count ? count : 1) * sizeof(T)));
        return 0;
    }
};

// Forced kernel variants (include/h2v.h: h2v_tuning; every field 0 = automatic).  Lives in the context; a launch copies it into
// the objects that choose (MsmWorkspace::tune, FrvmArgs, the pairing launcher's argument).  The library reads no environment variable.
struct Tuning {
    int frvm_streams = 0, frvm_lds_kb = 0;
    int msm_parts = 0, msm_global_sort = 0, msm_no_term_split = 0, msm_window_threads = 0, msm_window_wpw = 0, msm_window_slots = 0, msm_acc_waves = 0;
    int pairing_one_stream = 0;
    int upload_mode = 0;
};

// ------------------------------------------------------------------ MSM (msm.hip)
// Pippenger over pooled (scalar, base) terms.  Scalars: canonical little-endian 32-bit words, 8 per
// term; bases: affine Montgomery, (0,0) = identity (skipped).
struct MsmPlan {
    uint32_t n;        // terms
    uint32_t c;        // window bits
    uint32_t windows;  // ceil(129 / c): GLV halves are 128-bit magnitudes, +1 bit for the signed-digit carry
    uint32_t buckets;  // 2^(c-1) per window (signed digits)
};
MsmPlan msm_plan(uint32_t n, bool latency = false);

// One MSM of a multi-problem launch: term i < n1 reads scalars[i * sstride ..+8) and bases[i * bstride]; terms n1 <= i < n
// read a second segment (scalars2, bases2) at index i - n1 (a batch's VK-wide bases live apart from its own points).
struct MsmProblem {
    const uint32_t* scalars; const G1A* bases; G1J* out;
    uint32_t sstride, bstride, n;
    uint32_t n1;
    const uint32_t* scalars2; const G1A* bases2;
    const G1A* phi = nullptr; const G1A* phi2 = nullptr;   // optional: phi(P) = (beta x, y) of every base, arrays parallel to bases / bases2 (same stride).  Without them
                                                           // the launch makes its own table (msm_glv_prep); the verifier's points carry theirs from decompression
    uint32_t glv_off = 0;   // first record of this problem in the launch's GLV table (set by msm_enqueue_multi)
    uint32_t sub_first = 0, sub_count = 0;   // set by msm_enqueue_multi when it cuts a large problem into sub-problems (their window sums are merged)
    uint32_t nnz = 0;       // if non-zero: a promise that at most this many of the n scalars are non-zero (sizes msm_accumulate's grid; a broken
                            // promise costs time, not correctness)
    MsmProblem() : scalars(nullptr), bases(nullptr), out(nullptr), sstride(8), bstride(1), n(0), n1(0), scalars2(nullptr), bases2(nullptr) {}
    MsmProblem(const uint32_t* s, const G1A* b, G1J* o, uint32_t ss, uint32_t bs, uint32_t n_) : scalars(s), bases(b), out(o), sstride(ss), bstride(bs), n(n_), n1(n_), scalars2(nullptr), bases2(nullptr) {}
    MsmProblem(const uint32_t* s, const G1A* b, G1J* o, uint32_t ss, uint32_t bs, uint32_t n1_, const uint32_t* s2, const G1A* b2, uint32_t n2)
        : scalars(s), bases(b), out(o), sstride(ss), bstride(bs), n(n1_ + n2), n1(n1_), scalars2(s2), bases2(b2) {}
};
#define MSM_MAX_PROBLEMS 1024     // per launch (a grouped batch: two channels per group; SingleStrategy: one group per proof)
#define MSM_PROBLEM_CHUNK 44      // descriptors handed to the device per setter launch (kernel-argument space)
struct MsmProblemChunk { MsmProblem p[MSM_PROBLEM_CHUNK]; };
static_assert(sizeof(MsmProblemChunk) <= 4000, "a chunk of problem descriptors travels as one kernel argument (4 KB limit)");
struct MsmProblems { std::vector<MsmProblem> p; };

// A Jacobian point in its own 128-byte line: the MSM's intermediate arrays are written once per lane at scattered
// indices, and a 108-byte object that straddles lines costs partial-line writes on both (measured 2.5x the bytes)
struct alignas(128) G1JSlot {
    G1J p;
    uint32_t pad[5];   // written with the point: a store that leaves part of a 32-byte sector untouched is a read-modify-write in memory
    __host__ __device__ G1JSlot& operator=(const G1J& v) { p = v; for (int i = 0; i < 5; ++i) pad[i] = 0; return *this; }
    __host__ __device__ operator const G1J&() const { return p; }
};

#define MSM_MAX_PARTS 6
struct MsmSplit {
    uint32_t want_parts = 0;                        // in
    uint32_t parts = 0, shift = 0, count = 0;       // out
    const G1JSlot* pts = nullptr;                   // out: the workspace's piece array
    const G1JSlot* ready = nullptr;                 // out: the same pieces as (X Z, Y, Z^3), the form the Miller lines are evaluated at
};

struct MsmWorkspace {
    uint32_t cap_terms = 0, cap_problems = 0;
    uint32_t* counts = nullptr;   // [problems * windows * buckets + 3]  (last three words: number of heavy buckets, list cursor, number of straddling buckets)
    uint32_t* offsets = nullptr;  // [problems * windows * buckets]
    uint32_t* cursor = nullptr;   // scatter cursors, then the heavy-bucket list
    uint32_t* list = nullptr;     // term indices sorted by (problem, window, bucket)
    G1JSlot* bucket_pts = nullptr;  // [problems * windows * buckets]
    G1JSlot* window_sums = nullptr; // [problems * windows]
    G1JSlot* pieces = nullptr;      // [2][problems * MSM_MAX_PARTS] partial Horner sums (MsmSplit): Jacobian, then line-ready
    std::vector<MsmProblem> prepared; // the caller's problems msm_prepare_problems uploaded ahead of the next msm_enqueue_multi (empty: none)
    MsmProblem* problems = nullptr;  // [cap_problems] descriptors of the launch in flight (sub-problems when large problems are cut)
    MsmProblem* parents = nullptr;   // [cap_parents] the caller's problems when they were cut
    G1JSlot* merged_sums = nullptr;  // [cap_parents * 128] window sums of the caller's problems, merged over their sub-problems
    const MsmProblem* final_problems = nullptr;   // whichever of the two the last launch's Horner wrote through
    uint32_t cap_parents = 0;
    uint32_t* block_sums = nullptr;  // [cap_buckets / 1024 + 2] prefix-sum scratch
    G1JSlot* partial = nullptr;      // [2 * cap_list / chunk] head and tail pieces of the accumulation chunks
    uint32_t* redo = nullptr;        // [cap_list / chunk] chunks msm_accumulate left to msm_accumulate_redo (complete formulas)
    uint32_t* glv = nullptr;         // [cap_list / 2] signed window digits of the launch's terms, window-major per problem (LDS sort path)
    G1A* phi_pts = nullptr;          // [cap_terms] phi(P) = (beta x, y) of every base of the launch (LDS sort path): made once per term instead of once per list entry
    uint32_t* seg_total = nullptr;   // [problems * windows] entries per list segment
    uint32_t* seg_start = nullptr;   // [problems * windows + 1] logical start of every segment
    size_t cap_buckets = 0, cap_list = 0;
    // optional HIP events around msm_accumulate (the dominant kernel: bench.py's roofline.kernels), recorded when `profile` is set
    Tuning tune;                     // forced variants for the next launch (copied from the context by the caller)
    bool profile = false, profile_recorded = false;
    hipEvent_t ev_acc[2] = {nullptr, nullptr};
    // max_terms_per_problem sizes the bucket arrays (the window plan follows the largest problem of a launch)
    int alloc(uint32_t max_total_terms, uint32_t max_problems, uint32_t max_terms_per_problem = 0);
    void release();
};
// Enqueue all problems of `pr` (each: sum_i scalars[i] * bases[i] -> *out, Jacobian, device memory).  Asynchronous on `s`.
// With `split` (want_parts > 1) the Horner over the windows stops early: problem q is left as `parts` points
//   pts[q * parts + j],   sum_i scalars[i] * bases[i] = sum_j 2^(shift * j) * pts[q * parts + j]
// and *out is NOT written until msm_combine_enqueue (any stream ordered after `s`).  split->parts == 0 on return: the launch was
// not cut (no terms, or a single window) and *out is written as without `split`.
int msm_enqueue_multi(hipStream_t s, MsmWorkspace& ws, const MsmProblems& pr, MsmSplit* split = nullptr);
int msm_combine_enqueue(hipStream_t s, MsmWorkspace& ws, const MsmSplit& sp, size_t lds_reserve = 0);
// optional: the descriptors of the NEXT msm_enqueue_multi on this workspace, sent ahead on the same stream (msm.hip)
int msm_prepare_problems(hipStream_t s, MsmWorkspace& ws, const MsmProblems& pr);
// `lds_reserve` (msm_combine_enqueue, point_to_bytes_enqueue): bytes of LDS the launch asks for without using them.  The two kernels run
// on a batch's auxiliary stream BESIDE the pairing, as a handful of waves with a 0.3 ms chain of their own; where one of those waves
// landed on a SIMD of a pairing workgroup (which issues at raised priority) it crawled — msm_combine_parts took 0.29 ms alone and up to
// 0.53 ms beside k_pairing2, and the launch then waited for IT.  A workgroup that asks for more LDS than a CU has left beside a
// k_pairing2 workgroup (160 KB - 58 KB) is never placed on such a CU, and its CU takes no pairing workgroup afterwards.
#define H2V_AUX_LDS_RESERVE ((size_t)110 * 1024)
int msm_enqueue(hipStream_t s, MsmWorkspace& ws, const uint32_t* d_scalars, const G1A* d_bases, uint32_t n, G1J* d_out);

// ------------------------------------------------------------------ small helpers (util.hip)
// canonical x|y bytes (64 B each, all-zero = identity) -> affine Montgomery; flags[i] = 0 ok, 1 not canonical / not on curve
int bases_from_bytes_enqueue(hipStream_t s, const uint8_t* d_bytes, G1A* d_out, uint32_t* d_flags, uint32_t n);
// canonical 32-B scalars -> 8 x 32-bit words (validated < r); flags[i] = 1 when >= r
int scalars_from_bytes_enqueue(hipStream_t s, const uint8_t* d_bytes, uint32_t* d_out, uint32_t* d_flags, uint32_t n);
// Jacobian -> canonical x|y bytes (+ identity flag word after the 64 bytes: out is 68 B aligned to 4)
int copy_words_enqueue(hipStream_t s, const void* d_src, void* d_dst, size_t n_words, size_t lds_reserve = 0);   // by a kernel (the destination may be mapped host memory)
int point_to_bytes_enqueue(hipStream_t s, const G1J* d_in, uint8_t* d_out_xy64, uint32_t* d_is_identity, uint32_t n, size_t lds_reserve = 0);
// Sharded batches exchange H2V_ACC_RECORD_BYTES records per group: [failed, parts, shift, 0][left pieces][right pieces] (h2v.h).
// export: d_out[g] <- the group's accumulators — pieces [(2g + side) * parts + j] if d_pieces, else the whole points d_acc[2g], [2g+1] —
// and the number of non-zero statuses among the group's n / groups proofs
int export_records_enqueue(hipStream_t s, const G1J* d_acc, const G1JSlot* d_pieces, uint32_t parts, uint32_t shift, const int* d_status, uint32_t n, uint32_t groups, void* d_out);
// fold: the sums over the records [i][g] of each group's accumulators, cut into `parts` pieces of weight 2^(shift j) (-> d_pieces and,
// as (X Z, Y, Z^3), d_ready) or whole (parts <= 1 -> d_acc[2g + side]); d_fold_failed[g] = sum of the records' failure counts
int fold_records_enqueue(hipStream_t s, const void* d_recs, uint32_t n_recs, uint32_t groups, uint32_t parts, uint32_t shift, G1J* d_acc, G1JSlot* d_pieces, G1JSlot* d_ready,
                         uint32_t* d_fold_failed);

}  // namespace h2v
