// Pooled BN254 G1 multi-scalar multiplication for gfx950.
//
// Replaces MSMKZG::eval -> best_multiexp -> multiexp_serial (poly/kzg/msm.rs:81-86,
// arithmetic.rs:7-108), which is a serial fixed-window Pippenger (c in {1,3,4}, 256/c+1
// windows).  The result is the same group element; the schedule is chosen for the GPU and
// evaluates several independent MSMs ("problems": the left and right channel of a batch) in
// one set of launches:
//
//   0. (inside 1 and 3) GLV: k = k1 + k2*lambda with |k1|, |k2| < 2^128, k2 acting on phi(P) = (beta*x, y); both halves are
//                   recoded into SIGNED c-bit digits, so a 254-bit scalar costs 2*ceil(129/c) bucket entries over
//                   ceil(129/c) windows of 2^(c-1) buckets (instead of ceil(254/c) windows of 2^c - 1): half the windows
//                   to reduce and half the doublings in the final Horner chain
//   1. msm_count    one workgroup per tile of 1024 terms: signed digits of both GLV halves, histogram staged in LDS
//                   (one LDS atomic per entry), one global atomic per non-empty (tile, bucket)
//   2. msm_block_sums / msm_scan_sums / msm_offsets
//                   exclusive prefix sum of the histogram in three small launches (1024 bins per workgroup)
//   3. msm_scatter  same tiles: LDS histogram again, one global atomic per (tile, bucket) reserves the tile's span of
//                   the bucket's list, LDS atomics hand out positions inside it
//   4. msm_accumulate  the sorted list is cut into CHUNKS OF EQUAL LENGTH (16..64 entries, msm_chunk_len), one lane per chunk, whatever the
//                   bucket boundaries: every lane does the same number of mixed additions, so a wave is not held up by its
//                   fullest bucket (with ~13 entries per bucket on average a lane-per-bucket mapping idles ~45 % of the
//                   lanes) and no bucket is ever too big for a lane (the top window of a 254-bit scalar is only a few bits
//                   wide: its buckets collect hundreds of entries).  A bucket that lies inside one chunk is written
//                   directly; the head and tail pieces of a chunk go to a side array and
//                   the lane where such a bucket begins lists it (empty buckets are recognised by their count downstream);
//      msm_fixup    a second, dense launch sums the listed buckets' pieces (buckets spread over more than 64 chunks — skewed
//                   inputs — are summed by one wave each, in the same launch)
//   5. msm_window   sum_b (b+1) * bucket[b] per (problem, window) by per-lane running sums over a slice of buckets, then a cross-lane
//                   butterfly (wave shuffles) or LDS tree: one wave per window, or two waves (four-wave workgroups of two windows,
//                   so that every wave has a SIMD to itself), or four for more than 2048 buckets
//   6. msm_final    four lanes per problem: Horner over windows (c doublings + one add per window), every doubling split over the quad;
//      msm_final_parts / msm_combine_parts: the same Horner cut into pieces for a launch that ends in its own pairing checks
//                   (MsmSplit): the checks take the pieces, the whole point is put together beside them
//
// Steps 1-3 are a hand-written counting sort (no atomics on points, no library sort); the only
// atomics are 32-bit counters.  The order of additions inside a bucket depends on atomic
// arrival order, but the group element — hence the affine bytes — does not.
//
// Arithmetic intensity: one term = 96 algorithmic bytes (32 B canonical scalar + 64 B canonical affine base; in memory
// 32 B of scalar words + a 72 B affine point in 29-bit limbs) and 2 * windows mixed additions (7 Fq products + 4 squarings
// each, ~162 v_mad_u64_u32 per product), i.e. ~6 * 10^4 integer instructions per 96 bytes: the stage is bound by
// 32-bit integer multiply issue, not by HBM (DESIGN.md "Roofline").
#include "../../include/h2v.h"
#include <hip/hip_ext.h>
#include "batch.h"

namespace h2v {

// List entries per lane of msm_accumulate ("chunk"), chosen ON THE DEVICE from the number of entries E the sort produced: the
// kernel holds 3 waves per SIMD (196608 lanes per round; 2 until round 3, see msm_accumulate), and with a fixed chunk of 32 a 20-step
// launch (6.4 M entries) needed a second, mostly empty round.  The chunk is the smallest length that fits E into a whole number of
// rounds (k rounds of at most 64 entries per lane), at least 16.
#define MSM_LDS_SORT_MAX_TERMS 16384u
#define MSM_SORT_THREADS 512u
#define MSM_CHUNK_MIN 16u
#define MSM_CHUNK_MAX 64u
#define MSM_ACC_LANES_PER_ROUND 196608u   // 3 waves per SIMD x 1024 SIMDs x 64 lanes (msm_accumulate: 156 VGPRs since its slow path left the kernel)
// (R = lanes per round: MSM_ACC_LANES_PER_ROUND, or 4 waves per SIMD's worth when the launch runs that variant — MsmSeg::lanes_round)
// (A SHORT list — a batch of up to ~256 proofs; the MSM of ONE proof is ~900 entries — is cut into chunks of MSM_CHUNK_SMALL: 16 entries
// on each of 57 lanes took 0.21 ms where 4 entries on 228 lanes take 0.12.  Not beyond: shorter chunks spread a bucket over more of
// them, and from ~512 proofs on the longer fix-up chains of the top window's buckets cost more than the accumulation gains — one
// batch of 1 / 16 / 256 / 512 / 1024 proofs: 1.43 / 1.40 / 1.49 / 1.66 / 1.83 ms with chunks of 4 throughout, 1.49 / 1.45 / 1.57 /
// 1.55 / 1.67 ms with chunks of 16, tools/batch_latency_probe.py.)
#define MSM_CHUNK_SMALL 4u
#define MSM_SHORT_LIST 98304u
__host__ __device__ __forceinline__ uint32_t msm_chunk_len(uint32_t E, uint32_t R) {
    if (E <= MSM_SHORT_LIST) return MSM_CHUNK_SMALL;
    if (E <= R * MSM_CHUNK_MIN) return MSM_CHUNK_MIN;
    const uint32_t k = (uint32_t)(((uint64_t)E + (uint64_t)R * MSM_CHUNK_MAX - 1) / ((uint64_t)R * MSM_CHUNK_MAX));
    const uint32_t c = (uint32_t)(((uint64_t)E + (uint64_t)k * R - 1) / ((uint64_t)k * R));
    return c < MSM_CHUNK_MIN ? MSM_CHUNK_MIN : c;
}
// workgroups of msm_accumulate for at most `max_entries` list entries: whatever E <= max_entries the device finds, its chunk
// lanes ceil(E / msm_chunk_len(E)) stay within k whole rounds, k = the rounds of the bound itself
static inline uint32_t msm_accumulate_blocks(size_t max_entries, size_t R) {
    size_t lanes;
    if (max_entries <= R * MSM_CHUNK_MIN)   // E <= max_entries: chunks of 16 — or of 4 while E <= MSM_SHORT_LIST
        lanes = std::max<size_t>((max_entries + MSM_CHUNK_MIN - 1) / MSM_CHUNK_MIN, (std::min<size_t>(max_entries, MSM_SHORT_LIST) + MSM_CHUNK_SMALL - 1) / MSM_CHUNK_SMALL);
    else lanes = (max_entries + R * MSM_CHUNK_MAX - 1) / (R * MSM_CHUNK_MAX) * R;
    return (uint32_t)(((lanes + 63) / 64 + 1 + 7) / 8 * 8);
}
#define MSM_CONTROL_WORDS 8u     // counts[nb ..]: heavy buckets, E (list entries), straddling buckets, team buckets, chunks to redo
#define MSM_FIXUP_SERIAL 64u     // a bucket spread over more chunks than this is summed by a workgroup
#define MSM_FIXUP_TEAM 3u        // ... over more than this, by a team of eight lanes (the rest: one lane per bucket)
#define MSM_FIXUP_TEAM_BLOCKS 256u
#define MSM_FIXUP_HEAVY_BLOCKS 256u  // one-wave workgroups of msm_fixup for the buckets spread over more than MSM_FIXUP_SERIAL chunks
#define MSM_WIN_THREADS 256

// lanes that share a window's bucket reduction: one wave up to 2048 buckets, four beyond.  A window's reduction is a dependent
// chain of 2 * slice + ~c + log2(T) group additions on every lane (phases of one wave in the 20-step launch, by s_memtime: running
// sums 296 us, slice weighting 95 us, butterfly 56 us), so more lanes per window shorten the chain — but only while the launch has
// few workgroups.  Measured (gpurun_out/r02_winT*, r02_win4): 1 / 2 / 4 waves per window take 0.47 / 0.35 / 0.31 ms at 192
// workgroups, but 0.48 / 0.59 / 0.60 ms at 480; splitting each of the 480 windows between two ONE-wave workgroups (the later one
// adding the halves) also loses, 0.59 ms; so does a compact group law with out-of-line field products (0.65 ms) — the per-operation
// cost of this kernel rises with the number of its waves per CU, whatever their grouping.  Hence: four waves per window up to
// 256 workgroups, one wave beyond.
static inline uint32_t msm_window_threads(uint32_t buckets, uint32_t n_workgroups = 0xffffffffu, int forced = 0) {
    if (buckets <= 64) return std::max(1u, buckets);
    if (n_workgroups != 0xffffffffu && (forced == 64 || forced == 128 || forced == 256)) return (uint32_t)forced;   // h2v_tuning.msm_window_threads
    if (buckets > 2048 || (buckets >= 256 && n_workgroups <= 256)) return 256u;
    return 64u;
}

// Window width.  Throughput plan: a cost model in Fq products — 2n mixed additions (11) per window, and per window the
// reduction sum_b (b+1) B_b done by T lanes: 2 * slice running-sum additions, a c-bit double-and-add to weight the slice, a
// log2(T) tree — full additions (16 products), T lanes wide.  Latency plan (a launch whose problems are all tiny, e.g. one
// proof under SingleStrategy): the work is negligible whatever c is, what counts is the dependent chain — the window
// reduction's additions and msm_final's one addition per window on top of its ~130 doublings — so fewer, wider windows win.
MsmPlan msm_plan(uint32_t n, bool latency) {
    MsmPlan best{n, 2, 65, 2};
    double best_cost = 1e300;
    for (uint32_t c = 2; c <= 15; ++c) {
        uint32_t w = (130 + c - 1) / c;        // magnitudes < 2^128 (+1 bit of slack) + the carry of the signed recoding
        uint32_t b = 1u << (c - 1);
        uint32_t T = msm_window_threads(b), slice = (b + T - 1) / T;
        double lg = 0; for (uint32_t t = T; t > 1; t >>= 1) lg += 1;
        double cost = latency ? 9.0 * w + 9.0 * (2.0 * slice + 1.5 * c + 1.0 + lg) + 5.0 * std::min<double>(32.0, 2.0 * n)   // microseconds: final, window, one chunk
                              : (double)w * (11.0 * 2.0 * n + 16.0 * T * (2.0 * slice + 1.5 * c + 1.0 + lg));
        if (cost < best_cost) { best_cost = cost; best = MsmPlan{n, c, w, b}; }
    }
    return best;
}
// a launch is planned for latency when all its problems together are a few thousand terms
static inline bool msm_latency_bound(size_t total_terms) { return total_terms <= 4096; }

// Large problems are cut into sub-problems of at most MSM_LDS_SORT_MAX_TERMS terms (msm_enqueue_multi): the workspace is sized for the
// sub-problems, and for the uncut form too (h2v_tuning.msm_no_term_split).
static inline uint32_t msm_subproblems(uint32_t n) { return n > MSM_LDS_SORT_MAX_TERMS ? (n + MSM_LDS_SORT_MAX_TERMS - 1) / MSM_LDS_SORT_MAX_TERMS : 1u; }
int MsmWorkspace::alloc(uint32_t max_terms, uint32_t max_problems, uint32_t max_per_problem) {
    release();
    if (!max_per_problem || max_per_problem > max_terms) max_per_problem = max_terms;
    const uint32_t subs = msm_subproblems(max_per_problem);
    cap_terms = max_terms; cap_parents = max_problems;
    cap_problems = (uint32_t)std::min<size_t>((size_t)max_problems * subs, MSM_MAX_PROBLEMS);
    if (cap_problems < max_problems) cap_problems = max_problems;
    const uint32_t per_sub = subs > 1 ? MSM_LDS_SORT_MAX_TERMS : max_per_problem;   // a cut problem's pieces can be exactly the limit
    // (problems, largest problem) of the two forms a launch can take
    const uint32_t form_n[2] = {per_sub, max_per_problem}, form_q[2] = {cap_problems, max_problems};
    size_t mb = 0;
    cap_list = 0;
    for (int f = 0; f < 2; ++f) {
        size_t fb = 0;
        for (int lat = 0; lat < 2; ++lat) {
            for (uint32_t n = 1; n <= form_n[f]; n = n < 16 ? n + 1 : n + n / 8) { MsmPlan p = msm_plan(n, lat != 0); fb = std::max(fb, (size_t)p.windows * p.buckets); }
            MsmPlan p = msm_plan(form_n[f], lat != 0); fb = std::max(fb, (size_t)p.windows * p.buckets);
        }
        mb = std::max(mb, fb * form_q[f]);
        // list entries: 2 GLV halves x windows per term (segment-addressed: every problem owns 2 * nmax entries per window), for the plan of the largest problem of a launch
        for (uint32_t n = 1;; n = n < 16 ? n + 1 : n + n / 8) {
            if (n > form_n[f]) n = form_n[f];
            // (the LDS sort gives every problem 2 * nmax list slots per window: problems x largest problem, not the term total)
            size_t terms = n <= MSM_LDS_SORT_MAX_TERMS ? (size_t)n * form_q[f] : std::min<size_t>((size_t)max_terms, (size_t)n * form_q[f]);
            cap_list = std::max(cap_list, terms * 2 * std::max(msm_plan(n, false).windows, msm_plan(n, true).windows));
            if (n == form_n[f]) break;
        }
    }
    cap_buckets = mb;
    H2V_HIP_CHECK(hipMalloc(&counts, (mb + MSM_CONTROL_WORDS) * 4));
    H2V_HIP_CHECK(hipMalloc(&offsets, mb * 4));
    H2V_HIP_CHECK(hipMalloc(&cursor, 2 * mb * 4));   // scatter cursors; then the fix-up's work lists (second half: the team list)
    H2V_HIP_CHECK(hipMalloc(&list, cap_list * 4));
    H2V_HIP_CHECK(hipMalloc(&bucket_pts, mb * sizeof(G1JSlot)));
    H2V_HIP_CHECK(hipMalloc(&window_sums, (size_t)128 * cap_problems * sizeof(G1JSlot)));
    H2V_HIP_CHECK(hipMalloc(&merged_sums, (size_t)128 * cap_parents * sizeof(G1JSlot)));
    H2V_HIP_CHECK(hipMalloc(&pieces, (size_t)2 * MSM_MAX_PARTS * cap_problems * sizeof(G1JSlot)));
    H2V_HIP_CHECK(hipMalloc(&problems, (size_t)cap_problems * sizeof(MsmProblem)));
    H2V_HIP_CHECK(hipMalloc(&parents, (size_t)cap_parents * sizeof(MsmProblem)));
    H2V_HIP_CHECK(hipMalloc(&block_sums, (mb / 1024 + 2) * 4));
    // chunks of a launch at most: cap_list / 16, or (short lists, msm_chunk_len) a quarter of up to MSM_SHORT_LIST entries
    const size_t max_chunks = std::max<size_t>(cap_list / MSM_CHUNK_MIN + 1, std::min<size_t>(cap_list, MSM_SHORT_LIST) / MSM_CHUNK_SMALL + 1);
    H2V_HIP_CHECK(hipMalloc(&partial, max_chunks * 2 * sizeof(G1JSlot)));
    H2V_HIP_CHECK(hipMalloc(&redo, max_chunks * 4));
    H2V_HIP_CHECK(hipMalloc(&glv, (cap_list / 2 + 1) * 4));   // digit table: one word per (term, window)
    H2V_HIP_CHECK(hipMalloc(&phi_pts, ((size_t)cap_terms + 1) * sizeof(G1A)));
    H2V_HIP_CHECK(hipMalloc(&seg_total, ((size_t)128 * cap_problems + 2) * 4));
    H2V_HIP_CHECK(hipMalloc(&seg_start, ((size_t)128 * cap_problems + 2) * 4));
    for (int i = 0; i < 2; ++i) if (!ev_acc[i]) H2V_HIP_CHECK(hipEventCreate(&ev_acc[i]));
    return 0;
}
void MsmWorkspace::release() {
    if (counts) hipFree(counts);
    if (offsets) hipFree(offsets);
    if (cursor) hipFree(cursor);
    if (list) hipFree(list);
    if (bucket_pts) hipFree(bucket_pts);
    if (window_sums) hipFree(window_sums);
    if (pieces) hipFree(pieces);
    pieces = nullptr;
    if (problems) hipFree(problems);
    if (parents) hipFree(parents);
    if (merged_sums) hipFree(merged_sums);
    parents = nullptr; merged_sums = nullptr; final_problems = nullptr; cap_parents = 0;
    if (block_sums) hipFree(block_sums);
    if (partial) hipFree(partial);
    if (redo) hipFree(redo);
    redo = nullptr;
    if (glv) hipFree(glv);
    if (phi_pts) hipFree(phi_pts);
    phi_pts = nullptr;
    if (seg_total) hipFree(seg_total);
    if (seg_start) hipFree(seg_start);
    glv = seg_total = seg_start = nullptr;
    counts = offsets = cursor = list = block_sums = nullptr; bucket_pts = window_sums = partial = nullptr; problems = nullptr;
    for (int i = 0; i < 2; ++i) if (ev_acc[i]) { hipEventDestroy(ev_acc[i]); ev_acc[i] = nullptr; }
    cap_terms = 0; cap_problems = 0; profile_recorded = false;
}

// ---- GLV decomposition for BN254 G1 (constants derived in DESIGN.md section 4; lattice basis (a1, b1), (a2, b2) with
// a_i + b_i*lambda = 0 mod r; g_i = floor(2^256 * {b2, -b1} / r))
__device__ __forceinline__ void mul_lo5(uint32_t out[5], const uint32_t* a, int na, const uint32_t* b, int nb) {  // a*b mod 2^160
    uint64_t acc[5] = {0, 0, 0, 0, 0};
    for (int i = 0; i < na && i < 5; ++i) {
        uint64_t carry = 0;
        for (int j = 0; j < nb && i + j < 5; ++j) {
            uint64_t t = (uint64_t)a[i] * b[j] + (uint32_t)acc[i + j] + carry;
            acc[i + j] = (uint32_t)t; carry = t >> 32;
        }
    }
    for (int i = 0; i < 5; ++i) out[i] = (uint32_t)acc[i];
}
// (k * g) >> 256 for a 5-limb g; result < 2^130 in 5 limbs
__device__ __forceinline__ void mul_hi256(uint32_t out[5], const uint32_t k[8], const uint32_t g[5]) {
    uint32_t t[13];
    for (int i = 0; i < 13; ++i) t[i] = 0;
    for (int i = 0; i < 8; ++i) {
        uint64_t carry = 0;
        for (int j = 0; j < 5; ++j) {
            uint64_t v = (uint64_t)k[i] * g[j] + t[i + j] + carry;
            t[i + j] = (uint32_t)v; carry = v >> 32;
        }
        t[i + 5] = (uint32_t)carry;
    }
    for (int i = 0; i < 5; ++i) out[i] = t[8 + i];
}
struct GlvHalf { uint32_t mag[5]; bool neg; };  // |k_i| < 2^128 (limb 4 only ever holds the recoding carry)
__device__ __forceinline__ void glv_finish(GlvHalf& h, uint32_t t[5]) {  // t = two's complement mod 2^160
    h.neg = (t[4] >> 31) != 0;
    if (h.neg) { uint64_t c = 1; for (int i = 0; i < 5; ++i) { uint64_t v = (uint64_t)(~t[i]) + c; t[i] = (uint32_t)v; c = v >> 32; } }
    for (int i = 0; i < 5; ++i) h.mag[i] = t[i];
}
__device__ __forceinline__ void glv_decompose(const uint32_t* __restrict__ k, GlvHalf& h1, GlvHalf& h2) {
    const uint32_t A1[5] = {0x7d4f1128u, 0x8211bbebu, 0xeeb859fcu, 0x6f4d8248u, 0u};   //  a1
    const uint32_t B1N[5] = {0x94d213e3u, 0x89d32568u, 0u, 0u, 0u};                     // -b1 ( = a2 )
    const uint32_t B2[5] = {0x1221250bu, 0x0be4e154u, 0xeeb859fdu, 0x6f4d8248u, 0u};    //  b2
    const uint32_t G1[5] = {0x00ff6565u, 0x5398fd03u, 0xa773d2d2u, 0x4ccef014u, 0x2u};
    const uint32_t G2[5] = {0xc7e0b3d7u, 0xd91d232eu, 0x2u, 0u, 0u};
    uint32_t kk[8];
    for (int i = 0; i < 8; ++i) kk[i] = k[i];
    uint32_t c1[5], c2[5], p1[5], p2[5], t[5];
    mul_hi256(c1, kk, G1);
    mul_hi256(c2, kk, G2);
    // k1 = k - c1*a1 - c2*a2
    mul_lo5(p1, c1, 5, A1, 5); mul_lo5(p2, c2, 5, B1N, 5);
    {
        int64_t borrow = 0;
        for (int i = 0; i < 5; ++i) { int64_t v = (int64_t)kk[i] - p1[i] - p2[i] + borrow; t[i] = (uint32_t)v; borrow = v >> 32; }
    }
    glv_finish(h1, t);
    // k2 = -c1*b1 - c2*b2 = c1*(-b1) - c2*b2
    mul_lo5(p1, c1, 5, B1N, 5); mul_lo5(p2, c2, 5, B2, 5);
    {
        int64_t borrow = 0;
        for (int i = 0; i < 5; ++i) { int64_t v = (int64_t)p1[i] - p2[i] + borrow; t[i] = (uint32_t)v; borrow = v >> 32; }
    }
    glv_finish(h2, t);
}
__device__ __forceinline__ uint32_t raw_window(const uint32_t m[5], uint32_t w, uint32_t c) {
    uint32_t off = w * c, word = off >> 5, sh = off & 31;
    if (word >= 5) return 0;
    uint32_t v = m[word] >> sh;
    if (sh + c > 32 && word + 1 < 5) v |= m[word + 1] << (32 - sh);
    return v & ((1u << c) - 1);
}
// entry of the sorted list: term index | half << 30 | negate << 31
#define MSM_ENTRY_HALF 0x40000000u
#define MSM_ENTRY_NEG 0x80000000u
#define MSM_ENTRY_TERM 0x3fffffffu

// Counting sort, passes 1 and 3.  A workgroup owns a tile of MSM_TILE consecutive terms of one problem and keeps the
// histogram of (a group of) windows in LDS: every (term, half, window) entry costs one LDS atomic; global atomics
// are issued once per non-empty bin per tile — for the bench shape ~6x fewer than one global atomic per entry.
//   count:   LDS histogram -> counts[bin] += h
//   scatter: LDS histogram -> base = cursor[bin] += h (global, returns the tile's first position inside the bucket)
//            -> the LDS word becomes a running position -> every entry takes offsets[bin] + (LDS word)++
#define MSM_TILE 1024u
#define MSM_TILE_THREADS 256u
#define MSM_LDS_WORDS 16384u   // 64 KB of histogram per workgroup: as many windows per pass as fit

template <bool SCATTER>
__global__ void __launch_bounds__(MSM_TILE_THREADS) msm_count_or_scatter(const MsmProblem* __restrict__ prs, uint32_t n_problems, uint32_t tiles_per_problem, MsmPlan p, uint32_t windows_per_pass, uint32_t* __restrict__ counts,
                                                                         const uint32_t* __restrict__ offsets, uint32_t* __restrict__ cursor, uint32_t* __restrict__ list) {
    extern __shared__ uint32_t hist[];  // [windows_per_pass][buckets]
    // 1-D grid of 8 * ceil(problems / 8) * tiles workgroups; XCD x (= blockIdx % 8) takes the x-th eighth of the problems, so
    // that the bins and the list spans a problem's tiles write to stay in one L2 and partial-line stores merge there
    const uint32_t tiles = tiles_per_problem, per_xcd = (n_problems + 7) / 8;
    const uint32_t q = (blockIdx.x % 8) * per_xcd + (blockIdx.x / 8) / tiles, tile0 = ((blockIdx.x / 8) % tiles) * MSM_TILE, tid = threadIdx.x;
    if (q >= n_problems) return;  // whole workgroup
    const MsmProblem pq = prs[q];
    const uint32_t n = pq.n;
    if (tile0 >= n) return;  // whole workgroup: no barrier is skipped by part of a group
    const uint32_t nbq = p.windows * p.buckets;
    const uint32_t half_range = 1u << (p.c - 1);
    constexpr uint32_t PER = MSM_TILE / MSM_TILE_THREADS;
    // the lanes' terms: t = tile0 + tid + k * THREADS (coalesced); GLV halves stay in registers across the passes
    for (uint32_t w0 = 0; w0 < p.windows; w0 += windows_per_pass) {
        const uint32_t w1 = min(p.windows, w0 + windows_per_pass), nwords = (w1 - w0) * p.buckets;
        for (uint32_t i = tid; i < nwords; i += MSM_TILE_THREADS) hist[i] = 0;
        __syncthreads();
        for (int phase = 0; phase < (SCATTER ? 2 : 1); ++phase) {
            for (uint32_t k = 0; k < PER; ++k) {
                const uint32_t t = tile0 + tid + k * MSM_TILE_THREADS;
                if (t >= n) break;
                const uint32_t* s = t < pq.n1 ? pq.scalars + (size_t)pq.sstride * t : pq.scalars2 + (size_t)pq.sstride * (t - pq.n1);
                uint32_t nz = 0;
                for (int i = 0; i < 8; ++i) nz |= s[i];
                if (!nz) continue;  // zero scalar (e.g. a point slot the channel does not use)
                const uint32_t* bw = reinterpret_cast<const uint32_t*>(t < pq.n1 ? pq.bases + (size_t)pq.bstride * t : pq.bases2 + (size_t)pq.bstride * (t - pq.n1));
                uint32_t any = 0;
                for (int i = 0; i < (int)(sizeof(G1A) / 4); ++i) any |= bw[i];
                if (!any) continue;  // identity bases contribute nothing
                GlvHalf h[2];
                glv_decompose(s, h[0], h[1]);
                for (uint32_t hf = 0; hf < 2; ++hf) {
                    uint32_t carry = 0;
                    for (uint32_t w = 0; w < w1; ++w) {
                        uint32_t raw = raw_window(h[hf].mag, w, p.c) + carry;
                        bool neg_digit = raw > half_range;       // digit = raw - 2^c, carry into the next window
                        uint32_t mag = neg_digit ? (1u << p.c) - raw : raw;
                        carry = neg_digit ? 1u : 0u;
                        if (!mag || w < w0) continue;
                        uint32_t local = (w - w0) * p.buckets + mag - 1;
                        if (!SCATTER || phase == 0) atomicAdd(&hist[local], 1u);
                        else {
                            uint32_t b = q * nbq + w * p.buckets + mag - 1;
                            uint32_t pos = atomicAdd(&hist[local], 1u);
                            list[offsets[b] + pos] = t | (hf ? MSM_ENTRY_HALF : 0u) | ((neg_digit != h[hf].neg) ? MSM_ENTRY_NEG : 0u);
                        }
                    }
                }
            }
            __syncthreads();
            if (phase == 0) {
                for (uint32_t i = tid; i < nwords; i += MSM_TILE_THREADS) {
                    uint32_t v = hist[i];
                    if (!v) continue;
                    uint32_t b = q * nbq + w0 * p.buckets + i;
                    if (SCATTER) hist[i] = atomicAdd(&cursor[b], v);  // this tile's first position inside bucket b
                    else atomicAdd(&counts[b], v);
                }
                __syncthreads();
            }
        }
    }
}

// ---- LDS counting sort (round 2).  The global counting sort above stores every 4-byte list entry at a position of its own: each
// store costs a 32-byte sector (0.4 GB of write traffic for 41 MB of list), and it needs two passes over the scalars, a three-kernel
// prefix sum and one global atomic per (tile, bin).  When a problem is small enough for ONE window's entries to fit LDS as 16-bit
// records (at most 16 384 terms: every batch up to 1365 proofs of the toy VK), the sort runs per (problem, window) inside one workgroup:
//   msm_glv_prep   one lane per term: GLV halves and, from them, the signed c-bit digits of EVERY window, one word per (window, term),
//                  window-major; a term whose scalar is zero or whose base is the identity gets all-zero words and is skipped
//   msm_sort_lds   one workgroup per (problem, window): histogram of the window's signed digits in LDS, exclusive scan, scatter into
//                  an LDS list of 16-bit records, then ONE contiguous store of the window's list segment and of its bin table
//   msm_seg_scan   logical start of every segment (a 480-entry prefix sum for the 20-step launch) and the entry total
// digit table: for problem q (first word dig_off = glv_off * windows), window w, term t: word [dig_off + w * n + t] holds the two
// signed digits of the term's GLV halves, 16 bits each: magnitude (0 = no entry) | 0x8000 when the ENTRY is negated (digit sign xor
// the half's sign).  Written window-major so that the sort of window w reads n consecutive words.
__global__ void __launch_bounds__(256) msm_glv_prep(const MsmProblem* __restrict__ prs, uint32_t n_problems, MsmPlan p, uint32_t* __restrict__ dig, G1A* __restrict__ phi_pts) {
    const uint32_t q = blockIdx.y, t = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= n_problems) return;
    const MsmProblem pq = prs[q];
    if (t >= pq.n) return;
    uint32_t* out = dig + (size_t)pq.glv_off * p.windows + t;
    const uint32_t* sc = t < pq.n1 ? pq.scalars + (size_t)pq.sstride * t : pq.scalars2 + (size_t)pq.sstride * (t - pq.n1);
    uint32_t nz = 0;
    for (int i = 0; i < 8; ++i) nz |= sc[i];
    if (nz) {   // identity bases contribute nothing
        const uint32_t* bw = reinterpret_cast<const uint32_t*>(t < pq.n1 ? pq.bases + (size_t)pq.bstride * t : pq.bases2 + (size_t)pq.bstride * (t - pq.n1));
        uint32_t any = 0;
        for (int i = 0; i < (int)(sizeof(G1A) / 4); ++i) any |= bw[i];
        nz = any;
        if (any && !(t < pq.n1 ? pq.phi : pq.phi2)) phi_pts[(size_t)pq.glv_off + t] = g1_phi(*reinterpret_cast<const G1A*>(bw));   // (callers that bring phi(P) with their bases skip this)
    }
    if (!nz) { for (uint32_t w = 0; w < p.windows; ++w) out[(size_t)w * pq.n] = 0; return; }
    GlvHalf h[2];
    glv_decompose(sc, h[0], h[1]);
    const uint32_t half_range = 1u << (p.c - 1);
    uint32_t carry[2] = {0, 0};
    for (uint32_t w = 0; w < p.windows; ++w) {
        uint32_t word = 0;
#pragma unroll
        for (uint32_t hf = 0; hf < 2; ++hf) {
            const uint32_t raw = raw_window(h[hf].mag, w, p.c) + carry[hf];
            const bool neg_digit = raw > half_range;       // digit = raw - 2^c, carry into the next window
            const uint32_t mag = neg_digit ? (1u << p.c) - raw : raw;
            carry[hf] = neg_digit ? 1u : 0u;
            const uint32_t d16 = mag ? (mag | ((neg_digit != h[hf].neg) ? 0x8000u : 0u)) : 0u;
            word |= d16 << (16 * hf);
        }
        out[(size_t)w * pq.n] = word;
    }
}
__global__ void __launch_bounds__(MSM_SORT_THREADS) msm_sort_lds(const MsmProblem* __restrict__ prs, const uint32_t* __restrict__ dig, MsmPlan p, uint32_t stride,
                                                                 uint32_t* __restrict__ counts, uint32_t* __restrict__ offsets, uint32_t* __restrict__ list,
                                                                 uint32_t* __restrict__ seg_total) {
    extern __shared__ uint32_t sort_lds[];            // hist[buckets] | scan scratch[MSM_SORT_THREADS] | 16-bit records[2 n]
    uint32_t* hist = sort_lds;
    uint32_t* part = sort_lds + p.buckets;
    unsigned short* rec = reinterpret_cast<unsigned short*>(part + MSM_SORT_THREADS);
    const uint32_t w = blockIdx.x, q = blockIdx.y, tid = threadIdx.x;
    const MsmProblem pq = prs[q];
    const uint32_t n = pq.n, seg = q * p.windows + w;
    const uint32_t* dw = dig + (size_t)pq.glv_off * p.windows + (size_t)w * n;    // this window's digits, one word per term
    for (uint32_t i = tid; i < p.buckets; i += MSM_SORT_THREADS) hist[i] = 0;
    __syncthreads();
    for (int pass = 0; pass < 2; ++pass) {
        for (uint32_t t = tid; t < n; t += MSM_SORT_THREADS) {
            const uint32_t word = dw[t];
#pragma unroll
            for (uint32_t hf = 0; hf < 2; ++hf) {
                const uint32_t d16 = (word >> (16 * hf)) & 0xffffu, mag = d16 & 0x7fffu;
                if (!mag) continue;
                const uint32_t pos = atomicAdd(&hist[mag - 1], 1u);     // pass 0: count; pass 1: hist holds running positions
                if (pass == 1) rec[pos] = (unsigned short)(t | (hf ? 0x4000u : 0u) | (d16 & 0x8000u));
            }
        }
        __syncthreads();
        if (pass == 0) {
            // exclusive scan of the histogram: each thread owns a run of consecutive bins
            const uint32_t per = (p.buckets + MSM_SORT_THREADS - 1) / MSM_SORT_THREADS, b0 = tid * per;
            uint32_t local = 0;
            for (uint32_t i = 0; i < per; ++i) if (b0 + i < p.buckets) local += hist[b0 + i];
            part[tid] = local;
            __syncthreads();
            for (uint32_t d = 1; d < MSM_SORT_THREADS; d <<= 1) {
                const uint32_t v = tid >= d ? part[tid - d] : 0;
                __syncthreads();
                part[tid] += v;
                __syncthreads();
            }
            uint32_t run = part[tid] - local;
            const size_t bin0 = (size_t)seg * p.buckets;
            for (uint32_t i = 0; i < per; ++i) if (b0 + i < p.buckets) {
                const uint32_t cnt = hist[b0 + i];
                counts[bin0 + b0 + i] = cnt; offsets[bin0 + b0 + i] = run;
                hist[b0 + i] = run;            // becomes the bin's running position for the scatter pass
                run += cnt;
            }
            if (tid == MSM_SORT_THREADS - 1) seg_total[seg] = part[tid];
            __syncthreads();
        }
    }
    // the window's sorted list, one contiguous store; records widen to the list's 32-bit entries
    const uint32_t total = part[MSM_SORT_THREADS - 1];
    uint32_t* dst = list + (size_t)seg * stride;
    for (uint32_t i = tid; i < total; i += MSM_SORT_THREADS) {
        const uint32_t r = rec[i];
        dst[i] = (r & 0x3fffu) | ((r & 0x4000u) ? MSM_ENTRY_HALF : 0u) | ((r & 0x8000u) ? MSM_ENTRY_NEG : 0u);
    }
}
// seg_start[s] = logical start of segment s, seg_start[nseg] = counts[nb + 1] = E; zeroes the three control words
__global__ void __launch_bounds__(1024) msm_seg_scan(const uint32_t* __restrict__ seg_total, uint32_t nseg, uint32_t* __restrict__ seg_start, uint32_t* __restrict__ control) {
    __shared__ uint32_t part[1024];
    __shared__ uint32_t carry;
    const uint32_t t = threadIdx.x;
    if (t == 0) carry = 0;
    __syncthreads();
    for (uint32_t base = 0; base < nseg; base += 1024) {   // uniform trip count
        const uint32_t i = base + t, c = i < nseg ? seg_total[i] : 0;
        part[t] = c;
        __syncthreads();
        for (uint32_t d = 1; d < 1024; d <<= 1) {
            const uint32_t v = t >= d ? part[t - d] : 0;
            __syncthreads();
            part[t] += v;
            __syncthreads();
        }
        if (i < nseg) seg_start[i] = carry + part[t] - c;
        __syncthreads();
        if (t == 1023) carry += part[1023];
        __syncthreads();
    }
    if (t == 0) { seg_start[nseg] = carry; control[0] = 0; control[1] = carry; control[2] = 0; control[3] = 0; control[4] = 0; }
}

// the base an entry refers to: P, -P, phi(P) or -phi(P), phi(x, y) = (beta * x, y) — split into the load and the fix-up so that
// the load of the next entry can be issued before the additions of the current one
// `phi_pts` (may be null): the table of phi(P) = (beta x, y) per term that msm_glv_prep leaves on the LDS-sort path — an entry of the
// second GLV half then LOADS its base from there (one 72-byte gather either way) instead of paying a field product per list entry:
// with 64 chunks side by side some lane needs it in every iteration, so the whole wave paid it every time (9 % of msm_accumulate's
// multiply-adds).  A table PER PROBLEM costs L2 footprint: what an XCD gathers from grows from the points of ~2.5 groups (both
// channels of a group index the same array) to those plus a table per channel, and the fetches from memory double (0.36 -> 0.8 GB
// per 20-step launch).  So the verifier's launches bring phi(P) with the points themselves (MsmProblem::phi, written by k_decompress
// and shared by both channels); the table is for callers that bring bases only (h2v_msm_g1).  Variants measured with the table alone:
// beta x only (two gathers per entry: 0.68 ms), one (beta x | y | x) record per term serving both halves (0.695 ms, the same fetches).
__device__ __forceinline__ G1A msm_entry_load(const MsmProblem& q, uint32_t e, const G1A* __restrict__ phi_pts) {
    const uint32_t t = e & MSM_ENTRY_TERM;
    const bool first = t < q.n1;
    const size_t off = (size_t)(first ? t : t - q.n1) * q.bstride;
    const G1A* b = (first ? q.bases : q.bases2) + off;
    if (phi_pts && (e & MSM_ENTRY_HALF)) {
        const G1A* ph = first ? q.phi : q.phi2;   // the caller's own phi(P), parallel to its bases — else the launch's table
        b = ph ? ph + off : phi_pts + ((size_t)q.glv_off + t);
    }
    return *b;
}
__device__ __forceinline__ G1A msm_entry_apply(G1A b, uint32_t e, bool have_phi) {
    if (!have_phi && (e & MSM_ENTRY_HALF)) b.x = g1_beta_times(b.x);
    // -y as 2p - y without the correction step (27 instructions instead of 65; a wave has negated entries in every iteration): the
    // result lies in (0, 2p] and equals 2p only for y = 0, which no curve point has — an ordinary representative for everything the
    // group law does with it (products; the accumulator's Y when the accumulator was empty).  The identity (0, 0) is skipped before
    // its coordinates matter (g1_madd_fast tests x AND y for zero: (0, 2p) would not pass as the identity, so it is kept as it is).
    if ((e & MSM_ENTRY_NEG) && !b.y.is_zero()) b.y = Fq::lazy_neg(b.y);
    return b;
}
__device__ __forceinline__ G1A msm_entry_base(const MsmProblem& q, uint32_t e) { return msm_entry_apply(msm_entry_load(q, e, nullptr), e, false); }

// Exclusive prefix sum of counts[0..nb) -> offsets (bin order, so that a list position can be mapped back to its bin by
// binary search), in three launches: per-1024-bin sums, a scan of those sums by one workgroup, local scans + block base.
// counts[nb + 1] receives the total number of list entries.  Also zeroes the scatter cursors.
__global__ void __launch_bounds__(1024) msm_block_sums(const uint32_t* __restrict__ counts, uint32_t* __restrict__ block_sums, uint32_t nb) {
    __shared__ uint32_t part[1024];
    const uint32_t t = threadIdx.x, i = blockIdx.x * 1024 + t;
    part[t] = i < nb ? counts[i] : 0;
    __syncthreads();
    for (uint32_t d = 512; d > 0; d >>= 1) { if (t < d) part[t] += part[t + d]; __syncthreads(); }
    if (t == 0) block_sums[blockIdx.x] = part[0];
}
__global__ void __launch_bounds__(1024) msm_scan_sums(uint32_t* __restrict__ block_sums, uint32_t nblk, uint32_t* __restrict__ total) {
    __shared__ uint32_t part[1024];
    __shared__ uint32_t carry;
    const uint32_t t = threadIdx.x;
    if (t == 0) carry = 0;
    __syncthreads();
    for (uint32_t base = 0; base < nblk; base += 1024) {   // uniform trip count: no barrier is skipped
        const uint32_t i = base + t;
        const uint32_t c = i < nblk ? block_sums[i] : 0;
        part[t] = c;
        __syncthreads();
        for (uint32_t d = 1; d < 1024; d <<= 1) {
            uint32_t v = t >= d ? part[t - d] : 0;
            __syncthreads();
            part[t] += v;
            __syncthreads();
        }
        if (i < nblk) block_sums[i] = carry + part[t] - c;
        __syncthreads();
        if (t == 1023) carry += part[1023];
        __syncthreads();
    }
    if (t == 0) *total = carry;
}
__global__ void __launch_bounds__(1024) msm_offsets(const uint32_t* __restrict__ counts, const uint32_t* __restrict__ block_sums, uint32_t* __restrict__ offsets,
                                                    uint32_t* __restrict__ cursor, uint32_t nb) {
    __shared__ uint32_t part[1024];
    const uint32_t t = threadIdx.x, i = blockIdx.x * 1024 + t;
    const uint32_t c = i < nb ? counts[i] : 0;
    part[t] = c;
    __syncthreads();
    for (uint32_t d = 1; d < 1024; d <<= 1) {
        uint32_t v = t >= d ? part[t - d] : 0;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    if (i < nb) { offsets[i] = block_sums[blockIdx.x] + part[t] - c; cursor[i] = 0; }
}

// The sorted list is LOGICALLY dense — entry positions 0 .. E in bin order, which is what the equal-length chunks of msm_accumulate
// are cut from — but may be stored in SEGMENTS: segment s holds the bins [s * bps, (s + 1) * bps), its entries sit at
// list[s * stride + i], its first logical position is seg_start[s], and offsets[b] is the bin's start INSIDE its segment.  The LDS
// sort writes one segment per (problem, window) (each workgroup writes its own contiguous piece, no global prefix sum before the
// stores); the global counting sort is the one-segment case (bps = all bins, seg_start = {0, E}).
struct MsmSeg { const uint32_t* seg_start; uint32_t nseg, bps, stride, lanes_round; };
__device__ __forceinline__ uint32_t msm_bin_start(const MsmSeg& g, const uint32_t* __restrict__ offsets, uint32_t b) { return g.seg_start[b / g.bps] + offsets[b]; }
// the bin that holds logical position pos: the last bin whose start is <= pos (empty bins share the start of the next non-empty one)
__device__ __forceinline__ uint32_t msm_bin_of(const MsmSeg& g, const uint32_t* __restrict__ offsets, uint32_t pos) {
    uint32_t lo = 0, hi = g.nseg;
    while (hi - lo > 1) { uint32_t mid = (lo + hi) >> 1; if (g.seg_start[mid] <= pos) lo = mid; else hi = mid; }
    const uint32_t base = lo * g.bps, local = pos - g.seg_start[lo];
    lo = 0; hi = g.bps;
    while (hi - lo > 1) { uint32_t mid = (lo + hi) >> 1; if (offsets[base + mid] <= local) lo = mid; else hi = mid; }
    return base + lo;
}
// Where the pieces of chunk `lane` go: a piece that is a whole bucket -> bucket_pts[b]; otherwise the chunk's first piece ->
// partial[2*lane], its last -> partial[2*lane + 1] (a chunk has no other incomplete pieces).
__device__ __forceinline__ G1JSlot* msm_piece_dst(G1JSlot* __restrict__ bucket_pts, G1JSlot* __restrict__ partial, uint32_t lane, uint32_t b, uint32_t bin_lo, uint32_t bin_hi,
                                              uint32_t chunk_lo, uint32_t chunk_hi, bool first) {
    if (bin_lo >= chunk_lo && bin_hi <= chunk_hi) return bucket_pts + b;
    return partial + 2 * (size_t)lane + (first ? 0 : 1);
}
// complete (slow, call-based) group law for the rare chunk in which a point meets itself or its negative
__device__ __noinline__ void msm_chunk_slow(const MsmProblem* __restrict__ prs, uint32_t nbq, const uint32_t* __restrict__ counts, const uint32_t* __restrict__ offsets,
                                            const uint32_t* __restrict__ list, G1JSlot* __restrict__ bucket_pts, G1JSlot* __restrict__ partial, uint32_t nb, uint32_t lane, uint32_t E, MsmSeg g) {
    const uint32_t CH = msm_chunk_len(E, g.lanes_round);
    const uint32_t chunk_lo = lane * CH, chunk_hi = min(chunk_lo + CH, E);
    uint32_t b = msm_bin_of(g, offsets, chunk_lo);
    uint32_t bin_lo = msm_bin_start(g, offsets, b), bin_hi = bin_lo + counts[b];
    MsmProblem q = prs[b / nbq];
    G1J acc = G1J::identity();
    bool first = true;
    for (uint32_t pos = chunk_lo; pos < chunk_hi;) {
        const uint32_t sg = b / g.bps;
        acc = g1_add_affine(acc, msm_entry_base(q, list[(size_t)sg * g.stride + (pos - g.seg_start[sg])]));
        ++pos;
        if (pos == bin_hi || pos == chunk_hi) {
            *msm_piece_dst(bucket_pts, partial, lane, b, bin_lo, bin_hi, chunk_lo, chunk_hi, first) = acc;
            acc = G1J::identity(); first = false;
            if (pos == bin_hi && pos < chunk_hi) {
                { uint32_t tries = 0; do { ++b; } while (counts[b] == 0 && ++tries < 2); if (counts[b] == 0) b = msm_bin_of(g, offsets, pos); }   // (as in msm_accumulate_chunk)
                bin_lo = bin_hi; bin_hi = bin_lo + counts[b];
                q = prs[b / nbq];
            }
        }
    }
}
// (b, bin_lo, bin_hi) is the chunk's last bucket.  If it goes on past the chunk and BEGINS here, this lane enters it in the fix-up's
// work lists (`lists` = the scatter cursors, free by now): buckets that straddle chunks from the front (their number in control[2]),
// buckets spread over >= MSM_FIXUP_SERIAL chunks from the back (control[0]), the ones in between in the second half (control[3]).  A
// separate pass over all buckets to build the lists was 0.08 ms of a 20-step launch.
__device__ __forceinline__ void msm_chunk_tail(uint32_t b, uint32_t bin_lo, uint32_t bin_hi, uint32_t chunk_lo, uint32_t chunk_hi, uint32_t CH, uint32_t lane, uint32_t nb,
                                               uint32_t* __restrict__ control, uint32_t* __restrict__ lists) {
    if (bin_hi > chunk_hi && bin_lo >= chunk_lo) {
        const uint32_t span = (bin_hi - 1) / CH - lane;   // further chunks the bucket runs into
        if (span >= MSM_FIXUP_SERIAL) lists[nb - 1 - atomicAdd(&control[0], 1u)] = b;
        else if (span >= MSM_FIXUP_TEAM) lists[nb + atomicAdd(&control[3], 1u)] = b;   // summed by a team of lanes
        else lists[atomicAdd(&control[2], 1u)] = b;
    }
}
__device__ __forceinline__ void msm_accumulate_chunk(const MsmProblem* __restrict__ prs, uint32_t nbq, const uint32_t* __restrict__ counts, const uint32_t* __restrict__ offsets,
                                                     const uint32_t* __restrict__ list, G1JSlot* __restrict__ bucket_pts, G1JSlot* __restrict__ partial, uint32_t nb, const MsmSeg& g,
                                                     uint32_t* __restrict__ control, uint32_t* __restrict__ lists, const G1A* __restrict__ phi_pts, uint32_t* __restrict__ redo, uint32_t E, uint32_t CH,
                                                     uint32_t lane) {
    const uint32_t chunk_lo = lane * CH;
    if (chunk_lo >= E) return;
    const uint32_t chunk_hi = min(chunk_lo + CH, E);
    uint32_t b = msm_bin_of(g, offsets, chunk_lo);
    uint32_t sg = b / g.bps, seg_lo = g.seg_start[sg];      // the segment of the current bin and its first logical position
    uint32_t bin_lo = seg_lo + offsets[b], bin_hi = bin_lo + counts[b];
    uint32_t qi = b / nbq;
    MsmProblem q = prs[qi];
    G1J acc = G1J::identity();
    bool first = true, ok = true;
    // software pipeline: the (random-access) load of entry pos + 1 is in flight during the ~2500 instructions of addition pos
    uint32_t e_next = list[(size_t)sg * g.stride + (chunk_lo - seg_lo)];
    G1A raw_next = msm_entry_load(q, e_next, phi_pts);
    for (uint32_t pos = chunk_lo; pos < chunk_hi && ok;) {
        const uint32_t e = e_next;
        const G1A raw = raw_next;
        ++pos;
        const bool flush = pos == bin_hi || pos == chunk_hi;
        // bin (and problem) of the next entry
        uint32_t b_next = b, lo_next = bin_lo, hi_next = bin_hi, qn = qi;
        MsmProblem q_next = q;
        if (pos == bin_hi && pos < chunk_hi) {
            // pos < E: a later non-empty bin exists.  Normally the very next one; in a sparse problem (the one-term left channel of a
            // single proof: 2 entries among 64 bins per window) a bin-by-bin walk is a chain of dependent loads — 126 us of accumulation
            // for one proof where four proofs took 41 — so after two empty bins the bin of position pos is searched for instead
            uint32_t tries = 0;
            do { ++b_next; } while (counts[b_next] == 0 && ++tries < 2);
            if (counts[b_next] == 0) b_next = msm_bin_of(g, offsets, pos);
            lo_next = bin_hi; hi_next = lo_next + counts[b_next];   // logically the list is dense: the next bin starts where this one ends
            const uint32_t sn = b_next / g.bps;
            if (sn != sg) { sg = sn; seg_lo = g.seg_start[sn]; }
            qn = b_next / nbq;                                     // (the divisions only where the bin changes: once per ~25 entries)
            if (qn != qi) q_next = prs[qn];
        }
        // (unconditional: behind the chunk's last entry the same entry is loaded once more — inside an `if` the compiler waited for the
        // loads at the end of the block, i.e. BEFORE the addition they were meant to overlap.  Measured: 0.865 -> 0.857 ms; fetching the
        // list entry two iterations ahead as well, so that no iteration waits for a dependent pair of loads: no further change — the
        // second wave of the SIMD already covers these stalls)
        const uint32_t pos_n = min(pos, chunk_hi - 1);
        e_next = list[(size_t)sg * g.stride + (pos_n - seg_lo)]; raw_next = msm_entry_load(q_next, e_next, phi_pts);
        ok = g1_madd_fast(acc, msm_entry_apply(raw, e, phi_pts != nullptr));
        if (flush) {
            if (ok) *msm_piece_dst(bucket_pts, partial, lane, b, bin_lo, bin_hi, chunk_lo, chunk_hi, first) = acc;
            acc = G1J::identity(); first = false;
        }
        b = b_next; bin_lo = lo_next; bin_hi = hi_next; qi = qn; q = q_next;
    }
    // A chunk in which a point met itself or its negative is NOT redone here: the complete formulas live behind calls, and a kernel
    // is given the registers of its hungriest callee — with the slow path inside, this kernel was compiled for 252 VGPRs (two waves
    // per SIMD) although its loop needs 156 (three).  The lane lists its chunk (control[4]) and msm_accumulate_redo, a small launch
    // right behind this one, redoes the listed chunks with complete formulas (adversarial inputs only; the tests construct them).
    if (!ok) { redo[atomicAdd(&control[4], 1u)] = lane; return; }
    msm_chunk_tail(b, bin_lo, bin_hi, chunk_lo, chunk_hi, CH, lane, nb, control, lists);
}
// WPE = waves per SIMD the kernel is compiled for: 3 (156 registers, nothing spilled) is the default; 4 fits 128 registers with 30 of
// them spilled to scratch (h2v_tuning.msm_acc_waves; measured in DESIGN.md)
template <int WPE> __global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) msm_accumulate(const MsmProblem* __restrict__ prs, uint32_t nbq, const uint32_t* __restrict__ counts, const uint32_t* __restrict__ offsets,
                                                     const uint32_t* __restrict__ list, G1JSlot* __restrict__ bucket_pts, G1JSlot* __restrict__ partial, uint32_t nb, MsmSeg g,
                                                     uint32_t* __restrict__ control, uint32_t* __restrict__ lists, const G1A* __restrict__ phi_pts, uint32_t* __restrict__ redo) {
    const uint32_t E = counts[nb + 1];
    // Workgroups are dealt round-robin to the 8 XCDs, each with its own L2.  The list is sorted by (problem, window, bucket),
    // so giving XCD x the x-th eighth of the chunks keeps the bases an XCD gathers to one or two problems' points (~1 MB
    // each) instead of all of them (15 MB per 16-step launch): the gathers hit in L2 instead of going out to the fabric.
    const uint32_t CH = msm_chunk_len(E, g.lanes_round);
    const uint32_t blocks = ((E + CH - 1) / CH + 63) / 64, per_xcd = (blocks + 7) / 8;
    // The grid (a multiple of 8) covers the host's bound on the entry count (msm_accumulate_blocks); were there more entries than
    // promised (MsmProblem::nnz), a workgroup takes several blocks of chunks — slower, never wrong.
    for (uint32_t j = blockIdx.x / 8; j < per_xcd; j += gridDim.x / 8)
        msm_accumulate_chunk(prs, nbq, counts, offsets, list, bucket_pts, partial, nb, g, control, lists, phi_pts, redo, E, CH, ((blockIdx.x % 8) * per_xcd + j) * 64 + threadIdx.x);
}
// the chunks msm_accumulate gave up on (a point met itself or its negative inside a bucket), with the complete group law
__global__ void __launch_bounds__(64) msm_accumulate_redo(const MsmProblem* __restrict__ prs, uint32_t nbq, const uint32_t* __restrict__ counts, const uint32_t* __restrict__ offsets,
                                                          const uint32_t* __restrict__ list, G1JSlot* __restrict__ bucket_pts, G1JSlot* __restrict__ partial, uint32_t nb, MsmSeg g,
                                                          uint32_t* __restrict__ control, uint32_t* __restrict__ lists, const uint32_t* __restrict__ redo) {
    const uint32_t n_redo = control[4], E = counts[nb + 1], CH = msm_chunk_len(E, g.lanes_round);
    // every lane reaches the exit condition: the list is complete before this kernel starts
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n_redo; i += gridDim.x * blockDim.x) {
        const uint32_t lane = redo[i], chunk_lo = lane * CH, chunk_hi = min(chunk_lo + CH, E);
        msm_chunk_slow(prs, nbq, counts, offsets, list, bucket_pts, partial, nb, lane, E, g);
        const uint32_t b = msm_bin_of(g, offsets, chunk_hi - 1), bin_lo = msm_bin_start(g, offsets, b), bin_hi = bin_lo + counts[b];
        msm_chunk_tail(b, bin_lo, bin_hi, chunk_lo, chunk_hi, CH, lane, nb, control, lists);
    }
}

// the piece of bucket [off, off + cnt) that chunk i holds: its tail piece when the bucket starts inside the chunk, else its head piece
__device__ __forceinline__ const G1JSlot* msm_piece_src(const G1JSlot* __restrict__ partial, uint32_t i, uint32_t i0, uint32_t off, uint32_t CH) {
    return partial + 2 * (size_t)i + ((i == i0 && off > i0 * CH) ? 1 : 0);
}
__device__ __noinline__ void msm_fixup_slow(const G1JSlot* __restrict__ partial, uint32_t i0, uint32_t i1, uint32_t off, uint32_t CH, G1JSlot* __restrict__ out) {
    G1J acc = msm_piece_src(partial, i0, i0, off, CH)->p;
    for (uint32_t i = i0 + 1; i <= i1; ++i) acc = g1_add(acc, msm_piece_src(partial, i, i0, off, CH)->p);
    *out = acc;
}
// The work lists (buckets that straddle chunks) are built by msm_accumulate; the additions run in a second, dense launch that keeps
// its waves full: only ~40 % of the buckets straddle.
// Buckets that run over a few chunks (the top window's: a 7-bit digit, a few dozen buckets that take 1/64 of the entries each) are
// rare but scattered through the list: one lane per bucket, every second wave had one of them and waited for its 8 .. 16 sequential
// additions (0.13 - 0.45 ms for 0.03 ms of arithmetic).  They are summed by teams of eight lanes in the first workgroups of the launch.
// (Round 3 tried the teams as a kernel of their own — with their calls inside, msm_fixup is compiled for 280 + 32 registers where the
// lane path alone needs 113: the lane kernel then takes 48 us instead of sharing 77, but the team kernel is a 70 us chain of its own
// and the two run one after the other: 118 us.  Inside one launch the lane path hides under the teams' chain.)
__device__ __noinline__ void g1_add_to(G1J* dst, const G1J* a, const G1J* b);   // (defined with msm_window below)
__device__ __noinline__ void msm_fixup_team(const uint32_t* __restrict__ counts, const uint32_t* __restrict__ offsets, const G1JSlot* __restrict__ partial,
                                            const uint32_t* __restrict__ lists, G1JSlot* __restrict__ bucket_pts, uint32_t nb, const MsmSeg& g, G1J* team_acc) {
    const uint32_t n_team = counts[nb + 3], r = threadIdx.x & 7u;
    const uint32_t CH = msm_chunk_len(counts[nb + 1], g.lanes_round);
    // whole waves loop together (the shuffles below need all eight lanes of a team): the trip count is rounded up per wave
    for (uint32_t m0 = blockIdx.x * 8; m0 < n_team; m0 += MSM_FIXUP_TEAM_BLOCKS * 8) {
        const uint32_t m = m0 + (threadIdx.x >> 3);
        const bool live = m < n_team;
        const uint32_t b = live ? lists[nb + m] : 0;
        const uint32_t cnt = live ? counts[b] : 0, off = live ? msm_bin_start(g, offsets, b) : 0;
        const uint32_t i0 = off / CH, i1 = cnt ? (off + cnt - 1) / CH : i0;
        // (round 3, same box: taking a lane's first piece as it is instead of adding it to the identity made the kernel 84.5 us instead of
        // 77 — the complete addition returns early on an identity operand, the extra select did not pay; raising the teams' issue
        // priority over the lane path's waves changed nothing; at the end of the round the in-register fast addition was inlined here — one site
        // in a loop over a lane's own pieces and the three tree levels — in place of the calls: 198 us instead of 77, not kept;
        // 512 team blocks instead of 256 and two waves per SIMD for the kernel (248 registers): 79 us, no change)
        // The accumulators live in LDS and the additions are the out-of-line routine with explicit operands and destination (as in
        // msm_window): by value through g1_add, operands and result travelled through scratch memory, ~17 us per addition — a 70 us chain.
        // A tree level reads the neighbour's slot directly: a wave runs in lockstep, every lane has loaded its operands (first statement of
        // g1_add_to) before any lane stores its result.
        G1J* mine = team_acc + threadIdx.x;
        *mine = G1J::identity();
        if (live) for (uint32_t i = i0 + r; i <= i1; i += 8) g1_add_to(mine, mine, &msm_piece_src(partial, i, i0, off, CH)->p);
        for (uint32_t d = 4; d > 0; d >>= 1)
            g1_add_to(mine, mine, team_acc + min(threadIdx.x + d, 63u));   // lanes r >= 8 - d add a value they do not own: harmless, only r = 0 is kept
        if (live && r == 0) bucket_pts[b] = *mine;
    }
}
// Buckets spread over more than MSM_FIXUP_SERIAL chunks (skewed inputs: one digit value shared by thousands of scalars; none in a launch
// of ordinary proofs): one wave each, the pieces dealt to its 64 lanes, then a butterfly over the lanes.  (Until round 3 a launch of its
// own with 256-lane workgroups: 6 us of kernel boundary in every launch for a list that is almost always empty.)
__device__ __noinline__ void msm_fixup_heavy(const uint32_t* __restrict__ counts, const uint32_t* __restrict__ offsets, const G1JSlot* __restrict__ partial,
                                             const uint32_t* __restrict__ lists, G1JSlot* __restrict__ bucket_pts, uint32_t nb, const MsmSeg& g, uint32_t first_block) {
    const uint32_t n_heavy = counts[nb], t = threadIdx.x;
    const uint32_t CH = msm_chunk_len(counts[nb + 1], g.lanes_round);
    // every wave reaches the exit condition: the heavy list is complete before this kernel starts
    for (uint32_t h = blockIdx.x - first_block; h < n_heavy; h += MSM_FIXUP_HEAVY_BLOCKS) {
        const uint32_t b = lists[nb - 1 - h];
        const uint32_t cnt = counts[b], off = msm_bin_start(g, offsets, b);
        const uint32_t i0 = off / CH, i1 = (off + cnt - 1) / CH;
        G1J acc = G1J::identity();
        for (uint32_t i = i0 + t; i <= i1; i += 64) acc = g1_add(acc, msm_piece_src(partial, i, i0, off, CH)->p);
        for (uint32_t d = 32; d > 0; d >>= 1) {
            G1J other;
            uint32_t* dst = reinterpret_cast<uint32_t*>(&other);
            const uint32_t* src = reinterpret_cast<const uint32_t*>(&acc);
#pragma unroll
            for (uint32_t k = 0; k < sizeof(G1J) / 4; ++k) dst[k] = (uint32_t)__shfl_down((int)src[k], d, 64);
            acc = g1_add(acc, other);   // lanes t >= 64 - d add a value they do not own: harmless, only lane 0 is kept
        }
        if (t == 0) bucket_pts[b] = acc;
    }
}
__global__ void __launch_bounds__(64) msm_fixup(const uint32_t* __restrict__ counts, const uint32_t* __restrict__ offsets, const G1JSlot* __restrict__ partial,
                                                const uint32_t* __restrict__ lists, G1JSlot* __restrict__ bucket_pts, uint32_t nb, MsmSeg g) {
    __shared__ G1J team_acc[64];   // the teams' accumulators (7 KB; the lane path keeps its own in registers)
    if (blockIdx.x < MSM_FIXUP_TEAM_BLOCKS) { msm_fixup_team(counts, offsets, partial, lists, bucket_pts, nb, g, team_acc); return; }
    if (blockIdx.x < MSM_FIXUP_TEAM_BLOCKS + MSM_FIXUP_HEAVY_BLOCKS) { msm_fixup_heavy(counts, offsets, partial, lists, bucket_pts, nb, g, MSM_FIXUP_TEAM_BLOCKS); return; }
    const uint32_t k = (blockIdx.x - MSM_FIXUP_TEAM_BLOCKS - MSM_FIXUP_HEAVY_BLOCKS) * blockDim.x + threadIdx.x;
    if (k >= counts[nb + 2]) return;
    const uint32_t b = lists[k];
    const uint32_t cnt = counts[b], off = msm_bin_start(g, offsets, b);
    const uint32_t CH = msm_chunk_len(counts[nb + 1], g.lanes_round);
    const uint32_t i0 = off / CH, i1 = (off + cnt - 1) / CH;
    // in-register additions; pieces of one bucket can coincide or cancel (the same point in two chunks): complete formulas then
    G1J acc = msm_piece_src(partial, i0, i0, off, CH)->p;
    bool ok = true;
    for (uint32_t i = i0 + 1; i <= i1 && ok; ++i) ok = g1_add_fast(acc, msm_piece_src(partial, i, i0, off, CH)->p);
    if (!ok) { msm_fixup_slow(partial, i0, i1, off, CH, bucket_pts + b); return; }
    bucket_pts[b] = acc;
}
#define MSM_WIN_SLOTS 5   // LDS points per lane of msm_window: running sum, weighted sum, scaled (later the tree), 2 run, 3 run
// The out-of-line group law with explicit destinations: operands are loaded from wherever they live (LDS here), the
// result is stored where the caller says — no hidden return-value temporaries in scratch memory.
__device__ __noinline__ void g1_add_to(G1J* dst, const G1J* a, const G1J* b) { const G1J x = *a, y = *b; *dst = g1_add_inl(x, y); }
__device__ __noinline__ void g1_dbl_to(G1J* dst, const G1J* a) { const G1J x = *a; *dst = g1_dbl_inl(x); }

// A lone wave per window is latency-bound, and the two inlined additions of the running-sum step are ~70 KB of code —
// more than the instruction cache, so every iteration streamed its code from L2 (measured 2x the time of the arithmetic).
// Here the group law is the shared out-of-line routine (one 35 KB body, complete formulas) and the three accumulators of a
// lane live in LDS (dynamic: MSM_WIN_SLOTS x T points; the tree reuses the `scaled` slots): kept in private memory across the calls they cost
// 0.35 GB of scratch write-backs per launch.
// Empty buckets are recognised by their count: nobody writes an identity into their slots.
// A workgroup reduces `wpw` windows, T = blockDim.x / wpw lanes each (T a power of two, msm_window_threads).
__global__ void __launch_bounds__(MSM_WIN_THREADS) msm_window(const G1JSlot* __restrict__ bucket_pts, const uint32_t* __restrict__ counts, G1JSlot* __restrict__ window_sums, MsmPlan p,
                                                              uint32_t n_windows, uint32_t wpw, uint32_t slots) {
    __builtin_amdgcn_s_setprio(3);   // a latency chain: its waves win the issue arbitration over the throughput kernels of other launches in flight
    extern __shared__ G1J win_lds[];
    const uint32_t T = blockDim.x / wpw, sub = threadIdx.x / T, t = threadIdx.x % T;
    const uint32_t widx = blockIdx.x * wpw + sub;          // (problem, window) = widx / windows, widx % windows
    const bool live = widx < n_windows;
    G1J* mine = win_lds + (size_t)sub * slots * T;     // slots = MSM_WIN_SLOTS, or 3 (no digit table) when the launch wants many small workgroups per CU
    G1J* run = mine + t;
    G1J* sum = mine + T + t;
    G1J* scaled = mine + 2 * T + t;
    G1J* red = mine + 2 * T;              // the tree of T > 64 reuses the slots of `scaled` (each lane's own slot: dead by then)
    const uint32_t slice = (p.buckets + T - 1) / T;
    const uint32_t lo = live ? min(p.buckets, t * slice) : 0, hi = live ? min(p.buckets, lo + slice) : 0;
    *run = G1J::identity(); *sum = G1J::identity();
    const G1JSlot* bp = bucket_pts + (size_t)widx * p.buckets;
    const uint32_t* cn = counts + (size_t)widx * p.buckets;
    for (uint32_t b = hi; b > lo; --b) {
        if (cn[b - 1]) g1_add_to(run, run, &bp[b - 1].p);
        g1_add_to(sum, sum, run);
    }
    // sum = sum_{b in slice} (b - lo + 1) B_b ; the bucket's weight is (b + 1): add lo * (sum of the slice)
    if ((slice & (slice - 1)) == 0 && T > 1 && slots >= MSM_WIN_SLOTS) {
        // lo = t * slice with slice a power of two (buckets and T are): (t * run) by FIXED two-bit digits of t against the table
        // {run, 2 run, 3 run}, then log2(slice) doublings.  A wave runs in lockstep: with the bit-serial double-and-add below some lane
        // of the 64 had a one at every position, so the wave paid a doubling AND an addition per bit of its largest lo (10 of each at
        // T = 128) whatever the lanes' own bit counts (a non-adjacent form changed nothing, measured); two-bit digits are 5 additions
        // and 10 doublings for every lane alike.  (Measured without this phase: 0.255 of the kernel's 0.35 ms.)
        // (Round 3 replaced it by a suffix scan of the slice totals across the lanes — sum_t t run_t = sum_{t >= 1} sum_{u >= t} run_u:
        // 7 additions + 3 doublings + 1 addition per lane instead of 5 additions + 10 doublings — and measured no change, 0.34 - 0.37 ms
        // either way: a doubling is half an addition, and the scan needs eight workgroup barriers.  Not kept.)
        G1J* run2 = mine + 3 * T + t;
        G1J* run3 = mine + 4 * T + t;
        if (lo < hi && t > 0) {
            g1_dbl_to(run2, run);
            g1_add_to(run3, run2, run);
            *scaled = G1J::identity();
            const int nd = (32 - (int)__clz((int)(T - 1)) + 1) / 2;      // two-bit digits of t < T
            for (int d = nd - 1; d >= 0; --d) {
                if (d < nd - 1) { g1_dbl_to(scaled, scaled); g1_dbl_to(scaled, scaled); }
                const uint32_t dig = (t >> (2 * d)) & 3u;
                if (dig) g1_add_to(scaled, scaled, dig == 1 ? run : (dig == 2 ? run2 : run3));
            }
            for (uint32_t w = slice; w > 1; w >>= 1) g1_dbl_to(scaled, scaled);
            g1_add_to(sum, sum, scaled);
        }
    } else if (lo < hi && lo > 0) {
        // the doublings in front of lo's highest set bit would act on the identity: start there (the wave runs as many rounds as its largest lo needs)
        const int top = 31 - (int)__clz((int)lo);
        *scaled = G1J::identity();
        for (int i = top; i >= 0; --i) {
            if (i < top) g1_dbl_to(scaled, scaled);
            if ((lo >> i) & 1) g1_add_to(scaled, scaled, run);
        }
        g1_add_to(sum, sum, scaled);
    }
    if (blockDim.x <= 64) {
        // a single wave: butterfly over the lanes with cross-lane moves (27 dwords per step), no barrier;
        // lanes beyond T hold the identity
        for (uint32_t d = 32; d > 0; d >>= 1) {
            const G1J mine_sum = *sum;
            G1J other;
            uint32_t* dst = reinterpret_cast<uint32_t*>(&other);
            const uint32_t* src = reinterpret_cast<const uint32_t*>(&mine_sum);
#pragma unroll
            for (uint32_t k = 0; k < sizeof(G1J) / 4; ++k) dst[k] = (uint32_t)__shfl_down((int)src[k], d, 64);
            if (t + d >= T) other = G1J::identity();
            *scaled = other;
            g1_add_to(sum, sum, scaled);
        }
        if (t == 0 && live) window_sums[widx] = *sum;
        return;
    }
    // several waves per window: the same butterfly inside every wave (no barrier: the tree over all T lanes through LDS was seven
    // additions with a workgroup barrier each, 200 000 cycles where the additions alone are 147 000), then the waves' leaders through LDS
    const uint32_t W = T < 64u ? T : 64u;  // lanes of this window inside one wave (T is a power of two)
    for (uint32_t d = W / 2; d > 0; d >>= 1) {
        const G1J mine_sum = *sum;
        G1J other;
        uint32_t* dst = reinterpret_cast<uint32_t*>(&other);
        const uint32_t* src = reinterpret_cast<const uint32_t*>(&mine_sum);
#pragma unroll
        for (uint32_t k = 0; k < sizeof(G1J) / 4; ++k) dst[k] = (uint32_t)__shfl_down((int)src[k], d, 64);
        if ((t & (W - 1)) + d >= W) other = G1J::identity();
        *scaled = other;
        g1_add_to(sum, sum, scaled);
    }
    __syncthreads();                       // every lane is done with its `scaled` slot: the first T / 64 of them now carry the waves' sums
    if ((t & (W - 1)) == 0) red[t >> 6] = *sum;
    __syncthreads();
    if (t == 0) {
        for (uint32_t wv = 1; wv < (T >> 6); ++wv) g1_add_to(&red[0], &red[0], &red[wv]);
        if (live) window_sums[widx] = red[0];
    }
}

// ---- msm_final: Horner over the windows — c doublings and one addition per window, ~130 dependent group operations, nothing
// to parallelise ACROSS operations.  Four lanes (a quad) work on one problem and split the inside of a doubling instead:
//   level 1   lane 0: A = X^2     lane 1: B = Y^2         lane 2: Z3 = (2Y) Z
//   level 2   lane 0: C = B^2     lane 1: X B             lane 2: (3A)^2
//   level 3   every lane: X3 (a carry sweep), Y3 = E (D - X3) - 8C (one dot2)
// three products deep instead of seven, values exchanged by DPP quad broadcasts (no LDS).  X, Y, Z are replicated in the four
// lanes; the one addition per window is done redundantly by all of them (nothing to exchange).
template <int S> __device__ __forceinline__ Fq quad_bcast(const Fq& v) {
    Fq r;
#pragma unroll
    for (int l = 0; l < H2V_LIMBS; ++l) r.v[l] = (uint32_t)__builtin_amdgcn_mov_dpp((int)v.v[l], S * 0x55, 0xf, 0xf, true);   // quad_perm [S,S,S,S]
    return r;
}
__device__ __forceinline__ Fq fq_sel(bool c, const Fq& a, const Fq& b) {
    Fq r;
#pragma unroll
    for (int l = 0; l < H2V_LIMBS; ++l) r.v[l] = c ? a.v[l] : b.v[l];
    return r;
}
__device__ __forceinline__ void g1_dbl_quad(G1J& p, uint32_t r) {
    // g1_dbl_inl's formulas (curve.hip.h: lazy linear forms, D = 4 X B as a product, X3 through one carry sweep, Y3 one dot2), the
    // products of a level on different lanes.  The identity (Z = 0) stays the identity: Z3 = (2Y) Z.
    // level 1   lane 0: A = X^2   lane 1: B = Y^2   lanes 2, 3: Z3 = (2Y) Z
    const Fq Y2 = Fq::lazy_dbl(p.Y);
    const Fq p1 = Fq::mul_inl(fq_sel(r == 0, p.X, fq_sel(r == 1, p.Y, Y2)), fq_sel(r == 0, p.X, fq_sel(r == 1, p.Y, p.Z)));
    const Fq A = quad_bcast<0>(p1), B = quad_bcast<1>(p1), Z3 = quad_bcast<2>(p1);
    // level 2   lane 0: C = B^2   lane 1: X B   lane 2: E^2, E = 3A   (lane 3: E B, unused)
    const Fq E = Fq::lazy_add2(A, A);
    const Fq p2 = Fq::mul_inl(fq_sel(r == 0, B, fq_sel(r == 1, p.X, E)), fq_sel(r == 2, E, B));
    const Fq C = quad_bcast<0>(p2), XB = quad_bcast<1>(p2), F = quad_bcast<2>(p2);
    // level 3   every lane: X3 = E^2 - 8 X B, Y3 = E (4 X B - X3) - 8 C
    const Fq D = Fq::lazy_dbl(Fq::lazy_dbl(XB));
    int64_t acc[9];
#pragma unroll
    for (int l = 0; l < 9; ++l) acc[l] = (int64_t)F.v[l] + (int64_t)Fq::KP29(9, l) - 2 * (int64_t)D.v[l];
    p.X = Fq::from_wide(acc);
    p.Y = Fq::dot2_inl(E, Fq::lazy_sub(D, p.X), C, g1_minus_eight());
    p.Z = Z3;
}
// Horner over the points src[0 .. items) (src[i] weighs 2^(dbl * i)) by the quad that lane r belongs to
__device__ __forceinline__ G1J msm_horner_quad(const G1JSlot* __restrict__ src, uint32_t items, uint32_t dbl, uint32_t r) {
    G1J acc = src[items - 1].p;
    for (int w = (int)items - 2; w >= 0; --w) {
        const G1J cur = src[w].p;   // in flight during the doublings
#pragma unroll 1
        for (uint32_t i = 0; i < dbl; ++i) g1_dbl_quad(acc, r);
        acc = g1_add_inl(acc, cur);
    }
    return acc;
}
__global__ void __launch_bounds__(64) msm_final(const G1JSlot* __restrict__ window_sums, const MsmProblem* __restrict__ prs, uint32_t count, MsmPlan p) {
    __builtin_amdgcn_s_setprio(3);   // a latency chain: its waves win the issue arbitration over the throughput kernels of other launches in flight
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x, q = t >> 2, r = t & 3u;
    if (q >= count) return;   // whole quads
    G1J acc = G1J::identity();
    if (prs[q].n) acc = msm_horner_quad(window_sums + (size_t)q * p.windows, p.windows, p.c, r);
    if (r == 0) *prs[q].out = acc;
}
// The same Horner cut into `parts` pieces of `wpp` windows: part j = sum over its windows of 2^(c (w - j wpp)) S_w, so that
//   result = sum_j 2^(c wpp j) part_j.
// A batch's pairing check takes the pieces as they are — e(2^k A, Q) = e(A, 2^k Q), and the multiples of the two fixed G2 points
// are tables of the context (pairing.hip) — so the ~130 dependent doublings of the full Horner leave the launch's critical path:
// a piece is (wpp - 1) c of them.  The full sum (the accumulator a caller can read back) is put together beside the pairing.
__global__ void __launch_bounds__(64) msm_final_parts(const G1JSlot* __restrict__ window_sums, const MsmProblem* __restrict__ prs, uint32_t count, MsmPlan p,
                                                      uint32_t parts, uint32_t wpp, G1JSlot* __restrict__ out, G1JSlot* __restrict__ ready) {
    __builtin_amdgcn_s_setprio(3);   // a latency chain: its waves win the issue arbitration over the throughput kernels of other launches in flight
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x, quad = t >> 2, r = t & 3u;
    if (quad >= count * parts) return;   // whole quads
    const uint32_t q = quad / parts, j = quad % parts, lo = j * wpp, hi = min(p.windows, lo + wpp);
    G1J acc = G1J::identity();
    if (prs[q].n && lo < hi) acc = msm_horner_quad(window_sums + (size_t)q * p.windows + lo, hi - lo, p.c, r);
    if (r == 0) out[quad] = acc;
    // the form the pairing's lines are evaluated at, (X Z, Y, Z^3): two lanes of the quad, two products deep
    if (r == 0) { ready[quad].p.X = Fq::mul_inl(acc.X, acc.Z); ready[quad].p.Y = acc.Y; }
    if (r == 1) ready[quad].p.Z = Fq::mul_inl(acc.Z.sqr_inl(), acc.Z);
}
__global__ void __launch_bounds__(64) msm_combine_parts(const G1JSlot* __restrict__ pieces, const MsmProblem* __restrict__ prs, uint32_t count, uint32_t parts, uint32_t shift) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x, q = t >> 2, r = t & 3u;
    if (q >= count) return;
    const G1J acc = msm_horner_quad(pieces + (size_t)q * parts, parts, shift, r);
    if (r == 0) *prs[q].out = acc;
}

// window sums of a caller's problem = the sums of its sub-problems' window sums, window by window (the same weights): a team of
// eight lanes per (problem, window)
__global__ void __launch_bounds__(64) msm_merge_windows(const G1JSlot* __restrict__ sub_sums, const MsmProblem* __restrict__ parents, uint32_t n_parents, uint32_t windows,
                                                        G1JSlot* __restrict__ merged) {
    const uint32_t team = (blockIdx.x * blockDim.x + threadIdx.x) >> 3, r = threadIdx.x & 7u;
    const bool live = team < n_parents * windows;
    const uint32_t q = live ? team / windows : 0, w = live ? team % windows : 0;
    const uint32_t first = parents[q].sub_first, cnt = live ? parents[q].sub_count : 0;
    G1J acc = G1J::identity();
    for (uint32_t k = r; k < cnt; k += 8) acc = g1_add(acc, sub_sums[(size_t)(first + k) * windows + w].p);
    for (uint32_t d = 4; d > 0; d >>= 1) {
        G1J other;
        uint32_t* dst = reinterpret_cast<uint32_t*>(&other);
        const uint32_t* src = reinterpret_cast<const uint32_t*>(&acc);
#pragma unroll
        for (uint32_t k = 0; k < sizeof(G1J) / 4; ++k) dst[k] = (uint32_t)__shfl_down((int)src[k], d, 8);
        acc = g1_add(acc, other);   // lanes r >= 8 - d add a value they do not own: only r = 0 is kept
    }
    if (live && r == 0) merged[(size_t)q * windows + w] = acc;
}

// descriptors travel as kernel arguments, a chunk at a time: no host staging buffer whose lifetime the caller would have to manage
__global__ void msm_set_problems(MsmProblemChunk ch, uint32_t count, MsmProblem* __restrict__ dst) {
    uint32_t i = threadIdx.x;
    if (i < count) dst[i] = ch.p[i];
}

// terms [first, first + len) of a problem as a problem of its own (no output of its own)
static MsmProblem msm_problem_slice(const MsmProblem& q, uint32_t first, uint32_t len) {
    MsmProblem r = q;
    r.out = nullptr; r.n = len; r.nnz = 0; r.sub_first = 0; r.sub_count = 0;
    if (first < q.n1) {
        r.scalars = q.scalars + (size_t)q.sstride * first; r.bases = q.bases + (size_t)q.bstride * first;
        if (q.phi) r.phi = q.phi + (size_t)q.bstride * first;
        r.n1 = std::min(q.n1 - first, len);
    } else {
        r.scalars = q.scalars2 + (size_t)q.sstride * (first - q.n1); r.bases = q.bases2 + (size_t)q.bstride * (first - q.n1);
        r.phi = q.phi2 ? q.phi2 + (size_t)q.bstride * (first - q.n1) : nullptr;
        r.n1 = len; r.scalars2 = nullptr; r.bases2 = nullptr; r.phi2 = nullptr;
    }
    return r;
}
static void msm_upload_problems(hipStream_t s, const std::vector<MsmProblem>& v, bool assign_glv, MsmProblem* dst) {
    uint32_t glv_next = 0;
    const uint32_t count = (uint32_t)v.size();
    for (uint32_t q0 = 0; q0 < count; q0 += MSM_PROBLEM_CHUNK) {
        MsmProblemChunk ch;
        uint32_t k = std::min<uint32_t>(MSM_PROBLEM_CHUNK, count - q0);
        for (uint32_t i = 0; i < k; ++i) { ch.p[i] = v[q0 + i]; if (assign_glv) { ch.p[i].glv_off = glv_next; glv_next += v[q0 + i].n; } }
        hipLaunchKernelGGL(msm_set_problems, dim3(1), dim3(64), 0, s, ch, k, dst + q0);
    }
}

// the shape of a launch: the caller's problems as they are, or cut into sub-problems
struct MsmLaunchShape { std::vector<MsmProblem> launch_p, parents_p; bool cut = false; uint32_t nmax = 0; size_t total = 0, total_nz = 0; };
static int msm_shape(const MsmWorkspace& ws, const MsmProblems& pr, MsmLaunchShape& L) {
    const uint32_t n_callers = (uint32_t)pr.p.size();
    if (n_callers > MSM_MAX_PROBLEMS || n_callers > std::max(ws.cap_problems, ws.cap_parents)) { set_last_error("msm_enqueue_multi: too many problems"); return H2V_ERR_BAD_ARGUMENT; }
    for (uint32_t q = 0; q < n_callers; ++q) { L.nmax = std::max(L.nmax, pr.p[q].n); L.total += pr.p[q].n; L.total_nz += pr.p[q].nnz ? std::min(pr.p[q].nnz, pr.p[q].n) : pr.p[q].n; }
    // A problem too large for the per-window LDS sort is cut into equal sub-problems that are not: the launch then looks like a
    // grouped batch (LDS sort, short buckets, many narrow window reductions side by side) — for one 8192-proof batch the uncut form
    // spent 0.77 ms in the global counting sort, 0.59 in the fix-up and 0.52 in two 4096-bucket window reductions.  The window sums
    // of a problem's sub-problems are added up (msm_merge_windows) before the Horner.
    if (L.nmax > MSM_LDS_SORT_MAX_TERMS && !ws.tune.msm_no_term_split) {
        size_t subs = 0;
        for (uint32_t q = 0; q < n_callers; ++q) subs += msm_subproblems(pr.p[q].n);
        L.cut = subs <= ws.cap_problems && subs <= MSM_MAX_PROBLEMS && n_callers <= ws.cap_parents;
    }
    if (L.cut) {
        for (uint32_t q = 0; q < n_callers; ++q) {
            const MsmProblem& c = pr.p[q];
            const uint32_t k = msm_subproblems(c.n), per = (c.n + k - 1) / k;
            MsmProblem parent = c;
            parent.sub_first = (uint32_t)L.launch_p.size(); parent.sub_count = k;
            for (uint32_t i = 0; i < k; ++i) { const uint32_t first = std::min(c.n, i * per); L.launch_p.push_back(msm_problem_slice(c, first, std::min(per, c.n - first))); }
            L.parents_p.push_back(parent);
        }
        L.nmax = 0;
        for (const MsmProblem& q : L.launch_p) L.nmax = std::max(L.nmax, q.n);
    } else {
        if (n_callers > ws.cap_problems) { set_last_error("msm_enqueue_multi: too many problems"); return H2V_ERR_BAD_ARGUMENT; }
        L.launch_p = pr.p;
    }
    return 0;
}
static bool msm_same_problems(const std::vector<MsmProblem>& a, const std::vector<MsmProblem>& b) {
    if (a.size() != b.size()) return false;
    for (size_t i = 0; i < a.size(); ++i) {
        const MsmProblem &x = a[i], &y = b[i];
        if (x.scalars != y.scalars || x.bases != y.bases || x.out != y.out || x.sstride != y.sstride || x.bstride != y.bstride || x.n != y.n || x.n1 != y.n1 ||
            x.scalars2 != y.scalars2 || x.bases2 != y.bases2 || x.phi != y.phi || x.phi2 != y.phi2 || x.nnz != y.nnz) return false;
    }
    return true;
}
// The problem descriptors depend on addresses and sizes only: a caller that knows them before the scalars exist (the batch verifier,
// while the Fr program still runs) hands them to the device early — on the stream of the MSM in front of the kernel that produces the
// scalars, or on a stream that is joined into it before the MSM (the batch verifier's auxiliary stream, beside the decompression) — and
// msm_enqueue_multi finds them there (two 5 us launches and a kernel boundary off the chain behind the Fr program).
int msm_prepare_problems(hipStream_t s, MsmWorkspace& ws, const MsmProblems& pr) {
    ws.prepared.clear();
    if (pr.p.empty()) return 0;
    MsmLaunchShape L;
    int rc = msm_shape(ws, pr, L);
    if (rc) return rc;
    if (L.cut) msm_upload_problems(s, L.parents_p, false, ws.parents);
    msm_upload_problems(s, L.launch_p, true, ws.problems);
    H2V_HIP_CHECK(hipGetLastError());
    ws.prepared = pr.p;
    return 0;
}

int msm_enqueue_multi(hipStream_t s, MsmWorkspace& ws, const MsmProblems& pr, MsmSplit* split) {
    if (split) { split->parts = 0; split->shift = 0; split->count = 0; split->pts = nullptr; split->ready = nullptr; }
    const uint32_t n_callers = (uint32_t)pr.p.size();
    if (n_callers == 0) return 0;
    MsmLaunchShape L;
    { int rc = msm_shape(ws, pr, L); if (rc) return rc; }
    const bool cut = L.cut;
    uint32_t nmax = L.nmax; const size_t total = L.total, total_nz = L.total_nz;
    const std::vector<MsmProblem>& launch_p = L.launch_p;
    const bool uploaded = msm_same_problems(ws.prepared, pr.p);   // msm_prepare_problems, earlier in stream order
    ws.prepared.clear();
    if (!uploaded) {
        if (cut) msm_upload_problems(s, L.parents_p, false, ws.parents);
        msm_upload_problems(s, launch_p, true, ws.problems);
    }
    const uint32_t count = (uint32_t)launch_p.size();
    ws.final_problems = cut ? ws.parents : ws.problems;
    if (nmax == 0) {
        hipLaunchKernelGGL(msm_final, dim3((4 * count + 63) / 64), dim3(64), 0, s, ws.window_sums, ws.problems, count, MsmPlan{0, 2, 0, 3});
        H2V_HIP_CHECK(hipGetLastError());
        return 0;
    }
    if (nmax > MSM_ENTRY_TERM) { set_last_error("msm_enqueue_multi: more than 2^30 terms in one problem"); return H2V_ERR_BAD_ARGUMENT; }
    if (total > ws.cap_terms) { set_last_error("msm_enqueue_multi: terms exceed workspace capacity"); return H2V_ERR_BAD_ARGUMENT; }
    MsmPlan p = msm_plan(nmax, msm_latency_bound(total));
    uint32_t nbq = p.windows * p.buckets, nb = nbq * count;
    if (nb > ws.cap_buckets || total * 2 * p.windows > ws.cap_list) { set_last_error("msm_enqueue_multi: workspace too small"); return H2V_ERR_BAD_ARGUMENT; }
    // the sort: per (problem, window) inside LDS when a window's entries fit there, else the global counting sort
    const uint32_t stride = 2 * nmax;
    const size_t sort_lds = ((size_t)p.buckets + MSM_SORT_THREADS) * 4 + (size_t)stride * 2;
    const bool lds_sort = nmax <= MSM_LDS_SORT_MAX_TERMS && sort_lds <= 150 * 1024 && (size_t)count * p.windows * stride <= ws.cap_list &&
                          (size_t)count * p.windows <= (size_t)128 * ws.cap_problems && !ws.tune.msm_global_sort;
    MsmSeg g;
    const uint32_t lanes_round = ws.tune.msm_acc_waves == 4 ? 262144u : MSM_ACC_LANES_PER_ROUND;
    if (lds_sort) {
        hipLaunchKernelGGL(msm_glv_prep, dim3((nmax + 255) / 256, count), dim3(256), 0, s, ws.problems, count, p, ws.glv, ws.phi_pts);
        if (sort_lds > 64 * 1024) H2V_HIP_CHECK(hipFuncSetAttribute((const void*)msm_sort_lds, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sort_lds));
        hipLaunchKernelGGL(msm_sort_lds, dim3(p.windows, count), dim3(MSM_SORT_THREADS), sort_lds, s, ws.problems, ws.glv, p, stride, ws.counts, ws.offsets, ws.list, ws.seg_total);
        hipLaunchKernelGGL(msm_seg_scan, dim3(1), dim3(1024), 0, s, ws.seg_total, p.windows * count, ws.seg_start, ws.counts + nb);
        g = MsmSeg{ws.seg_start, p.windows * count, p.buckets, stride, lanes_round};
    } else {
    H2V_HIP_CHECK(hipMemsetAsync(ws.counts, 0, ((size_t)nb + MSM_CONTROL_WORDS) * 4, s));
    H2V_HIP_CHECK(hipMemsetAsync(ws.seg_start, 0, 4, s));   // one segment that starts at 0
    const uint32_t tiles = (nmax + MSM_TILE - 1) / MSM_TILE;
    dim3 gt(8 * ((count + 7) / 8) * tiles);
    const uint32_t wpp = std::max<uint32_t>(1u, std::min<uint32_t>(p.windows, MSM_LDS_WORDS / p.buckets));
    const size_t lds = (size_t)wpp * p.buckets * 4;
    hipLaunchKernelGGL(msm_count_or_scatter<false>, gt, dim3(MSM_TILE_THREADS), lds, s, ws.problems, count, tiles, p, wpp, ws.counts, ws.offsets, ws.cursor, ws.list);
    const uint32_t nblk = (nb + 1023) / 1024;
    hipLaunchKernelGGL(msm_block_sums, dim3(nblk), dim3(1024), 0, s, ws.counts, ws.block_sums, nb);
    hipLaunchKernelGGL(msm_scan_sums, dim3(1), dim3(1024), 0, s, ws.block_sums, nblk, ws.counts + nb + 1);
    hipLaunchKernelGGL(msm_offsets, dim3(nblk), dim3(1024), 0, s, ws.counts, ws.block_sums, ws.offsets, ws.cursor, nb);
    hipLaunchKernelGGL(msm_count_or_scatter<true>, gt, dim3(MSM_TILE_THREADS), lds, s, ws.problems, count, tiles, p, wpp, ws.counts, ws.offsets, ws.cursor, ws.list);
    g = MsmSeg{ws.seg_start, 1, nb, 0, lanes_round};
    }
    // one lane per chunk of the sorted list; the entry count is only known on the device, the grid covers the host's bound on it.
    // Surplus workgroups are not free: the kernel holds exactly its occupancy in working workgroups (3 waves per SIMD), so the
    // surplus is dispatched after they retire, ~7 ns each — the old bound (every term non-zero, shortest chunk) cost 9 500 empty
    // workgroups, 0.07 ms, at the end of every 20-step launch.
    const uint32_t acc_blocks = msm_accumulate_blocks(total_nz * 2 * p.windows, lanes_round);
    {
        const G1A* phi_tab = lds_sort ? (const G1A*)ws.phi_pts : (const G1A*)nullptr;
        auto kern = ws.tune.msm_acc_waves == 4 ? msm_accumulate<4> : msm_accumulate<3>;
        // (the profiling events are attached to the dispatch itself — its own start and stop timestamps — instead of being recorded around it:
        // a recorded event is a barrier packet, ~6 us of idle stream on either side of the kernel)
        if (ws.profile) {
            hipExtLaunchKernelGGL(kern, dim3(acc_blocks), dim3(64), 0, s, ws.ev_acc[0], ws.ev_acc[1], 0, ws.problems, nbq, ws.counts, ws.offsets, ws.list, ws.bucket_pts, ws.partial, nb, g, ws.counts + nb, ws.cursor, phi_tab, ws.redo);
            ws.profile_recorded = true;
        } else hipLaunchKernelGGL(kern, dim3(acc_blocks), dim3(64), 0, s, ws.problems, nbq, ws.counts, ws.offsets, ws.list, ws.bucket_pts, ws.partial, nb, g, ws.counts + nb, ws.cursor, phi_tab, ws.redo);
    }
    hipLaunchKernelGGL(msm_accumulate_redo, dim3(256), dim3(64), 0, s, ws.problems, nbq, ws.counts, ws.offsets, ws.list, ws.bucket_pts, ws.partial, nb, g, ws.counts + nb, ws.cursor, ws.redo);
    hipLaunchKernelGGL(msm_fixup, dim3((nb + 63) / 64 + MSM_FIXUP_TEAM_BLOCKS + MSM_FIXUP_HEAVY_BLOCKS), dim3(64), 0, s, ws.counts, ws.offsets, ws.partial, ws.cursor, ws.bucket_pts, nb, g);
    {
        // Two-wave workgroups land on overlapping SIMD pairs when a CU holds two of them (measured: 0.57 ms for what one wave per
        // window does in 0.48), so beyond 256 windows a workgroup is FOUR waves reducing two windows, two waves each: every wave
        // has a SIMD of its own as long as the launch has at most one workgroup per CU.
        const uint32_t nw = p.windows * count;
        uint32_t T = msm_window_threads(p.buckets, nw, ws.tune.msm_window_threads), wpw = 1;
        if (T == 64 && p.buckets >= 256 && nw > 256 && nw <= 512 && !ws.tune.msm_window_threads) { T = 128; wpw = 2; }
        const uint32_t wpw_forced = (uint32_t)ws.tune.msm_window_wpw;   // h2v_tuning.msm_window_wpw
        if ((wpw_forced == 1 || wpw_forced == 2 || wpw_forced == 4) && T * wpw_forced <= MSM_WIN_THREADS) wpw = wpw_forced;
        // (one-wave workgroups beyond four per CU — launches of more than ~40 groups — keep the 20 KB form without the digit table: 34 KB each would not fit side by side)
        const uint32_t slots = ((T * wpw <= 64 && nw > 1024) || ws.tune.msm_window_slots == 3) ? 3u : (uint32_t)MSM_WIN_SLOTS;
        const size_t win_lds = (size_t)slots * T * wpw * sizeof(G1J);   // 34 KB for one wave, 135 KB for four
        if (win_lds > 64 * 1024) H2V_HIP_CHECK(hipFuncSetAttribute((const void*)msm_window, hipFuncAttributeMaxDynamicSharedMemorySize, (int)win_lds));
        hipLaunchKernelGGL(msm_window, dim3((nw + wpw - 1) / wpw), dim3(T * wpw), win_lds, s, ws.bucket_pts, ws.counts, ws.window_sums, p, nw, wpw, slots);
    }
    const G1JSlot* sums = ws.window_sums;
    if (cut) {
        hipLaunchKernelGGL(msm_merge_windows, dim3((8 * n_callers * p.windows + 63) / 64), dim3(64), 0, s, ws.window_sums, ws.parents, n_callers, p.windows, ws.merged_sums);
        sums = ws.merged_sums;
    }
    if (split && split->want_parts > 1 && p.windows > 1) {
        const uint32_t want = std::min<uint32_t>(split->want_parts, MSM_MAX_PARTS);
        const uint32_t wpp = (p.windows + want - 1) / want, parts = (p.windows + wpp - 1) / wpp;
        hipLaunchKernelGGL(msm_final_parts, dim3((4 * n_callers * parts + 63) / 64), dim3(64), 0, s, sums, ws.final_problems, n_callers, p, parts, wpp, ws.pieces, ws.pieces + (size_t)MSM_MAX_PARTS * ws.cap_problems);
        split->parts = parts; split->shift = p.c * wpp; split->count = n_callers; split->pts = ws.pieces; split->ready = ws.pieces + (size_t)MSM_MAX_PARTS * ws.cap_problems;
    } else {
        hipLaunchKernelGGL(msm_final, dim3((4 * n_callers + 63) / 64), dim3(64), 0, s, sums, ws.final_problems, n_callers, p);
    }
    H2V_HIP_CHECK(hipGetLastError());
    return 0;
}
int msm_combine_enqueue(hipStream_t s, MsmWorkspace& ws, const MsmSplit& sp, size_t lds_reserve) {
    if (!sp.parts) return 0;
    if (lds_reserve > 64 * 1024) H2V_HIP_CHECK(hipFuncSetAttribute((const void*)msm_combine_parts, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_reserve));
    hipLaunchKernelGGL(msm_combine_parts, dim3((4 * sp.count + 63) / 64), dim3(64), lds_reserve, s, sp.pts, ws.final_problems, sp.count, sp.parts, sp.shift);
    H2V_HIP_CHECK(hipGetLastError());
    return 0;
}

int msm_enqueue(hipStream_t s, MsmWorkspace& ws, const uint32_t* d_scalars, const G1A* d_bases, uint32_t n, G1J* d_out) {
    MsmProblems pr;
    pr.p.push_back(MsmProblem(d_scalars, d_bases, d_out, 8, 1, n));
    return msm_enqueue_multi(s, ws, pr);
}

}  // namespace h2v
