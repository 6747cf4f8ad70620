// C-ABI batch verification entry points (include/h2v.h): staged upload / launch / finish, the
// one-shot h2v_verify_batch / h2v_verify_each / h2v_guard_msm built on them, and h2v_fold_check.
#include "../../include/h2v.h"
#include "batch.h"
#include <string.h>
#include <algorithm>
#include <map>
#include <stdio.h>

using namespace h2v;

namespace {

template <class T> int dev_alloc(T*& p, size_t count) {
    if (p) { hipFree(p); p = nullptr; }
    H2V_HIP_CHECK(hipMalloc(&p, (count ? count : 1) * sizeof(T)));
    return 0;
}

// (re)allocate the per-batch device workspace for a given plan
int ensure_buffers(h2v_batch* b, PlanDevice* pd) {
    const Plan& pl = pd->host;
    size_t N = b->max_proofs, G = b->groups;
    size_t sig = pl.guard_term_order.size() * 977u + G * 7919u + pl.inst_queries.size() * 31u + (size_t)pl.opts.transcript * 77u + (size_t)pl.opts.multiopen * 131u + (size_t)pl.n_points * 1000003u + (size_t)pl.n_slots * 10007u + ((size_t)pl.n_slots_k[0] + pl.n_slots_k[1] + pl.n_slots_k[2]) * 1009u + pl.n_shared * 101u + pl.stream.size() + pl.proof_len * 7u + pl.n_instance_values * 13u + pl.n_challenges;
    if (b->cap_plan_sig == sig && b->pts) return 0;
    int rc;
    uint32_t words = (uint32_t)((pl.stream.size() + 7) / 8);
    const uint32_t blockw = pl.opts.transcript == H2V_TRANSCRIPT_KECCAK256 ? 17 : 16;   // 136-byte Keccak / 128-byte Blake2b blocks
    words = (words + blockw) / blockw * blockw;  // whole blocks, plus room for a final partial one
    b->stream_words = words;
    if ((rc = dev_alloc(b->proofs, N * pl.proof_len))) return rc;
    if ((rc = dev_alloc(b->inst, N * (size_t)pl.n_instance_values * 32))) return rc;
    if ((rc = dev_alloc(b->pts, N * pl.n_points + pl.n_shared))) return rc;
    if ((rc = dev_alloc(b->phi, N * pl.n_points + pl.n_shared))) return rc;
    if ((rc = dev_alloc(b->ycanon, N * pl.n_points * 32))) return rc;
    if ((rc = dev_alloc(b->words, (size_t)words * N))) return rc;
    if ((rc = dev_alloc(b->chal, (size_t)pl.squeeze_at.size() * N))) return rc;
    if ((rc = dev_alloc(b->mult, N))) return rc;
    if ((rc = dev_alloc(b->slots, (size_t)std::max(std::max(pl.n_slots, pl.n_slots_k[0]), std::max(pl.n_slots_k[1], pl.n_slots_k[2])) * N))) return rc;
    if ((rc = dev_alloc(b->msm_scal, (N * pl.n_points + G * pl.n_shared) * 8))) return rc;
    if ((rc = dev_alloc(b->shared, (size_t)pl.n_shared * N))) return rc;
    if ((rc = dev_alloc(b->left_scal, N * pl.n_points * 8))) return rc;
    if ((rc = dev_alloc(b->insteval, N * pl.inst_queries.size()))) return rc;
    if ((rc = dev_alloc(b->guard_scal, N * pl.guard_term_order.size() * 8))) return rc;
    if ((rc = dev_alloc(b->acc, 2 * G))) return rc;
    // everything h2v_batch_finish reads back sits in ONE block, mirrored by one pinned host buffer: a single copy per launch (four
    // separate copies into pageable memory were ~0.13 ms of a 20-step launch): [ok G][fold_failed G][out_ident 2 G][out_bytes 128 G][status N]
    // (the device block keeps the place of `ok`, unused: the offsets of the two blocks agree)
    b->results_bytes = 144 * G + 4 * N;
    if ((rc = dev_alloc(b->results, b->results_bytes))) return rc;
    if (b->results_host) { hipHostFree(b->results_host); b->results_host = nullptr; }
    H2V_HIP_CHECK(hipHostMalloc((void**)&b->results_host, b->results_bytes ? b->results_bytes : 1, hipHostMallocMapped));
    // the verdicts are the LAST thing a launch produces: the pairing kernel writes them into the host block itself (one word per group over
    // PCIe) and no copy follows it; everything else in the block is final before the pairing starts and is copied beside it (close_enqueue)
    { void* dp = nullptr; H2V_HIP_CHECK(hipHostGetDevicePointer(&dp, b->results_host, 0)); b->ok = reinterpret_cast<uint32_t*>(dp); }
    b->fold_failed = reinterpret_cast<uint32_t*>(b->results) + G;
    b->out_ident = reinterpret_cast<uint32_t*>(b->results) + 2 * G;
    b->out_bytes = b->results + 16 * G;
    b->status = reinterpret_cast<int*>(b->results + 144 * G);
    if ((rc = b->ws.alloc((uint32_t)(2 * (N * pl.n_points + G * pl.n_shared)), (uint32_t)(2 * G), (uint32_t)((N + G - 1) / G * pl.n_points + pl.n_shared)))) return rc;
    b->cap_plan_sig = sig;
    return 0;
}

bool scalar_is_canonical(const uint8_t* s) {
    uint32_t raw[8];
    for (int j = 0; j < 8; ++j) raw[j] = (uint32_t)s[4 * j] | ((uint32_t)s[4 * j + 1] << 8) | ((uint32_t)s[4 * j + 2] << 16) | ((uint32_t)s[4 * j + 3] << 24);
    return !Fr::geq_p(raw);
}

// Fr::random(getrandom_or_panic()) of AccumulatorStrategy::process (kzg/strategy.rs:129): 64 OS-random bytes reduced mod r
int os_random_scalars(std::vector<uint8_t>& out, size_t n) {
    out.resize(32 * n);
    FILE* f = fopen("/dev/urandom", "rb");
    if (!f) { set_last_error("cannot open /dev/urandom"); return H2V_ERR_DEVICE; }
    for (size_t i = 0; i < n; ++i) {
        uint8_t buf[64];
        if (fread(buf, 1, 64, f) != 64) { fclose(f); set_last_error("short read from /dev/urandom"); return H2V_ERR_DEVICE; }
        uint32_t w[16];
        for (int j = 0; j < 16; ++j) w[j] = (uint32_t)buf[4 * j] | ((uint32_t)buf[4 * j + 1] << 8) | ((uint32_t)buf[4 * j + 2] << 16) | ((uint32_t)buf[4 * j + 3] << 24);
        Fr::from_uniform_words(w).to_bytes(&out[32 * i]);
    }
    fclose(f);
    return 0;
}

int join_tail(h2v_batch* b);
// `overlap`: the copies run on the batch's copy stream in chunks and the decompression of every chunk is enqueued on the batch's own
// stream behind that chunk's event (h2v_batch_upload_launch); otherwise everything is copied on the batch's stream (h2v_batch_upload).
int upload_impl(h2v_batch* b, size_t n, const uint8_t* proofs_flat, size_t proof_len, const uint8_t* instances_flat, size_t ncols, const size_t* col_lens,
                const uint8_t* rand_tail, size_t n_tail, bool overlap = false) {
    if (!b || (n && !proofs_flat)) { set_last_error("h2v_batch_upload: null argument"); return H2V_ERR_BAD_ARGUMENT; }
    if (n > b->max_proofs) { set_last_error("h2v_batch_upload: n exceeds the batch capacity"); return H2V_ERR_BAD_ARGUMENT; }
    h2v_ctx* ctx = b->ctx;
    { H2V_HIP_CHECK(hipSetDevice(ctx->device)); int rcj = join_tail(b); if (rcj) return rcj; }
    if (!ctx->vk) { set_last_error("the context was created without a VerifyingKey"); return H2V_ERR_BAD_ARGUMENT; }
    if (ncols != ctx_total_instance_columns(ctx)) { set_last_error("instances do not match the VK's instance column count"); return H2V_ERR_INVALID_INSTANCES; }  // lib.rs:51-55
    std::vector<size_t> lens(col_lens, col_lens + ncols);
    PlanPin pin(ctx);
    int rc = pin.get(lens, b->want_guard);
    if (rc) return rc;
    PlanDevice* pd = pin.pd;
    const Plan& pl = pd->host;
    if (proof_len < pl.proof_len) { set_last_error("h2v_batch_upload: proof_len is shorter than this VK's proof"); return H2V_ERR_BAD_ARGUMENT; }
    if (pl.n_instance_values && n && !instances_flat) { set_last_error("h2v_batch_upload: instances missing"); return H2V_ERR_BAD_ARGUMENT; }
    if (rand_tail && n_tail < n) { set_last_error("h2v_batch_upload: n_tail < n"); return H2V_ERR_BAD_ARGUMENT; }
    if (b->groups > 1 && (n % b->groups || (rand_tail && n_tail % b->groups))) { set_last_error("h2v_batch_upload: n and n_tail must be multiples of the group count"); return H2V_ERR_BAD_ARGUMENT; }
    H2V_HIP_CHECK(hipSetDevice(ctx->device));
    if ((rc = ensure_buffers(b, pd))) return rc;
    if (b->plan) ctx_put_plan(ctx, b->plan);   // the batch holds its plan from upload to the next upload (or its destruction)
    b->plan = pin.take(); b->n = (uint32_t)n; b->launched = false; b->decompressed = false;
    std::vector<uint8_t> os_rand;
    if (!rand_tail) { if ((rc = os_random_scalars(os_rand, n))) return rc; rand_tail = os_rand.data(); n_tail = n; }
    for (size_t i = 0; i < n_tail; ++i) if (!scalar_is_canonical(rand_tail + 32 * i)) { set_last_error("h2v_batch_upload: rand32 scalar not canonical"); return H2V_ERR_BAD_ARGUMENT; }
    if (n_tail > b->cap_tail) { if ((rc = dev_alloc(b->tail, 32 * n_tail))) return rc; b->cap_tail = n_tail; }
    b->n_tail = (uint32_t)n_tail;
    hipStream_t s = b->stream;
    if (!n) { H2V_HIP_CHECK(hipStreamSynchronize(s)); return 0; }
    auto copy_proofs = [&](hipStream_t cs, size_t p0, size_t p1) -> int {
        if (proof_len == pl.proof_len) H2V_HIP_CHECK(hipMemcpyAsync(b->proofs + p0 * pl.proof_len, proofs_flat + p0 * proof_len, (p1 - p0) * proof_len, hipMemcpyHostToDevice, cs));
        else H2V_HIP_CHECK(hipMemcpy2DAsync(b->proofs + p0 * pl.proof_len, pl.proof_len, proofs_flat + p0 * proof_len, proof_len, pl.proof_len, p1 - p0, hipMemcpyHostToDevice, cs));
        return 0;
    };
    auto copy_rest = [&](hipStream_t cs) -> int {
        if (pl.n_instance_values) H2V_HIP_CHECK(hipMemcpyAsync(b->inst, instances_flat, n * (size_t)pl.n_instance_values * 32, hipMemcpyHostToDevice, cs));
        H2V_HIP_CHECK(hipMemcpyAsync(b->tail, rand_tail, 32 * n_tail, hipMemcpyHostToDevice, cs));
        // VK-wide bases sit behind the batch's own points so that one MSM covers both
        H2V_HIP_CHECK(hipMemcpyAsync(b->pts + n * (size_t)pl.n_points, pd->shared_bases, sizeof(G1A) * pl.n_shared, hipMemcpyDeviceToDevice, cs));
        H2V_HIP_CHECK(hipMemcpyAsync(b->phi + n * (size_t)pl.n_points, pd->shared_phi, sizeof(G1A) * pl.n_shared, hipMemcpyDeviceToDevice, cs));
        return 0;
    };
    if (!overlap) {
        if ((rc = copy_proofs(s, 0, n)) || (rc = copy_rest(s))) return rc;
        H2V_HIP_CHECK(hipStreamSynchronize(s));  // the host buffers are the caller's again
        return 0;
    }
    // Overlapped form (h2v_batch_upload_launch).  What was measured on the way (profiles/r03_h2d_microbench.txt, r03_upload_timeline.txt):
    //  * a 26 MB copy takes 0.47 ms from pageable and from pinned memory alike, and an "asynchronous" copy out of pageable memory
    //    returns only when the data is on the device — the calling thread is the one thing it blocks;
    //  * chunks on a second stream with an event per chunk for the kernels to wait on: erratic (0.64 ms best, 1.8 ms mean for eight
    //    chunks) — cross-stream event waits set the pace, slower than one blocking copy;
    //  * decompression in chunks, each enqueued when its chunk has arrived: point decompression is ONE round of ~0.5 ms of dependent
    //    work per lane whatever the launch size, so eight chunk launches are eight rounds (4.2 ms per launch instead of 3.6);
    //  * a decompression kernel reading the caller's registered buffer over the link: 0.82 ms instead of 0.60; a gather kernel for
    //    the point bytes alone: 0.23 ms — and hipHostUnregister waits for EVERY kernel in flight on the device (3.0 ms behind a 3 ms
    //    kernel, tools/unregister_probe.hip), so a registration cannot be dropped before the launch it helped has finished.
    // What is left: the decompression needs only the POINT bytes of a proof (12 x 32 of 1024 bytes for the headline VK), and those lie in
    // a few runs at fixed offsets.  The thread copies the point runs first (strided copies, a third of the bytes), enqueues the ONE
    // decompression launch — when the blocking copy has returned the bytes are in device memory, so no event is needed — and copies
    // everything (whole proofs, instances, draws) while the GPU decompresses; the later stages are enqueued after that copy returned.
    std::vector<std::pair<uint32_t, uint32_t>> runs;   // (offset, length) of the maximal runs of point bytes inside a proof
    {
        std::vector<uint32_t> offs(pl.point_offsets);
        std::sort(offs.begin(), offs.end());
        for (uint32_t o : offs) { if (!runs.empty() && runs.back().first + runs.back().second == o) runs.back().second += 32; else runs.push_back({o, 32u}); }
    }
    size_t point_bytes = 0;
    for (auto& r : runs) point_bytes += r.second;
    int mode = ctx->tuning.upload_mode;
    if (mode == 0) mode = 1;   // (one launch + finish, best of 15, tools/h2d_probe.py: resident 2.98 ms, points-first 3.29, two halves 3.33, plain 3.47;
                               //  in the benchmark's loop, PCIe-inclusive over resident: 0.905 / 0.885 / 0.835, tools/r03_ab.sh)
    if (mode == 1 && (runs.size() > 4 || 2 * point_bytes > pl.proof_len)) mode = 3;   // points all over the proof, or most of it: nothing to gain
    if (n < 2048) mode = 3;                                                           // a copy of a megabyte or two is not worth two launches
    if (mode == 3) {
        if ((rc = copy_proofs(s, 0, n)) || (rc = copy_rest(s))) return rc;
        H2V_HIP_CHECK(hipStreamSynchronize(s));
        return 0;
    }
    if (!b->copy) H2V_HIP_CHECK(hipStreamCreateWithFlags(&b->copy, hipStreamNonBlocking));
    H2V_HIP_CHECK(hipStreamSynchronize(s));   // an earlier launch of this batch may still read the buffers (normally long finished: h2v_batch_finish)
    if (mode == 2) {
        // the proofs in two halves: [first half] -> its decompression is enqueued -> [second half, instances, draws] travel while the GPU
        // decompresses the first -> the second half's decompression.  Two rounds of the decompression kernel at half occupancy take about
        // what one round at full occupancy takes, and the second copy hides behind the first round.
        StageArgs g{(uint32_t)n, &pl, pd, b->proofs, b->inst, b->pts, b->phi, b->ycanon, b->status, b->words, b->stream_words, b->chal};
        if ((rc = decompress_begin_enqueue(s, g))) return rc;
        const size_t half = (n / 2 + 15) / 16 * 16;
        if ((rc = copy_proofs(b->copy, 0, half))) return rc;
        H2V_HIP_CHECK(hipStreamSynchronize(b->copy));
        if ((rc = decompress_range_enqueue(s, g, 0, (uint32_t)half))) return rc;
        if ((rc = copy_proofs(b->copy, half, n)) || (rc = copy_rest(b->copy))) return rc;
        H2V_HIP_CHECK(hipStreamSynchronize(b->copy));   // everything is on the device; the host buffers are the caller's again
        if ((rc = decompress_range_enqueue(s, g, (uint32_t)half, (uint32_t)n))) return rc;
        if ((rc = decompress_finish_enqueue(s, g))) return rc;
        b->decompressed = true;
        return 0;
    }
    for (auto& r : runs) H2V_HIP_CHECK(hipMemcpy2DAsync(b->proofs + r.first, pl.proof_len, proofs_flat + r.first, proof_len, r.second, n, hipMemcpyHostToDevice, b->copy));
    H2V_HIP_CHECK(hipStreamSynchronize(b->copy));
    StageArgs g{(uint32_t)n, &pl, pd, b->proofs, b->inst, b->pts, b->phi, b->ycanon, b->status, b->words, b->stream_words, b->chal};
    if ((rc = decompress_begin_enqueue(s, g))) return rc;
    if ((rc = decompress_range_enqueue(s, g, 0, (uint32_t)n))) return rc;
    // (the whole proofs again, point bytes included: identical bytes over the ones the kernel is reading)
    if ((rc = copy_proofs(b->copy, 0, n)) || (rc = copy_rest(b->copy))) return rc;
    H2V_HIP_CHECK(hipStreamSynchronize(b->copy));   // everything is on the device; the host buffers are the caller's again
    if ((rc = decompress_finish_enqueue(s, g))) return rc;
    b->decompressed = true;
    return 0;
}

int close_enqueue(h2v_batch* b, bool with_pairing);
int ensure_whole(h2v_batch* b);
int export_batch_records(h2v_batch* b, void* device_dst);
#define H2V_SPLIT_MAX_GROUPS 64u

int launch_impl(h2v_batch* b, int with_pairing) {
    if (!b || !b->plan) { set_last_error("h2v_batch_launch: nothing uploaded"); return H2V_ERR_BAD_ARGUMENT; }
    h2v_ctx* ctx = b->ctx;
    H2V_HIP_CHECK(hipSetDevice(ctx->device));
    PlanDevice* pd = b->plan;
    const Plan& pl = pd->host;
    hipStream_t s = b->stream;
    uint32_t n = b->n;
    const uint32_t G = b->groups, gs = n / G;
    b->with_pairing = with_pairing != 0; b->launched = true;
    int rc;
    if ((rc = join_tail(b))) return rc;
    int ev = 0;
    auto mark = [&]() { if (b->profiling >= 2) hipEventRecord(b->ev[ev], s); ++ev; };   // (an event record is a barrier packet: ~6 us of idle stream each)
    mark();
    StageArgs g{n, &pl, pd, b->proofs, b->inst, b->pts, b->phi, b->ycanon, b->status, b->words, b->stream_words, b->chal};
    // stage 1: point decompression + canonicity checks (already on the stream, behind its chunked upload, after h2v_batch_upload_launch);
    // stage 2: absorbed stream, Blake2b challenges, batch multipliers.  The status words are cleared first, then the auxiliary stream is
    // forked: the scalar canonicity check (proof bytes only) runs there beside the decompression, with the multipliers
    const bool run_decompress = !b->decompressed;
    // cleared per launch: fold_failed (set by h2v_batch_fold_check_enqueue only) and — unless the upload already did (h2v_batch_upload_launch) —
    // the status words.  The results block is [ok][fold_failed][out_ident][out_bytes][status]: one fill from fold_failed to the last status word
    // (the output bytes in between are written later in the launch) instead of two
    if (run_decompress && n) H2V_HIP_CHECK(hipMemsetAsync(b->fold_failed, 0, (size_t)((uint8_t*)(b->status + n) - (uint8_t*)b->fold_failed), s));
    else H2V_HIP_CHECK(hipMemsetAsync(b->fold_failed, 0, 4 * (size_t)G, s));
    H2V_HIP_CHECK(hipEventRecord(b->ev_fork0, s));   // everything enqueued before this point (uploads, the cleared status words) is visible to the auxiliary stream
    if (run_decompress && (rc = decompress_range_enqueue(s, g, 0, n))) return rc;
    b->decompressed = false;   // (a later h2v_batch_launch on the same upload runs the stage again: every launch does all of its work)
    mark();
    if ((rc = transcript_stage_enqueue(s, g))) return rc;
    // both channels of every group in one set of launches: [2g] left (SHPLONK: sum_p m_p * h2_p; GWC: the witness points),
    // [2g+1] right = the group's pooled Guard terms + its folded VK-wide bases.  Both index the same point array; unused
    // slots have zero scalars and cost nothing.  The descriptors are addresses and sizes: they go to the device on the auxiliary stream, beside the decompression (msm_prepare_problems below).
    MsmProblems pr;
    {
        const uint32_t np = pl.n_points;
        for (uint32_t g = 0; g < G; ++g) {
            const size_t first = (size_t)g * gs * np;
            if (pl.left_term_order.size() == 1 && !pl.left_term_order[0].first) {
                // SHPLONK: one left term per proof (its h2): the problem is that slot of every proof — a strided view of gs terms, not the
                // group's gs * np slots with gs of them non-zero (msm_glv_prep wrote twelve zero words for each of the other slots)
                const size_t at = first + pl.left_term_order[0].second;
                pr.p.push_back(MsmProblem(b->left_scal + at * 8, b->pts + at, b->acc + 2 * g, 8 * np, np, gs));
                pr.p.back().phi = b->phi + at;
            } else {
                pr.p.push_back(MsmProblem(b->left_scal + first * 8, b->pts + first, b->acc + 2 * g, 8, 1, gs * np));
                pr.p.back().phi = b->phi + first;
                pr.p.back().nnz = gs * (uint32_t)pl.left_term_order.size();   // the program writes only these slots, the rest stay zero
            }
            pr.p.push_back(MsmProblem(b->msm_scal + first * 8, b->pts + first, b->acc + 2 * g + 1, 8, 1, gs * np,
                                      b->msm_scal + ((size_t)n * np + (size_t)g * pl.n_shared) * 8, b->pts + (size_t)n * np, n ? pl.n_shared : 0));
            pr.p.back().phi = b->phi + first; pr.p.back().phi2 = b->phi + (size_t)n * np;
        }
    }
    b->ws.tune = ctx->tuning;
    // the batch multipliers depend only on the uploaded draws: they run on the auxiliary stream beside decompression and transcript
    if (n) {
        H2V_HIP_CHECK(hipStreamWaitEvent(b->aux, b->ev_fork0, 0));
        hipStream_t sm = b->aux;
        if (run_decompress && (rc = decompress_finish_enqueue(sm, g))) return rc;   // k_check_scalars
        // multipliers: suffix products of the uploaded draws — or, for a batch that is a non-contiguous subset of a larger
        // accumulation (h2v_verify_batch_shapes), gathered from the multipliers of the whole sequence
        if (b->ext_mult) { if ((rc = gather_multipliers_enqueue(sm, b->ext_mult, b->ext_idx, n, b->mult))) return rc; }
        else if ((rc = multipliers_enqueue(sm, b->tail, b->n_tail, n, G, b->mult))) return rc;
        // the program writes only the slots the left channel uses; with ONE left term per proof the MSM reads exactly those (the strided problem above)
        if (!(pl.left_term_order.size() == 1 && !pl.left_term_order[0].first)) H2V_HIP_CHECK(hipMemsetAsync(b->left_scal, 0, (size_t)n * pl.n_points * 32, sm));
        if ((rc = msm_prepare_problems(sm, b->ws, pr))) return rc;   // (5 us of launch + kernel boundary that the main stream's chain no longer carries)
        H2V_HIP_CHECK(hipEventRecord(b->ev_join0, sm));
        H2V_HIP_CHECK(hipStreamWaitEvent(s, b->ev_join0, 0));   // joined before the Fr program reads them
    }
    mark();
    FrvmArgs a{pd->code, (uint32_t)pl.code.size(), pd->consts, b->slots, n, b->proofs, pl.proof_len, pd->scalar_offsets, b->inst, pl.n_instance_values,
               b->chal, b->mult, b->status, b->msm_scal, pl.n_points, b->shared, b->left_scal, b->insteval, b->guard_scal, (uint32_t)pl.guard_term_order.size()};
    if (n && pl.wide_instances) {
        Fr step = pl.omega;
        for (int i = 0; i < 8; ++i) step = step.sqr();   // omega^256: a thread's stride through the column
        const Fr step_inv = step.inv();
        for (size_t q = 0; q < pl.inst_queries.size(); ++q) {
            InstEvalArgs ia{b->inst, pl.n_instance_values, b->chal, pl.x_chal, n, pl.domain_k, pl.inst_queries[q].base, pl.inst_queries[q].len,
                            pl.inst_queries[q].w_start, pl.omega, step, step_inv, pl.n_inv, b->insteval + q * (size_t)n, b->status};
            if ((rc = instance_eval_enqueue(s, ia))) return rc;
        }
    }
    a.force_streams = ctx->tuning.frvm_streams; a.force_lds_kb = ctx->tuning.frvm_lds_kb;
    for (int k = 0; k < 3; ++k) { for (int q = 0; q < k + 2; ++q) { a.code_k[k][q] = pd->code_k[k][q]; a.n_code_k[k][q] = (uint32_t)pl.code_k[k][q].size(); } a.n_slots_k[k] = pl.n_slots_k[k]; }
    if ((rc = frvm_enqueue(s, a, pl.n_slots))) return rc;
    mark();
    if (n) { if ((rc = fold_shared_enqueue(s, b->shared, n, pl.n_points, pl.n_shared, G, b->msm_scal))) return rc; }
    mark();
    {
        b->ws.profile = b->profiling >= 1; b->ws.profile_recorded = false;
        // A launch leaves the accumulators in pieces (MsmSplit): its own pairing checks take the pieces and the whole points are put
        // together beside them (close_enqueue); a launch without a pairing (a shard) exports the pieces, the folded pairing takes them,
        // and the whole points are only made if somebody reads them (ensure_whole).  Worth it while the launch is a latency chain, i.e. few groups.
        b->split = MsmSplit();
        const uint32_t parts_knob = ctx->tuning.msm_parts > 0 ? (uint32_t)ctx->tuning.msm_parts : MSM_MAX_PARTS;   // h2v_tuning.msm_parts
        b->ws.tune = ctx->tuning;
        if (n && G <= H2V_SPLIT_MAX_GROUPS && parts_knob > 1) {
            if (b->line_ws_groups < G) {
                if (b->line_ws) { hipStreamSynchronize(s); hipFree(b->line_ws); b->line_ws = nullptr; b->line_ws_groups = 0; }
                H2V_HIP_CHECK(hipMalloc(&b->line_ws, (size_t)G * H2V_PAIRING_LINE_WS_BYTES));
                b->line_ws_groups = G;
            }
            b->split.want_parts = parts_knob;
        }
        if ((rc = msm_enqueue_multi(s, b->ws, pr, b->split.want_parts > 1 ? &b->split : nullptr))) return rc;
    }
    mark();
    if ((rc = close_enqueue(b, with_pairing != 0))) return rc;
    mark();
    return 0;
}

// The end of a launch: the pairing checks and the conversion of the accumulators to affine bytes only READ the accumulators, and
// both are latency chains on a few waves (0.5 ms and 0.35 ms) — they run side by side, the conversion and the copy of the result block
// on the batch's auxiliary stream.  That stream is NOT joined back into the main one (its last event, ev_join, is what join_tail makes
// the main stream wait for if anything but h2v_batch_finish comes next).
int close_enqueue(h2v_batch* b, bool with_pairing) {
    hipStream_t s = b->stream;
    const uint32_t G = b->groups;
    int rc;
    b->acc_stale = false;
    if (!with_pairing) {
        if (b->split.parts) { b->acc_stale = true; return 0; }   // pieces only for now
        return point_to_bytes_enqueue(s, b->acc, b->out_bytes, b->out_ident, 2 * G);
    }
    H2V_HIP_CHECK(hipEventRecord(b->ev_fork, s));
    H2V_HIP_CHECK(hipStreamWaitEvent(b->aux, b->ev_fork, 0));
    // (beside the pairing: kept off the pairing workgroups' CUs by an LDS request, internal.h)
    if (b->split.parts && (rc = msm_combine_enqueue(b->aux, b->ws, b->split, H2V_AUX_LDS_RESERVE))) return rc;   // acc <- the whole points
    if ((rc = point_to_bytes_enqueue(b->aux, b->acc, b->out_bytes, b->out_ident, 2 * G, H2V_AUX_LDS_RESERVE))) return rc;
    // the result block (all but the verdicts, which the pairing kernel writes to the host itself) goes back on the auxiliary stream too, and the
    // main stream does NOT wait for it: its last operation is the pairing kernel — the join (a barrier packet) and the copy behind it were
    // 16 us at the end of every launch.  h2v_batch_finish waits for both streams; anything else that touches the batch first calls join_tail.
    // (by a kernel, not hipMemcpyAsync: a copy enqueued now, behind kernels that end a launch later, can hold up an SDMA queue — util.hip)
    if ((rc = copy_words_enqueue(b->aux, b->results + 4 * (size_t)G, b->ok + G, 35 * (size_t)G + (size_t)b->n, H2V_AUX_LDS_RESERVE))) return rc;
    H2V_HIP_CHECK(hipEventRecord(b->ev_join, b->aux));
    b->tail_on_aux = true;
    if (b->split.parts) { if ((rc = pairing_check_split_enqueue(s, b->ctx->pairing, b->split.ready, G, b->split.parts, b->split.shift, b->line_ws, b->ok, b->ctx->tuning.pairing_one_stream != 0))) return rc; }
    else if ((rc = pairing_check_enqueue(s, b->ctx->pairing, b->acc, G, b->ok))) return rc;
    return 0;
}
// the main stream waits for what the last launch left on the auxiliary stream (before anything new reads or overwrites it)
int join_tail(h2v_batch* b) {
    if (!b->tail_on_aux) return 0;
    b->tail_on_aux = false;
    H2V_HIP_CHECK(hipStreamWaitEvent(b->stream, b->ev_join, 0));
    return 0;
}
// the whole accumulators (acc) and their affine bytes, if the last launch left pieces only
int ensure_whole(h2v_batch* b) {
    if (!b->acc_stale) return 0;
    int rc;
    if ((rc = msm_combine_enqueue(b->stream, b->ws, b->split))) return rc;
    if ((rc = point_to_bytes_enqueue(b->stream, b->acc, b->out_bytes, b->out_ident, 2 * b->groups))) return rc;
    b->acc_stale = false;
    return 0;
}
// the batch's accumulator records (pieces if the launch left pieces)
int export_batch_records(h2v_batch* b, void* device_dst) {
    { int rcj = join_tail(b); if (rcj) return rcj; }
    if (b->split.parts) return export_records_enqueue(b->stream, nullptr, b->split.pts, b->split.parts, b->split.shift, b->status, b->n, b->groups, device_dst);
    return export_records_enqueue(b->stream, b->acc, nullptr, 1, 0, b->status, b->n, b->groups, device_dst);
}

// group_ok / out_left / out_right hold one entry (64 bytes) per group
int finish_impl(h2v_batch* b, int* per_proof_status, int* group_ok, uint8_t* out_left, uint8_t* out_right) {
    if (!b || !b->launched) { set_last_error("h2v_batch_finish: nothing launched"); return H2V_ERR_BAD_ARGUMENT; }
    H2V_HIP_CHECK(hipSetDevice(b->ctx->device));
    hipStream_t s = b->stream;
    const uint32_t n = b->n, G = b->groups, gs = n / G;
    { int rcw = ensure_whole(b); if (rcw) return rcw; }
    hipError_t e;
    if (b->tail_on_aux) {
        // a launch that ended in its own pairing checks: the block is on its way on the auxiliary stream, the verdicts come from the kernel
        b->tail_on_aux = false;
        e = hipStreamSynchronize(s);
        if (e == hipSuccess) e = hipStreamSynchronize(b->aux);   // (not hipEventSynchronize on its last event: that wait goes through the runtime's event thread, and a host that re-uploads per launch lost 40 % to it)
    } else {
        const size_t nbytes = 144 * (size_t)G + 4 * (size_t)n;
        H2V_HIP_CHECK(hipMemcpyAsync(b->results_host + 4 * (size_t)G, b->results + 4 * (size_t)G, nbytes - 4 * (size_t)G, hipMemcpyDeviceToHost, s));
        e = hipStreamSynchronize(s);
    }
    if (e != hipSuccess) { set_last_error(std::string("h2v_batch_finish: ") + hipGetErrorString(e)); return H2V_ERR_DEVICE; }
    const uint32_t* okv = reinterpret_cast<const uint32_t*>(b->results_host);
    const uint32_t* foldf = okv + G;
    const uint8_t* outb = b->results_host + 16 * (size_t)G;
    const int* st = reinterpret_cast<const int*>(b->results_host + 144 * (size_t)G);
    for (int i = 0; i < 7; ++i) b->last_ms[i] = 0;
    if (b->profiling >= 2) {
        // events: 0 start, 1 after decompression, 2 after transcript + multipliers, 3 after Fr program, 4 after fold, 5 after MSMs, 6 after pairing
        float t01 = 0, t12 = 0, t23 = 0, t34 = 0, t45 = 0, t56 = 0;
        hipEventElapsedTime(&t01, b->ev[0], b->ev[1]); hipEventElapsedTime(&t12, b->ev[1], b->ev[2]); hipEventElapsedTime(&t23, b->ev[2], b->ev[3]);
        hipEventElapsedTime(&t34, b->ev[3], b->ev[4]); hipEventElapsedTime(&t45, b->ev[4], b->ev[5]); hipEventElapsedTime(&t56, b->ev[5], b->ev[6]);
        b->last_ms[0] = t01; b->last_ms[1] = t12; b->last_ms[2] = t23; b->last_ms[3] = t34; b->last_ms[4] = t45; b->last_ms[5] = t56;
    }
    if (b->profiling >= 1) {
        float tacc = 0;
        if (b->ws.profile_recorded) hipEventElapsedTime(&tacc, b->ws.ev_acc[0], b->ws.ev_acc[1]);
        b->last_ms[6] = tacc;
    }
    std::vector<char> all_ok(G, 1);
    // (the common case — no proof of the launch was rejected — is found by a word-wide scan: decoding 20 480 statuses one by one, with a
    // division each for the group, was 30 us of host time behind the GPU's last kernel)
    uint32_t any = 0;
    for (uint32_t i = 0; i < n; ++i) any |= (uint32_t)st[i];
    if (!any) { if (per_proof_status && n) memset(per_proof_status, 0, sizeof(int) * (size_t)n); }
    else for (uint32_t g = 0, i = 0; g < G; ++g) for (uint32_t k = 0; k < gs; ++k, ++i) {
        const int v = status_decode(st[i]);
        if (per_proof_status) per_proof_status[i] = v;
        if (v != 0) all_ok[g] = 0;
    }
    for (uint32_t g = 0; g < G; ++g) {
        // a sharded group is accepted only if no shard reported a failed proof (their terms are zeroed out of the accumulators)
        if (group_ok) group_ok[g] = (all_ok[g] && !foldf[g] && (!b->with_pairing || okv[g])) ? 1 : 0;
        if (out_left) memcpy(out_left + 64 * (size_t)g, &outb[128 * (size_t)g], 64);
        if (out_right) memcpy(out_right + 64 * (size_t)g, &outb[128 * (size_t)g + 64], 64);
    }
    return 0;
}

// one-shot calls reuse a batch object kept in the context (up to H2V_SCRATCH_BATCH_MAX proofs of capacity)
#define H2V_SCRATCH_BATCH_MAX 1024u
int scratch_batch_take(h2v_ctx* ctx, size_t capacity, size_t max_inst, h2v_batch** out) {
    h2v_batch* b = ctx->scratch_batch;
    ctx->scratch_batch = nullptr;
    if (b && (b->max_proofs < capacity || b->max_inst < max_inst)) { h2v_batch_destroy(b); b = nullptr; }
    if (!b) { int rc = h2v_batch_create(ctx, capacity, max_inst, &b); if (rc) return rc; }
    *out = b;
    return 0;
}
void scratch_batch_give(h2v_ctx* ctx, h2v_batch* b) {
    if (!b) return;
    if (b->max_proofs > H2V_SCRATCH_BATCH_MAX || ctx->scratch_batch) { h2v_batch_destroy(b); return; }
    ctx->scratch_batch = b;
}

// pack pointer-array proofs / instances into the flat layout; proofs shorter than the VK's proof are
// the reader running dry: "failed to fill whole buffer" -> Error::Transcript, or Opening inside the multi-open part
int pack_inputs(const Plan& pl, size_t n, const uint8_t* const* proofs, const size_t* proof_lens, const uint8_t* const* instances32,
                std::vector<uint8_t>& flat, std::vector<uint8_t>& iflat, std::vector<int>& forced) {
    const size_t per_inst = (size_t)pl.n_instance_values * 32;
    flat.assign(n * pl.proof_len, 0); iflat.assign(n * per_inst, 0); forced.assign(n, 0);
    // byte offset where the multi-open part starts: h1 is the first point after all scalars
    size_t opening_at = pl.opening_offset;
    for (size_t i = 0; i < n; ++i) {
        if (!proofs[i]) { set_last_error("null proof pointer"); return H2V_ERR_BAD_ARGUMENT; }
        if (proof_lens[i] < pl.proof_len) {
            // the reader runs dry; every point of the packed copy is made undecodable (x = 2^254-1 >= p) so that the proof
            // contributes nothing, and the status is set to what the reference reports for the place where it ran dry
            forced[i] = proof_lens[i] < opening_at ? H2V_ERR_TRANSCRIPT : H2V_ERR_OPENING;
            memset(&flat[i * pl.proof_len], 0xff, pl.proof_len);
        } else memcpy(&flat[i * pl.proof_len], proofs[i], pl.proof_len);
        if (per_inst) { if (!instances32 || !instances32[i]) { set_last_error("null instances pointer"); return H2V_ERR_BAD_ARGUMENT; } memcpy(&iflat[i * per_inst], instances32[i], per_inst); }
    }
    return 0;
}

int pack_and_run(h2v_ctx* ctx, size_t n, const uint8_t* const* proofs, const size_t* proof_lens, const uint8_t* const* instances32, size_t ncols,
                 const size_t* col_lens, const uint8_t* rand32, bool single, int with_pairing, int* per_proof_status, int* batch_ok, uint8_t* out_left,
                 uint8_t* out_right, h2v_batch** keep, bool guard = false) {
    if (!ctx || (n && (!proofs || !proof_lens)) || (ncols && !col_lens)) { set_last_error("null argument"); return H2V_ERR_BAD_ARGUMENT; }
    if (!ctx->vk) { set_last_error("the context was created without a VerifyingKey"); return H2V_ERR_BAD_ARGUMENT; }
    if (ncols != ctx_total_instance_columns(ctx)) { set_last_error("instances do not match the VK's instance column count"); return H2V_ERR_INVALID_INSTANCES; }
    std::lock_guard<std::mutex> lock(ctx->mu);   // one one-shot call per context at a time (it owns the context's scratch batch)
    std::vector<size_t> lens(col_lens, col_lens + ncols);
    PlanPin pin(ctx);
    int rc = pin.get(lens);
    if (rc) return rc;
    const Plan& pl = pin.pd->host;
    size_t per_inst = (size_t)pl.n_instance_values * 32;
    std::vector<uint8_t> flat, iflat;
    std::vector<int> forced;
    if ((rc = pack_inputs(pl, n, proofs, proof_lens, instances32, flat, iflat, forced))) return rc;
    h2v_batch* b = nullptr;
    if (single) {
        // SingleStrategy (kzg/strategy.rs:143-181) = an accumulator of ONE proof with multiplier 1 and its own pairing: run the
        // proofs as one-proof groups of grouped launches, at most MSM_MAX_PROBLEMS / 2 per launch
        const size_t per = MSM_MAX_PROBLEMS / 2;
        if ((rc = scratch_batch_take(ctx, std::min(n ? n : 1, per), pl.n_instance_values, &b))) return rc;
        b->want_guard = false;
        bool all = true;
        for (size_t off = 0; off < n && !rc; off += per) {
            const size_t m = std::min(per, n - off);
            std::vector<uint8_t> ones(32 * m, 0);
            for (size_t i = 0; i < m; ++i) ones[32 * i] = 1;
            std::vector<int> st(m, 0), gok(m, 0);
            if ((rc = h2v_batch_set_groups(b, m))) break;
            if ((rc = upload_impl(b, m, flat.data() + off * pl.proof_len, pl.proof_len, iflat.data() + off * per_inst, ncols, col_lens, ones.data(), m))) break;
            if ((rc = launch_impl(b, 1))) break;
            if ((rc = finish_impl(b, st.data(), gok.data(), nullptr, nullptr))) break;
            for (size_t i = 0; i < m; ++i) {
                int v = forced[off + i] ? forced[off + i] : st[i];
                if (v == 0 && !gok[i]) v = H2V_ERR_CONSTRAINT_SYSTEM_FAILURE;  // kzg/strategy.rs:171-175
                if (per_proof_status) per_proof_status[off + i] = v;
                if (v != 0) all = false;
            }
        }
        if (batch_ok && !rc) *batch_ok = all ? 1 : 0;
        scratch_batch_give(ctx, b);
        return rc;
    }
    if ((rc = scratch_batch_take(ctx, n ? n : 1, pl.n_instance_values, &b))) return rc;
    if ((rc = h2v_batch_set_groups(b, 1))) { h2v_batch_destroy(b); return rc; }
    b->want_guard = guard;
    do {
        if ((rc = upload_impl(b, n, flat.data(), pl.proof_len, iflat.data(), ncols, col_lens, rand32, rand32 ? n : 0))) break;
        if ((rc = launch_impl(b, with_pairing))) break;
        std::vector<int> st(n ? n : 1, 0);
        int ok = 0;
        if ((rc = finish_impl(b, st.data(), &ok, out_left, out_right))) break;
        for (size_t i = 0; i < n; ++i) if (forced[i]) { st[i] = forced[i]; ok = 0; }
        if (per_proof_status) for (size_t i = 0; i < n; ++i) per_proof_status[i] = st[i];
        if (batch_ok) *batch_ok = ok;
    } while (0);
    if (keep && !rc) *keep = b; else scratch_batch_give(ctx, b);
    return rc;
}

}  // namespace

extern "C" {

int h2v_random_scalars(uint8_t* out32, size_t n) {
    if (n && !out32) { set_last_error("h2v_random_scalars: null argument"); return H2V_ERR_BAD_ARGUMENT; }
    std::vector<uint8_t> v;
    int rc = os_random_scalars(v, n);
    if (rc) return rc;
    if (n) memcpy(out32, v.data(), 32 * n);
    return 0;
}

int h2v_ctx_proof_shape(const h2v_ctx* ctx, size_t* proof_len, size_t* n_points, size_t* n_scalars, size_t* n_right_terms, size_t* n_instance_columns) {
    if (!ctx || !ctx->vk) { set_last_error("the context was created without a VerifyingKey"); return H2V_ERR_BAD_ARGUMENT; }
    // the layout does not depend on instance lengths; compile (or fetch) the plan for empty columns of the right count
    std::vector<size_t> lens(ctx_total_instance_columns(ctx), 0);
    PlanPin pin(const_cast<h2v_ctx*>(ctx));
    int rc = pin.get(lens);
    if (rc) return rc;
    PlanDevice* pd = pin.pd;
    if (proof_len) *proof_len = pd->host.proof_len;
    if (n_points) *n_points = pd->host.n_points;
    if (n_scalars) *n_scalars = pd->host.n_scalars;
    if (n_right_terms) *n_right_terms = pd->host.right_term_order.size();
    if (n_instance_columns) *n_instance_columns = ctx_total_instance_columns(ctx);
    return 0;
}

int h2v_batch_create(h2v_ctx* ctx, size_t max_proofs, size_t max_instance_values_per_proof, h2v_batch** out) {
    if (!ctx || !out || !max_proofs) { set_last_error("h2v_batch_create: bad argument"); return H2V_ERR_BAD_ARGUMENT; }
    if (max_proofs > (1u << 22)) { set_last_error("h2v_batch_create: max_proofs too large"); return H2V_ERR_BAD_ARGUMENT; }
    H2V_HIP_CHECK(hipSetDevice(ctx->device));
    h2v_batch* b = new h2v_batch();
    b->ctx = ctx; b->max_proofs = max_proofs; b->max_inst = max_instance_values_per_proof;
    if (hipStreamCreateWithFlags(&b->stream, hipStreamNonBlocking) != hipSuccess || hipStreamCreateWithFlags(&b->aux, hipStreamNonBlocking) != hipSuccess) {
        if (b->stream) hipStreamDestroy(b->stream);
        delete b; set_last_error("hipStreamCreate failed"); return H2V_ERR_DEVICE;
    }
    for (int i = 0; i < 8; ++i) hipEventCreate(&b->ev[i]);
    hipEventCreateWithFlags(&b->ev_fork, hipEventDisableTiming); hipEventCreateWithFlags(&b->ev_join, hipEventDisableTiming);
    hipEventCreateWithFlags(&b->ev_fork0, hipEventDisableTiming); hipEventCreateWithFlags(&b->ev_join0, hipEventDisableTiming);
    *out = b;
    return 0;
}

void h2v_batch_destroy(h2v_batch* b) {
    if (!b) return;
    hipSetDevice(b->ctx->device);
    if (b->stream) hipStreamSynchronize(b->stream);
    if (b->aux) hipStreamSynchronize(b->aux);   // (the tail of the last launch may still be running there)
    if (b->plan) { ctx_put_plan(b->ctx, b->plan); b->plan = nullptr; }
    hipFree(b->proofs); hipFree(b->inst); hipFree(b->tail); hipFree(b->pts); hipFree(b->phi); hipFree(b->ycanon); hipFree(b->results); hipFree(b->words); hipFree(b->chal);
    hipFree(b->mult); hipFree(b->slots); hipFree(b->msm_scal); hipFree(b->shared); hipFree(b->left_scal); hipFree(b->insteval); hipFree(b->guard_scal); hipFree(b->acc);
    hipFree(b->line_ws);
    if (b->results_host) hipHostFree(b->results_host);
    b->ws.release();
    for (int i = 0; i < 8; ++i) if (b->ev[i]) hipEventDestroy(b->ev[i]);
    if (b->aux) { hipStreamSynchronize(b->aux); hipStreamDestroy(b->aux); }
    if (b->copy) { hipStreamSynchronize(b->copy); hipStreamDestroy(b->copy); }

    if (b->ev_fork) hipEventDestroy(b->ev_fork);
    if (b->ev_join) hipEventDestroy(b->ev_join);
    if (b->ev_fork0) hipEventDestroy(b->ev_fork0);
    if (b->ev_join0) hipEventDestroy(b->ev_join0);
    if (b->stream && b->owns_stream) hipStreamDestroy(b->stream);
    delete b;
}

int h2v_batch_upload(h2v_batch* b, size_t n, const uint8_t* proofs_flat, size_t proof_len, const uint8_t* instances_flat, size_t n_instance_columns,
                     const size_t* col_lens, const uint8_t* rand32_tail, size_t n_tail) {
    return upload_impl(b, n, proofs_flat, proof_len, instances_flat, n_instance_columns, col_lens, rand32_tail, n_tail);
}
int h2v_batch_launch(h2v_batch* b, int with_pairing) { return launch_impl(b, with_pairing); }
int h2v_batch_upload_launch(h2v_batch* b, size_t n, const uint8_t* proofs_flat, size_t proof_len, const uint8_t* instances_flat, size_t n_instance_columns,
                            const size_t* col_lens, const uint8_t* rand32_tail, size_t n_tail, int with_pairing) {
    int rc = upload_impl(b, n, proofs_flat, proof_len, instances_flat, n_instance_columns, col_lens, rand32_tail, n_tail, true);
    if (rc) return rc;
    return launch_impl(b, with_pairing);
}
int h2v_batch_finish(h2v_batch* b, int* per_proof_status, int* batch_ok, uint8_t out_left_xy[64], uint8_t out_right_xy[64]) {
    if (b && b->groups > 1) { set_last_error("h2v_batch_finish: the batch is grouped, use h2v_batch_finish_groups"); return H2V_ERR_BAD_ARGUMENT; }
    return finish_impl(b, per_proof_status, batch_ok, out_left_xy, out_right_xy);
}
int h2v_batch_set_groups(h2v_batch* b, size_t groups) {
    if (!b || !groups || groups > MSM_MAX_PROBLEMS / 2 || groups > b->max_proofs) { set_last_error("h2v_batch_set_groups: bad group count"); return H2V_ERR_BAD_ARGUMENT; }
    if (b->stream) hipStreamSynchronize(b->stream);
    if (b->aux) hipStreamSynchronize(b->aux);   // (the tail of the last launch may still be running there)
    if (b->plan) { ctx_put_plan(b->ctx, b->plan); b->plan = nullptr; }
    b->groups = (uint32_t)groups; b->launched = false;  // the next upload re-sizes the workspace
    return 0;
}
int h2v_batch_finish_groups(h2v_batch* b, int* per_proof_status, int* group_ok, uint8_t* out_left_xy, uint8_t* out_right_xy, size_t n_groups) {
    if (!b || n_groups != b->groups) { set_last_error("h2v_batch_finish_groups: n_groups does not match h2v_batch_set_groups"); return H2V_ERR_BAD_ARGUMENT; }
    return finish_impl(b, per_proof_status, group_ok, out_left_xy, out_right_xy);
}
int h2v_batch_accumulators(h2v_batch* b, void** device_ptr, size_t* nbytes) {
    if (!b || !b->acc || !device_ptr) { set_last_error("h2v_batch_accumulators: nothing uploaded"); return H2V_ERR_BAD_ARGUMENT; }
    if (b->launched) { H2V_HIP_CHECK(hipSetDevice(b->ctx->device)); int rcw = join_tail(b); if (!rcw) rcw = ensure_whole(b); if (rcw) return rcw; }
    *device_ptr = b->acc;
    if (nbytes) *nbytes = 2 * sizeof(G1J) * b->groups;   // raw points, no failure word: see h2v_batch_export_accumulators
    return 0;
}
void* h2v_batch_stream(h2v_batch* b) { return b ? (void*)b->stream : nullptr; }
int h2v_batch_set_stream(h2v_batch* b, void* hip_stream) {
    if (!b) return H2V_ERR_BAD_ARGUMENT;
    hipSetDevice(b->ctx->device);
    if (b->stream) hipStreamSynchronize(b->stream);
    if (b->owns_stream && b->stream) hipStreamDestroy(b->stream);
    b->stream = (hipStream_t)hip_stream; b->owns_stream = false;
    return join_tail(b);   // (the new stream waits for what the last launch left on the auxiliary stream)
}
int h2v_batch_export_accumulators(h2v_batch* b, void* device_dst) {
    if (!b || !b->launched || !device_dst) { set_last_error("h2v_batch_export_accumulators: nothing launched"); return H2V_ERR_BAD_ARGUMENT; }
    H2V_HIP_CHECK(hipSetDevice(b->ctx->device));
    return export_batch_records(b, device_dst);
}
int h2v_batch_fold_check_enqueue(h2v_batch* b, const void* device_accumulators, size_t n_parts) {
    if (!b || !b->launched || !device_accumulators || !n_parts) { set_last_error("h2v_batch_fold_check_enqueue: bad argument"); return H2V_ERR_BAD_ARGUMENT; }
    H2V_HIP_CHECK(hipSetDevice(b->ctx->device));
    int rc;
    if ((rc = join_tail(b))) return rc;
    const uint32_t G = b->groups;
    // the fold keeps the cut of this rank's own launch: records cut the same way add up piece by piece, the pairing takes the pieces
    // (the folded pieces replace the rank's own in the workspace: they were exported before the collective that brought these records)
    if (b->split.parts) {
        G1JSlot* pieces = b->ws.pieces; G1JSlot* ready = b->ws.pieces + (size_t)MSM_MAX_PARTS * b->ws.cap_problems;
        if ((rc = fold_records_enqueue(b->stream, device_accumulators, (uint32_t)n_parts, G, b->split.parts, b->split.shift, b->acc, pieces, ready, b->fold_failed))) return rc;
    } else if ((rc = fold_records_enqueue(b->stream, device_accumulators, (uint32_t)n_parts, G, 1, 0, b->acc, nullptr, nullptr, b->fold_failed))) return rc;
    if ((rc = close_enqueue(b, true))) return rc;
    b->with_pairing = true;
    return 0;
}
int h2v_batch_set_profiling(h2v_batch* b, int level) { if (!b) return H2V_ERR_BAD_ARGUMENT; b->profiling = level == 0 ? 0 : (level == H2V_PROFILE_KERNEL ? 1 : 2); return 0; }
int h2v_batch_timings(h2v_batch* b, float* ms, int cap) {
    if (!b || !ms) return H2V_ERR_BAD_ARGUMENT;
    int k = cap < 7 ? cap : 7;
    for (int i = 0; i < k; ++i) ms[i] = b->last_ms[i];
    return k;
}

// (the caller holds ctx->mu)
static int fold_check_locked(h2v_ctx* ctx, const void* device_accumulators, size_t n_parts, int* ok, uint8_t* out_left_xy, uint8_t* out_right_xy) {
    H2V_HIP_CHECK(hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    DevBuf<G1J> acc; DevBuf<uint32_t> d_ok, d_ident, d_failed; DevBuf<uint8_t> d_out;   // freed on every return path
    int rc;
    if ((rc = acc.alloc(2)) || (rc = d_ok.alloc(1)) || (rc = d_out.alloc(128)) || (rc = d_ident.alloc(2)) || (rc = d_failed.alloc(1))) return rc;
    uint32_t okv = 0, failed = 0; uint8_t outb[128];
    if ((rc = fold_records_enqueue(s, device_accumulators, (uint32_t)n_parts, 1, 1, 0, acc.p, nullptr, nullptr, d_failed.p))) return rc;   // whole points: records in pieces are put together
    if ((rc = pairing_check_enqueue(s, ctx->pairing, acc.p, 1, d_ok.p))) return rc;
    if ((rc = point_to_bytes_enqueue(s, acc.p, d_out.p, d_ident.p, 2))) return rc;
    H2V_HIP_CHECK(hipMemcpyAsync(&okv, d_ok.p, 4, hipMemcpyDeviceToHost, s));
    H2V_HIP_CHECK(hipMemcpyAsync(&failed, d_failed.p, 4, hipMemcpyDeviceToHost, s));
    H2V_HIP_CHECK(hipMemcpyAsync(outb, d_out.p, 128, hipMemcpyDeviceToHost, s));
    H2V_HIP_CHECK(hipStreamSynchronize(s));
    *ok = (okv && !failed) ? 1 : 0;
    if (out_left_xy) memcpy(out_left_xy, outb, 64);
    if (out_right_xy) memcpy(out_right_xy, outb + 64, 64);
    return 0;
}
int h2v_fold_check(h2v_ctx* ctx, const void* device_accumulators, size_t n_parts, int* ok, uint8_t out_left_xy[64], uint8_t out_right_xy[64]) {
    if (!ctx || !device_accumulators || !n_parts || !ok) { set_last_error("h2v_fold_check: bad argument"); return H2V_ERR_BAD_ARGUMENT; }
    std::lock_guard<std::mutex> lock(ctx->mu);
    return fold_check_locked(ctx, device_accumulators, n_parts, ok, out_left_xy, out_right_xy);
}

int h2v_verify_batch(h2v_ctx* ctx, size_t n, const uint8_t* const* proofs, const size_t* proof_lens, const uint8_t* const* instances32, size_t n_instance_columns,
                     const size_t* col_lens, const uint8_t* rand32, int* per_proof_status, int* batch_ok, uint8_t out_left_xy[64], uint8_t out_right_xy[64]) {
    return pack_and_run(ctx, n, proofs, proof_lens, instances32, n_instance_columns, col_lens, rand32, false, 1, per_proof_status, batch_ok, out_left_xy, out_right_xy, nullptr);
}

// AccumulatorStrategy::with(msm_accumulator) (kzg/strategy.rs:75-78): the strategy starts from an existing DualMSM — the
// reference's only pause / resume hook — instead of an empty one.  Every later process() scales the WHOLE accumulator by its fresh
// draw before the proof's Guard joins (strategy.rs:129), so the seed ends up scaled by the product of ALL n draws of this call:
// the seed's two channels are evaluated (two pooled MSMs with the scalars already multiplied by that product), written as a
// record, and folded with the batch's own record into the one pairing.
int h2v_verify_batch_seeded(h2v_ctx* ctx, size_t n, const uint8_t* const* proofs, const size_t* proof_lens, const uint8_t* const* instances32, size_t n_instance_columns,
                            const size_t* col_lens, const uint8_t* rand32,
                            const uint8_t* seed_left_scalars32, const uint8_t* seed_left_bases64, size_t n_seed_left,
                            const uint8_t* seed_right_scalars32, const uint8_t* seed_right_bases64, size_t n_seed_right,
                            int* per_proof_status, int* batch_ok, uint8_t out_left_xy[64], uint8_t out_right_xy[64]) {
    if (!ctx || (n_seed_left && (!seed_left_scalars32 || !seed_left_bases64)) || (n_seed_right && (!seed_right_scalars32 || !seed_right_bases64))) {
        set_last_error("h2v_verify_batch_seeded: null argument"); return H2V_ERR_BAD_ARGUMENT;
    }
    if (n_seed_left > (1u << 24) || n_seed_right > (1u << 24)) { set_last_error("h2v_verify_batch_seeded: seed too large"); return H2V_ERR_BAD_ARGUMENT; }
    int rc;
    std::vector<uint8_t> os_rand;
    if (!rand32 && n) { if ((rc = os_random_scalars(os_rand, n))) return rc; rand32 = os_rand.data(); }
    // the product of this call's draws, and the seed's scalars times it (host: a few hundred Fr products)
    Fr M = Fr::one();
    for (size_t i = 0; i < n; ++i) { Fr r; if (!Fr::from_bytes(rand32 + 32 * i, r)) { set_last_error("h2v_verify_batch_seeded: rand32 scalar not canonical"); return H2V_ERR_BAD_ARGUMENT; } M = M * r; }
    std::vector<uint8_t> sc[2];
    const uint8_t* in_s[2] = {seed_left_scalars32, seed_right_scalars32};
    const size_t ns[2] = {n_seed_left, n_seed_right};
    for (int side = 0; side < 2; ++side) {
        sc[side].resize(32 * ns[side]);
        for (size_t j = 0; j < ns[side]; ++j) {
            Fr v;
            if (!Fr::from_bytes(in_s[side] + 32 * j, v)) { set_last_error("h2v_verify_batch_seeded: seed scalar not canonical"); return H2V_ERR_BAD_ARGUMENT; }
            (v * M).to_bytes(&sc[side][32 * j]);
        }
    }
    uint8_t seed_xy[128];
    int ident = 0;
    if ((rc = h2v_msm_g1(ctx, sc[0].data(), seed_left_bases64, n_seed_left, seed_xy, &ident))) return rc;         // (rejects bases that are not on the curve)
    if ((rc = h2v_msm_g1(ctx, sc[1].data(), seed_right_bases64, n_seed_right, seed_xy + 64, &ident))) return rc;
    // the proofs: one batch without its pairing, kept for the fold
    h2v_batch* b = nullptr;
    int ok_unused = 0;
    if ((rc = pack_and_run(ctx, n, proofs, proof_lens, instances32, n_instance_columns, col_lens, rand32, false, 0, per_proof_status, &ok_unused, nullptr, nullptr, &b))) return rc;
    std::lock_guard<std::mutex> lock(ctx->mu);
    do {
        if (hipSetDevice(ctx->device) != hipSuccess) { rc = H2V_ERR_DEVICE; break; }
        DevBuf<uint8_t> d_xy, d_records; DevBuf<G1A> d_aff; DevBuf<G1J> d_jac; DevBuf<uint32_t> d_flags;
        if ((rc = d_xy.alloc(128)) || (rc = d_records.alloc(2 * H2V_ACC_RECORD_BYTES)) || (rc = d_aff.alloc(2)) || (rc = d_jac.alloc(2)) || (rc = d_flags.alloc(2))) break;
        hipStream_t s = b->stream;
        if (hipMemcpyAsync(d_xy.p, seed_xy, 128, hipMemcpyHostToDevice, s) != hipSuccess) { set_last_error("h2v_verify_batch_seeded: copy failed"); rc = H2V_ERR_DEVICE; break; }
        if ((rc = bases_from_bytes_enqueue(s, d_xy.p, d_aff.p, d_flags.p, 2))) break;
        if ((rc = affine_to_jacobian_enqueue(s, d_aff.p, d_jac.p, 2))) break;
        if ((rc = export_batch_records(b, d_records.p))) break;                                                                  // record 0: the proofs of this call
        if ((rc = export_records_enqueue(s, d_jac.p, nullptr, 1, 0, nullptr, 0, 1, d_records.p + H2V_ACC_RECORD_BYTES))) break;   // record 1: the scaled seed
        if ((rc = h2v_batch_fold_check_enqueue(b, d_records.p, 2))) break;
        int ok = 0;
        if ((rc = finish_impl(b, nullptr, &ok, out_left_xy, out_right_xy))) break;   // (synchronises: the scoped buffers outlive their use)
        if (per_proof_status) for (size_t i = 0; i < n; ++i) if (per_proof_status[i] != 0) ok = 0;   // (short proofs: statuses forced on the host)
        if (batch_ok) *batch_ok = ok;
    } while (0);
    if (rc) h2v_batch_destroy(b); else scratch_batch_give(ctx, b);
    return rc;
}

// N x verify_proof with per-proof instance shapes (lib.rs:33-49 takes `instances` per call): proofs are grouped by shape
// (one compiled plan each), every group runs as its own batch without a pairing, and the groups' accumulator records are
// folded into the single pairing.  The multiplier of proof i is the product of the draws of ALL later proofs in call order
// (kzg/strategy.rs:129, msm.rs:173-176), whatever group they fall in: the suffix products are computed once over the whole
// sequence and every group gathers its own.
// The instance shapes of a call are chosen by whoever supplies the proofs, and every distinct shape costs a plan compilation
// (O(program length^2) host work, ~10 device uploads) and a resize of the batch workspace: a call takes at most
// H2V_MAX_SHAPES_PER_CALL distinct shapes (H2V_ERR_UNSUPPORTED beyond), the groups share ONE batch object, and the plans go
// through the context's bounded cache (H2V_MAX_CACHED_PLANS, least recently used out).
#define H2V_MAX_SHAPES_PER_CALL 64
int h2v_verify_batch_shapes(h2v_ctx* ctx, size_t n, const uint8_t* const* proofs, const size_t* proof_lens, const uint8_t* const* instances32,
                            size_t n_instance_columns, const size_t* col_lens_per_proof, const uint8_t* rand32, int* per_proof_status, int* batch_ok,
                            uint8_t out_left_xy[64], uint8_t out_right_xy[64]) {
    if (!ctx || (n && (!proofs || !proof_lens)) || (n && n_instance_columns && !col_lens_per_proof)) { set_last_error("h2v_verify_batch_shapes: null argument"); return H2V_ERR_BAD_ARGUMENT; }
    if (!ctx->vk) { set_last_error("the context was created without a VerifyingKey"); return H2V_ERR_BAD_ARGUMENT; }
    if (n_instance_columns != ctx_total_instance_columns(ctx)) { set_last_error("instances do not match the VK's instance column count"); return H2V_ERR_INVALID_INSTANCES; }
    const size_t nc = n_instance_columns;
    std::vector<std::pair<std::vector<size_t>, std::vector<size_t>>> groups;   // (shape, proof indices) in first-appearance order
    std::map<std::vector<size_t>, size_t> group_of;
    for (size_t i = 0; i < n; ++i) {
        std::vector<size_t> shape(col_lens_per_proof + i * nc, col_lens_per_proof + (i + 1) * nc);
        auto it = group_of.find(shape);
        if (it == group_of.end()) {
            if (groups.size() == H2V_MAX_SHAPES_PER_CALL) { set_last_error("h2v_verify_batch_shapes: more than 64 distinct instance shapes in one call"); return H2V_ERR_UNSUPPORTED; }
            it = group_of.emplace(shape, groups.size()).first;
            groups.push_back({shape, {}});
        }
        groups[it->second].second.push_back(i);
    }
    if (groups.size() <= 1)
        return h2v_verify_batch(ctx, n, proofs, proof_lens, instances32, nc, n ? col_lens_per_proof : nullptr, rand32, per_proof_status, batch_ok, out_left_xy, out_right_xy);
    std::vector<uint8_t> os_rand;
    int rc;
    if (!rand32) { if ((rc = os_random_scalars(os_rand, n))) return rc; rand32 = os_rand.data(); }
    for (size_t i = 0; i < n; ++i) if (!scalar_is_canonical(rand32 + 32 * i)) { set_last_error("h2v_verify_batch_shapes: rand32 scalar not canonical"); return H2V_ERR_BAD_ARGUMENT; }
    std::lock_guard<std::mutex> lock(ctx->mu);   // a one-shot entry point: it owns the context's stream and scratch batch for its duration
    H2V_HIP_CHECK(hipSetDevice(ctx->device));
    // whole-sequence multipliers on the context's stream
    DevBuf<uint8_t> d_rand; DevBuf<Fr> d_mult; DevBuf<uint8_t> d_records; DevBuf<uint32_t> d_idx;
    if ((rc = d_rand.alloc(32 * n)) || (rc = d_mult.alloc(n)) || (rc = d_records.alloc(H2V_ACC_RECORD_BYTES * groups.size())) || (rc = d_idx.alloc(n))) return rc;
    H2V_HIP_CHECK(hipMemcpyAsync(d_rand.p, rand32, 32 * n, hipMemcpyHostToDevice, ctx->stream));
    if ((rc = multipliers_enqueue(ctx->stream, d_rand.p, (uint32_t)n, (uint32_t)n, 1, d_mult.p))) return rc;
    H2V_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    // one batch object serves every shape group (its workspace follows the group's plan: ensure_buffers)
    size_t max_group = 1, max_inst = 0;
    for (auto& g : groups) { max_group = std::max(max_group, g.second.size()); size_t t = 0; for (size_t l : g.first) t += l; max_inst = std::max(max_inst, t); }
    h2v_batch* b = nullptr;
    if ((rc = scratch_batch_take(ctx, max_group, max_inst, &b))) return rc;
    b->want_guard = false;
    bool all_ok = true;
    size_t idx_off = 0;
    for (size_t gi = 0; gi < groups.size() && !rc; ++gi) {
        const std::vector<size_t>& idx = groups[gi].second;
        const size_t m = idx.size();
        PlanPin pin(ctx);
        if ((rc = pin.get(groups[gi].first))) break;
        const Plan& pl = pin.pd->host;
        std::vector<const uint8_t*> pp(m), ip(m); std::vector<size_t> plen(m);
        for (size_t j = 0; j < m; ++j) { pp[j] = proofs[idx[j]]; plen[j] = proof_lens[idx[j]]; ip[j] = instances32 ? instances32[idx[j]] : nullptr; }
        std::vector<uint8_t> flat, iflat; std::vector<int> forced;
        if ((rc = pack_inputs(pl, m, pp.data(), plen.data(), ip.data(), flat, iflat, forced))) break;
        std::vector<uint32_t> idx32(idx.begin(), idx.end());
        if (hipMemcpy(d_idx.p + idx_off, idx32.data(), 4 * m, hipMemcpyHostToDevice) != hipSuccess) { set_last_error("h2v_verify_batch_shapes: hipMemcpy failed"); rc = H2V_ERR_DEVICE; break; }
        std::vector<uint8_t> ones(32 * m, 0);
        for (size_t j = 0; j < m; ++j) ones[32 * j] = 1;      // placeholder draws: the multipliers come from d_mult
        std::vector<int> st(m, 0); int gok = 0;
        if ((rc = h2v_batch_set_groups(b, 1))) break;
        if ((rc = upload_impl(b, m, flat.data(), pl.proof_len, iflat.data(), nc, groups[gi].first.data(), ones.data(), m))) break;
        b->ext_mult = d_mult.p; b->ext_idx = d_idx.p + idx_off;
        rc = launch_impl(b, 0);
        if (!rc) rc = export_batch_records(b, d_records.p + gi * H2V_ACC_RECORD_BYTES);
        if (!rc) rc = finish_impl(b, st.data(), &gok, nullptr, nullptr);
        b->ext_mult = nullptr; b->ext_idx = nullptr;
        if (rc) break;
        for (size_t j = 0; j < m; ++j) {
            int v = forced[j] ? forced[j] : st[j];
            if (per_proof_status) per_proof_status[idx[j]] = v;
            if (v != 0) all_ok = false;
        }
        idx_off += m;
    }
    if (rc) { h2v_batch_destroy(b); return rc; }
    scratch_batch_give(ctx, b);
    int ok = 0;
    if ((rc = fold_check_locked(ctx, d_records.p, groups.size(), &ok, out_left_xy, out_right_xy))) return rc;
    if (batch_ok) *batch_ok = (ok && all_ok) ? 1 : 0;
    return 0;
}

int h2v_verify_each(h2v_ctx* ctx, size_t n, const uint8_t* const* proofs, const size_t* proof_lens, const uint8_t* const* instances32, size_t n_instance_columns,
                    const size_t* col_lens, int* per_proof_status) {
    int ok = 0;
    return pack_and_run(ctx, n, proofs, proof_lens, instances32, n_instance_columns, col_lens, nullptr, true, 1, per_proof_status, &ok, nullptr, nullptr, nullptr);
}

int h2v_guard_msm(h2v_ctx* ctx, const uint8_t* proof, size_t proof_len, const uint8_t* instances32, size_t n_instance_columns, const size_t* col_lens,
                  uint8_t* right_scalars32, uint8_t* right_bases64, size_t* n_right, uint8_t* left_scalars32, uint8_t* left_bases64, size_t* n_left,
                  uint8_t* challenges32, size_t* n_challenges) {
    if (!ctx || !proof || !n_right || !n_left) { set_last_error("h2v_guard_msm: null argument"); return H2V_ERR_BAD_ARGUMENT; }
    uint8_t one[32] = {1};
    const uint8_t* pp[1] = {proof}; size_t pl1[1] = {proof_len}; const uint8_t* ip[1] = {instances32};
    int st = 0, ok = 0;
    h2v_batch* b = nullptr;
    int rc = pack_and_run(ctx, 1, pp, pl1, ip, n_instance_columns, col_lens, one, false, 0, &st, &ok, nullptr, nullptr, &b, true);
    if (rc) return rc;
    if (st != 0) { h2v_batch_destroy(b); return st; }
    const Plan& pl = b->plan->host;
    size_t T = pl.right_term_order.size();
    do {
        size_t TL = pl.left_term_order.size();
        if (T > *n_right || TL > *n_left) { set_last_error("h2v_guard_msm: output capacity too small"); rc = H2V_ERR_BAD_ARGUMENT; break; }
        std::vector<uint32_t> lscal((size_t)pl.n_points * 8);
        if (hipMemcpy(lscal.data(), b->left_scal, lscal.size() * 4, hipMemcpyDeviceToHost) != hipSuccess) { rc = H2V_ERR_DEVICE; break; }
        std::vector<uint32_t> scal((size_t)pl.n_points * 8);
        std::vector<Fr> shared(pl.n_shared);
        std::vector<G1A> pts(pl.n_points + pl.n_shared);
        std::vector<Fr> chal(pl.squeeze_at.size());
        if (hipMemcpy(scal.data(), b->msm_scal, scal.size() * 4, hipMemcpyDeviceToHost) != hipSuccess || hipMemcpy(shared.data(), b->shared, sizeof(Fr) * pl.n_shared, hipMemcpyDeviceToHost) != hipSuccess ||
            hipMemcpy(pts.data(), b->pts, sizeof(G1A) * pts.size(), hipMemcpyDeviceToHost) != hipSuccess || hipMemcpy(chal.data(), b->chal, sizeof(Fr) * chal.size(), hipMemcpyDeviceToHost) != hipSuccess) { rc = H2V_ERR_DEVICE; break; }
        auto put_pt = [](const G1A& p, uint8_t* o) { if (p.is_identity()) memset(o, 0, 64); else { p.x.to_bytes(o); p.y.to_bytes(o + 32); } };
        if (!pl.guard_term_order.empty()) {
            // GWC: term by term as the reference appends them (gwc.rs:86-132), each with its own scalar
            T = pl.guard_term_order.size();
            if (T > *n_right) { set_last_error("h2v_guard_msm: output capacity too small"); rc = H2V_ERR_BAD_ARGUMENT; break; }
            std::vector<uint32_t> gs(T * 8);
            if (hipMemcpy(gs.data(), b->guard_scal, gs.size() * 4, hipMemcpyDeviceToHost) != hipSuccess) { rc = H2V_ERR_DEVICE; break; }
            for (size_t t = 0; t < T; ++t) {
                auto w = pl.guard_term_order[t];
                memcpy(right_scalars32 + 32 * t, &gs[t * 8], 32);
                put_pt(pts[w.first ? pl.n_points + w.second : w.second], right_bases64 + 64 * t);
            }
        } else
        for (size_t t = 0; t < T; ++t) {
            auto w = pl.right_term_order[t];
            if (w.first) { shared[w.second].to_bytes(right_scalars32 + 32 * t); put_pt(pts[pl.n_points + w.second], right_bases64 + 64 * t); }
            else { memcpy(right_scalars32 + 32 * t, &scal[(size_t)w.second * 8], 32); put_pt(pts[w.second], right_bases64 + 64 * t); }
        }
        *n_right = T;
        for (size_t t = 0; t < TL; ++t) {
            uint32_t slot = pl.left_term_order[t].second;
            memcpy(left_scalars32 + 32 * t, &lscal[(size_t)slot * 8], 32);
            put_pt(pts[slot], left_bases64 + 64 * t);
        }
        *n_left = TL;
        if (challenges32 && n_challenges) {
            // reorder squeeze order -> [user challenges.., theta, beta, gamma, y, x, y', v, u]
            size_t nc = pl.n_challenges;
            if (nc > *n_challenges) { set_last_error("h2v_guard_msm: challenge capacity too small"); rc = H2V_ERR_BAD_ARGUMENT; break; }
            // squeeze order is the transcript order; user challenges are interleaved by phase. Rebuild the map as compile_plan did.
            const VkHost& vk = ctx->vk->vk;
            std::vector<uint32_t> order;
            uint8_t max_phase = 0; for (uint8_t p2 : vk.advice_column_phase) max_phase = std::max(max_phase, p2);
            for (unsigned ph = 0; ph <= max_phase; ++ph) for (uint32_t i = 0; i < vk.num_challenges; ++i) if (vk.challenge_phase[i] == ph) order.push_back(i);
            for (uint32_t i = 0; i + vk.num_challenges < nc; ++i) order.push_back(vk.num_challenges + i);
            for (size_t q = 0; q < order.size() && q < chal.size(); ++q) chal[q].to_bytes(challenges32 + 32 * order[q]);
            *n_challenges = nc;
        }
    } while (0);
    h2v_batch_destroy(b);
    return rc;
}

}  // extern "C"
