// The batch object behind h2v_batch and the argument blocks of the per-proof kernels.
#pragma once
#include "ctx.h"
#include "vkplan.h"

namespace h2v {

// Per-proof status words on the device.  The reference stops at the FIRST failing step of verify_proof, so when several
// kernels (or several lanes of one) find different faults in one proof the earliest step of the reference's sequence must
// win, whatever order the lanes run in: instance values are typed Fr before the call (lib.rs:33-49), then the main transcript
// reads (Error::Transcript, lib.rs:91-253), then the inversions of the evaluation part (the reference panics: vanishing.rs:100,
// domain.rs:187-212), then the multi-open reads (Error::Opening, lib.rs:420-424).  Every writer uses atomicMin on these
// rank-coded values; the host translates them back to the ABI's codes (status_decode).
#define H2V_DEV_ST_INVALID_INSTANCES (-40)
#define H2V_DEV_ST_TRANSCRIPT (-30)
#define H2V_DEV_ST_PANIC (-20)
#define H2V_DEV_ST_OPENING (-10)
__device__ __forceinline__ void status_set(int* status, uint32_t p, int dev_code) { atomicMin(&status[p], dev_code); }
// a status word that another wave of the same kernel may have set (the Fr program's inversion, ordered before this read by a
// workgroup barrier): read at device scope, past the CU's vector cache, like the atomic that wrote it
__device__ __forceinline__ int status_get(const int* status, uint32_t p) { return __hip_atomic_load(&status[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
inline int status_decode(int dev) {
    switch (dev) {
        case 0: return 0;
        case H2V_DEV_ST_INVALID_INSTANCES: return H2V_ERR_INVALID_INSTANCES;
        case H2V_DEV_ST_TRANSCRIPT: return H2V_ERR_TRANSCRIPT;
        case H2V_DEV_ST_PANIC: return H2V_ERR_REFERENCE_PANIC;
        case H2V_DEV_ST_OPENING: return H2V_ERR_OPENING;
        default: return dev;
    }
}

struct FrvmArgs {
    const VmInstr* code; uint32_t n_code;   // the program as ONE stream (k_frvm)
    const Fr* consts;
    Fr* slots;
    uint32_t n;
    const uint8_t* proofs; uint32_t proof_len; const uint32_t* scalar_offsets;
    const uint8_t* inst; uint32_t ninst;
    const Fr* chal; const Fr* mult;
    int* status;
    uint32_t* msm_scal; uint32_t np;
    Fr* shared;
    uint32_t* left_scal;
    const Fr* insteval;   // [query][proof], wide instance vectors only
    uint32_t* guard_scal; uint32_t n_guard;   // [proof][term][8], guard variant of a GWC plan only (h2v_guard_msm)
    // the same program as 2 / 3 / 4 instruction streams per proof (k_frvm2; index K - 2, slot numbering of its own): frvm_enqueue chooses
    const VmInstr* code_k[3][FRVM_MAX_STREAMS] = {{nullptr}}; uint32_t n_code_k[3][FRVM_MAX_STREAMS] = {{0}}; uint32_t n_slots_k[3] = {0, 0, 0};
    uint32_t streams = 0;   // set by frvm_enqueue: the K the launch uses
    int force_streams = 0, force_lds_kb = 0;   // h2v_tuning (0 = automatic)
};

// sum_j inst[base + j] * l_{j - rot}(x) for one instance query of every proof (lib.rs:173-218; l_i_range poly/domain.rs:187-212)
struct InstEvalArgs {
    const uint8_t* inst; uint32_t ninst;      // canonical instance bytes [proof][ninst][32]
    const Fr* chal; uint32_t x_chal;          // challenges [c][proof]; index of x
    uint32_t n, k;                            // proofs; domain size 2^k
    uint32_t base, len;                       // the query's column inside a proof's instance values
    Fr w_start, omega, omega_step, omega_step_inv, n_inv;   // omega^(-rot), omega, omega^256, omega^(-256), 1/2^k
    Fr* out;                                  // [proof]
    int* status;
};
int instance_eval_enqueue(hipStream_t s, const InstEvalArgs& a);

struct StageArgs {
    uint32_t n;
    const Plan* plan; const PlanDevice* pd;
    const uint8_t* proofs; const uint8_t* inst;
    G1A* pts; G1A* phi; uint8_t* ycanon; int* status;
    unsigned long long* words; uint32_t stream_words;
    Fr* chal;
};

int decompress_stage_enqueue(hipStream_t s, const StageArgs& g);
// the same stage in pieces (h2v_batch_upload_launch): reset the status words; decompress the points of proofs [p0, p1); check the
// scalars of all proofs
int decompress_begin_enqueue(hipStream_t s, const StageArgs& g);
// (src / src_stride: read the proof bytes from there instead of g.proofs — the caller's host buffer, h2v_batch_upload_launch)
int decompress_range_enqueue(hipStream_t s, const StageArgs& g, uint32_t p0, uint32_t p1, const uint8_t* src = nullptr, uint32_t src_stride = 0);
int decompress_finish_enqueue(hipStream_t s, const StageArgs& g);
int transcript_stage_enqueue(hipStream_t s, const StageArgs& g);
// groups > 1: group g owns proofs [g*n/groups, ..) and the draws tail[g*n_tail/groups, ..)
int multipliers_enqueue(hipStream_t s, const uint8_t* d_tail, uint32_t n_tail, uint32_t n, uint32_t groups, Fr* d_mult);
// out[i] = src[idx[i]]: the multipliers of a non-contiguous subset of a larger accumulation
int gather_multipliers_enqueue(hipStream_t s, const Fr* d_src, const uint32_t* d_idx, uint32_t n, Fr* d_out);
int frvm_enqueue(hipStream_t s, const FrvmArgs& a, uint32_t n_slots);
int fold_shared_enqueue(hipStream_t s, const Fr* d_shared, uint32_t n, uint32_t np, uint32_t n_shared, uint32_t groups, uint32_t* d_msm_scal);
}  // namespace h2v

struct h2v_batch {
    h2v_ctx* ctx = nullptr;
    hipStream_t stream = nullptr;
    bool owns_stream = true;
    hipStream_t aux = nullptr;        // the accumulators' affine conversion runs here, beside the pairing (both only read them)
    hipEvent_t ev_fork = nullptr, ev_join = nullptr, ev_fork0 = nullptr, ev_join0 = nullptr;
    // h2v_batch_upload_launch: the host -> device copies run on `copy`, chunk by chunk, each followed (after the blocking copy has
    // returned) by the decompression of that chunk on `stream`
    hipStream_t copy = nullptr;
    bool decompressed = false;        // the next launch finds its points already decompressed (set by h2v_batch_upload_launch)
    size_t max_proofs = 0, max_inst = 0;
    h2v::PlanDevice* plan = nullptr;  // set at upload (depends on the instance shape)
    uint32_t n = 0, n_tail = 0;
    uint32_t groups = 1;              // independent accumulator batches inside this launch (h2v_batch_set_groups)
    const h2v::Fr* ext_mult = nullptr; const uint32_t* ext_idx = nullptr;   // multipliers gathered from a larger sequence (h2v_verify_batch_shapes)
    bool launched = false, with_pairing = false;
    // device buffers (sized for max_proofs with the plan of the first upload; re-allocated if a later plan needs more)
    uint8_t* proofs = nullptr; uint8_t* inst = nullptr; uint8_t* tail = nullptr;
    h2v::G1A* pts = nullptr; h2v::G1A* phi = nullptr;   // the batch's points + the VK-wide bases, and their images under the GLV endomorphism (same shape)
    uint8_t* ycanon = nullptr; int* status = nullptr;
    unsigned long long* words = nullptr; h2v::Fr* chal = nullptr; h2v::Fr* mult = nullptr; h2v::Fr* slots = nullptr;
    uint32_t* msm_scal = nullptr; h2v::Fr* shared = nullptr; uint32_t* left_scal = nullptr;
    h2v::Fr* insteval = nullptr;  // [query][proof] (wide instance vectors)
    uint32_t* guard_scal = nullptr;   // [proof][guard term][8] (h2v_guard_msm with GWC)
    bool want_guard = false;          // the next upload takes the guard variant of the plan
    h2v::G1J* acc = nullptr;      // per group: [2g] left, [2g+1] right
    uint32_t* ok = nullptr;       // [groups] — in the pinned host block (results_host): the pairing kernels write their verdicts straight to the host
    uint8_t* out_bytes = nullptr; uint32_t* out_ident = nullptr;
    uint32_t* fold_failed = nullptr;  // [groups] failed proofs reported by the folded shards (h2v_batch_fold_check_enqueue)
    uint8_t* results = nullptr; uint8_t* results_host = nullptr; size_t results_bytes = 0;   // ok / fold_failed / out_ident / out_bytes / status live in `results`
    h2v::MsmWorkspace ws;
    h2v::MsmSplit split;              // how the last launch left its accumulators to the pairing (parts == 0: whole points in acc)
    bool acc_stale = false;           // a launch without a pairing left pieces only: acc / out_bytes are put together on demand (ensure_whole)
    bool tail_on_aux = false;         // the last launch's whole accumulators, their bytes and the result copy are still the auxiliary stream's business (close_enqueue): join_tail before the main stream touches them
    void* line_ws = nullptr; size_t line_ws_groups = 0;   // k_pair_lines' output, H2V_PAIRING_LINE_WS_BYTES per group
    size_t cap_proof_bytes = 0, cap_inst_bytes = 0, cap_tail = 0, cap_plan_sig = 0;
    uint32_t stream_words = 0;
    // profiling
    int profiling = 0;                // 0 off, 1: the dominant kernel's own events (msm_accumulate), 2: + an event between the stages
    hipEvent_t ev[8] = {nullptr};
    float last_ms[7] = {0, 0, 0, 0, 0, 0, 0};
};
