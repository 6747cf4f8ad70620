// Per-VK host data model and the compiled "verification plan".
//
// The reference re-derives everything below for every proof inside verify_proof (query list,
// rotation sets, which evaluation feeds which expression).  All of it depends only on the
// VerifyingKey, so it is compiled once per context into:
//   * the proof layout        (byte offset of every point / scalar the transcript reads, lib.rs:91-253, shplonk.rs:195-200)
//   * the transcript stream   (which bytes Blake2b absorbs, where challenges are squeezed, transcript/mod.rs:205-232)
//   * a straight-line Fr program (expression evaluation lib.rs:273-346 + SHPLONK scalar preparation shplonk.rs:202-264)
//     executed by a SIMT interpreter with one proof per lane
//   * the MSM term map        (which point slot / shared base each output scalar multiplies)
#pragma once
#include "ctx.h"
#include <map>
#include <mutex>

namespace h2v {

// ---------------------------------------------------------------- VK data model (plonk/vk.rs:16-26,173-211)
static const uint8_t COL_INSTANCE = 254, COL_FIXED = 255;  // 0..2 = advice phase (plonk/circuit.rs:36-65)
struct ColumnH { uint32_t index; uint8_t type; };
struct QueryH { ColumnH column; int32_t rotation; };
struct TermH { uint16_t coeff_idx; std::vector<std::pair<uint32_t, uint32_t>> factors; };
struct ExprH { uint32_t num_vars = 0; std::vector<TermH> terms; };
struct LookupH { std::vector<ExprH> input, table; };
struct ShuffleH { std::vector<ExprH> input, shuffle; };

struct VkHost {
    uint32_t k = 0, cs_degree = 0;
    std::vector<G1A> fixed_commitments, permutation_commitments;
    Fr transcript_repr;
    uint32_t num_fixed_columns = 0, num_advice_columns = 0, num_instance_columns = 0, num_selectors = 0, num_challenges = 0;
    std::vector<uint8_t> advice_column_phase, challenge_phase;
    std::vector<uint32_t> num_advice_queries;
    std::vector<QueryH> advice_queries, instance_queries, fixed_queries;
    std::vector<ColumnH> permutation_columns;
    std::vector<ExprH> gates;
    std::vector<LookupH> lookups;
    std::vector<ShuffleH> shuffles;
    std::vector<Fr> coeff_vals;
    std::vector<uint8_t> selector_bytes;   // the selector bitmaps as read (num_selectors x ceil(2^k / 8) bytes): unused by verification, kept for VerifyingKey::write
    size_t blinding_factors() const;  // plonk/vk.rs:396-401
};
// VerifyingKey::read (plonk/vk.rs:76-115, 274-365); false + message when the bytes are rejected
bool vk_from_bytes(const uint8_t* data, size_t len, int format, VkHost& out, std::string& err);
// VerifyingKey::write (plonk/vk.rs:41-64); layout: H2V_VK_LAYOUT_WRITER / _READER (serde.hip)
void vk_to_bytes(const VkHost& vk, int format, int layout, std::vector<uint8_t>& out);
void params_to_bytes(const ParamsHost& p, int format, std::vector<uint8_t>& out);

// ---------------------------------------------------------------- Fr program
enum VmOp : uint32_t {
    OP_CONST = 1,      // d <- consts[a]
    OP_MUL = 2,        // d <- a * b
    OP_ADD = 3,
    OP_SUB = 4,
    OP_NEG = 5,        // d <- -a
    OP_INV = 6,        // d <- 1/a ; a == 0 sets the proof's status to H2V_ERR_REFERENCE_PANIC
    OP_POW = 7,        // d <- a ^ b (b immediate)
    OP_SQRN = 8,       // d <- a ^ (2^b) (b immediate)
    OP_LOAD_SCALAR = 9,   // d <- proof scalar #a
    OP_LOAD_INST = 10,    // d <- instance value #a (flat index over the proof's instance columns)
    OP_LOAD_CHAL = 11,    // d <- challenge #a
    OP_LOAD_MULT = 12,    // d <- batch multiplier of this proof
    OP_STORE_MSM = 13,    // msm_scalars[proof][point slot b] <- canonical(a)
    OP_STORE_SHARED = 14, // shared[proof][b] <- a (Montgomery)
    OP_STORE_LEFT = 15,   // left_scalars[proof][point slot b] <- canonical(a)
    OP_LOAD_INSTEVAL = 16, // d <- instance-query evaluation #a of this proof, computed by k_instance_eval (wide instance vectors)
    OP_STORE_GUARD = 17,   // guard_scalars[proof][term b] <- canonical(a): one scalar per term of the Guard in reference order (h2v_guard_msm, GWC)
    OP_BARRIER = 18,       // two-stream programs: both streams of a proof meet here (values cross between them only over a barrier)
};
struct VmInstr { uint32_t op, d, a, b; };
#define FRVM_MAX_STREAMS 4
#define VM_CONST_OPERAND 0x80000000u   // operand a / b of MUL, ADD, SUB: consts[index] instead of a slot

struct TranscriptSrc {  // one byte of the absorbed stream
    enum Kind : uint8_t { CONST = 0, PROOF = 1, PROOF_MASKED = 2, YCOORD = 3, INSTANCE = 4 };
    uint8_t kind; uint8_t value; uint32_t offset;
};

struct PlanOptions { int multiopen = 0; int transcript = 0; int circuit_instances = 1; bool guard_terms = false; int instance_kernel_threshold = 0; };  // h2v_options (+ the h2v_guard_msm variant)

struct Plan {
    PlanOptions opts;
    double dag_work = 0, dag_critical_path = 0, makespan_k[3] = {0, 0, 0};   // diagnostics: Fr-program DAG in products; the scheduler's estimate for 2 / 3 / 4 streams
    // proof layout
    uint32_t n_points = 0, n_scalars = 0, proof_len = 0;
    std::vector<uint32_t> point_offsets, scalar_offsets;  // byte offsets into the proof
    uint32_t n_main_points = 0;                           // points read before the multi-open part (errors there are Transcript, after: Opening)
    uint32_t slot_h1 = 0, slot_h2 = 0;                    // SHPLONK only
    uint32_t opening_offset = 0;                          // byte offset of the first point of the multi-open part
    std::vector<std::pair<uint8_t, uint32_t>> left_term_order;  // reference order of the left channel's terms
    // transcript
    std::vector<TranscriptSrc> stream;     // absorbed byte stream incl. the 0x00 challenge markers
    std::vector<uint32_t> squeeze_at;      // stream length at which challenge i is squeezed
    uint32_t n_challenges = 0;             // user challenges + theta, beta, gamma, y, x, y', v, u
    uint32_t n_user_challenges = 0;
    // program
    std::vector<VmInstr> code;
    std::vector<Fr> consts;
    uint32_t n_slots = 0;
    // the same program as 2, 3 and 4 instruction streams per proof (k_frvm2; index K - 2), each with its own slot numbering
    std::vector<VmInstr> code_k[3][FRVM_MAX_STREAMS];
    uint32_t n_slots_k[3] = {0, 0, 0};
    // MSM map
    uint32_t n_shared = 0;                  // fixed (queried) + permutation + g
    std::vector<G1A> shared_bases;
    // reference term order of the right channel (shplonk.rs:256-264): (is_shared, index)
    std::vector<std::pair<uint8_t, uint32_t>> right_term_order;
    // GWC, guard variant only: the right channel term by term as the reference appends it (gwc.rs:86-132: witness_with_aux, then
    // commitment_multi query by query, then (eval_multi, -g)) — a commitment opened at several points occurs once per query here,
    // while the MSM above carries it once with the scalars summed; each term's own scalar is stored by OP_STORE_GUARD
    std::vector<std::pair<uint8_t, uint32_t>> guard_term_order;
    // instance shape this plan was compiled for
    std::vector<size_t> col_lens;
    uint32_t n_instance_values = 0;
    // Wide instance vectors (more than 1024 values; h2v_options.instance_kernel_threshold): sum_j inst[j] * l_{j-rot}(x) (lib.rs:173-218) is not unrolled
    // into the Fr program (9 instructions and a slot per public input) but evaluated by k_instance_eval, one workgroup per proof
    struct InstQuery { uint32_t base, len; Fr w_start; };   // flat offset / length of the column, omega^(-rotation)
    bool wide_instances = false;
    std::vector<InstQuery> inst_queries;
    uint32_t x_chal = 0, domain_k = 0;
    Fr omega, n_inv;
};
// Returns 0 or an H2V error code (InstanceTooLarge, ReferencePanic for an empty gate polynomial, ...)
int compile_plan(const VkHost& vk, const ParamsHost& params, const std::vector<size_t>& col_lens, PlanOptions opts, Plan& out, std::string& err);

struct PlanDevice {
    Plan host;
    VmInstr* code = nullptr;
    VmInstr* code_k[3][FRVM_MAX_STREAMS] = {{nullptr}};
    Fr* consts = nullptr;
    TranscriptSrc* stream = nullptr;
    uint32_t* squeeze_at = nullptr;
    uint32_t* point_offsets = nullptr;
    uint32_t* scalar_offsets = nullptr;
    G1A* shared_bases = nullptr; G1A* shared_phi = nullptr;   // the VK-wide bases and their images under phi (MsmProblem::phi2)
    int pins = 0;             // users that hold the plan (batches between upload and their next upload / destruction, entry points while
                              // they read it); guarded by VkDevice::mu.  Only unpinned plans are evicted.
    uint64_t last_use = 0;    // VkDevice::clock at the last ctx_get_plan
    int upload();
    void release();
};

// Compiled plans of one VK, keyed by instance column lengths.  A plan costs a compilation (one single-stream emit + three list
// schedules, O(n^2) in the program length) and ~10 device allocations, and callers choose the key — the instance shapes of
// UNTRUSTED proofs (h2v_verify_batch_shapes) — so the cache is bounded: beyond H2V_MAX_CACHED_PLANS the least recently used plan
// that nobody holds is released.
#define H2V_MAX_CACHED_PLANS 32
struct VkDevice {
    VkHost vk;
    std::map<std::vector<size_t>, PlanDevice*> plans;
    uint64_t clock = 0;
    std::mutex mu;
};
// The plan comes back PINNED: every successful ctx_get_plan is paired with one ctx_put_plan (PlanPin does it for a scope).
int ctx_get_plan(h2v_ctx* ctx, const std::vector<size_t>& col_lens, PlanDevice** out, bool guard_terms = false);
void ctx_put_plan(h2v_ctx* ctx, PlanDevice* pd);
struct PlanPin {
    h2v_ctx* ctx; PlanDevice* pd = nullptr;
    explicit PlanPin(h2v_ctx* c) : ctx(c) {}
    PlanPin(const PlanPin&) = delete;
    PlanPin& operator=(const PlanPin&) = delete;
    ~PlanPin() { if (pd) ctx_put_plan(ctx, pd); }
    int get(const std::vector<size_t>& col_lens, bool guard_terms = false) { if (pd) { ctx_put_plan(ctx, pd); pd = nullptr; } return ctx_get_plan(ctx, col_lens, &pd, guard_terms); }
    PlanDevice* take() { PlanDevice* p = pd; pd = nullptr; return p; }   // the pin moves to the caller
};

}  // namespace h2v
