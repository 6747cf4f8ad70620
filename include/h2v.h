/* h2v.h — C ABI of the MI355X-native Halo2/KZG/SHPLONK batch verifier.
 *
 * The reference (ChainSafe/halo2-verifier, pure Rust, no FFI of its own) exposes the hot path
 * through Rust traits; this header is what an `extern "C"` block on the Rust side binds
 * (INTEGRATION.md shows the shim).  Each entry point cites the reference item it replaces;
 * paths are relative to the reference repository root.
 *
 * Conventions
 *   - All buffers are caller-owned; the library never retains a host pointer past the call
 *     that received it.  No exceptions or panics cross the boundary.
 *   - Return value 0 = OK; negative = error.  -1..-6 mirror plonk::Error
 *     (halo2_verifier/src/plonk/mod.rs:19-32) in declaration order.
 *   - Field elements: 32 bytes little-endian canonical (== ff::PrimeField::to_repr).
 *   - G1 points at this boundary: x | y, 64 bytes canonical, all-zero = identity.
 *   - Proofs, VerifyingKey and ParamsKZG bytes are in the reference's own formats
 *     (VerifyingKey::write  halo2_verifier/src/plonk/vk.rs:41-64;
 *      ParamsKZG::write_custom  halo2_verifier/src/poly/kzg/commitment.rs:142-152).
 *   - serde format codes follow helpers.rs:7-19: 0 Processed, 1 RawBytes, 2 RawBytesUnchecked.
 *   - Threading: a context may be shared between host threads: its VK / params / compiled plans are
 *     immutable after creation, and the one-shot entry points (h2v_verify_batch, h2v_verify_each,
 *     h2v_guard_msm, h2v_msm_g1, h2v_pairing_check, h2v_fold_check) serialise themselves on an internal
 *     lock (they share the context's stream and a cached workspace).  A batch object owns one HIP stream
 *     and its device workspace and must be used from one thread at a time; several batches may be in
 *     flight on one context.
 */
#ifndef H2V_H
#define H2V_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define H2V_OK 0
#define H2V_ERR_INVALID_INSTANCES (-1)         /* Error::InvalidInstances          lib.rs:51-55          */
#define H2V_ERR_CONSTRAINT_SYSTEM_FAILURE (-2) /* Error::ConstraintSystemFailure   kzg/strategy.rs:171-175 */
#define H2V_ERR_BOUNDS_FAILURE (-3)            /* Error::BoundsFailure                                     */
#define H2V_ERR_OPENING (-4)                   /* Error::Opening                   lib.rs:420-424        */
#define H2V_ERR_TRANSCRIPT (-5)                /* Error::Transcript                plonk/mod.rs:34-39    */
#define H2V_ERR_INSTANCE_TOO_LARGE (-6)        /* Error::InstanceTooLarge                                  */
#define H2V_ERR_REFERENCE_PANIC (-7)  /* inputs on which the reference panics: zero inverse (vanishing.rs:100, shplonk.rs:215) */
#define H2V_ERR_BAD_ARGUMENT (-16)
#define H2V_ERR_FORMAT (-17)          /* VK / params bytes rejected (io::Error in VerifyingKey::read / ParamsKZG::read_custom) */
#define H2V_ERR_DEVICE (-18)          /* HIP runtime error; h2v_last_error() has the text */
#define H2V_ERR_UNSUPPORTED (-19)

#define H2V_SERDE_PROCESSED 0
#define H2V_SERDE_RAW_BYTES 1
#define H2V_SERDE_RAW_BYTES_UNCHECKED 2

typedef struct h2v_ctx h2v_ctx;
typedef struct h2v_batch h2v_batch;

/* Library / device probe: number of HIP devices visible (0 if none; never fails). */
int h2v_device_count(void);
/* Text of the most recent error on this thread (never NULL). */
const char* h2v_last_error(void);

/* Context = ParamsKZG + VerifyingKey resident on one GPU, plus everything derived from them
 * once per VK (evaluation domain constants, the compiled per-proof program, G2 line
 * coefficients).
 *   replaces: ParamsKZG::read_custom (poly/kzg/commitment.rs:155-207),
 *             VerifyingKey::read (plonk/vk.rs:76-115) -> EvaluationDomain::new (poly/domain.rs:34-140),
 *             G2Prepared::from (poly/kzg/msm.rs:186-187).
 * vk may be NULL (vk_len 0) for a context that only serves h2v_msm_g1 / h2v_pairing_check. */
int h2v_ctx_create(const uint8_t* params, size_t params_len, int params_format,
                   const uint8_t* vk, size_t vk_len, int vk_format,
                   int device, h2v_ctx** out);
/* The two generic parameters of the reference's verify_proof that change what is computed (lib.rs:33-40):
 *   multiopen:  0 = VerifierSHPLONK (poly/kzg/multiopen/shplonk.rs), 1 = VerifierGWC (poly/kzg/multiopen/gwc.rs)
 *   transcript: 0 = Blake2bRead, 1 = Keccak256Read (transcript/mod.rs:104-116)
 * and the length of its `instances: &[&[&[Fr]]]` argument (lib.rs:43,51-55,63):
 *   circuit_instances: how many circuit instances share ONE proof transcript (0 or 1 = one, what every caller inside the reference
 *                      passes).  With M > 1 a proof carries M sets of advice / permutation / lookup / shuffle commitments and
 *                      evaluations in the reference's interleaved order (lib.rs:91-161, 220-253), and every entry point below takes
 *                      n_instance_columns = M x the VK's instance columns, col_lens and instances32 instance-major.
 *                      (Parity for M > 1 is pinned by the repository's two restatements only — oracle/ and oracle/pyref.py,
 *                      the _m2 fixtures under tests/golden — no caller inside the reference passes M > 1.)
 *   struct_size:       sizeof(h2v_options) as the CALLER's header declares it.  The library reads only the fields that lie inside
 *                      it (later fields default to 0), and rejects a value that is not a layout it knows with H2V_ERR_BAD_ARGUMENT:
 *                      a caller built against another revision of this struct gets an error, not another proof layout.
 *   instance_kernel_threshold: debug / test.  0 = the default: instance columns of more than 1024 values in total are summed by the
 *                      wide-instance kernel instead of being unrolled into the per-proof program; n > 0 sets that bound to n - 1
 *                      (1 = every circuit takes the kernel path).
 * h2v_ctx_create == h2v_ctx_create_ex with {sizeof(h2v_options), 0, 0, 1, 0}. */
typedef struct h2v_options { size_t struct_size; int multiopen; int transcript; int circuit_instances; int instance_kernel_threshold; } h2v_options;
#define H2V_OPTIONS_INIT { sizeof(h2v_options), 0, 0, 1, 0 }
#define H2V_ABI_VERSION 3   /* bumped whenever a struct of this header changes layout; h2v_abi_version() returns the library's */
int h2v_abi_version(void);
#define H2V_MULTIOPEN_SHPLONK 0
#define H2V_MULTIOPEN_GWC 1
#define H2V_TRANSCRIPT_BLAKE2B 0
#define H2V_TRANSCRIPT_KECCAK256 1
int h2v_ctx_create_ex(const uint8_t* params, size_t params_len, int params_format,
                      const uint8_t* vk, size_t vk_len, int vk_format,
                      int device, const h2v_options* options, h2v_ctx** out);
/* Debug / test: force the kernel variants that the library otherwise chooses from the shape of a launch (how many proofs, groups,
 * MSM terms it carries).  Every field 0 = automatic, which is what production runs; the library reads no environment variable.
 * The bit-exact suite uses this to put every variant under test at sizes the CPU oracle can follow.  The setting belongs to the
 * context and is read when a launch is enqueued; it is NOT synchronised with launches of other threads (set it while the context is
 * idle).  A NULL pointer restores automatic choice.
 *   frvm_streams        1..4: instruction streams per proof of the Fr program (automatic: 4 up to 341 waves per launch, else 2)
 *   frvm_lds_kb         LDS slice for the program's slots, KB (automatic: 156 / 78 / 36 by launch size)
 *   msm_parts           pieces the accumulators are left in for the pairing (automatic: 6); 1 = whole points (full Horner, whole-point pairing)
 *   msm_global_sort     1: the global counting sort instead of the per-window LDS sort
 *   msm_no_term_split   1: do not cut problems of more than 16 384 terms into sub-problems
 *   msm_window_threads  lanes per window reduction (64, 128, 256; automatic by bucket and window count)
 *   msm_window_wpw      windows per workgroup of the window reduction (1, 2, 4)
 *   msm_window_slots    3: the 20 KB form of the window reduction without the two-bit digit table (automatic: beyond 1024 windows)
 *   msm_acc_waves       3 or 4: waves per SIMD msm_accumulate is compiled for (automatic: 3 — 156 registers, no spills)
 *   pairing_one_stream  1: the single-stream pairing table over split accumulators instead of the two-stream one
 *   upload_mode         h2v_batch_upload_launch: 1 = point bytes first, then one decompression launch under the full copy;
 *                       2 = the proofs in two halves, each decompressed as soon as it has arrived; 3 = plain upload, then launch
 *                       (automatic: see the function) */
typedef struct h2v_tuning {
    size_t struct_size;
    int frvm_streams, frvm_lds_kb;
    int msm_parts, msm_global_sort, msm_no_term_split, msm_window_threads, msm_window_wpw, msm_window_slots, msm_acc_waves;
    int pairing_one_stream;
    int upload_mode;
} h2v_tuning;
int h2v_ctx_set_tuning(h2v_ctx* ctx, const h2v_tuning* tuning);
/* Destroys the context and everything compiled for it.  Every h2v_batch created on it must have been destroyed before. */
void h2v_ctx_destroy(h2v_ctx* ctx);

/* Re-serialisation of a VerifyingKey / ParamsKZG in another SerdeFormat — host only, no device needed:
 *   replaces: VerifyingKey::read(from_format) followed by VerifyingKey::write / to_bytes(to_format)   (plonk/vk.rs:41-123)
 *             ParamsKZG::read_custom followed by write_custom / to_bytes                               (poly/kzg/commitment.rs:142-224)
 * out == NULL: *out_len receives the size.  Otherwise *out_len holds the capacity on entry and the size on return.
 * layout (VerifyingKey only): the reference's writer emits a lookup / shuffle argument as all first expressions, then all second
 * ones (lookup.rs:36-49, shuffle.rs:70-84), while its reader takes them in pairs (lookup.rs:51-68, shuffle.rs:86-102).  They
 * agree for arguments of one expression pair; for more, the reference's read does not invert its write.
 *   H2V_VK_LAYOUT_WRITER: exactly the bytes VerifyingKey::write produces;
 *   H2V_VK_LAYOUT_READER: the bytes that VerifyingKey::read (and h2v_ctx_create) read back as the SAME key. */
#define H2V_VK_LAYOUT_WRITER 0
#define H2V_VK_LAYOUT_READER 1
int h2v_vk_convert(const uint8_t* vk, size_t vk_len, int from_format, int to_format, int layout, uint8_t* out, size_t* out_len);
int h2v_params_convert(const uint8_t* params, size_t params_len, int from_format, int to_format, uint8_t* out, size_t* out_len);

/* Shape of one proof for this VK (Np points, Ns scalars, T_R right-channel terms). */
int h2v_ctx_proof_shape(const h2v_ctx* ctx, size_t* proof_len, size_t* n_points, size_t* n_scalars,
                        size_t* n_right_terms, size_t* n_instance_columns);

/* sum_i scalars[i] * bases[i] in G1.
 *   replaces: MSMKZG::eval + to_affine (poly/kzg/msm.rs:81-86) == best_multiexp (arithmetic.rs:102-108). */
int h2v_msm_g1(h2v_ctx* ctx, const uint8_t* scalars32, const uint8_t* bases64, size_t n,
               uint8_t out_xy[64], int* out_is_identity);

/* e(left, s_g2) * e(right, -g2) == 1 ?
 *   replaces: DualMSM::check after both channels are evaluated (poly/kzg/msm.rs:185-203). */
int h2v_pairing_check(h2v_ctx* ctx, const uint8_t left_xy[64], const uint8_t right_xy[64], int* ok);

/* N x verify_proof under AccumulatorStrategy, then finalize():
 *   replaces: the loop  s = verify_proof(&params, &vk, s, instances_i, &mut Blake2bRead::init(proof_i))?
 *             followed by s.finalize()   (lib.rs:33-425, poly/kzg/strategy.rs:125-140).
 * proofs[i] / proof_lens[i]: proof byte strings.  Instances: instances32[i] is the concatenation of proof i's instance
 * columns — of all its circuit instances, instance by instance, when the context was created with circuit_instances > 1 —
 * col_lens[c] the number of values in column c (same for every proof of the batch; h2v_verify_batch_shapes lifts that),
 * n_instance_columns must equal circuit_instances x the VK's instance columns (else H2V_ERR_INVALID_INSTANCES, lib.rs:51-55).
 * rand32: the n scalars that AccumulatorStrategy::process draws with Fr::random
 * (kzg/strategy.rs:129), in call order; NULL = draw from the OS RNG.
 * per_proof_status[i]: 0 or the plonk::Error the reference's verify_proof returns for proof i;
 * a failing proof contributes nothing to the accumulator.  (An instance value that is not a canonical
 * field element cannot be expressed in the reference, whose instances are typed Fr; here it is
 * reported as H2V_ERR_INVALID_INSTANCES for that proof.)
 * batch_ok: all statuses OK and the single pairing check passed.
 * out_left_xy / out_right_xy: the two evaluated channels of the final DualMSM (may be NULL). */
int h2v_verify_batch(h2v_ctx* ctx, size_t n,
                     const uint8_t* const* proofs, const size_t* proof_lens,
                     const uint8_t* const* instances32, size_t n_instance_columns, const size_t* col_lens,
                     const uint8_t* rand32,
                     int* per_proof_status, int* batch_ok,
                     uint8_t out_left_xy[64], uint8_t out_right_xy[64]);

/* As h2v_verify_batch, but every proof brings its own instance column lengths — what N independent calls of the reference's
 * verify_proof allow (`instances: &[&[&[Fr]]]` is an argument of each call, lib.rs:33-49).  col_lens_per_proof is
 * [n][n_instance_columns]; instances32[i] is the concatenation of proof i's columns.  Proofs are grouped by shape inside the
 * library (one compiled plan per shape); the multipliers follow call order over the whole batch and ONE pairing closes it, so
 * the result equals n calls of verify_proof on one AccumulatorStrategy followed by finalize().
 * Cost and limits: every distinct shape compiles a plan (host work quadratic in the per-proof program's length, ~10 device uploads)
 * and resizes the workspace, and the shapes are chosen by whoever supplies the proofs — so one call takes at most 64 distinct
 * shapes (H2V_ERR_UNSUPPORTED beyond) and a context keeps at most 32 compiled plans (least recently used out; plans held by a
 * batch object stay). */
int h2v_verify_batch_shapes(h2v_ctx* ctx, size_t n,
                            const uint8_t* const* proofs, const size_t* proof_lens,
                            const uint8_t* const* instances32, size_t n_instance_columns, const size_t* col_lens_per_proof,
                            const uint8_t* rand32,
                            int* per_proof_status, int* batch_ok,
                            uint8_t out_left_xy[64], uint8_t out_right_xy[64]);

/* As h2v_verify_batch, starting from an existing accumulator instead of an empty one:
 *   replaces: AccumulatorStrategy::with(msm_accumulator) (poly/kzg/strategy.rs:75-78) — the reference's only pause / resume hook —
 *             followed by the same loop of verify_proof calls and finalize().
 * The seed is a DualMSM as the reference holds it: two lists of (scalar, base) terms, left and right channel (either may be empty).
 * Every verify_proof of this call scales the whole accumulator by its fresh draw before its Guard joins (strategy.rs:129), so the
 * seed's terms end up multiplied by the product of all n draws — which is what makes
 *     verify_batch(first half) -> (L, R);  verify_batch_seeded(second half, seed = {(1, L)}, {(1, R)})
 * equal, bit for bit, to ONE verify_batch over both halves with the draws concatenated.
 * Seed scalars: 32-byte canonical; seed bases: 64-byte x | y (all-zero = identity), rejected with H2V_ERR_BAD_ARGUMENT when not
 * on the curve.  out_left_xy / out_right_xy: the evaluated channels of the final DualMSM, seed included. */
int h2v_verify_batch_seeded(h2v_ctx* ctx, size_t n,
                            const uint8_t* const* proofs, const size_t* proof_lens,
                            const uint8_t* const* instances32, size_t n_instance_columns, const size_t* col_lens,
                            const uint8_t* rand32,
                            const uint8_t* seed_left_scalars32, const uint8_t* seed_left_bases64, size_t n_seed_left,
                            const uint8_t* seed_right_scalars32, const uint8_t* seed_right_bases64, size_t n_seed_right,
                            int* per_proof_status, int* batch_ok,
                            uint8_t out_left_xy[64], uint8_t out_right_xy[64]);

/* N x verify_proof under SingleStrategy (one pairing per proof; poly/kzg/strategy.rs:164-176).
 * per_proof_status[i] = 0, or H2V_ERR_CONSTRAINT_SYSTEM_FAILURE when that proof's pairing fails,
 * or the transcript/opening error. */
int h2v_verify_each(h2v_ctx* ctx, size_t n,
                    const uint8_t* const* proofs, const size_t* proof_lens,
                    const uint8_t* const* instances32, size_t n_instance_columns, const size_t* col_lens,
                    int* per_proof_status);

/* Debug / parity: the Guard of one proof term by term in the order the reference appends them (shplonk.rs:256-264;
 * gwc.rs:86-132: witness_with_aux, commitment_multi query by query — a commitment opened at several points occurs once per
 * query, each time with that query's own scalar — then (eval_multi, -g)), and the
 * Fiat-Shamir challenges [user challenges.., theta, beta, gamma, y, x, y', v, u] (GWC: [.., x, v, u]).
 * On entry *n_right / *n_left / *n_challenges hold the capacities (in elements). */
int h2v_guard_msm(h2v_ctx* ctx, const uint8_t* proof, size_t proof_len,
                  const uint8_t* instances32, size_t n_instance_columns, const size_t* col_lens,
                  uint8_t* right_scalars32, uint8_t* right_bases64, size_t* n_right,
                  uint8_t* left_scalars32, uint8_t* left_bases64, size_t* n_left,
                  uint8_t* challenges32, size_t* n_challenges);

/* n x Fr::random(OsRng) as AccumulatorStrategy::process draws them (kzg/strategy.rs:129): 64 bytes of OS randomness reduced mod r,
 * 32 canonical bytes each.  What rand32 = NULL makes the entry points above draw internally; a SHARDED batch needs the one stream
 * on every rank (draw on one rank, broadcast, pass each rank its tail: h2v_batch_upload). */
int h2v_random_scalars(uint8_t* out32, size_t n);

/* ---- staged interface: inputs resident in HBM, asynchronous execution on the batch's stream.
 * h2v_verify_batch == upload + launch + finish.  A sharded (multi-GPU) run uses
 * h2v_batch_launch(b, 0) on every rank, exchanges the accumulator records (h2v_batch_export_accumulators),
 * and closes with h2v_batch_fold_check_enqueue (or h2v_fold_check). */
int h2v_batch_create(h2v_ctx* ctx, size_t max_proofs, size_t max_instance_values_per_proof, h2v_batch** out);
void h2v_batch_destroy(h2v_batch* b);
/* Host -> device copy of one shard.  proofs_flat = n * proof_len bytes, instances_flat = n * (sum col_lens) * 32 bytes.
 * rand32_tail: the Fr::random draws of proofs [first_index, total) of the whole (possibly sharded)
 * batch — the multiplier of proof i is the product of the draws of all later proofs
 * (kzg/strategy.rs:129, msm.rs:173-176) — n_tail = total - first_index >= n. NULL = OS RNG (unsharded only). */
int h2v_batch_upload(h2v_batch* b, size_t n, const uint8_t* proofs_flat, size_t proof_len,
                     const uint8_t* instances_flat, size_t n_instance_columns, const size_t* col_lens,
                     const uint8_t* rand32_tail, size_t n_tail);
/* Enqueue decompress -> transcript -> Fr program -> fold -> MSM (-> pairing if with_pairing). */
int h2v_batch_launch(h2v_batch* b, int with_pairing);
/* h2v_batch_upload followed by h2v_batch_launch, with most of the host -> device copy HIDDEN behind the first stage: only the point
 * bytes of the proofs (a few runs at fixed offsets: 12 x 32 of 1024 bytes for the headline VK) are copied first, the point
 * decompression is enqueued, and the calling thread copies everything — whole proofs, instances, draws — while the GPU decompresses.
 * Same results as the two calls; returns when the host buffers are the caller's again, with the launch still running
 * (h2v_batch_finish waits for it).  For callers whose proofs arrive from the host for every batch (the C ABI hands over host
 * memory): a single launch has nothing else to hide the PCIe copy behind. */
int h2v_batch_upload_launch(h2v_batch* b, size_t n, const uint8_t* proofs_flat, size_t proof_len,
                            const uint8_t* instances_flat, size_t n_instance_columns, const size_t* col_lens,
                            const uint8_t* rand32_tail, size_t n_tail, int with_pairing);
/* Wait for the launch and fetch results (any pointer may be NULL).  A launch with its own pairing checks ends on two streams — the
 * batch's stream (last: the pairing kernel, which writes the verdicts into pinned host memory itself) and an internal auxiliary stream (the
 * whole accumulators, their bytes, the copy of the result block); this call waits for both.  The batch's stream alone going idle does
 * NOT mean the results are there; every other h2v_batch_* call orders its work behind that auxiliary stream by itself. */
int h2v_batch_finish(h2v_batch* b, int* per_proof_status, int* batch_ok, uint8_t out_left_xy[64], uint8_t out_right_xy[64]);
/* Grouped batches: one upload / launch carries `groups` INDEPENDENT AccumulatorStrategy batches (kzg/strategy.rs:99-141 each):
 * group g owns proofs [g*n/groups, (g+1)*n/groups) and the draws rand32_tail[g*n_tail/groups, (g+1)*n_tail/groups), has its own
 * pair of accumulators and its own pairing check — exactly what `groups` separate h2v_batch objects would compute, but with
 * every kernel launched once for all of them (the pairing and the tail of the MSM are latency-bound single-wave kernels, so G
 * of them side by side cost the time of one).  n and n_tail of later uploads must be multiples of `groups`.  Call before upload. */
int h2v_batch_set_groups(h2v_batch* b, size_t groups);
/* As h2v_batch_finish for a grouped batch: group_ok[n_groups], out_left_xy / out_right_xy = n_groups x 64 bytes. */
int h2v_batch_finish_groups(h2v_batch* b, int* per_proof_status, int* group_ok, uint8_t* out_left_xy, uint8_t* out_right_xy, size_t n_groups);
/* Device address of this batch's accumulator points after launch: per group [left, right], 2 x 108 bytes each, Jacobian
 * (X, Y, Z) in the library's Montgomery limb layout (debug / inspection; the record a sharded run exchanges is written by
 * h2v_batch_export_accumulators). */
int h2v_batch_accumulators(h2v_batch* b, void** device_ptr, size_t* nbytes);
/* What a shard contributes to a sharded batch, per group — opaque bytes to be moved by a collective (all-gather):
 *   [u32 failed][u32 parts][u32 shift][u32 0][left piece 0 .. 5][right piece 0 .. 5]      (a piece: 108 B, Jacobian X, Y, Z)
 * `failed` = number of this shard's proofs with a non-zero status.  It matters: a failed proof is zeroed out of its shard's
 * accumulators, so the folded pairing alone would accept a batch in which another shard rejected a proof;
 * h2v_batch_fold_check_enqueue / h2v_fold_check clear `ok` when any folded record reports failures, so every rank reaches the same
 * verdict without a second collective.
 * The accumulators travel the way a launch leaves them — in `parts` pieces, accumulator = sum_j 2^(shift j) piece_j — because the
 * folded pairing takes them in pieces too (the doublings move to precomputed multiples of the two G2 points; the whole point would
 * cost ~120 dependent doublings on every rank before AND the slower pairing after the exchange).  parts = 1, shift = 0 is a whole
 * point.  Records of ranks whose launches chose another (parts, shift) — shards of very different size — are folded correctly all
 * the same (their pieces are put together first). */
#define H2V_ACC_RECORD_PIECES 6
#define H2V_ACC_RECORD_BYTES 1312   /* 16 + 2 * H2V_ACC_RECORD_PIECES * 108 */
/* The HIP stream (hipStream_t) the batch runs on, for event timing and stream-ordered interop. */
void* h2v_batch_stream(h2v_batch* b);
/* Run the batch on a caller-owned stream (e.g. a torch.cuda.Stream's cuda_stream) instead of its own,
 * so that collectives issued by the caller on that stream are ordered with the batch's kernels without
 * host synchronisation.  The caller keeps the stream alive while the batch uses it. */
int h2v_batch_set_stream(h2v_batch* b, void* hip_stream);
/* Stream-ordered write of the batch's accumulator records (groups x H2V_ACC_RECORD_BYTES) into caller device memory. */
int h2v_batch_export_accumulators(h2v_batch* b, void* device_dst);
/* Stream-ordered version of h2v_fold_check on the batch's stream: fold n_parts gathered accumulator
 * sets (each laid out as h2v_batch_export_accumulators writes it: [group][left, right]) group by group and enqueue one
 * pairing per group; the result is fetched by h2v_batch_finish / h2v_batch_finish_groups (ok, left, right): a group is ok
 * when its pairing passes, its local proofs are all ok AND no folded record reports a failed proof. */
int h2v_batch_fold_check_enqueue(h2v_batch* b, const void* device_accumulators, size_t n_parts);
/* Fold n_parts accumulator records (as written by h2v_batch_export_accumulators for an ungrouped batch, contiguous in
 * device memory) with G1 additions and run the single pairing check; ok = pairing passed and no record reports a failed proof.
 *   replaces: DualMSM::add_msm + check across shards (poly/kzg/msm.rs:178-203). */
int h2v_fold_check(h2v_ctx* ctx, const void* device_accumulators, size_t n_parts, int* ok,
                   uint8_t out_left_xy[64], uint8_t out_right_xy[64]);
/* Per-stage device time of the last finished launch, milliseconds, measured with HIP events on
 * the batch's stream: [decompress, transcript, fr_program, fold, msm, pairing, msm_accumulate (the dominant kernel
 * inside the msm stage)]; returns the number of entries written (<= cap).  Entries the profiling level does not record are 0. */
int h2v_batch_timings(h2v_batch* b, float* ms, int cap);
/* level 0: off; 1 (or any other non-zero value): an event between the stages and around the dominant kernel — every event is a
 * barrier packet on the stream, ~6 us of idle time each, ~0.06 ms per launch; H2V_PROFILE_KERNEL: only the dominant kernel's own
 * start / stop timestamps (attached to its dispatch, no extra packet). */
#define H2V_PROFILE_KERNEL 3
int h2v_batch_set_profiling(h2v_batch* b, int level);

#ifdef __cplusplus
}
#endif
#endif /* H2V_H */
