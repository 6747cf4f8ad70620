//! `extern "C"` binding of include/h2v.h for the halo2_verifier crate (feature "mi355x", needs std + the HIP runtime).
//! UNCOMPILED in this repository: the build image has no Rust toolchain.  The tested callers of the same ABI are
//! halo2_verifier_amd/verifier.py and bench.py (ctypes) and tests/cpp/harness.cpp (C++).  tests/test_cabi_symbols.py checks that
//! every function include/h2v.h declares is bound here.  See INTEGRATION.md.
#![allow(non_camel_case_types)]
use core::ffi::{c_char, c_float, c_int, c_void};

#[repr(C)] pub struct h2v_ctx { _p: [u8; 0] }
#[repr(C)] pub struct h2v_batch { _p: [u8; 0] }
/// struct_size: `core::mem::size_of::<h2v_options>()` (the library rejects a layout it does not know);
/// multiopen: VerifierSHPLONK / VerifierGWC; transcript: Blake2bRead / Keccak256Read; circuit_instances: `instances.len()` of
/// verify_proof (0 or 1 = one circuit instance per transcript); instance_kernel_threshold: debug, 0 = default
#[repr(C)] pub struct h2v_options { pub struct_size: usize, pub multiopen: c_int, pub transcript: c_int, pub circuit_instances: c_int, pub instance_kernel_threshold: c_int }
/// debug / test: forced kernel variants, every field 0 = automatic (h2v_ctx_set_tuning)
#[repr(C)] pub struct h2v_tuning { pub struct_size: usize, pub frvm_streams: c_int, pub frvm_lds_kb: c_int, pub msm_parts: c_int, pub msm_global_sort: c_int,
                                   pub msm_no_term_split: c_int, pub msm_window_threads: c_int, pub msm_window_wpw: c_int, pub msm_window_slots: c_int,
                                   pub msm_acc_waves: c_int, pub pairing_one_stream: c_int, pub upload_mode: c_int }
pub const H2V_ABI_VERSION: c_int = 3;
pub const H2V_VK_LAYOUT_WRITER: c_int = 0;
pub const H2V_VK_LAYOUT_READER: c_int = 1;

pub const H2V_OK: c_int = 0;
pub const H2V_ERR_INVALID_INSTANCES: c_int = -1;
pub const H2V_ERR_CONSTRAINT_SYSTEM_FAILURE: c_int = -2;
pub const H2V_ERR_BOUNDS_FAILURE: c_int = -3;
pub const H2V_ERR_OPENING: c_int = -4;
pub const H2V_ERR_TRANSCRIPT: c_int = -5;
pub const H2V_ERR_INSTANCE_TOO_LARGE: c_int = -6;
pub const H2V_ERR_REFERENCE_PANIC: c_int = -7;
pub const H2V_ERR_BAD_ARGUMENT: c_int = -16;
pub const H2V_ERR_FORMAT: c_int = -17;
pub const H2V_ERR_DEVICE: c_int = -18;
pub const H2V_ERR_UNSUPPORTED: c_int = -19;
pub const H2V_SERDE_PROCESSED: c_int = 0;
pub const H2V_SERDE_RAW_BYTES: c_int = 1;
pub const H2V_SERDE_RAW_BYTES_UNCHECKED: c_int = 2;
pub const H2V_MULTIOPEN_SHPLONK: c_int = 0;
pub const H2V_MULTIOPEN_GWC: c_int = 1;
pub const H2V_TRANSCRIPT_BLAKE2B: c_int = 0;
pub const H2V_TRANSCRIPT_KECCAK256: c_int = 1;
/// accumulator pieces per side in a record
pub const H2V_ACC_RECORD_PIECES: usize = 6;
/// bytes one shard contributes per group to a sharded batch ([failed, parts, shift, 0] + 2 x 6 Jacobian pieces)
pub const H2V_ACC_RECORD_BYTES: usize = 1312;
/// h2v_batch_set_profiling: only the dominant kernel's own timestamps
pub const H2V_PROFILE_KERNEL: c_int = 3;

#[link(name = "h2v_amd")]
extern "C" {
    pub fn h2v_device_count() -> c_int;
    pub fn h2v_last_error() -> *const c_char;
    pub fn h2v_ctx_create(params: *const u8, params_len: usize, params_format: c_int, vk: *const u8, vk_len: usize, vk_format: c_int,
                          device: c_int, out: *mut *mut h2v_ctx) -> c_int;
    pub fn h2v_ctx_create_ex(params: *const u8, params_len: usize, params_format: c_int, vk: *const u8, vk_len: usize, vk_format: c_int,
                             device: c_int, options: *const h2v_options, out: *mut *mut h2v_ctx) -> c_int;
    pub fn h2v_ctx_destroy(ctx: *mut h2v_ctx);
    pub fn h2v_abi_version() -> c_int;
    pub fn h2v_ctx_set_tuning(ctx: *mut h2v_ctx, tuning: *const h2v_tuning) -> c_int;
    pub fn h2v_vk_convert(vk: *const u8, vk_len: usize, from_format: c_int, to_format: c_int, layout: c_int, out: *mut u8, out_len: *mut usize) -> c_int;
    pub fn h2v_params_convert(params: *const u8, params_len: usize, from_format: c_int, to_format: c_int, out: *mut u8, out_len: *mut usize) -> c_int;
    pub fn h2v_ctx_proof_shape(ctx: *const h2v_ctx, proof_len: *mut usize, n_points: *mut usize, n_scalars: *mut usize,
                               n_right_terms: *mut usize, n_instance_columns: *mut usize) -> c_int;
    pub fn h2v_msm_g1(ctx: *mut h2v_ctx, scalars32: *const u8, bases64: *const u8, n: usize, out_xy: *mut u8, out_is_identity: *mut c_int) -> c_int;
    pub fn h2v_pairing_check(ctx: *mut h2v_ctx, left_xy: *const u8, right_xy: *const u8, ok: *mut c_int) -> c_int;
    pub fn h2v_verify_batch(ctx: *mut h2v_ctx, n: usize, proofs: *const *const u8, proof_lens: *const usize,
                            instances32: *const *const u8, n_instance_columns: usize, col_lens: *const usize, rand32: *const u8,
                            per_proof_status: *mut c_int, batch_ok: *mut c_int, out_left_xy: *mut u8, out_right_xy: *mut u8) -> c_int;
    pub fn h2v_verify_batch_shapes(ctx: *mut h2v_ctx, n: usize, proofs: *const *const u8, proof_lens: *const usize,
                                   instances32: *const *const u8, n_instance_columns: usize, col_lens_per_proof: *const usize, rand32: *const u8,
                                   per_proof_status: *mut c_int, batch_ok: *mut c_int, out_left_xy: *mut u8, out_right_xy: *mut u8) -> c_int;
    pub fn h2v_verify_batch_seeded(ctx: *mut h2v_ctx, n: usize, proofs: *const *const u8, proof_lens: *const usize,
                                   instances32: *const *const u8, n_instance_columns: usize, col_lens: *const usize, rand32: *const u8,
                                   seed_left_scalars32: *const u8, seed_left_bases64: *const u8, n_seed_left: usize,
                                   seed_right_scalars32: *const u8, seed_right_bases64: *const u8, n_seed_right: usize,
                                   per_proof_status: *mut c_int, batch_ok: *mut c_int, out_left_xy: *mut u8, out_right_xy: *mut u8) -> c_int;
    pub fn h2v_verify_each(ctx: *mut h2v_ctx, n: usize, proofs: *const *const u8, proof_lens: *const usize,
                           instances32: *const *const u8, n_instance_columns: usize, col_lens: *const usize, per_proof_status: *mut c_int) -> c_int;
    pub fn h2v_guard_msm(ctx: *mut h2v_ctx, proof: *const u8, proof_len: usize, instances32: *const u8, n_instance_columns: usize, col_lens: *const usize,
                         right_scalars32: *mut u8, right_bases64: *mut u8, n_right: *mut usize,
                         left_scalars32: *mut u8, left_bases64: *mut u8, n_left: *mut usize,
                         challenges32: *mut u8, n_challenges: *mut usize) -> c_int;
    pub fn h2v_random_scalars(out32: *mut u8, n: usize) -> c_int;
    pub fn h2v_batch_create(ctx: *mut h2v_ctx, max_proofs: usize, max_instance_values_per_proof: usize, out: *mut *mut h2v_batch) -> c_int;
    pub fn h2v_batch_destroy(b: *mut h2v_batch);
    pub fn h2v_batch_upload(b: *mut h2v_batch, n: usize, proofs_flat: *const u8, proof_len: usize, instances_flat: *const u8,
                            n_instance_columns: usize, col_lens: *const usize, rand32_tail: *const u8, n_tail: usize) -> c_int;
    pub fn h2v_batch_launch(b: *mut h2v_batch, with_pairing: c_int) -> c_int;
    pub fn h2v_batch_upload_launch(b: *mut h2v_batch, n: usize, proofs_flat: *const u8, proof_len: usize, instances_flat: *const u8,
                                   n_instance_columns: usize, col_lens: *const usize, rand32_tail: *const u8, n_tail: usize, with_pairing: c_int) -> c_int;
    pub fn h2v_batch_finish(b: *mut h2v_batch, per_proof_status: *mut c_int, batch_ok: *mut c_int, out_left_xy: *mut u8, out_right_xy: *mut u8) -> c_int;
    pub fn h2v_batch_set_groups(b: *mut h2v_batch, groups: usize) -> c_int;
    pub fn h2v_batch_finish_groups(b: *mut h2v_batch, per_proof_status: *mut c_int, group_ok: *mut c_int, out_left_xy: *mut u8, out_right_xy: *mut u8,
                                   n_groups: usize) -> c_int;
    pub fn h2v_batch_accumulators(b: *mut h2v_batch, device_ptr: *mut *mut c_void, nbytes: *mut usize) -> c_int;
    pub fn h2v_batch_stream(b: *mut h2v_batch) -> *mut c_void;
    pub fn h2v_batch_set_stream(b: *mut h2v_batch, hip_stream: *mut c_void) -> c_int;
    pub fn h2v_batch_export_accumulators(b: *mut h2v_batch, device_dst: *mut c_void) -> c_int;
    pub fn h2v_batch_fold_check_enqueue(b: *mut h2v_batch, device_accumulators: *const c_void, n_parts: usize) -> c_int;
    pub fn h2v_fold_check(ctx: *mut h2v_ctx, device_accumulators: *const c_void, n_parts: usize, ok: *mut c_int,
                          out_left_xy: *mut u8, out_right_xy: *mut u8) -> c_int;
    pub fn h2v_batch_timings(b: *mut h2v_batch, ms: *mut c_float, cap: c_int) -> c_int;
    pub fn h2v_batch_set_profiling(b: *mut h2v_batch, level: c_int) -> c_int;
}
