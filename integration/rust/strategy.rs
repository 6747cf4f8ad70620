//! GPU-backed drop-ins at the reference's two seams (UNCOMPILED here — no rustc in the build image).
//!
//! * `GpuBatchVerifier`        — whole-batch seam: N x `verify_proof` + `AccumulatorStrategy::finalize`
//!                               (halo2_verifier/src/lib.rs:33-49, poly/kzg/strategy.rs:125-140) in one call.
//! * `GpuAccumulatorStrategy`  — trait seam: `impl VerificationStrategy` (poly/strategy.rs:12-31) whose `finalize`
//!                               evaluates the two `MSMKZG` channels and the pairing on the GPU
//!                               (poly/kzg/msm.rs:81-86, 185-203) while `verify_proof` itself stays on the CPU.
use super::ffi::*;
use crate::{
    helpers::SerdeFormat,
    plonk::Error,
    poly::{
        commitment::MSM,
        kzg::{commitment::{KZGCommitmentScheme, ParamsKZG}, msm::DualMSM, multiopen::VerifierSHPLONK, strategy::GuardKZG},
        strategy::VerificationStrategy,
    },
    VerifyingKey,
};
use ff::{Field, PrimeField};
use group::Curve;
use halo2curves::bn256::{Bn256, Fr, G1Affine};
use halo2curves::CurveAffine;

fn map_err(rc: i32) -> Error {
    match rc { -1 => Error::InvalidInstances, -2 => Error::ConstraintSystemFailure, -3 => Error::BoundsFailure,
               -5 => Error::Transcript("h2v"), -6 => Error::InstanceTooLarge, _ => Error::Opening }
}

pub struct GpuBatchVerifier<'p> {
    ctx: *mut h2v_ctx,
    proofs: Vec<&'p [u8]>,
    instances: Vec<Vec<u8>>,
    col_lens: Vec<usize>,
}

impl<'p> GpuBatchVerifier<'p> {
    pub fn new(params: &ParamsKZG<Bn256>, vk: &VerifyingKey<G1Affine>, device: i32) -> Result<Self, Error> {
        let (mut pb, mut vb) = (Vec::new(), Vec::new());
        params.write_custom(&mut pb, SerdeFormat::RawBytes).map_err(|_| Error::Opening)?;
        vk.write(&mut vb, SerdeFormat::RawBytes).map_err(|_| Error::Opening)?;
        let mut ctx = core::ptr::null_mut();
        let rc = unsafe { h2v_ctx_create(pb.as_ptr(), pb.len(), H2V_SERDE_RAW_BYTES, vb.as_ptr(), vb.len(), H2V_SERDE_RAW_BYTES, device, &mut ctx) };
        if rc != 0 { return Err(map_err(rc)); }
        Ok(Self { ctx, proofs: Vec::new(), instances: Vec::new(), col_lens: Vec::new() })
    }

    /// One `verify_proof(&params, &vk, strategy, &[instances], &mut Blake2bRead::init(proof))` call.
    pub fn push(&mut self, proof: &'p [u8], instances: &[&[Fr]]) {
        if self.col_lens.is_empty() { self.col_lens = instances.iter().map(|c| c.len()).collect(); }
        let mut flat = Vec::with_capacity(32 * instances.iter().map(|c| c.len()).sum::<usize>());
        for col in instances { for v in col.iter() { flat.extend_from_slice(v.to_repr().as_ref()); } }
        self.proofs.push(proof);
        self.instances.push(flat);
    }

    /// `strategy.finalize()`: true iff every proof is well formed and the single pairing check passes.
    pub fn finalize(self) -> Result<bool, Error> {
        let n = self.proofs.len();
        let ptrs: Vec<*const u8> = self.proofs.iter().map(|p| p.as_ptr()).collect();
        let lens: Vec<usize> = self.proofs.iter().map(|p| p.len()).collect();
        let iptrs: Vec<*const u8> = self.instances.iter().map(|i| i.as_ptr()).collect();
        let (mut status, mut ok) = (vec![0i32; n], 0i32);
        let rc = unsafe { h2v_verify_batch(self.ctx, n, ptrs.as_ptr(), lens.as_ptr(), iptrs.as_ptr(), self.col_lens.len(), self.col_lens.as_ptr(),
                                           core::ptr::null(), status.as_mut_ptr(), &mut ok, core::ptr::null_mut(), core::ptr::null_mut()) };
        if rc != 0 { return Err(map_err(rc)); }
        Ok(ok == 1)
    }
}
impl<'p> Drop for GpuBatchVerifier<'p> { fn drop(&mut self) { unsafe { h2v_ctx_destroy(self.ctx) } } }

/// Trait seam: same `process` as AccumulatorStrategy (kzg/strategy.rs:125-136); `finalize` on the GPU.
pub struct GpuAccumulatorStrategy<'params> { acc: DualMSM<'params, Bn256>, ctx: *mut h2v_ctx }

impl<'params> VerificationStrategy<'params, KZGCommitmentScheme<Bn256>, VerifierSHPLONK<'params, Bn256>> for GpuAccumulatorStrategy<'params> {
    type Output = Self;
    fn new(params: &'params ParamsKZG<Bn256>) -> Self {
        let mut pb = Vec::new();
        params.write_custom(&mut pb, SerdeFormat::RawBytes).expect("vec write");
        let mut ctx = core::ptr::null_mut();
        let rc = unsafe { h2v_ctx_create(pb.as_ptr(), pb.len(), H2V_SERDE_RAW_BYTES, core::ptr::null(), 0, 0, 0, &mut ctx) };
        assert_eq!(rc, 0, "h2v_ctx_create");
        Self { acc: DualMSM::new(params), ctx }
    }
    fn process(mut self, f: impl FnOnce(DualMSM<'params, Bn256>) -> Result<GuardKZG<'params, Bn256>, Error>) -> Result<Self, Error> {
        self.acc.scale(Fr::random(getrandom_or_panic::getrandom_or_panic()));
        let guard = f(self.acc)?;
        Ok(Self { acc: guard.msm_accumulator, ctx: self.ctx })
    }
    fn finalize(self) -> bool {
        let eval = |m: &dyn Fn() -> (Vec<Fr>, Vec<<G1Affine as CurveAffine>::CurveExt>)| -> [u8; 64] {
            let (scalars, bases) = m();
            let sb: Vec<u8> = scalars.iter().flat_map(|s| s.to_repr().as_ref().to_vec()).collect();
            let bb: Vec<u8> = bases.iter().flat_map(|b| { let a = b.to_affine(); let c = a.coordinates();
                if bool::from(c.is_some()) { let c = c.unwrap(); [c.x().to_repr().as_ref(), c.y().to_repr().as_ref()].concat() } else { vec![0u8; 64] } }).collect();
            let (mut out, mut ident) = ([0u8; 64], 0i32);
            let rc = unsafe { h2v_msm_g1(self.ctx, sb.as_ptr(), bb.as_ptr(), scalars.len(), out.as_mut_ptr(), &mut ident) };
            assert_eq!(rc, 0, "h2v_msm_g1");
            out
        };
        let left = eval(&|| (self.acc.left.scalars(), self.acc.left.bases()));
        let right = eval(&|| (self.acc.right.scalars(), self.acc.right.bases()));
        let mut ok = 0i32;
        let rc = unsafe { h2v_pairing_check(self.ctx, left.as_ptr(), right.as_ptr(), &mut ok) };
        rc == 0 && ok == 1
    }
}
