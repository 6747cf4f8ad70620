// ORACLE — TEST INFRASTRUCTURE ONLY (see bn254_fp.hpp header).
//
// CPU restatement of the reference verifier, following it step for step so that it can serve
// both as the parity checker and as the timed single-thread CPU baseline:
//   best_multiexp / multiexp_serial   arithmetic.rs:7-108   (serial Pippenger, c in {1,3,4})
//   lagrange_interpolate, eval_polynomial, evaluate_vanishing_polynomial  arithmetic.rs:137-206
//   MSMKZG / DualMSM                  poly/kzg/msm.rs:17-204
//   EvaluationDomain (used fields)    poly/domain.rs:34-140,172-212
//   IndexedExpressionPoly::evaluate   plonk/vk.rs:478-512,579-586
//   permutation / lookup / shuffle / vanishing   plonk/{permutation,lookup,shuffle,vanishing}.rs
//   verify_proof                      lib.rs:33-425
//   VerifierSHPLONK::verify_proof     poly/kzg/multiopen/shplonk.rs:58-267
//   VerifierGWC::verify_proof         poly/kzg/multiopen/gwc.rs:54-163
//   SingleStrategy / AccumulatorStrategy   poly/kzg/strategy.rs:55-181
// PARITY PINNING: the reference holds no proof/VK/MSM golden values; what it
// does hold — params/kzg_bn254_8.srs — pins field/curve encodings, omega, 256 MSM answers and
// 255 pairing relations (tests/test_oracle_srs_kat.py).  Everything above that level (transcript
// order, expression evaluation, SHPLONK) is "parity unpinned" by reference data and rests on
// two independent restatements agreeing (this file and oracle/pyref) plus the accept/reject
// semantics of the reference's own tests.
#pragma once
#include "vk.hpp"

namespace h2o {

// plonk/mod.rs:19-32
enum Error {
    OK = 0,
    InvalidInstances = -1,
    ConstraintSystemFailure = -2,
    BoundsFailure = -3,
    Opening = -4,
    Transcript = -5,
    InstanceTooLarge = -6,
    // Conditions under which the reference panics (unwrap on a zero inverse, vanishing.rs:100,
    // shplonk.rs:215; empty gate polynomial, multilinear.rs:65).
    ReferencePanic = -7,
};

G1 best_multiexp(const Fr* coeffs, const G1Affine* bases, size_t n);

struct MSMKZG {
    std::vector<Fr> scalars;
    std::vector<G1> bases;
    void append_term(const Fr& s, const G1& p) { scalars.push_back(s); bases.push_back(p); }
    void add_msm(const MSMKZG& o) {
        scalars.insert(scalars.end(), o.scalars.begin(), o.scalars.end());
        bases.insert(bases.end(), o.bases.begin(), o.bases.end());
    }
    void scale(const Fr& f) { for (auto& s : scalars) s = s * f; }
    G1 eval() const;
    bool check() const { return eval().is_identity(); }
};

struct DualMSM {
    MSMKZG left, right;
    void scale(const Fr& e) { left.scale(e); right.scale(e); }
    void add_msm(const DualMSM& o) { left.add_msm(o.left); right.add_msm(o.right); }
    bool check(const ParamsKZG& params) const;
};

struct Domain {
    uint32_t k; uint64_t n;
    Fr omega, omega_inv, barycentric_weight;
    uint64_t quotient_poly_degree;
    Domain(uint32_t j, uint32_t k);
    Fr rotate_omega(const Fr& value, int32_t rotation) const;
    std::vector<Fr> l_i_range(const Fr& x, const Fr& xn, int32_t from, int32_t to_exclusive) const;
};

Fr eval_expr(const ExprPoly& poly, const std::vector<Fr>& coeffs, const std::vector<Fr>& advice, const std::vector<Fr>& fixed,
             const std::vector<Fr>& instance, const std::vector<Fr>& challenges);

// Intermediate values exposed for parity tests against the GPU path and the golden fixtures.
struct VerifyTrace {
    std::vector<Fr> challenges;   // user challenges
    Fr theta, beta, gamma, y, x;  // lib.rs:115,129-132,166,172
    Fr sh_y, sh_v, sh_u;          // shplonk.rs:195-199
    Fr expected_h_eval;
    std::vector<Fr> expressions;
};

// lib.rs:33-425, one circuit instance per transcript (instances.len() == 1, as in every reference caller; verify_proof_multi
// below takes the general `instances: &[&[&[Fr]]]`).  Appends this proof's terms
// to `acc` exactly as the closure passed to strategy.process does (shplonk.rs:256-264).
// The generic parameters of verify_proof the reference instantiates (lib.rs:33-40): V in {VerifierSHPLONK, VerifierGWC}
// (poly/kzg/multiopen/{shplonk,gwc}.rs), T in {Blake2bRead, Keccak256Read} (transcript/mod.rs:104-116).
enum MultiOpen { MO_SHPLONK = 0, MO_GWC = 1 };
struct VerifyOptions { int multiopen = MO_SHPLONK; int transcript = TR_BLAKE2B; };

Error verify_proof(const ParamsKZG& params, const VerifyingKey& vk, const std::vector<std::vector<Fr>>& instances,
                   const uint8_t* proof, size_t proof_len, DualMSM& acc, VerifyTrace* trace = nullptr,
                   const char** err_msg = nullptr, VerifyOptions opts = VerifyOptions());

// The general form: `insts` is the reference's `instances: &[&[&[Fr]]]`, one entry per circuit instance of the transcript.
Error verify_proof_multi(const ParamsKZG& params, const VerifyingKey& vk, const std::vector<std::vector<std::vector<Fr>>>& insts,
                         const uint8_t* proof, size_t proof_len, DualMSM& acc, VerifyTrace* trace = nullptr,
                         const char** err_msg = nullptr, VerifyOptions opts = VerifyOptions());

// poly/kzg/strategy.rs:164-176
Error verify_single(const ParamsKZG& params, const VerifyingKey& vk, const std::vector<std::vector<Fr>>& instances,
                    const uint8_t* proof, size_t proof_len, VerifyOptions opts = VerifyOptions());

// poly/kzg/strategy.rs:125-140.  `rand` holds the Fr::random draw of each process() call
// (strategy.rs:129), injectable so results are reproducible.
struct AccumulatorStrategy {
    DualMSM acc;
    Error process(const ParamsKZG& params, const VerifyingKey& vk, const std::vector<std::vector<Fr>>& instances,
                  const uint8_t* proof, size_t proof_len, const Fr& rand, VerifyOptions opts = VerifyOptions());
    bool finalize(const ParamsKZG& params) const { return acc.check(params); }
};

}  // namespace h2o
