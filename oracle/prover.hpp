// ORACLE — TEST INFRASTRUCTURE ONLY (see bn254_fp.hpp header).
//
// Test-only key generator and honest PLONKish/KZG/SHPLONK prover.  The reference ships no
// prover and no proof bytes: its tests borrow halo2_proofs::plonk::{keygen_vk, create_proof}
// (halo2_verifier/tests/helpers.rs:31-61), which is not vendored.  This file produces the
// inputs (VK bytes, params bytes, proof bytes) that the verifier under test consumes, in the
// transcript order that lib.rs:33-425 and shplonk.rs:175-267 read them.
// It is deliberately written independently of verifier.cpp (own constraint evaluation over
// the extended coset, own rotation-set construction) so that prover and verifier do not share
// a bug by construction.
#pragma once
#include "vk.hpp"
#include <functional>

namespace h2o {

struct Rng {  // splitmix64 -> uniform Fr (deterministic test randomness)
    u64 s;
    explicit Rng(u64 seed) : s(seed) {}
    u64 next() { u64 z = (s += 0x9e3779b97f4a7c15ULL); z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL; z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL; return z ^ (z >> 31); }
    Fr fr() { uint8_t b[64]; for (int i = 0; i < 8; ++i) { u64 w = next(); for (int j = 0; j < 8; ++j) b[8 * i + j] = (uint8_t)(w >> (8 * j)); } return Fr::from_uniform_bytes(b); }
};

// KZG commitment key.  Two modes:
//  * known-s test SRS: commitments are single fixed-base multiplications [p(s)]G;
//  * the reference's own SRS file (halo2_proofs RawBytes layout: k | g[n] | g_lagrange[n] | g2 | s_g2,
//    SURVEY.md §2 row 21): commitments are real MSMs over g / g_lagrange.
struct CommitKey {
    uint32_t k = 0; uint64_t n = 0;
    bool known_s = false;
    Fr s;
    std::vector<Fr> lagrange_at_s;           // L_i(s), known-s mode
    std::vector<G1Affine> g, g_lagrange;     // file mode
    ParamsKZG params;                        // verifier-side params (g, g2, s_g2)
    static CommitKey from_secret(uint32_t k, const Fr& s);
    static CommitKey from_srs_file(const uint8_t* data, size_t len);
    G1Affine commit_lagrange(const std::vector<Fr>& values) const;
    G1Affine commit_coeff(const std::vector<Fr>& coeffs) const;
};

struct CopyConstraint { uint32_t col_a, row_a, col_b, row_b; };  // indices into cs.permutation_columns

struct Circuit {
    uint32_t k = 0;
    uint32_t cs_degree = 3;
    ConstraintSystem cs;
    std::vector<std::vector<Fr>> fixed;   // [num_fixed_columns][n]
    std::vector<CopyConstraint> copies;
    uint64_t n() const { return 1ULL << k; }
    size_t usable_rows() const { return n() - (cs.blinding_factors() + 1); }
};

// advice[col][row] for the usable rows; called once per phase with the challenges squeezed so far.
typedef std::function<void(unsigned phase, const std::vector<Fr>& challenges, std::vector<std::vector<Fr>>& advice)> WitnessFn;

struct ProvingKey {
    Circuit circuit;
    VerifyingKey vk;
    std::vector<std::vector<Fr>> sigma;          // permutation polynomials, Lagrange values
    std::vector<std::vector<Fr>> fixed_coeff, sigma_coeff;  // coefficient form
    std::vector<std::vector<Fr>> fixed_ext, sigma_ext;      // extended-coset evaluations
    std::vector<Fr> l0_ext, llast_ext, lactive_ext;         // l_0, l_last, 1-(l_last+l_blind) on the coset
    uint32_t ext_k = 0;
};

ProvingKey keygen(const Circuit& c, const CommitKey& ck);

// Produces proof bytes for one circuit instance (instances[col][row]).
// multiopen: 0 SHPLONK, 1 GWC; transcript: 0 Blake2b, 1 Keccak256 (the generic parameters of the reference's prover/verifier pair)
std::vector<uint8_t> create_proof(const ProvingKey& pk, const CommitKey& ck, const std::vector<std::vector<Fr>>& instances,
                                  const WitnessFn& witness, Rng& rng, int multiopen = 0, int transcript = 0);
// several circuit instances in one transcript (the reference's `instances: &[&[&[Fr]]]`, lib.rs:33-49): instances[m], witnesses[m]
std::vector<uint8_t> create_proof_multi(const ProvingKey& pk, const CommitKey& ck, const std::vector<std::vector<std::vector<Fr>>>& instances,
                                        const std::vector<WitnessFn>& witnesses, Rng& rng, int multiopen = 0, int transcript = 0);

// ---- synthetic circuits used by the tests and the bench
// config 1/2/3: the tests/vector_mul.rs shape — 3 advice, 1 instance, 1 fixed (selector), gate s*(a*b-c),
// equality over [instance, a0, a1, a2]; `n_mul` multiplications, products exposed as public inputs.
Circuit circuit_vector_mul(uint32_t k, size_t n_mul);
WitnessFn witness_vector_mul(const std::vector<Fr>& a, const std::vector<Fr>& b);
// tests/shuffle.rs shape: W original + W shuffled first-phase advice, z in the second phase, two user challenges,
// three selectors, no permutation argument.  `shuffled` may be an invalid shuffle (reject case).
Circuit circuit_two_phase_shuffle(uint32_t k, size_t W, size_t H);
WitnessFn witness_two_phase_shuffle(const std::vector<std::vector<Fr>>& original, const std::vector<std::vector<Fr>>& shuffled);
// config 4 style: A advice columns (every 4th also queried at -1 and +1), F fixed, L lookups with 2-column
// input/table expressions, Sh shuffle arguments, a degree-`deg` gate, permutation over all advice columns.
Circuit circuit_wide(uint32_t k, size_t A, size_t F, size_t L, size_t Sh, uint32_t gate_degree, u64 seed);
WitnessFn witness_wide(const Circuit& c, u64 seed);

}  // namespace h2o
