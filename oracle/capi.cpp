// ORACLE — TEST INFRASTRUCTURE ONLY (see bn254_fp.hpp header).
//
// C entry points for ctypes (tests/, bench.py cpu_baseline, __graft_entry__.smoke()).
// Field elements cross as 32-byte little-endian canonical values, G1 points as x|y canonical
// 64 bytes (all-zero = identity), proofs / VK / params in the reference byte formats.
#include "prover.hpp"
#include "verifier.hpp"
#include <atomic>
#include <thread>

using namespace h2o;

namespace {
G1Affine g1_from_xy(const uint8_t b[64]) {
    G1Affine p; bool z = true;
    for (int i = 0; i < 64; ++i) if (b[i]) { z = false; break; }
    if (z) return G1Affine::identity();
    Fq::from_bytes(b, p.x); Fq::from_bytes(b + 32, p.y); p.inf = false;
    return p;
}
void g1_to_xy(const G1Affine& p, uint8_t b[64]) {
    if (p.inf) { memset(b, 0, 64); return; }
    p.x.to_bytes(b); p.y.to_bytes(b + 32);
}
std::vector<std::vector<Fr>> parse_instances(const uint8_t* inst32, const size_t* col_lens, size_t ncols) {
    std::vector<std::vector<Fr>> out(ncols);
    size_t off = 0;
    for (size_t c = 0; c < ncols; ++c)
        for (size_t i = 0; i < col_lens[c]; ++i, ++off) { Fr v; Fr::from_bytes(inst32 + 32 * off, v); out[c].push_back(v); }
    return out;
}
struct Setup {
    int multiopen = 0, transcript = 0;   // options of the prover side (h2o_setup_set_options)
    CommitKey ck;
    ProvingKey pk;
    int kind;  // 0 vector_mul, 1 two-phase shuffle, 2 wide
    size_t n_mul, W, H;
    u64 wide_seed;
};
size_t copy_out(const std::vector<uint8_t>& v, uint8_t* buf, size_t cap) {
    if (buf && cap >= v.size()) memcpy(buf, v.data(), v.size());
    return v.size();
}
}  // namespace

static thread_local VerifyOptions g_opts;
static thread_local size_t g_circuit_instances = 1;   // h2o_set_circuit_instances: `instances.len()` of the verifier-side calls
namespace {
// ncols = M x (columns per circuit instance), instance-major: the reference's `instances: &[&[&[Fr]]]` flattened
std::vector<std::vector<std::vector<Fr>>> parse_multi(const uint8_t* inst32, const size_t* col_lens, size_t ncols) {
    const size_t M = g_circuit_instances ? g_circuit_instances : 1, per = ncols / M;
    std::vector<std::vector<std::vector<Fr>>> out(M, std::vector<std::vector<Fr>>(per));
    size_t off = 0;
    for (size_t c = 0; c < M * per; ++c)
        for (size_t i = 0; i < col_lens[c]; ++i, ++off) { Fr v; Fr::from_bytes(inst32 + 32 * off, v); out[c / per][c % per].push_back(v); }
    return out;
}
}  // namespace

extern "C" {

// ---------------------------------------------------------------- primitives
int h2o_keccak256(const uint8_t* data, size_t len, uint8_t out[32]) { Keccak256 k; k.update(data, len); k.finalize(out); return 0; }
int h2o_fr_from_uniform(const uint8_t in[64], uint8_t out[32]) { Fr::from_uniform_bytes(in).to_bytes(out); return 0; }
int h2o_blake2b_personal(const uint8_t personal[16], const uint8_t* data, size_t len, uint8_t out[64]) {
    Blake2b h((const char*)personal); h.update(data, len); h.finalize(out); return 0;
}
int h2o_g1_decompress(const uint8_t in[32], uint8_t out[64], int* is_identity) {
    G1Affine p; if (!g1_from_bytes(in, p)) return -1;
    g1_to_xy(p, out); *is_identity = p.inf; return 0;
}
int h2o_g1_compress(const uint8_t in[64], uint8_t out[32]) { g1_to_bytes(g1_from_xy(in), out); return 0; }
// == MSMKZG::eval + to_affine (poly/kzg/msm.rs:81-86, arithmetic.rs:7-108)
int h2o_g1_msm(const uint8_t* scalars32, const uint8_t* bases64, size_t n, uint8_t out[64], int* is_identity) {
    std::vector<Fr> s(n); std::vector<G1Affine> b(n);
    for (size_t i = 0; i < n; ++i) { if (!Fr::from_bytes(scalars32 + 32 * i, s[i])) return -1; b[i] = g1_from_xy(bases64 + 64 * i); if (!b[i].on_curve()) return -2; }
    G1Affine r = best_multiexp(s.data(), b.data(), n).to_affine();
    g1_to_xy(r, out); *is_identity = r.inf; return 0;
}
// == DualMSM::check on two already-evaluated channels (poly/kzg/msm.rs:185-203)
int h2o_pairing_check(const uint8_t* params, size_t plen, int pfmt, const uint8_t left[64], const uint8_t right[64], int* ok) {
    try {
        ParamsKZG p = read_params(params, plen, (SerdeFormat)pfmt);
        *ok = pairing_product_is_one(g1_from_xy(left), p.s_g2, g1_from_xy(right), p.g2.neg());
        return 0;
    } catch (...) { return -1; }
}
// raw G2 interface for the SRS known-answer tests: e(a, q1) * e(b, q2) == 1, G2 as x.c0|x.c1|y.c0|y.c1 canonical
int h2o_pairing_product_is_one(const uint8_t a[64], const uint8_t q1[128], const uint8_t b[64], const uint8_t q2[128], int* ok) {
    G2Affine Q1, Q2; Q1.inf = Q2.inf = false;
    Fq::from_bytes(q1, Q1.x.c0); Fq::from_bytes(q1 + 32, Q1.x.c1); Fq::from_bytes(q1 + 64, Q1.y.c0); Fq::from_bytes(q1 + 96, Q1.y.c1);
    Fq::from_bytes(q2, Q2.x.c0); Fq::from_bytes(q2 + 32, Q2.x.c1); Fq::from_bytes(q2 + 64, Q2.y.c0); Fq::from_bytes(q2 + 96, Q2.y.c1);
    if (!Q1.on_curve() || !Q2.on_curve()) return -1;
    *ok = pairing_product_is_one(g1_from_xy(a), Q1, g1_from_xy(b), Q2);
    return 0;
}
// convert params between serde formats (exercises G2 (de)compression)
size_t h2o_params_convert(const uint8_t* params, size_t plen, int from_fmt, int to_fmt, uint8_t* buf, size_t cap) {
    try { return copy_out(write_params(read_params(params, plen, (SerdeFormat)from_fmt), (SerdeFormat)to_fmt), buf, cap); } catch (...) { return 0; }
}
size_t h2o_vk_convert(const uint8_t* vk, size_t len, int from_fmt, int to_fmt, uint8_t* buf, size_t cap) {
    try { return copy_out(write_vk(read_vk(vk, len, (SerdeFormat)from_fmt), (SerdeFormat)to_fmt), buf, cap); } catch (...) { return 0; }
}

// ---------------------------------------------------------------- verifier
int h2o_verify_single(const uint8_t* params, size_t plen, int pfmt, const uint8_t* vkb, size_t vlen, int vfmt,
                      const uint8_t* inst32, const size_t* col_lens, size_t ncols, const uint8_t* proof, size_t proof_len) {
    try {
        ParamsKZG p = read_params(params, plen, (SerdeFormat)pfmt);
        VerifyingKey vk = read_vk(vkb, vlen, (SerdeFormat)vfmt);
        DualMSM msm;
        Error e = verify_proof_multi(p, vk, parse_multi(inst32, col_lens, ncols), proof, proof_len, msm, nullptr, nullptr, g_opts);
        if (e != OK) return e;
        return msm.check(p) ? OK : ConstraintSystemFailure;   // SingleStrategy (strategy.rs:164-176)
    } catch (...) { return -100; }
}

// Per-proof Guard in reference term order + the Fiat-Shamir challenges (parity/debug).
// challenges32 receives: user challenges..., theta, beta, gamma, y, x, shplonk y, v, u.
int h2o_guard_msm(const uint8_t* params, size_t plen, int pfmt, const uint8_t* vkb, size_t vlen, int vfmt,
                  const uint8_t* inst32, const size_t* col_lens, size_t ncols, const uint8_t* proof, size_t proof_len,
                  uint8_t* right_scalars32, uint8_t* right_bases64, size_t* n_right,
                  uint8_t* left_scalars32, uint8_t* left_bases64, size_t* n_left,
                  uint8_t* challenges32, size_t* n_challenges) {
    try {
        ParamsKZG p = read_params(params, plen, (SerdeFormat)pfmt);
        VerifyingKey vk = read_vk(vkb, vlen, (SerdeFormat)vfmt);
        DualMSM acc; VerifyTrace t;
        Error e = verify_proof_multi(p, vk, parse_multi(inst32, col_lens, ncols), proof, proof_len, acc, &t, nullptr, g_opts);
        if (e != OK) return e;
        size_t cap_r = *n_right, cap_l = *n_left;
        *n_right = acc.right.scalars.size(); *n_left = acc.left.scalars.size();
        if (*n_right > cap_r || *n_left > cap_l) return -101;
        for (size_t i = 0; i < *n_right; ++i) { acc.right.scalars[i].to_bytes(right_scalars32 + 32 * i); g1_to_xy(acc.right.bases[i].to_affine(), right_bases64 + 64 * i); }
        for (size_t i = 0; i < *n_left; ++i) { acc.left.scalars[i].to_bytes(left_scalars32 + 32 * i); g1_to_xy(acc.left.bases[i].to_affine(), left_bases64 + 64 * i); }
        if (challenges32) {
            std::vector<Fr> ch = t.challenges;
            for (const Fr& f : {t.theta, t.beta, t.gamma, t.y, t.x}) ch.push_back(f);
            if (g_opts.multiopen == MO_SHPLONK) ch.push_back(t.sh_y);   // GWC has no y' (gwc.rs:73-83: v, then u)
            ch.push_back(t.sh_v); ch.push_back(t.sh_u);
            if (ch.size() > *n_challenges) return -101;
            *n_challenges = ch.size();
            for (size_t i = 0; i < ch.size(); ++i) ch[i].to_bytes(challenges32 + 32 * i);
        }
        return 0;
    } catch (...) { return -100; }
}

// == N x verify_proof under AccumulatorStrategy + finalize (poly/kzg/strategy.rs:125-140).
// All proofs share one instance shape.  rand32: the n Fr::random draws (strategy.rs:129).
// A proof whose verify_proof returns an error is reported in statuses[] and contributes nothing.
int h2o_verify_batch(const uint8_t* params, size_t plen, int pfmt, const uint8_t* vkb, size_t vlen, int vfmt,
                     size_t n, const uint8_t* proofs, size_t proof_len, const uint8_t* inst32, const size_t* col_lens, size_t ncols,
                     const uint8_t* rand32, int* statuses, int* batch_ok, uint8_t out_left[64], uint8_t out_right[64]) {
    try {
        ParamsKZG p = read_params(params, plen, (SerdeFormat)pfmt);
        VerifyingKey vk = read_vk(vkb, vlen, (SerdeFormat)vfmt);
        size_t per = 0; for (size_t c = 0; c < ncols; ++c) per += col_lens[c];
        AccumulatorStrategy st; bool all_ok = true;
        for (size_t i = 0; i < n; ++i) {
            Fr r; if (!Fr::from_bytes(rand32 + 32 * i, r)) return -1;
            DualMSM saved = st.acc;
            st.acc.scale(r);  // strategy.rs:129 — before the closure runs
            Error e = verify_proof_multi(p, vk, parse_multi(inst32 + 32 * per * i, col_lens, ncols), proofs + proof_len * i, proof_len, st.acc, nullptr, nullptr, g_opts);
            statuses[i] = e;
            if (e != OK) { all_ok = false; saved.scale(r); st.acc = saved; }
        }
        G1Affine l = st.acc.left.eval().to_affine(), r = st.acc.right.eval().to_affine();
        g1_to_xy(l, out_left); g1_to_xy(r, out_right);
        *batch_ok = all_ok && st.finalize(p);
        return 0;
    } catch (...) { return -100; }
}

// N x verify_proof under SingleStrategy (one pairing per proof), single thread; returns #accepted
int h2o_verify_each(const uint8_t* params, size_t plen, int pfmt, const uint8_t* vkb, size_t vlen, int vfmt,
                    size_t n, const uint8_t* proofs, size_t proof_len, const uint8_t* inst32, const size_t* col_lens, size_t ncols, int* statuses) {
    try {
        ParamsKZG p = read_params(params, plen, (SerdeFormat)pfmt);
        VerifyingKey vk = read_vk(vkb, vlen, (SerdeFormat)vfmt);
        size_t per = 0; for (size_t c = 0; c < ncols; ++c) per += col_lens[c];
        int acc = 0;
        for (size_t i = 0; i < n; ++i) {
            DualMSM msm;
            Error e = verify_proof_multi(p, vk, parse_multi(inst32 + 32 * per * i, col_lens, ncols), proofs + proof_len * i, proof_len, msm, nullptr, nullptr, g_opts);
            statuses[i] = e != OK ? e : (msm.check(p) ? OK : ConstraintSystemFailure);
            acc += statuses[i] == OK;
        }
        return acc;
    } catch (...) { return -100; }
}

// ---------------------------------------------------------------- test-only keygen / prover
// srs: NULL => known-s test SRS with s derived from s_seed; else the reference SRS file bytes.
static Setup* make_setup(const Circuit& c, const uint8_t* srs, size_t srs_len, uint64_t s_seed) {
    Setup* s = new Setup();
    if (srs) s->ck = CommitKey::from_srs_file(srs, srs_len);
    else { Rng r(s_seed); s->ck = CommitKey::from_secret(c.k, r.fr()); }
    if (s->ck.k != c.k) { delete s; return nullptr; }
    s->pk = keygen(c, s->ck);
    return s;
}
void* h2o_setup_vector_mul(uint32_t k, size_t n_mul, const uint8_t* srs, size_t srs_len, uint64_t s_seed) {
    if (k > 24 || 3 * n_mul + 8 > (size_t(1) << k)) return nullptr;  // the circuit uses 3 rows per multiplication and needs the blinding rows free
    try { Setup* s = make_setup(circuit_vector_mul(k, n_mul), srs, srs_len, s_seed); if (s) { s->kind = 0; s->n_mul = n_mul; } return s; } catch (...) { return nullptr; }
}
void* h2o_setup_shuffle(uint32_t k, size_t W, size_t H, const uint8_t* srs, size_t srs_len, uint64_t s_seed) {
    try { Setup* s = make_setup(circuit_two_phase_shuffle(k, W, H), srs, srs_len, s_seed); if (s) { s->kind = 1; s->W = W; s->H = H; } return s; } catch (...) { return nullptr; }
}
void* h2o_setup_wide(uint32_t k, size_t A, size_t F, size_t L, size_t Sh, uint32_t deg, uint64_t seed, const uint8_t* srs, size_t srs_len, uint64_t s_seed) {
    try { Setup* s = make_setup(circuit_wide(k, A, F, L, Sh, deg, seed), srs, srs_len, s_seed); if (s) { s->kind = 2; s->wide_seed = seed; } return s; } catch (...) { return nullptr; }
}
void h2o_setup_free(void* h) { delete (Setup*)h; }
void h2o_setup_set_options(void* h, int multiopen, int transcript) { ((Setup*)h)->multiopen = multiopen; ((Setup*)h)->transcript = transcript; }
// verifier-side options for every h2o_verify_* / h2o_guard_msm call of this thread (0/0 = SHPLONK + Blake2b)
void h2o_set_verify_options(int multiopen, int transcript) { g_opts.multiopen = multiopen; g_opts.transcript = transcript; }
// `instances.len()` for every h2o_verify_* / h2o_guard_msm call of this thread: their ncols / col_lens then describe M x columns
void h2o_set_circuit_instances(size_t m) { g_circuit_instances = m ? m : 1; }
size_t h2o_setup_vk(void* h, int fmt, uint8_t* buf, size_t cap) { return copy_out(write_vk(((Setup*)h)->pk.vk, (SerdeFormat)fmt), buf, cap); }
size_t h2o_setup_params(void* h, int fmt, uint8_t* buf, size_t cap) { return copy_out(write_params(((Setup*)h)->ck.params, (SerdeFormat)fmt), buf, cap); }

// vector_mul: a32/b32 hold n_mul scalars each; instances_out receives the n_mul products (the public inputs)
size_t h2o_prove_vector_mul(void* h, const uint8_t* a32, const uint8_t* b32, uint64_t rng_seed, uint8_t* proof, size_t cap, uint8_t* instances_out) {
    Setup* s = (Setup*)h;
    std::vector<Fr> a(s->n_mul), b(s->n_mul), c(s->n_mul);
    for (size_t i = 0; i < s->n_mul; ++i) { Fr::from_bytes(a32 + 32 * i, a[i]); Fr::from_bytes(b32 + 32 * i, b[i]); c[i] = a[i] * b[i]; if (instances_out) c[i].to_bytes(instances_out + 32 * i); }
    Rng rng(rng_seed);
    return copy_out(create_proof(s->pk, s->ck, {c}, witness_vector_mul(a, b), rng, s->multiopen, s->transcript), proof, cap);
}
// vector_mul with a SHORTER instance vector: the public inputs are the first inst_len products; the caller makes the remaining
// products zero (b_i = 0), so the circuit's copy constraints hold against an instance column that is zero beyond inst_len.
// The prover absorbs exactly inst_len values — what verify_proof is handed per call (`instances`, lib.rs:33-49, 76-82).
size_t h2o_prove_vector_mul_len(void* h, const uint8_t* a32, const uint8_t* b32, size_t inst_len, uint64_t rng_seed, uint8_t* proof, size_t cap, uint8_t* instances_out) {
    Setup* s = (Setup*)h;
    if (inst_len > s->n_mul) return 0;
    std::vector<Fr> a(s->n_mul), b(s->n_mul), c(inst_len);
    for (size_t i = 0; i < s->n_mul; ++i) {
        Fr::from_bytes(a32 + 32 * i, a[i]); Fr::from_bytes(b32 + 32 * i, b[i]);
        Fr prod = a[i] * b[i];
        if (i < inst_len) { c[i] = prod; if (instances_out) c[i].to_bytes(instances_out + 32 * i); }
        else if (!(prod == Fr::zero())) return 0;
    }
    Rng rng(rng_seed);
    return copy_out(create_proof(s->pk, s->ck, {c}, witness_vector_mul(a, b), rng, s->multiopen, s->transcript), proof, cap);
}
// M circuit instances in ONE transcript: a32 / b32 hold M x n_mul scalars, instances_out receives the M x n_mul products
size_t h2o_prove_vector_mul_multi(void* h, size_t M, const uint8_t* a32, const uint8_t* b32, uint64_t rng_seed, uint8_t* proof, size_t cap, uint8_t* instances_out) {
    Setup* s = (Setup*)h;
    std::vector<std::vector<std::vector<Fr>>> insts(M);
    std::vector<WitnessFn> ws;
    for (size_t q = 0; q < M; ++q) {
        std::vector<Fr> a(s->n_mul), b(s->n_mul), c(s->n_mul);
        for (size_t i = 0; i < s->n_mul; ++i) {
            Fr::from_bytes(a32 + 32 * (q * s->n_mul + i), a[i]); Fr::from_bytes(b32 + 32 * (q * s->n_mul + i), b[i]); c[i] = a[i] * b[i];
            if (instances_out) c[i].to_bytes(instances_out + 32 * (q * s->n_mul + i));
        }
        insts[q] = {c};
        ws.push_back(witness_vector_mul(a, b));
    }
    Rng rng(rng_seed);
    return copy_out(create_proof_multi(s->pk, s->ck, insts, ws, rng, s->multiopen, s->transcript), proof, cap);
}
// batch of `count` distinct proofs with pseudo-random a, b derived from seed+i; nthreads workers
size_t h2o_prove_vector_mul_batch(void* h, size_t count, uint64_t seed, unsigned nthreads, uint8_t* proofs, size_t proof_len, uint8_t* instances_out) {
    Setup* s = (Setup*)h;
    std::atomic<size_t> next(0); std::atomic<size_t> bad(0);
    auto work = [&]() {
        for (;;) {
            size_t i = next.fetch_add(1); if (i >= count) return;
            Rng wr(seed * 0x9e3779b97f4a7c15ULL + i);
            std::vector<Fr> a(s->n_mul), b(s->n_mul), c(s->n_mul);
            for (size_t j = 0; j < s->n_mul; ++j) { a[j] = wr.fr(); b[j] = wr.fr(); c[j] = a[j] * b[j]; c[j].to_bytes(instances_out + 32 * (i * s->n_mul + j)); }
            Rng rng(seed ^ (0xabcdef12345ULL + i));
            std::vector<uint8_t> p = create_proof(s->pk, s->ck, {c}, witness_vector_mul(a, b), rng, s->multiopen, s->transcript);
            if (p.size() != proof_len) { bad++; continue; }
            memcpy(proofs + i * proof_len, p.data(), proof_len);
        }
    };
    std::vector<std::thread> th;
    for (unsigned t = 0; t < (nthreads ? nthreads : 1); ++t) th.emplace_back(work);
    for (auto& t : th) t.join();
    return bad.load() ? 0 : count;
}
// two-phase shuffle: random W x H table from data_seed, shuffled by rows; break_it != 0 swaps two
// cells of shuffled column 0 (the reference's negative test, tests/shuffle.rs:291-308)
size_t h2o_prove_shuffle(void* h, uint64_t data_seed, int break_it, uint64_t rng_seed, uint8_t* proof, size_t cap) {
    Setup* s = (Setup*)h;
    Rng dr(data_seed);
    std::vector<std::vector<Fr>> orig(s->W, std::vector<Fr>(s->H)), shuf;
    for (auto& col : orig) for (auto& v : col) v = dr.fr();
    shuf = orig;
    for (size_t row = s->H - 1; row >= 1; --row) { size_t r = dr.next() % row; for (auto& col : shuf) std::swap(col[row], col[r]); }
    if (break_it) std::swap(shuf[0][0], shuf[0][1]);
    Rng rng(rng_seed);
    return copy_out(create_proof(s->pk, s->ck, {}, witness_two_phase_shuffle(orig, shuf), rng, s->multiopen, s->transcript), proof, cap);
}
// M shuffles in one transcript (data_seed + q each); break_at in [0, M) breaks that instance's shuffle, -1 none
size_t h2o_prove_shuffle_multi(void* h, size_t M, uint64_t data_seed, int break_at, uint64_t rng_seed, uint8_t* proof, size_t cap) {
    Setup* s = (Setup*)h;
    std::vector<WitnessFn> ws;
    for (size_t q = 0; q < M; ++q) {
        Rng dr(data_seed + 1000003ULL * q);
        std::vector<std::vector<Fr>> orig(s->W, std::vector<Fr>(s->H)), shuf;
        for (auto& col : orig) for (auto& v : col) v = dr.fr();
        shuf = orig;
        for (size_t row = s->H - 1; row >= 1; --row) { size_t r = dr.next() % row; for (auto& col : shuf) std::swap(col[row], col[r]); }
        if ((int)q == break_at) std::swap(shuf[0][0], shuf[0][1]);
        ws.push_back(witness_two_phase_shuffle(orig, shuf));
    }
    Rng rng(rng_seed);
    return copy_out(create_proof_multi(s->pk, s->ck, std::vector<std::vector<std::vector<Fr>>>(M), ws, rng, s->multiopen, s->transcript), proof, cap);
}
// M instances of the wide circuit in one transcript (witness_seed + q each); tamper_at in [0, M) corrupts one lookup input cell
// of that instance, -1 none; instances_out receives M x 8 public inputs
size_t h2o_prove_wide_multi(void* h, size_t M, uint64_t witness_seed, int tamper_at, uint64_t rng_seed, uint8_t* proof, size_t cap, uint8_t* instances_out) {
    Setup* s = (Setup*)h;
    const Circuit& c = s->pk.circuit;
    std::vector<std::vector<std::vector<Fr>>> insts(M);
    std::vector<WitnessFn> ws;
    for (size_t q = 0; q < M; ++q) {
        WitnessFn base = witness_wide(c, witness_seed + 7919ULL * q);
        std::vector<std::vector<Fr>> scratch(c.cs.num_advice_columns, std::vector<Fr>(c.n(), Fr::zero()));
        base(0, {}, scratch);
        std::vector<Fr> inst(8);
        for (size_t i = 0; i < 8; ++i) { inst[i] = scratch[2][i + 1]; if (instances_out) inst[i].to_bytes(instances_out + 32 * (8 * q + i)); }
        insts[q] = {inst};
        if ((int)q == tamper_at) ws.push_back([base](unsigned ph, const std::vector<Fr>& ch, std::vector<std::vector<Fr>>& adv) { base(ph, ch, adv); if (ph == 0) adv[1][5] = adv[1][5] + Fr::from_u64(123456789); });
        else ws.push_back(base);
    }
    Rng rng(rng_seed);
    return copy_out(create_proof_multi(s->pk, s->ck, insts, ws, rng, s->multiopen, s->transcript), proof, cap);
}
// wide circuit: instances_out receives the 8 public inputs; tamper != 0 corrupts one lookup input cell
size_t h2o_prove_wide(void* h, uint64_t witness_seed, int tamper, uint64_t rng_seed, uint8_t* proof, size_t cap, uint8_t* instances_out) {
    Setup* s = (Setup*)h;
    const Circuit& c = s->pk.circuit;
    WitnessFn base = witness_wide(c, witness_seed);
    std::vector<std::vector<Fr>> scratch(c.cs.num_advice_columns, std::vector<Fr>(c.n(), Fr::zero()));
    base(0, {}, scratch);
    std::vector<Fr> inst(8);
    for (size_t i = 0; i < 8; ++i) { inst[i] = scratch[2][i + 1]; if (instances_out) inst[i].to_bytes(instances_out + 32 * i); }
    WitnessFn w = base;
    if (tamper) w = [base](unsigned ph, const std::vector<Fr>& ch, std::vector<std::vector<Fr>>& adv) { base(ph, ch, adv); if (ph == 0) adv[1][5] = adv[1][5] + Fr::from_u64(123456789); };
    Rng rng(rng_seed);
    return copy_out(create_proof(s->pk, s->ck, {inst}, w, rng, s->multiopen, s->transcript), proof, cap);
}

}  // extern "C"
