// ORACLE — TEST INFRASTRUCTURE ONLY (see prover.hpp header).
#include "prover.hpp"
#include <algorithm>
#include <array>
#include <map>
#include <numeric>

namespace h2o {

// ------------------------------------------------------------------------------ NTT
namespace {
struct Ntt {
    uint32_t logm; size_t m;
    std::vector<Fr> tw, tw_inv;  // omega^i, omega^-i for i < m/2
    Fr m_inv, omega;
    explicit Ntt(uint32_t lg) : logm(lg), m((size_t)1 << lg) {
        Fr w = fr_consts().root_of_unity;
        for (uint32_t i = lg; i < (uint32_t)FrConsts::S; ++i) w = w.sqr();
        omega = w;
        Fr wi = w.inv();
        tw.resize(m / 2 ? m / 2 : 1); tw_inv.resize(tw.size());
        Fr a = Fr::one(), b = Fr::one();
        for (size_t i = 0; i < tw.size(); ++i) { tw[i] = a; tw_inv[i] = b; a *= w; b *= wi; }
        m_inv = Fr::from_u64(m).inv();
    }
    void run(std::vector<Fr>& a, bool inverse) const {
        for (size_t i = 1, j = 0; i < m; ++i) {
            size_t bit = m >> 1;
            for (; j & bit; bit >>= 1) j ^= bit;
            j ^= bit;
            if (i < j) std::swap(a[i], a[j]);
        }
        const std::vector<Fr>& t = inverse ? tw_inv : tw;
        for (size_t len = 2; len <= m; len <<= 1) {
            size_t step = m / len;
            for (size_t i = 0; i < m; i += len)
                for (size_t j = 0; j < len / 2; ++j) {
                    Fr u = a[i + j], v = a[i + j + len / 2] * t[j * step];
                    a[i + j] = u + v; a[i + j + len / 2] = u - v;
                }
        }
        if (inverse) for (auto& x : a) x *= m_inv;
    }
};
const Ntt& ntt_ctx(uint32_t lg) {
    static std::map<uint32_t, Ntt*> cache;
    auto it = cache.find(lg);
    if (it == cache.end()) it = cache.emplace(lg, new Ntt(lg)).first;
    return *it->second;
}
const u64 COSET_GEN = 7;

std::vector<Fr> lagrange_to_coeff(const std::vector<Fr>& v, uint32_t k) {
    std::vector<Fr> c = v; ntt_ctx(k).run(c, true); return c;
}
// evaluations of a degree < n polynomial on the coset  zeta * <omega_ext>
std::vector<Fr> coeff_to_ext(const std::vector<Fr>& c, uint32_t ext_k) {
    std::vector<Fr> a((size_t)1 << ext_k, Fr::zero());
    Fr z = Fr::from_u64(COSET_GEN), p = Fr::one();
    for (size_t i = 0; i < c.size(); ++i) { a[i] = c[i] * p; p *= z; }
    ntt_ctx(ext_k).run(a, false);
    return a;
}
Fr horner(const std::vector<Fr>& c, const Fr& x) {
    Fr acc = Fr::zero();
    for (size_t i = c.size(); i-- > 0;) acc = acc * x + c[i];
    return acc;
}
// windowed MSM for the SRS-file commit key (prover-side convenience, not the reference's MSM)
G1 msm_window(const std::vector<Fr>& s, const std::vector<G1Affine>& b) {
    const int c = 8; size_t n = s.size();
    std::vector<uint8_t> repr(32 * n);
    for (size_t i = 0; i < n; ++i) s[i].to_bytes(&repr[32 * i]);
    G1 acc = G1::identity();
    for (int w = 31; w >= 0; --w) {
        for (int i = 0; i < c; ++i) acc = acc.dbl();
        std::vector<G1> buckets(255, G1::identity());
        for (size_t i = 0; i < n; ++i) { uint8_t d = repr[32 * i + w]; if (d) buckets[d - 1] = buckets[d - 1].add_affine(b[i]); }
        G1 run = G1::identity(), sum = G1::identity();
        for (int j = 254; j >= 0; --j) { run = run.add(buckets[j]); sum = sum.add(run); }
        acc = acc.add(sum);
    }
    return acc;
}
}  // namespace

// ------------------------------------------------------------------------------ commit key
CommitKey CommitKey::from_secret(uint32_t k, const Fr& s) {
    CommitKey ck; ck.k = k; ck.n = 1ULL << k; ck.known_s = true; ck.s = s;
    // L_i(s) = omega^i (s^n - 1) / (n (s - omega^i))
    const Ntt& nt = ntt_ctx(k);
    std::vector<Fr> den(ck.n), wi(ck.n);
    Fr w = Fr::one();
    for (size_t i = 0; i < ck.n; ++i) { wi[i] = w; den[i] = s - w; w *= nt.omega; }
    batch_invert(den.data(), den.size());
    u64 e[4] = {ck.n, 0, 0, 0};
    Fr common = (s.pow(e) - Fr::one()) * nt.m_inv;
    ck.lagrange_at_s.resize(ck.n);
    for (size_t i = 0; i < ck.n; ++i) ck.lagrange_at_s[i] = wi[i] * common * den[i];
    ck.params.k = k; ck.params.g = G1Affine::generator(); ck.params.g2 = g2_generator();
    ck.params.s_g2 = g2_generator().mul(s);
    return ck;
}
CommitKey CommitKey::from_srs_file(const uint8_t* data, size_t len) {
    ByteReader r(data, len);
    CommitKey ck; ck.k = r.u32le(); ck.n = 1ULL << ck.k; ck.known_s = false;
    for (size_t i = 0; i < ck.n; ++i) ck.g.push_back(read_g1(r, RawBytes));
    for (size_t i = 0; i < ck.n; ++i) ck.g_lagrange.push_back(read_g1(r, RawBytes));
    ck.params.k = ck.k; ck.params.g = ck.g[0];
    ck.params.g2 = read_g2(r, RawBytes); ck.params.s_g2 = read_g2(r, RawBytes);
    return ck;
}
G1Affine CommitKey::commit_lagrange(const std::vector<Fr>& v) const {
    if (known_s) {
        Fr acc = Fr::zero();
        for (size_t i = 0; i < v.size(); ++i) if (!v[i].is_zero()) acc += v[i] * lagrange_at_s[i];
        return G1::from_affine(params.g).mul(acc).to_affine();
    }
    return msm_window(v, g_lagrange).to_affine();
}
G1Affine CommitKey::commit_coeff(const std::vector<Fr>& c) const {
    if (known_s) return G1::from_affine(params.g).mul(horner(c, s)).to_affine();
    std::vector<G1Affine> b(g.begin(), g.begin() + c.size());
    return msm_window(c, b).to_affine();
}

// ------------------------------------------------------------------------------ keygen
static uint32_t extended_k(uint32_t k, uint32_t d) {
    uint32_t e = k;
    while ((1ULL << e) < (1ULL << k) * (u64)(d - 1)) ++e;
    return e;
}

ProvingKey keygen(const Circuit& c, const CommitKey& ck) {
    ProvingKey pk; pk.circuit = c;
    const size_t n = c.n();
    const ConstraintSystem& cs = c.cs;
    const Ntt& nt = ntt_ctx(c.k);
    pk.ext_k = extended_k(c.k, c.cs_degree);
    // permutation: cycles over (perm column, row), merged by swapping successors
    size_t P = cs.permutation_columns.size();
    std::vector<uint32_t> succ(P * n), parent(P * n);
    std::iota(succ.begin(), succ.end(), 0u); std::iota(parent.begin(), parent.end(), 0u);
    std::function<uint32_t(uint32_t)> find = [&](uint32_t a) { while (parent[a] != a) { parent[a] = parent[parent[a]]; a = parent[a]; } return a; };
    for (const CopyConstraint& cc : c.copies) {
        uint32_t a = cc.col_a * n + cc.row_a, b = cc.col_b * n + cc.row_b;
        uint32_t ra = find(a), rb = find(b);
        if (ra == rb) continue;
        std::swap(succ[a], succ[b]); parent[ra] = rb;
    }
    std::vector<Fr> omega_pow(n), delta_pow(P ? P : 1);
    { Fr w = Fr::one(); for (size_t i = 0; i < n; ++i) { omega_pow[i] = w; w *= nt.omega; } }
    { Fr d = Fr::one(); for (size_t i = 0; i < delta_pow.size(); ++i) { delta_pow[i] = d; d *= fr_consts().delta; } }
    pk.sigma.assign(P, std::vector<Fr>(n));
    for (size_t j = 0; j < P; ++j)
        for (size_t i = 0; i < n; ++i) { uint32_t t = succ[j * n + i]; pk.sigma[j][i] = delta_pow[t / n] * omega_pow[t % n]; }

    VerifyingKey& vk = pk.vk;
    vk.k = c.k; vk.cs = cs; vk.cs_degree = c.cs_degree;
    for (const auto& f : c.fixed) vk.fixed_commitments.push_back(ck.commit_lagrange(f));
    for (const auto& s : pk.sigma) vk.permutation_commitments.push_back(ck.commit_lagrange(s));
    // selectors bitmaps (unused by verification): the first num_selectors fixed columns, non-zero = set
    for (uint32_t s = 0; s < cs.num_selectors; ++s) {
        std::vector<uint8_t> bits((n + 7) / 8, 0);
        for (size_t i = 0; i < n; ++i) if (!c.fixed[s][i].is_zero()) bits[i / 8] |= (uint8_t)(1u << (i % 8));
        vk.selectors.push_back(bits);
    }
    // transcript_repr: in halo2_proofs a Blake2b hash of the VK's pinned description; any
    // Fr works for verification.  Here: Blake2b("Halo2-Verify-Key") over the serialized VK.
    vk.transcript_repr = Fr::zero();
    { std::vector<uint8_t> bytes = write_vk(vk, RawBytes); Blake2b h("Halo2-Verify-Key"); h.update(bytes.data(), bytes.size()); uint8_t out[64]; h.finalize(out); vk.transcript_repr = Fr::from_uniform_bytes(out); }

    for (const auto& f : c.fixed) { pk.fixed_coeff.push_back(lagrange_to_coeff(f, c.k)); pk.fixed_ext.push_back(coeff_to_ext(pk.fixed_coeff.back(), pk.ext_k)); }
    for (const auto& s : pk.sigma) { pk.sigma_coeff.push_back(lagrange_to_coeff(s, c.k)); pk.sigma_ext.push_back(coeff_to_ext(pk.sigma_coeff.back(), pk.ext_k)); }
    size_t u = c.usable_rows();
    std::vector<Fr> l0(n, Fr::zero()), ll(n, Fr::zero()), la(n, Fr::zero());
    l0[0] = Fr::one(); ll[u] = Fr::one();
    for (size_t i = 0; i < u; ++i) la[i] = Fr::one();  // 1 - (l_last + l_blind)
    pk.l0_ext = coeff_to_ext(lagrange_to_coeff(l0, c.k), pk.ext_k);
    pk.llast_ext = coeff_to_ext(lagrange_to_coeff(ll, c.k), pk.ext_k);
    pk.lactive_ext = coeff_to_ext(lagrange_to_coeff(la, c.k), pk.ext_k);
    return pk;
}

// ------------------------------------------------------------------------------ prover
namespace {
struct VarSrc { const std::vector<Fr>* arr; int32_t rot; Fr chal; };

// value of an expression polynomial at index i of a cyclic table of size `size` where one row step = `stride`
Fr eval_expr_at(const ExprPoly& e, const std::vector<Fr>& coeffs, const std::vector<VarSrc>& vars, size_t i, size_t size, size_t stride) {
    Fr sum = Fr::zero();
    for (const ExprTerm& t : e.terms) {
        Fr prod = coeffs[t.coeff_idx];
        for (const auto& f : t.factors) {
            const VarSrc& v = vars[f.first];
            Fr val = v.arr ? (*v.arr)[(i + size + (int64_t)v.rot * (int64_t)stride) % size] : v.chal;
            prod *= val.pow_u64(f.second);
        }
        sum += prod;
    }
    return sum;
}

struct PolyQuery { const std::vector<Fr>* poly; Fr point, eval; };
struct FrLess { bool operator()(const Fr& a, const Fr& b) const { return Fr::cmp(a, b) < 0; } };

// quotient of p(X) by (X - a), assuming p(a) == 0 (remainder discarded)
void divide_by_linear(std::vector<Fr>& p, const Fr& a) {
    Fr carry = Fr::zero();
    for (size_t i = p.size(); i-- > 0;) { Fr t = p[i] + carry * a; p[i] = carry; carry = t; }
    // now p[i] holds quotient coefficient of X^i (top coefficient slot is zero)
}
// coefficients of the interpolant through (pts[i], vals[i]) — Newton form expanded
std::vector<Fr> interpolate(const std::vector<Fr>& pts, const std::vector<Fr>& vals) {
    size_t n = pts.size();
    std::vector<Fr> dd = vals;
    for (size_t lvl = 1; lvl < n; ++lvl)
        for (size_t i = n - 1; i >= lvl; --i) dd[i] = (dd[i] - dd[i - 1]) * (pts[i] - pts[i - lvl]).inv();
    std::vector<Fr> res(n, Fr::zero()), basis = {Fr::one()};
    for (size_t i = 0; i < n; ++i) {
        for (size_t t = 0; t < basis.size(); ++t) res[t] += dd[i] * basis[t];
        std::vector<Fr> nb(basis.size() + 1, Fr::zero());
        for (size_t t = 0; t < basis.size(); ++t) { nb[t + 1] += basis[t]; nb[t] -= basis[t] * pts[i]; }
        basis.swap(nb);
    }
    return res;
}
}  // namespace

// One circuit instance's share of the prover state (the reference's verifier loops over `num_proofs` of these inside one
// transcript, lib.rs:63-161,220-253)
struct InstState {
    std::vector<std::vector<Fr>> inst, advice;
    std::vector<std::vector<Fr>> lkA, lkS, lkAp, lkSp, lkZ, permZ, shA, shS, shZ;
    std::vector<std::vector<Fr>> adv_c, inst_c, pz_c, lkAp_c, lkSp_c, lkZ_c, shZ_c;
    std::vector<std::vector<Fr>> adv_e, inst_e, pz_e, lkAp_e, lkSp_e, lkZ_e, shZ_e;
    std::vector<VarSrc> row_vars, ext_vars;
    std::vector<Fr> adv_evals;
    std::vector<std::array<Fr, 3>> pz_ev;
    std::vector<std::array<Fr, 5>> lk_ev;
    std::vector<std::array<Fr, 2>> sh_ev;
};

std::vector<uint8_t> create_proof(const ProvingKey& pk, const CommitKey& ck, const std::vector<std::vector<Fr>>& instances,
                                  const WitnessFn& witness, Rng& rng, int multiopen, int transcript) {
    return create_proof_multi(pk, ck, {instances}, {witness}, rng, multiopen, transcript);
}

// `instances[m]` / `witnesses[m]`: circuit instance m of the transcript (`instances: &[&[&[Fr]]]`, lib.rs:33-49).  Every step
// that the verifier repeats per instance is written per instance in the same order (lib.rs:91-161, 220-253, 349-391).
std::vector<uint8_t> create_proof_multi(const ProvingKey& pk, const CommitKey& ck, const std::vector<std::vector<std::vector<Fr>>>& instances,
                                        const std::vector<WitnessFn>& witnesses, Rng& rng, int multiopen, int transcript) {
    const Circuit& c = pk.circuit; const ConstraintSystem& cs = c.cs;
    const size_t n = c.n(), u = c.usable_rows(), bf = cs.blinding_factors();
    const uint32_t k = c.k, ek = pk.ext_k;
    const size_t m = (size_t)1 << ek, stride = m / n;
    const size_t M = instances.size();
    const Ntt& nt = ntt_ctx(k);
    const Fr omega = nt.omega, omega_inv = omega.inv();
    TranscriptWrite tr(transcript);
    tr.common_scalar(pk.vk.transcript_repr);
    for (const auto& one : instances) for (const auto& col : one) for (const Fr& v : col) tr.common_scalar(v);

    std::vector<InstState> S(M);
    // instance columns as Lagrange polynomials
    for (size_t q = 0; q < M; ++q) {
        S[q].inst.assign(cs.num_instance_columns, std::vector<Fr>(n, Fr::zero()));
        for (size_t j = 0; j < S[q].inst.size(); ++j) for (size_t i = 0; i < instances[q][j].size(); ++i) S[q].inst[j][i] = instances[q][j][i];
        S[q].advice.assign(cs.num_advice_columns, std::vector<Fr>(n, Fr::zero()));
    }

    // phases: the advice commitments of every instance, then the phase's challenges
    std::vector<Fr> challenges(cs.num_challenges, Fr::zero());
    for (unsigned phase = 0; phase <= cs.max_phase(); ++phase) {
        for (size_t q = 0; q < M; ++q) {
            witnesses[q](phase, challenges, S[q].advice);
            for (size_t j = 0; j < cs.num_advice_columns; ++j) {
                if (cs.advice_column_phase[j] != phase) continue;
                for (size_t i = u; i < n; ++i) S[q].advice[j][i] = rng.fr();
                tr.write_point(ck.commit_lagrange(S[q].advice[j]));
            }
        }
        for (size_t j = 0; j < cs.num_challenges; ++j) if (cs.challenge_phase[j] == phase) challenges[j] = tr.squeeze_challenge();
    }

    // variable table for expression polynomials over the n-row domain
    auto make_vars = [&](const std::vector<std::vector<Fr>>& adv, const std::vector<std::vector<Fr>>& fix, const std::vector<std::vector<Fr>>& ins) {
        std::vector<VarSrc> vars;
        for (const Query& q : cs.advice_queries) vars.push_back({&adv[q.column.index], q.rotation, Fr::zero()});
        for (const Query& q : cs.fixed_queries) vars.push_back({&fix[q.column.index], q.rotation, Fr::zero()});
        for (const Query& q : cs.instance_queries) vars.push_back({&ins[q.column.index], q.rotation, Fr::zero()});
        for (const Fr& ch : challenges) vars.push_back({nullptr, 0, ch});
        return vars;
    };
    for (size_t q = 0; q < M; ++q) S[q].row_vars = make_vars(S[q].advice, c.fixed, S[q].inst);

    Fr theta = tr.squeeze_challenge();
    auto compress_rows = [&](const std::vector<VarSrc>& row_vars, const std::vector<ExprPoly>& es) {
        std::vector<Fr> out(n, Fr::zero());
        for (size_t i = 0; i < n; ++i) {
            Fr acc = Fr::zero();
            for (const ExprPoly& e : es) acc = acc * theta + eval_expr_at(e, cs.coeff_vals, row_vars, i, n, 1);
            out[i] = acc;
        }
        return out;
    };
    // lookups: permuted input / table columns
    size_t L = cs.lookups.size(), Sh = cs.shuffles.size();
    for (size_t q = 0; q < M; ++q) {
        InstState& T = S[q];
        T.lkA.resize(L); T.lkS.resize(L); T.lkAp.resize(L); T.lkSp.resize(L); T.lkZ.resize(L);
        for (size_t l = 0; l < L; ++l) {
            T.lkA[l] = compress_rows(T.row_vars, cs.lookups[l].input);
            T.lkS[l] = compress_rows(T.row_vars, cs.lookups[l].table);
            std::vector<Fr> a(T.lkA[l].begin(), T.lkA[l].begin() + u);
            std::sort(a.begin(), a.end(), FrLess());
            // S': a "new" value of A' takes the matching table entry; repeats take whatever is left over
            std::multimap<Fr, int, FrLess> leftover;
            for (size_t i = 0; i < u; ++i) leftover.emplace(T.lkS[l][i], 0);
            std::vector<Fr> sp(u); std::vector<size_t> holes;
            for (size_t i = 0; i < u; ++i) {
                if (i == 0 || !(a[i] == a[i - 1])) {
                    sp[i] = a[i];
                    auto it = leftover.find(a[i]);
                    if (it != leftover.end()) leftover.erase(it);  // absent => invalid witness; the proof will be rejected
                } else holes.push_back(i);
            }
            {
                auto it = leftover.begin();
                for (size_t hidx : holes) { if (it == leftover.end()) break; sp[hidx] = it->first; ++it; }
            }
            T.lkAp[l].assign(n, Fr::zero()); T.lkSp[l].assign(n, Fr::zero());
            for (size_t i = 0; i < u; ++i) { T.lkAp[l][i] = a[i]; T.lkSp[l][i] = sp[i]; }
            for (size_t i = u; i < n; ++i) { T.lkAp[l][i] = rng.fr(); T.lkSp[l][i] = rng.fr(); }
            tr.write_point(ck.commit_lagrange(T.lkAp[l]));
            tr.write_point(ck.commit_lagrange(T.lkSp[l]));
        }
    }
    Fr beta = tr.squeeze_challenge();
    Fr gamma = tr.squeeze_challenge();

    // permutation grand products
    size_t P = cs.permutation_columns.size(), chunk = c.cs_degree - 2;
    size_t nsets = P == 0 ? 0 : (P + chunk - 1) / chunk;
    std::vector<Fr> omega_pow(n); { Fr w = Fr::one(); for (size_t i = 0; i < n; ++i) { omega_pow[i] = w; w *= omega; } }
    for (size_t q = 0; q < M; ++q) {
        InstState& T = S[q];
        auto perm_col_values = [&](size_t j) -> const std::vector<Fr>& {
            const Column& col = cs.permutation_columns[j];
            if (col.is_advice()) return T.advice[col.index];
            if (col.type == COL_FIXED) return c.fixed[col.index];
            return T.inst[col.index];
        };
        T.permZ.assign(nsets, std::vector<Fr>(n));
        Fr last = Fr::one();
        for (size_t s2 = 0; s2 < nsets; ++s2) {
            size_t lo = s2 * chunk, hi = std::min(P, lo + chunk);
            std::vector<Fr> num(u, Fr::one()), den(u, Fr::one());
            for (size_t j = lo; j < hi; ++j) {
                const std::vector<Fr>& v = perm_col_values(j);
                Fr dj = fr_consts().delta.pow_u64(j);
                for (size_t i = 0; i < u; ++i) {
                    num[i] *= v[i] + beta * dj * omega_pow[i] + gamma;
                    den[i] *= v[i] + beta * pk.sigma[j][i] + gamma;
                }
            }
            batch_invert(den.data(), den.size());
            T.permZ[s2][0] = last;
            for (size_t i = 0; i < u; ++i) T.permZ[s2][i + 1] = T.permZ[s2][i] * num[i] * den[i];
            last = T.permZ[s2][u];
            for (size_t i = u + 1; i < n; ++i) T.permZ[s2][i] = rng.fr();
            tr.write_point(ck.commit_lagrange(T.permZ[s2]));
        }
    }
    for (size_t q = 0; q < M; ++q) {
        InstState& T = S[q];
        for (size_t l = 0; l < L; ++l) {
            std::vector<Fr> den(u);
            for (size_t i = 0; i < u; ++i) den[i] = (T.lkAp[l][i] + beta) * (T.lkSp[l][i] + gamma);
            batch_invert(den.data(), den.size());
            T.lkZ[l].assign(n, Fr::zero()); T.lkZ[l][0] = Fr::one();
            for (size_t i = 0; i < u; ++i) T.lkZ[l][i + 1] = T.lkZ[l][i] * (T.lkA[l][i] + beta) * (T.lkS[l][i] + gamma) * den[i];
            for (size_t i = u + 1; i < n; ++i) T.lkZ[l][i] = rng.fr();
            tr.write_point(ck.commit_lagrange(T.lkZ[l]));
        }
    }
    for (size_t q = 0; q < M; ++q) {
        InstState& T = S[q];
        T.shA.resize(Sh); T.shS.resize(Sh); T.shZ.resize(Sh);
        for (size_t s2 = 0; s2 < Sh; ++s2) {
            T.shA[s2] = compress_rows(T.row_vars, cs.shuffles[s2].input);
            T.shS[s2] = compress_rows(T.row_vars, cs.shuffles[s2].shuffle);
            std::vector<Fr> den(u);
            for (size_t i = 0; i < u; ++i) den[i] = T.shS[s2][i] + gamma;
            batch_invert(den.data(), den.size());
            T.shZ[s2].assign(n, Fr::zero()); T.shZ[s2][0] = Fr::one();
            for (size_t i = 0; i < u; ++i) T.shZ[s2][i + 1] = T.shZ[s2][i] * (T.shA[s2][i] + gamma) * den[i];
            for (size_t i = u + 1; i < n; ++i) T.shZ[s2][i] = rng.fr();
            tr.write_point(ck.commit_lagrange(T.shZ[s2]));
        }
    }
    // vanishing: random polynomial
    std::vector<Fr> random_poly(n);
    for (auto& x : random_poly) x = rng.fr();
    tr.write_point(ck.commit_coeff(random_poly));
    Fr y = tr.squeeze_challenge();

    // coefficient forms and extended-coset evaluations
    auto to_coeff = [&](const std::vector<std::vector<Fr>>& cols) { std::vector<std::vector<Fr>> o; for (const auto& v : cols) o.push_back(lagrange_to_coeff(v, k)); return o; };
    auto to_ext = [&](const std::vector<std::vector<Fr>>& cf) { std::vector<std::vector<Fr>> o; for (const auto& v : cf) o.push_back(coeff_to_ext(v, ek)); return o; };
    for (size_t q = 0; q < M; ++q) {
        InstState& T = S[q];
        T.adv_c = to_coeff(T.advice); T.inst_c = to_coeff(T.inst); T.pz_c = to_coeff(T.permZ);
        T.lkAp_c = to_coeff(T.lkAp); T.lkSp_c = to_coeff(T.lkSp); T.lkZ_c = to_coeff(T.lkZ); T.shZ_c = to_coeff(T.shZ);
        T.adv_e = to_ext(T.adv_c); T.inst_e = to_ext(T.inst_c); T.pz_e = to_ext(T.pz_c);
        T.lkAp_e = to_ext(T.lkAp_c); T.lkSp_e = to_ext(T.lkSp_c); T.lkZ_e = to_ext(T.lkZ_c); T.shZ_e = to_ext(T.shZ_c);
        T.ext_vars = make_vars(T.adv_e, pk.fixed_ext, T.inst_e);
    }

    // numerator of h on the coset, folded with y in the verifier's expression order (lib.rs:273-346): instance by instance
    std::vector<Fr> hnum(m);
    {
        const Ntt& ne = ntt_ctx(ek);
        Fr zeta = Fr::from_u64(COSET_GEN), pt = zeta;
        std::vector<Fr> delta_pow(P ? P : 1); { Fr d = Fr::one(); for (auto& x : delta_pow) { x = d; d *= fr_consts().delta; } }
        const int64_t last_rot = -(int64_t)(bf + 1);
        for (size_t i = 0; i < m; ++i, pt *= ne.omega) {
            auto at = [&](const std::vector<Fr>& e, int64_t rot) -> const Fr& { return e[(i + m + rot * (int64_t)stride) % m]; };
            Fr acc = Fr::zero();
            auto push = [&](const Fr& v) { acc = acc * y + v; };
            const Fr &l0 = pk.l0_ext[i], &ll = pk.llast_ext[i], &act = pk.lactive_ext[i];
            for (size_t q = 0; q < M; ++q) {
                const InstState& T = S[q];
                auto perm_col_ext = [&](size_t j) -> const std::vector<Fr>& {
                    const Column& col = cs.permutation_columns[j];
                    if (col.is_advice()) return T.adv_e[col.index];
                    if (col.type == COL_FIXED) return pk.fixed_ext[col.index];
                    return T.inst_e[col.index];
                };
                for (const ExprPoly& g : cs.gates) push(eval_expr_at(g, cs.coeff_vals, T.ext_vars, i, m, stride));
                if (nsets > 0) {
                    push(l0 * (Fr::one() - T.pz_e[0][i]));
                    push(ll * (T.pz_e[nsets - 1][i].sqr() - T.pz_e[nsets - 1][i]));
                    for (size_t s2 = 1; s2 < nsets; ++s2) push(l0 * (T.pz_e[s2][i] - at(T.pz_e[s2 - 1], last_rot)));
                    for (size_t s2 = 0; s2 < nsets; ++s2) {
                        size_t lo = s2 * chunk, hi = std::min(P, lo + chunk);
                        Fr left = at(T.pz_e[s2], 1), right = T.pz_e[s2][i];
                        for (size_t j = lo; j < hi; ++j) {
                            const Fr& v = perm_col_ext(j)[i];
                            left *= v + beta * pk.sigma_ext[j][i] + gamma;
                            right *= v + beta * delta_pow[j] * pt + gamma;
                        }
                        push((left - right) * act);
                    }
                }
                for (size_t l = 0; l < L; ++l) {
                    Fr ca = Fr::zero(), ct = Fr::zero();
                    for (const ExprPoly& e : cs.lookups[l].input) ca = ca * theta + eval_expr_at(e, cs.coeff_vals, T.ext_vars, i, m, stride);
                    for (const ExprPoly& e : cs.lookups[l].table) ct = ct * theta + eval_expr_at(e, cs.coeff_vals, T.ext_vars, i, m, stride);
                    const Fr &z = T.lkZ_e[l][i], &ap = T.lkAp_e[l][i], &sp = T.lkSp_e[l][i];
                    push(l0 * (Fr::one() - z));
                    push(ll * (z.sqr() - z));
                    push((at(T.lkZ_e[l], 1) * (ap + beta) * (sp + gamma) - z * (ca + beta) * (ct + gamma)) * act);
                    push(l0 * (ap - sp));
                    push((ap - sp) * (ap - at(T.lkAp_e[l], -1)) * act);
                }
                for (size_t s2 = 0; s2 < Sh; ++s2) {
                    Fr ca = Fr::zero(), csh = Fr::zero();
                    for (const ExprPoly& e : cs.shuffles[s2].input) ca = ca * theta + eval_expr_at(e, cs.coeff_vals, T.ext_vars, i, m, stride);
                    for (const ExprPoly& e : cs.shuffles[s2].shuffle) csh = csh * theta + eval_expr_at(e, cs.coeff_vals, T.ext_vars, i, m, stride);
                    const Fr& z = T.shZ_e[s2][i];
                    push(l0 * (Fr::one() - z));
                    push(ll * (z.sqr() - z));
                    push((at(T.shZ_e[s2], 1) * (csh + gamma) - z * (ca + gamma)) * act);
                }
            }
            hnum[i] = acc;
        }
        // divide by t(X) = X^n - 1 on the coset: t repeats with period m/n
        std::vector<Fr> tinv(stride);
        u64 e4[4] = {n, 0, 0, 0};
        Fr zn = zeta.pow(e4), wn = ne.omega.pow(e4), cur = zn;
        for (size_t i = 0; i < stride; ++i) { tinv[i] = cur - Fr::one(); cur *= wn; }
        batch_invert(tinv.data(), tinv.size());
        for (size_t i = 0; i < m; ++i) hnum[i] *= tinv[i % stride];
        ne.run(hnum, true);
        Fr zi = zeta.inv(), p = Fr::one();
        for (size_t i = 0; i < m; ++i) { hnum[i] *= p; p *= zi; }
    }
    size_t H = c.cs_degree - 1;
    std::vector<std::vector<Fr>> h_pieces(H, std::vector<Fr>(n, Fr::zero()));
    for (size_t i = 0; i < H; ++i) for (size_t j = 0; j < n; ++j) if (i * n + j < m) h_pieces[i][j] = hnum[i * n + j];
    for (size_t i = 0; i < H; ++i) tr.write_point(ck.commit_coeff(h_pieces[i]));
    Fr x = tr.squeeze_challenge();

    // evaluations, in the order lib.rs:220-253 reads them
    auto rot_pt = [&](int64_t r) { return r >= 0 ? x * omega.pow_u64((u64)r) : x * omega_inv.pow_u64((u64)(-r)); };
    std::vector<PolyQuery> queries;  // in the verifier's query order (lib.rs:349-414), built below
    std::vector<Fr> fix_evals, sig_evals;
    for (size_t q = 0; q < M; ++q)
        for (const Query& qq : cs.advice_queries) { Fr e = horner(S[q].adv_c[qq.column.index], rot_pt(qq.rotation)); S[q].adv_evals.push_back(e); tr.write_scalar(e); }
    for (const Query& q : cs.fixed_queries) { Fr e = horner(pk.fixed_coeff[q.column.index], rot_pt(q.rotation)); fix_evals.push_back(e); tr.write_scalar(e); }
    Fr random_eval = horner(random_poly, x); tr.write_scalar(random_eval);
    for (size_t j = 0; j < P; ++j) { Fr e = horner(pk.sigma_coeff[j], x); sig_evals.push_back(e); tr.write_scalar(e); }
    Fr x_next = rot_pt(1), x_prev = rot_pt(-1), x_last = rot_pt(-(int64_t)(bf + 1));
    for (size_t q = 0; q < M; ++q) {
        InstState& T = S[q];
        T.pz_ev.resize(nsets);
        for (size_t s2 = 0; s2 < nsets; ++s2) {
            T.pz_ev[s2][0] = horner(T.pz_c[s2], x); tr.write_scalar(T.pz_ev[s2][0]);
            T.pz_ev[s2][1] = horner(T.pz_c[s2], x_next); tr.write_scalar(T.pz_ev[s2][1]);
            if (s2 + 1 < nsets) { T.pz_ev[s2][2] = horner(T.pz_c[s2], x_last); tr.write_scalar(T.pz_ev[s2][2]); }
        }
    }
    for (size_t q = 0; q < M; ++q) {
        InstState& T = S[q];
        T.lk_ev.resize(L);
        for (size_t l = 0; l < L; ++l) {
            T.lk_ev[l] = {horner(T.lkZ_c[l], x), horner(T.lkZ_c[l], x_next), horner(T.lkAp_c[l], x), horner(T.lkAp_c[l], x_prev), horner(T.lkSp_c[l], x)};
            for (const Fr& e : T.lk_ev[l]) tr.write_scalar(e);
        }
    }
    for (size_t q = 0; q < M; ++q) {
        InstState& T = S[q];
        T.sh_ev.resize(Sh);
        for (size_t s2 = 0; s2 < Sh; ++s2) { T.sh_ev[s2] = {horner(T.shZ_c[s2], x), horner(T.shZ_c[s2], x_next)}; for (const Fr& e : T.sh_ev[s2]) tr.write_scalar(e); }
    }

    // combined quotient polynomial  sum_i xn^i h_i(X)  (vanishing.rs:102-112)
    u64 e4[4] = {n, 0, 0, 0};
    Fr xn = x.pow(e4);
    std::vector<Fr> h_comb(n, Fr::zero());
    for (size_t i = H; i-- > 0;) for (size_t j = 0; j < n; ++j) h_comb[j] = h_comb[j] * xn + h_pieces[i][j];
    Fr h_eval = horner(h_comb, x);

    for (size_t q = 0; q < M; ++q) {
        const InstState& T = S[q];
        for (size_t qi = 0; qi < cs.advice_queries.size(); ++qi) queries.push_back({&T.adv_c[cs.advice_queries[qi].column.index], rot_pt(cs.advice_queries[qi].rotation), T.adv_evals[qi]});
        for (size_t s2 = 0; s2 < nsets; ++s2) { queries.push_back({&T.pz_c[s2], x, T.pz_ev[s2][0]}); queries.push_back({&T.pz_c[s2], x_next, T.pz_ev[s2][1]}); }
        for (size_t s2 = nsets; s2-- > 0;) { if (s2 + 1 == nsets) continue; queries.push_back({&T.pz_c[s2], x_last, T.pz_ev[s2][2]}); }
        for (size_t l = 0; l < L; ++l) {
            queries.push_back({&T.lkZ_c[l], x, T.lk_ev[l][0]}); queries.push_back({&T.lkAp_c[l], x, T.lk_ev[l][2]}); queries.push_back({&T.lkSp_c[l], x, T.lk_ev[l][4]});
            queries.push_back({&T.lkAp_c[l], x_prev, T.lk_ev[l][3]}); queries.push_back({&T.lkZ_c[l], x_next, T.lk_ev[l][1]});
        }
        for (size_t s2 = 0; s2 < Sh; ++s2) { queries.push_back({&T.shZ_c[s2], x, T.sh_ev[s2][0]}); queries.push_back({&T.shZ_c[s2], x_next, T.sh_ev[s2][1]}); }
    }
    for (size_t qi = 0; qi < cs.fixed_queries.size(); ++qi) queries.push_back({&pk.fixed_coeff[cs.fixed_queries[qi].column.index], rot_pt(cs.fixed_queries[qi].rotation), fix_evals[qi]});
    for (size_t j = 0; j < P; ++j) queries.push_back({&pk.sigma_coeff[j], x, sig_evals[j]});
    queries.push_back({&h_comb, x, h_eval});
    queries.push_back({&random_poly, x, random_eval});

    if (multiopen == 1) {
        // GWC opening (gwc.rs:54-135): one witness W_i(X) = sum_j v^j (p_j(X) - e_j) / (X - z_i) per distinct point z_i
        Fr gv = tr.squeeze_challenge();
        std::vector<std::pair<Fr, std::vector<const PolyQuery*>>> by_point;
        for (const PolyQuery& q : queries) {
            bool found = false;
            for (auto& e : by_point) if (e.first == q.point) { e.second.push_back(&q); found = true; break; }
            if (!found) by_point.push_back({q.point, {&q}});
        }
        for (auto& e : by_point) {
            std::vector<Fr> acc(n, Fr::zero());
            Fr pv = Fr::one();
            for (const PolyQuery* q : e.second) {
                for (size_t t = 0; t < n; ++t) acc[t] += pv * (*q->poly)[t];
                acc[0] -= pv * q->eval;
                pv *= gv;
            }
            divide_by_linear(acc, e.first);
            tr.write_point(ck.commit_coeff(acc));
        }
        tr.squeeze_challenge();  // u: the prover sends nothing after it
        return tr.out;
    }
    // SHPLONK opening
    struct PolyPts { const std::vector<Fr>* poly; std::vector<Fr> pts; };
    std::vector<PolyPts> by_poly; std::vector<Fr> super;
    auto insert_sorted = [](std::vector<Fr>& v, const Fr& p) { auto it = std::lower_bound(v.begin(), v.end(), p, FrLess()); if (it == v.end() || !(*it == p)) v.insert(it, p); };
    for (const PolyQuery& q : queries) {
        insert_sorted(super, q.point);
        auto it = std::find_if(by_poly.begin(), by_poly.end(), [&](const PolyPts& e) { return e.poly == q.poly; });
        if (it == by_poly.end()) { by_poly.push_back({q.poly, {}}); it = by_poly.end() - 1; }
        insert_sorted(it->pts, q.point);
    }
    struct RSet { std::vector<Fr> pts; std::vector<const std::vector<Fr>*> polys; };
    std::vector<RSet> rsets;
    for (const PolyPts& e : by_poly) {
        auto it = std::find_if(rsets.begin(), rsets.end(), [&](const RSet& r) { return r.pts == e.pts; });
        if (it == rsets.end()) { rsets.push_back({e.pts, {}}); it = rsets.end() - 1; }
        it->polys.push_back(e.poly);
    }
    auto eval_of = [&](const std::vector<Fr>* poly, const Fr& pt) { for (const PolyQuery& q : queries) if (q.poly == poly && q.point == pt) return q.eval; return Fr::zero(); };

    Fr sy = tr.squeeze_challenge();
    Fr sv = tr.squeeze_challenge();
    std::vector<Fr> f_poly(n, Fr::zero());
    std::vector<std::vector<Fr>> set_comb(rsets.size());            // sum_j y^j p_ij(X)
    std::vector<std::vector<Fr>> set_r(rsets.size());               // sum_j y^j r_ij(X)
    {
        Fr vp = Fr::one();
        for (size_t i = 0; i < rsets.size(); ++i, vp *= sv) {
            std::vector<Fr> comb(n, Fr::zero()), rcomb(rsets[i].pts.size(), Fr::zero());
            Fr yp = Fr::one();
            for (const auto* poly : rsets[i].polys) {
                for (size_t t = 0; t < n; ++t) comb[t] += yp * (*poly)[t];
                std::vector<Fr> vals; for (const Fr& p : rsets[i].pts) vals.push_back(eval_of(poly, p));
                std::vector<Fr> r = interpolate(rsets[i].pts, vals);
                for (size_t t = 0; t < r.size(); ++t) rcomb[t] += yp * r[t];
                yp *= sy;
            }
            set_comb[i] = comb; set_r[i] = rcomb;
            std::vector<Fr> num = comb;
            for (size_t t = 0; t < rcomb.size(); ++t) num[t] -= rcomb[t];
            for (const Fr& p : rsets[i].pts) divide_by_linear(num, p);
            for (size_t t = 0; t < n; ++t) f_poly[t] += vp * num[t];
        }
    }
    tr.write_point(ck.commit_coeff(f_poly));
    Fr su = tr.squeeze_challenge();
    {
        auto vanish = [&](const std::vector<Fr>& roots) { Fr a = Fr::one(); for (const Fr& r : roots) a *= su - r; return a; };
        auto complement = [&](const std::vector<Fr>& pts) { std::vector<Fr> o; for (const Fr& p : super) if (std::find(pts.begin(), pts.end(), p) == pts.end()) o.push_back(p); return o; };
        Fr zdiff0_inv = vanish(complement(rsets[0].pts)).inv();
        Fr z0 = vanish(rsets[0].pts);
        std::vector<Fr> lpoly(n, Fr::zero());
        Fr vp = Fr::one();
        for (size_t i = 0; i < rsets.size(); ++i, vp *= sv) {
            Fr zhat = i == 0 ? Fr::one() : vanish(complement(rsets[i].pts)) * zdiff0_inv;
            Fr w = vp * zhat;
            for (size_t t = 0; t < n; ++t) lpoly[t] += w * set_comb[i][t];
            lpoly[0] -= w * horner(set_r[i], su);
        }
        for (size_t t = 0; t < n; ++t) lpoly[t] -= z0 * f_poly[t];
        divide_by_linear(lpoly, su);
        tr.write_point(ck.commit_coeff(lpoly));
    }
    return tr.out;
}

// ------------------------------------------------------------------------------ circuits
static Column adv_col(uint32_t i, uint8_t phase = 0) { return {i, phase}; }
static ExprTerm term(uint16_t c, std::initializer_list<std::pair<uint32_t, uint32_t>> f) { ExprTerm t; t.coeff_idx = c; t.factors.assign(f.begin(), f.end()); return t; }

Circuit circuit_vector_mul(uint32_t k, size_t n_mul) {
    Circuit c; c.k = k; c.cs_degree = 3;
    ConstraintSystem& cs = c.cs;
    cs.num_fixed_columns = 1; cs.num_advice_columns = 3; cs.num_instance_columns = 1; cs.num_selectors = 1; cs.num_challenges = 0;
    cs.advice_column_phase = {0, 0, 0};
    cs.num_advice_queries = {1, 1, 1};
    for (uint32_t i = 0; i < 3; ++i) cs.advice_queries.push_back({adv_col(i), 0});
    cs.instance_queries.push_back({{0, COL_INSTANCE}, 0});
    cs.fixed_queries.push_back({{0, COL_FIXED}, 0});
    cs.permutation_columns = {{0, COL_INSTANCE}, adv_col(0), adv_col(1), adv_col(2)};
    cs.coeff_vals = {Fr::one(), Fr::one().neg()};
    // s * (a*b - c): variables a0=0, a1=1, a2=2, f0=3, i0=4  (plonk/vk.rs:486-508 index space)
    ExprPoly g; g.num_vars = 5;
    g.terms.push_back(term(0, {{0, 1}, {1, 1}, {3, 1}}));
    g.terms.push_back(term(1, {{2, 1}, {3, 1}}));
    cs.gates.push_back(g);
    size_t n = c.n();
    c.fixed.assign(1, std::vector<Fr>(n, Fr::zero()));
    // rows [0, n_mul): load a (col 0); rows [n_mul, 2 n_mul): load b (col 0); rows [2 n_mul, 3 n_mul): mul rows
    for (size_t i = 0; i < n_mul; ++i) {
        size_t r = 2 * n_mul + i;
        c.fixed[0][r] = Fr::one();
        c.copies.push_back({1, (uint32_t)i, 1, (uint32_t)r});              // a0[i]        == a0[mul row]
        c.copies.push_back({1, (uint32_t)(n_mul + i), 2, (uint32_t)r});    // a0[n_mul+i]  == a1[mul row]
        c.copies.push_back({3, (uint32_t)r, 0, (uint32_t)i});              // a2[mul row]  == instance[i]
    }
    return c;
}
WitnessFn witness_vector_mul(const std::vector<Fr>& a, const std::vector<Fr>& b) {
    return [a, b](unsigned phase, const std::vector<Fr>&, std::vector<std::vector<Fr>>& adv) {
        if (phase != 0) return;
        size_t m = a.size();
        for (size_t i = 0; i < m; ++i) {
            adv[0][i] = a[i]; adv[0][m + i] = b[i];
            adv[0][2 * m + i] = a[i]; adv[1][2 * m + i] = b[i]; adv[2][2 * m + i] = a[i] * b[i];
        }
    };
}

Circuit circuit_two_phase_shuffle(uint32_t k, size_t W, size_t Hrows) {
    Circuit c; c.k = k; c.cs_degree = 3;
    ConstraintSystem& cs = c.cs;
    uint32_t A = (uint32_t)(2 * W + 1), zc = (uint32_t)(2 * W);
    cs.num_fixed_columns = 3; cs.num_advice_columns = A; cs.num_instance_columns = 0; cs.num_selectors = 3; cs.num_challenges = 2;
    cs.advice_column_phase.assign(A, 0); cs.advice_column_phase[zc] = 1;
    cs.challenge_phase = {0, 0};
    cs.num_advice_queries.assign(A, 1); cs.num_advice_queries[zc] = 2;
    for (uint32_t i = 0; i < zc; ++i) cs.advice_queries.push_back({adv_col(i, 0), 0});
    cs.advice_queries.push_back({adv_col(zc, 1), 0});
    cs.advice_queries.push_back({adv_col(zc, 1), 1});
    for (uint32_t i = 0; i < 3; ++i) cs.fixed_queries.push_back({{i, COL_FIXED}, 0});
    // variables: advice queries 0..2W-1 (orig, shuf), z = 2W, z_w = 2W+1; fixed q_shuffle = 2W+2, q_first = 2W+3, q_last = 2W+4;
    // challenges theta = 2W+5, gamma = 2W+6
    uint32_t vz = zc, vzw = zc + 1, qs = zc + 2, qf = zc + 3, ql = zc + 4, th = zc + 5, ga = zc + 6;
    cs.coeff_vals = {Fr::one(), Fr::one().neg()};
    ExprPoly g1; g1.num_vars = ga + 1; g1.terms = {term(0, {{qf, 1}}), term(1, {{qf, 1}, {vz, 1}})};
    ExprPoly g2; g2.num_vars = ga + 1; g2.terms = {term(0, {{ql, 1}}), term(1, {{ql, 1}, {vz, 1}})};
    ExprPoly g3; g3.num_vars = ga + 1;
    for (uint32_t w = 0; w < W; ++w) {   // q * z * o_w * theta^(W-1-w)   and   - q * z_w * s_w * theta^(W-1-w)
        uint32_t pw = (uint32_t)(W - 1 - w);
        ExprTerm t1 = term(0, {{qs, 1}, {vz, 1}, {w, 1}}), t2 = term(1, {{qs, 1}, {vzw, 1}, {(uint32_t)(W + w), 1}});
        if (pw) { t1.factors.push_back({th, pw}); t2.factors.push_back({th, pw}); }
        g3.terms.push_back(t1); g3.terms.push_back(t2);
    }
    g3.terms.push_back(term(0, {{qs, 1}, {vz, 1}, {ga, 1}}));
    g3.terms.push_back(term(1, {{qs, 1}, {vzw, 1}, {ga, 1}}));
    cs.gates = {g1, g2, g3};
    size_t n = c.n();
    c.fixed.assign(3, std::vector<Fr>(n, Fr::zero()));
    for (size_t i = 0; i < Hrows; ++i) c.fixed[0][i] = Fr::one();
    c.fixed[1][0] = Fr::one();
    c.fixed[2][Hrows] = Fr::one();
    return c;
}
WitnessFn witness_two_phase_shuffle(const std::vector<std::vector<Fr>>& original, const std::vector<std::vector<Fr>>& shuffled) {
    return [original, shuffled](unsigned phase, const std::vector<Fr>& ch, std::vector<std::vector<Fr>>& adv) {
        size_t W = original.size(), Hn = original[0].size();
        if (phase == 0) {
            for (size_t w = 0; w < W; ++w) for (size_t i = 0; i < Hn; ++i) { adv[w][i] = original[w][i]; adv[W + w][i] = shuffled[w][i]; }
            return;
        }
        const Fr &theta = ch[0], &gamma = ch[1];
        std::vector<Fr>& z = adv[2 * W];
        z[0] = Fr::one();
        for (size_t i = 0; i < Hn; ++i) {
            Fr co = Fr::zero(), csf = Fr::zero();
            for (size_t w = 0; w < W; ++w) { co = co * theta + original[w][i]; csf = csf * theta + shuffled[w][i]; }
            z[i + 1] = z[i] * (co + gamma) * (csf + gamma).inv();
        }
    };
}

Circuit circuit_wide(uint32_t k, size_t A, size_t F, size_t L, size_t Sh, uint32_t deg, u64 seed) {
    Circuit c; c.k = k;
    ConstraintSystem& cs = c.cs;
    size_t blocks = A / 4;
    if (L + Sh > blocks || F < 3 + blocks || deg < 3) throw std::runtime_error("circuit_wide: need A/4 >= L+Sh, F >= 3 + A/4, deg >= 3");
    c.cs_degree = std::max<uint32_t>(deg, L > 0 ? 4 : 3);
    cs.num_fixed_columns = (uint32_t)F; cs.num_advice_columns = (uint32_t)A; cs.num_instance_columns = 1; cs.num_selectors = 1; cs.num_challenges = 0;
    cs.advice_column_phase.assign(A, 0);
    cs.num_advice_queries.assign(A, 1);
    // advice queries: every column at 0; block base columns (4j) additionally at -1 and +1
    std::vector<uint32_t> q0(A), qm(blocks), qp(blocks);
    for (uint32_t i = 0; i < A; ++i) { q0[i] = (uint32_t)cs.advice_queries.size(); cs.advice_queries.push_back({adv_col(i), 0}); }
    for (uint32_t j = 0; j < blocks; ++j) {
        qm[j] = (uint32_t)cs.advice_queries.size(); cs.advice_queries.push_back({adv_col(4 * j), -1});
        qp[j] = (uint32_t)cs.advice_queries.size(); cs.advice_queries.push_back({adv_col(4 * j), 1});
        cs.num_advice_queries[4 * j] = 3;
    }
    uint32_t nq = (uint32_t)cs.advice_queries.size();
    for (uint32_t i = 0; i < F; ++i) cs.fixed_queries.push_back({{i, COL_FIXED}, 0});
    cs.instance_queries.push_back({{0, COL_INSTANCE}, 0});
    auto fvar = [&](uint32_t f) { return nq + f; };
    uint32_t nvars = nq + (uint32_t)F + 1;
    cs.coeff_vals = {Fr::one(), Fr::one().neg()};
    // fixed: f0 = q (gate selector), f1/f2 = lookup table columns, f3+j = per-block constant column
    for (uint32_t j = 0; j < blocks; ++j) {
        uint32_t b = 4 * j;
        // q * ( a_b * a_{b+1}^(deg-2) + c_j * a_b(wX) - a_b(w^-1 X) - a_{b+2} )
        ExprPoly g; g.num_vars = nvars;
        g.terms.push_back(term(0, {{fvar(0), 1}, {q0[b], 1}, {q0[b + 1], deg - 2}}));
        g.terms.push_back(term(0, {{fvar(0), 1}, {fvar(3 + j), 1}, {qp[j], 1}}));
        g.terms.push_back(term(1, {{fvar(0), 1}, {qm[j], 1}}));
        g.terms.push_back(term(1, {{fvar(0), 1}, {q0[b + 2], 1}}));
        cs.gates.push_back(g);
    }
    auto single = [&](uint32_t var) { ExprPoly e; e.num_vars = nvars; e.terms.push_back(term(0, {{var, 1}})); return e; };
    for (uint32_t l = 0; l < L; ++l) {
        LookupArg a; a.input = {single(q0[4 * l + 1]), single(q0[4 * l + 3])}; a.table = {single(fvar(1)), single(fvar(2))};
        cs.lookups.push_back(a);
    }
    for (uint32_t s = 0; s < Sh; ++s) {
        uint32_t j = (uint32_t)L + s;
        ShuffleArg a; a.input = {single(q0[4 * j + 1])}; a.shuffle = {single(q0[4 * j + 3])};
        cs.shuffles.push_back(a);
    }
    cs.permutation_columns.push_back({0, COL_INSTANCE});
    for (uint32_t i = 0; i < A; ++i) cs.permutation_columns.push_back(adv_col(i));
    size_t n = c.n(), u = c.usable_rows();
    Rng rng(seed);
    c.fixed.assign(F, std::vector<Fr>(n, Fr::zero()));
    for (size_t i = 1; i + 1 < u; ++i) c.fixed[0][i] = Fr::one();
    for (size_t i = 0; i < u; ++i) { c.fixed[1][i] = Fr::from_u64(i); c.fixed[2][i] = Fr::from_u64(i * i + 7); }
    for (size_t f = 3; f < F; ++f) for (size_t i = 0; i < u; ++i) c.fixed[f][i] = rng.fr();
    // copies: rows 2,3 of column 4j+1 are made equal by the witness; instance[i] == a_{2}[i+1] for 8 public inputs
    for (uint32_t j = 0; j < blocks; ++j) c.copies.push_back({1 + 4 * j + 1, 2, 1 + 4 * j + 1, 3});
    for (uint32_t i = 0; i < 8; ++i) c.copies.push_back({0, i, 1 + 2, i + 1});
    return c;
}
WitnessFn witness_wide(const Circuit& c, u64 seed) {
    return [c, seed](unsigned phase, const std::vector<Fr>&, std::vector<std::vector<Fr>>& adv) {
        if (phase != 0) return;
        const ConstraintSystem& cs = c.cs;
        size_t A = cs.num_advice_columns, blocks = A / 4, L = cs.lookups.size(), Sh = cs.shuffles.size(), n = c.n(), u = c.usable_rows();
        uint32_t deg = 0;
        for (const auto& f : cs.gates[0].terms[0].factors) if (f.second > deg) deg = f.second;
        deg += 2;
        Rng rng(seed);
        for (size_t j = 0; j < blocks; ++j) {
            size_t b = 4 * j;
            for (size_t i = 0; i < n; ++i) { adv[b][i] = rng.fr(); adv[b + 1][i] = rng.fr(); adv[b + 3][i] = rng.fr(); }
            if (j < L) for (size_t i = 0; i < u; ++i) { size_t r = (i == 3) ? 0 : rng.next() % u; if (i == 2) r = 0; adv[b + 1][i] = c.fixed[1][r]; adv[b + 3][i] = c.fixed[2][r]; }
            else if (j < L + Sh) {
                adv[b + 1][3] = adv[b + 1][2];
                std::vector<size_t> perm(u); for (size_t i = 0; i < u; ++i) perm[i] = i;
                for (size_t i = u - 1; i > 0; --i) std::swap(perm[i], perm[rng.next() % (i + 1)]);
                for (size_t i = 0; i < u; ++i) adv[b + 3][i] = adv[b + 1][perm[i]];
            } else adv[b + 1][3] = adv[b + 1][2];
            for (size_t i = 1; i + 1 < u; ++i)
                adv[b + 2][i] = adv[b][i] * adv[b + 1][i].pow_u64(deg - 2) + c.fixed[3 + j][i] * adv[b][i + 1] - adv[b][i - 1];
        }
        for (size_t col = 4 * blocks; col < A; ++col) for (size_t i = 0; i < u; ++i) adv[col][i] = rng.fr();
    };
}

}  // namespace h2o
