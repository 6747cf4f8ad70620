// ORACLE — TEST INFRASTRUCTURE ONLY (see bn254_fp.hpp / verifier.hpp headers).
#include "verifier.hpp"
#include <algorithm>

namespace h2o {

// ------------------------------------------------------------------ arithmetic.rs:7-108
static inline size_t get_at(size_t segment, size_t c, const uint8_t bytes[32]) {
    size_t skip_bits = segment * c, skip_bytes = skip_bits / 8;
    if (skip_bytes >= 32) return 0;
    u64 tmp = 0;
    for (size_t i = 0; i < 8 && skip_bytes + i < 32; ++i) tmp |= (u64)bytes[skip_bytes + i] << (8 * i);
    tmp >>= skip_bits - skip_bytes * 8;
    return (size_t)(tmp % (1ULL << c));
}

G1 best_multiexp(const Fr* coeffs, const G1Affine* bases, size_t n) {
    std::vector<uint8_t> repr(32 * n);
    for (size_t i = 0; i < n; ++i) coeffs[i].to_bytes(&repr[32 * i]);
    size_t c = n < 4 ? 1 : (n < 32 ? 3 : 4);
    size_t segments = 256 / c + 1;
    G1 acc = G1::identity();
    std::vector<G1> buckets((1u << c) - 1);
    for (size_t seg = segments; seg-- > 0;) {
        for (size_t i = 0; i < c; ++i) acc = acc.dbl();
        for (auto& b : buckets) b = G1::identity();
        for (size_t i = 0; i < n; ++i) {
            size_t d = get_at(seg, c, &repr[32 * i]);
            if (d != 0) buckets[d - 1] = buckets[d - 1].add_affine(bases[i]);
        }
        G1 running = G1::identity();
        for (size_t b = buckets.size(); b-- > 0;) {
            running = running.add(buckets[b]);
            acc = acc.add(running);
        }
    }
    return acc;
}

// poly/kzg/msm.rs:81-86
G1 MSMKZG::eval() const {
    std::vector<G1Affine> aff(bases.size());
    g1_batch_normalize(bases.data(), aff.data(), bases.size());
    return best_multiexp(scalars.data(), aff.data(), scalars.size());
}

// poly/kzg/msm.rs:185-203
bool DualMSM::check(const ParamsKZG& params) const {
    G2Prepared s_g2(params.s_g2), n_g2(params.g2.neg());
    G1Affine ps[2] = {left.eval().to_affine(), right.eval().to_affine()};
    const G2Prepared* qs[2] = {&s_g2, &n_g2};
    return final_exponentiation(multi_miller_loop(ps, qs, 2)).is_one();
}

// ------------------------------------------------------------------ arithmetic.rs:137-210
static Fr eval_polynomial(const std::vector<Fr>& poly, const Fr& point) {
    Fr acc = Fr::zero();
    for (size_t i = poly.size(); i-- > 0;) acc = acc * point + poly[i];
    return acc;
}
static std::vector<Fr> lagrange_interpolate(const std::vector<Fr>& points, const std::vector<Fr>& evals) {
    size_t n = points.size();
    if (n == 1) return {evals[0]};
    std::vector<Fr> denoms;  // denoms[j][k'] flattened, k != j in order
    for (size_t j = 0; j < n; ++j)
        for (size_t k = 0; k < n; ++k) if (k != j) denoms.push_back(points[j] - points[k]);
    batch_invert(denoms.data(), denoms.size());
    std::vector<Fr> final_poly(n, Fr::zero());
    size_t di = 0;
    for (size_t j = 0; j < n; ++j) {
        std::vector<Fr> tmp = {Fr::one()}, product;
        for (size_t k = 0; k < n; ++k) {
            if (k == j) continue;
            const Fr& denom = denoms[di++];
            product.assign(tmp.size() + 1, Fr::zero());
            Fr c0 = (denom * points[k]).neg();  // -denom * x_k
            for (size_t t = 0; t <= tmp.size(); ++t) {
                Fr a = t < tmp.size() ? tmp[t] : Fr::zero();
                Fr b = t > 0 ? tmp[t - 1] : Fr::zero();
                product[t] = a * c0 + b * denom;
            }
            tmp.swap(product);
        }
        for (size_t t = 0; t < n; ++t) final_poly[t] += tmp[t] * evals[j];
    }
    return final_poly;
}
static Fr evaluate_vanishing_polynomial(const std::vector<Fr>& roots, const Fr& z) {
    Fr acc = Fr::one();
    for (const Fr& r : roots) acc = (z - r) * acc;
    return acc;
}

// ------------------------------------------------------------------ poly/domain.rs
Domain::Domain(uint32_t j, uint32_t k_) : k(k_), n(1ULL << k_) {
    quotient_poly_degree = j - 1;
    Fr w = fr_consts().root_of_unity;
    for (uint32_t i = k; i < (uint32_t)FrConsts::S; ++i) w = w.sqr();
    omega = w;
    omega_inv = w.inv();
    barycentric_weight = Fr::from_u64(n).inv();
}
Fr Domain::rotate_omega(const Fr& value, int32_t rotation) const {
    if (rotation >= 0) return value * omega.pow_u64((u64)rotation);
    return value * omega_inv.pow_u64((u64)(-(int64_t)rotation));
}
std::vector<Fr> Domain::l_i_range(const Fr& x, const Fr& xn, int32_t from, int32_t to_exclusive) const {
    std::vector<Fr> results;
    for (int32_t r = from; r < to_exclusive; ++r) results.push_back(x - rotate_omega(Fr::one(), r));
    batch_invert(results.data(), results.size());
    Fr common = (xn - Fr::one()) * barycentric_weight;
    size_t i = 0;
    for (int32_t r = from; r < to_exclusive; ++r, ++i) results[i] = rotate_omega(results[i] * common, r);
    return results;
}

// ------------------------------------------------------------------ plonk/vk.rs:478-512,579-586
struct PanicEquiv { const char* what; };

Fr eval_expr(const ExprPoly& poly, const std::vector<Fr>& coeffs, const std::vector<Fr>& advice, const std::vector<Fr>& fixed,
             const std::vector<Fr>& instance, const std::vector<Fr>& challenges) {
    if (poly.terms.empty()) throw PanicEquiv{"called `Option::unwrap()` on a `None` value (multilinear.rs:65)"};
    size_t ar = advice.size(), fr_ = ar + fixed.size(), ir = fr_ + instance.size(), cr = ir + challenges.size();
    Fr result = Fr::zero();
    bool first = true;
    for (const ExprTerm& t : poly.terms) {
        if (t.coeff_idx >= coeffs.size()) throw PanicEquiv{"index out of bounds (vk.rs:490)"};
        Fr prod = Fr::one();
        for (const auto& f : t.factors) {
            size_t idx = f.first;
            Fr var;
            if (idx < ar) var = advice[idx];
            else if (idx < fr_) var = fixed[idx - ar];
            else if (idx < ir) var = instance[idx - fr_];
            else if (idx < cr) var = challenges[idx - ir];
            else throw PanicEquiv{"index out of range (vk.rs:501)"};
            prod = prod * var.pow_u64(f.second);
        }
        Fr term = coeffs[t.coeff_idx] * prod;
        result = first ? term : result + term;
        first = false;
    }
    return result;
}

// ------------------------------------------------------------------ queries (poly/query.rs)
namespace {
enum CommitKind { K_ADVICE, K_PERM_PRODUCT, K_LOOKUP, K_SHUFFLE, K_FIXED, K_PERM_COMMON, K_H_MSM, K_RANDOM };
struct CommitRef {
    int kind, idx, inst;   // inst: which circuit instance's commitment (advice / permutation product / lookup / shuffle); 0 for VK-wide ones
    bool operator==(const CommitRef& o) const { return kind == o.kind && idx == o.idx && inst == o.inst; }  // pointer identity, query.rs:63-74
};
struct VQuery { CommitRef c; Fr point, eval; };

struct FrLess { bool operator()(const Fr& a, const Fr& b) const { return Fr::cmp(a, b) < 0; } };

// sorted, de-duplicated point set (BTreeSet<Fr>)
struct PointSet {
    std::vector<Fr> pts;
    void insert(const Fr& p) {
        auto it = std::lower_bound(pts.begin(), pts.end(), p, FrLess());
        if (it != pts.end() && *it == p) return;
        pts.insert(it, p);
    }
    bool contains(const Fr& p) const { return std::binary_search(pts.begin(), pts.end(), p, FrLess()); }
    bool operator==(const PointSet& o) const { return pts == o.pts; }
};
struct RotCommitment { CommitRef c; std::vector<Fr> evals; };
struct RotationSet { std::vector<RotCommitment> commitments; std::vector<Fr> points; };
}  // namespace

struct OpeningError { const char* what; };

// shplonk.rs:58-149
static void construct_intermediate_sets(const std::vector<VQuery>& queries, std::vector<RotationSet>& rotation_sets, PointSet& super) {
    std::vector<std::pair<CommitRef, PointSet>> cmap;
    for (const VQuery& q : queries) {
        super.insert(q.point);
        bool found = false;
        for (auto& e : cmap) if (e.first == q.c) { e.second.insert(q.point); found = true; break; }
        if (!found) { PointSet s; s.insert(q.point); cmap.push_back({q.c, s}); }
    }
    std::vector<std::pair<PointSet, std::vector<CommitRef>>> rmap;
    for (auto& e : cmap) {
        bool found = false;
        for (auto& r : rmap) if (r.first == e.second) { r.second.push_back(e.first); found = true; break; }
        if (!found) rmap.push_back({e.second, {e.first}});
    }
    for (auto& r : rmap) {
        RotationSet rs; rs.points = r.first.pts;
        for (const CommitRef& c : r.second) {
            RotCommitment rc; rc.c = c;
            for (const Fr& p : rs.points) {
                const VQuery* hit = nullptr;
                for (const VQuery& q : queries) if (q.c == c && q.point == p) { hit = &q; break; }
                rc.evals.push_back(hit->eval);
            }
            rs.commitments.push_back(rc);
        }
        rotation_sets.push_back(rs);
    }
}

// ------------------------------------------------------------------ lib.rs:33-425
// `insts` = the reference's `instances: &[&[&[Fr]]]`: one entry per circuit instance sharing this transcript (lib.rs:51-55).
Error verify_proof_multi(const ParamsKZG& params, const VerifyingKey& vk, const std::vector<std::vector<std::vector<Fr>>>& insts,
                         const uint8_t* proof, size_t proof_len, DualMSM& acc, VerifyTrace* trace, const char** err_msg, VerifyOptions opts) {
    const ConstraintSystem& cs = vk.cs;
    for (const auto& instances : insts) if (instances.size() != cs.num_instance_columns) return InvalidInstances;  // lib.rs:51-55
    const size_t M = insts.size();   // num_proofs (lib.rs:63)
    Domain domain(vk.cs_degree, vk.k);
    TranscriptRead tr(proof, proof_len, opts.transcript);
    bool in_opening = false;
    try {
        tr.common_scalar(vk.transcript_repr);                                   // vk.rs:145-152
        for (const auto& instances : insts) for (const auto& col : instances) for (const Fr& v : col) tr.common_scalar(v);  // lib.rs:76-82

        // lib.rs:86-112: per phase, the advice commitments of EVERY instance, then the phase's challenges
        std::vector<std::vector<G1Affine>> advice_commitments(M, std::vector<G1Affine>(cs.num_advice_columns, G1Affine::identity()));
        std::vector<Fr> challenges(cs.num_challenges, Fr::zero());
        for (unsigned phase = 0; phase <= cs.max_phase(); ++phase) {
            for (size_t m = 0; m < M; ++m)
                for (size_t i = 0; i < cs.num_advice_columns; ++i)
                    if (cs.advice_column_phase[i] == phase) advice_commitments[m][i] = tr.read_point();
            for (size_t i = 0; i < cs.num_challenges; ++i)
                if (cs.challenge_phase[i] == phase) challenges[i] = tr.squeeze_challenge();
        }
        Fr theta = tr.squeeze_challenge();                                       // lib.rs:115
        size_t L = cs.lookups.size(), Sh = cs.shuffles.size();
        std::vector<std::vector<G1Affine>> lk_input(M, std::vector<G1Affine>(L)), lk_table(M, std::vector<G1Affine>(L)), lk_product(M, std::vector<G1Affine>(L)),
            sh_product(M, std::vector<G1Affine>(Sh));
        for (size_t m = 0; m < M; ++m)
            for (size_t i = 0; i < L; ++i) { lk_input[m][i] = tr.read_point(); lk_table[m][i] = tr.read_point(); }  // lib.rs:117-126, lookup.rs:82-97
        Fr beta = tr.squeeze_challenge();
        Fr gamma = tr.squeeze_challenge();
        size_t chunk_len = vk.cs_degree - 2;                                     // permutation.rs:72
        size_t P = cs.permutation_columns.size();
        size_t nsets = P == 0 ? 0 : (P + chunk_len - 1) / chunk_len;
        std::vector<std::vector<G1Affine>> perm_product(M, std::vector<G1Affine>(nsets));
        for (size_t m = 0; m < M; ++m) for (size_t i = 0; i < nsets; ++i) perm_product[m][i] = tr.read_point();   // lib.rs:134-139
        for (size_t m = 0; m < M; ++m) for (size_t i = 0; i < L; ++i) lk_product[m][i] = tr.read_point();         // lib.rs:141-150
        for (size_t m = 0; m < M; ++m) for (size_t i = 0; i < Sh; ++i) sh_product[m][i] = tr.read_point();        // lib.rs:152-161
        G1Affine random_poly_commitment = tr.read_point();                        // vanishing.rs:49-58
        Fr y = tr.squeeze_challenge();
        size_t H = domain.quotient_poly_degree;
        std::vector<G1Affine> h_commitments(H);
        for (size_t i = 0; i < H; ++i) h_commitments[i] = tr.read_point();        // vanishing.rs:61-74
        Fr x = tr.squeeze_challenge();

        // lib.rs:173-218 instance evaluations
        u64 nexp[4] = {params.n(), 0, 0, 0};
        std::vector<std::vector<Fr>> instance_evals(M);
        {
            Fr xn = x.pow(nexp);
            int32_t min_rot = 0, max_rot = 0;
            for (const Query& q : cs.instance_queries) {
                if (q.rotation < min_rot) min_rot = q.rotation;
                else if (q.rotation > max_rot) max_rot = q.rotation;
            }
            size_t max_len = 0;
            for (const auto& instances : insts) for (const auto& col : instances) max_len = std::max(max_len, col.size());
            std::vector<Fr> l_i_s = domain.l_i_range(x, xn, -max_rot, (int32_t)max_len + std::abs(min_rot));
            for (size_t m = 0; m < M; ++m)
                for (const Query& q : cs.instance_queries) {
                    const std::vector<Fr>& inst = insts[m][q.column.index];
                    size_t offset = (size_t)(max_rot - q.rotation);
                    Fr s = Fr::zero();
                    for (size_t i = 0; i < inst.size(); ++i) s += inst[i] * l_i_s[offset + i];
                    instance_evals[m].push_back(s);
                }
        }

        // lib.rs:220-253
        std::vector<std::vector<Fr>> advice_evals(M, std::vector<Fr>(cs.advice_queries.size()));
        std::vector<Fr> fixed_evals(cs.fixed_queries.size());
        for (size_t m = 0; m < M; ++m) for (auto& e : advice_evals[m]) e = tr.read_scalar();
        for (auto& e : fixed_evals) e = tr.read_scalar();
        Fr random_eval = tr.read_scalar();
        std::vector<Fr> perm_common(P);
        for (auto& e : perm_common) e = tr.read_scalar();
        struct PermSet { Fr eval, next_eval, last_eval; bool has_last; };
        std::vector<std::vector<PermSet>> psets(M, std::vector<PermSet>(nsets));
        for (size_t m = 0; m < M; ++m)
            for (size_t i = 0; i < nsets; ++i) {                                 // permutation.rs:105-131
                psets[m][i].eval = tr.read_scalar();
                psets[m][i].next_eval = tr.read_scalar();
                psets[m][i].has_last = i + 1 < nsets;
                if (psets[m][i].has_last) psets[m][i].last_eval = tr.read_scalar();
            }
        struct LkEval { Fr product, product_next, input, input_inv, table; };
        std::vector<std::vector<LkEval>> lk(M, std::vector<LkEval>(L));
        for (size_t m = 0; m < M; ++m)
            for (auto& e : lk[m]) {                                              // lookup.rs:127-146
                e.product = tr.read_scalar(); e.product_next = tr.read_scalar();
                e.input = tr.read_scalar(); e.input_inv = tr.read_scalar(); e.table = tr.read_scalar();
            }
        struct ShEval { Fr product, product_next; };
        std::vector<std::vector<ShEval>> sh(M, std::vector<ShEval>(Sh));
        for (size_t m = 0; m < M; ++m) for (auto& e : sh[m]) { e.product = tr.read_scalar(); e.product_next = tr.read_scalar(); }

        // lib.rs:257-346
        Fr xn = x.pow(nexp);
        size_t bf = cs.blinding_factors();
        std::vector<Fr> l_evals = domain.l_i_range(x, xn, -(int32_t)(bf + 1), 1);
        Fr l_last = l_evals[0];
        Fr l_blind = Fr::zero();
        for (size_t i = 1; i < 1 + bf; ++i) l_blind += l_evals[i];
        Fr l_0 = l_evals[1 + bf];
        Fr active_rows = Fr::one() - (l_last + l_blind);

        std::vector<Fr> exprs;
        for (size_t m = 0; m < M; ++m) {   // flat_map over the instances, each: gates, permutation, lookups, shuffles
            const std::vector<Fr>&adv = advice_evals[m], &ins = instance_evals[m];
            for (const ExprPoly& g : cs.gates) exprs.push_back(eval_expr(g, cs.coeff_vals, adv, fixed_evals, ins, challenges));
            // permutation.rs:189-288
            auto column_eval = [&](const Column& c) -> Fr {
                size_t qi = cs.get_any_query_index(c, 0);
                if (c.is_advice()) return adv[qi];
                if (c.type == COL_FIXED) return fixed_evals[qi];
                return ins[qi];
            };
            const std::vector<PermSet>& ps = psets[m];
            if (nsets > 0) {
                exprs.push_back(l_0 * (Fr::one() - ps[0].eval));
                exprs.push_back((ps[nsets - 1].eval.sqr() - ps[nsets - 1].eval) * l_last);
                for (size_t i = 1; i < nsets; ++i) exprs.push_back((ps[i].eval - ps[i - 1].last_eval) * l_0);
                for (size_t ci = 0; ci < nsets; ++ci) {
                    size_t lo = ci * chunk_len, hi = std::min(P, lo + chunk_len);
                    Fr left = ps[ci].next_eval;
                    for (size_t j = lo; j < hi; ++j) left *= column_eval(cs.permutation_columns[j]) + beta * perm_common[j] + gamma;
                    Fr right = ps[ci].eval;
                    Fr current_delta = (beta * x) * fr_consts().delta.pow_u64(ci * chunk_len);
                    for (size_t j = lo; j < hi; ++j) {
                        right *= column_eval(cs.permutation_columns[j]) + current_delta + gamma;
                        current_delta *= fr_consts().delta;
                    }
                    exprs.push_back((left - right) * (Fr::one() - (l_last + l_blind)));
                }
            }
            auto compress = [&](const std::vector<ExprPoly>& es) {
                Fr acc2 = Fr::zero();
                for (const ExprPoly& e : es) acc2 = acc2 * theta + eval_expr(e, cs.coeff_vals, adv, fixed_evals, ins, challenges);
                return acc2;
            };
            for (size_t i = 0; i < L; ++i) {                                      // lookup.rs:159-230
                const LkEval& e = lk[m][i];
                exprs.push_back(l_0 * (Fr::one() - e.product));
                exprs.push_back(l_last * (e.product.sqr() - e.product));
                Fr left = e.product_next * (e.input + beta) * (e.table + gamma);
                Fr right = e.product * (compress(cs.lookups[i].input) + beta) * (compress(cs.lookups[i].table) + gamma);
                exprs.push_back((left - right) * active_rows);
                exprs.push_back(l_0 * (e.input - e.table));
                exprs.push_back((e.input - e.table) * (e.input - e.input_inv) * active_rows);
            }
            for (size_t i = 0; i < Sh; ++i) {                                     // shuffle.rs:148-203
                const ShEval& e = sh[m][i];
                exprs.push_back(l_0 * (Fr::one() - e.product));
                exprs.push_back(l_last * (e.product.sqr() - e.product));
                Fr left = e.product_next * (compress(cs.shuffles[i].shuffle) + gamma);
                Fr right = e.product * (compress(cs.shuffles[i].input) + gamma);
                exprs.push_back((left - right) * active_rows);
            }
        }
        // vanishing.rs:92-121
        Fr expected_h_eval = Fr::zero();
        for (const Fr& v : exprs) expected_h_eval = expected_h_eval * y + v;
        Fr xn_m1 = xn - Fr::one();
        if (xn_m1.is_zero()) throw PanicEquiv{"called `Option::unwrap()` on a `None` value (vanishing.rs:100)"};
        expected_h_eval = expected_h_eval * xn_m1.inv();
        MSMKZG h_commitment;
        for (size_t i = H; i-- > 0;) {
            h_commitment.scale(xn);
            h_commitment.append_term(Fr::one(), G1::from_affine(h_commitments[i]));
        }

        // lib.rs:349-414 query list: per instance advice / permutation / lookups / shuffles, then the VK-wide ones
        std::vector<VQuery> queries;
        for (size_t m = 0; m < M; ++m) {
            const int im = (int)m;
            for (size_t qi = 0; qi < cs.advice_queries.size(); ++qi) {
                const Query& q = cs.advice_queries[qi];
                queries.push_back({{K_ADVICE, (int)q.column.index, im}, domain.rotate_omega(x, q.rotation), advice_evals[m][qi]});
            }
            {   // permutation.rs:290-325
                Fr x_next = domain.rotate_omega(x, 1);
                Fr x_last = domain.rotate_omega(x, -(int32_t)(bf + 1));
                for (size_t i = 0; i < nsets; ++i) {
                    queries.push_back({{K_PERM_PRODUCT, (int)i, im}, x, psets[m][i].eval});
                    queries.push_back({{K_PERM_PRODUCT, (int)i, im}, x_next, psets[m][i].next_eval});
                }
                for (size_t i = nsets; i-- > 0;) {
                    if (i + 1 == nsets) continue;  // rev().skip(1)
                    queries.push_back({{K_PERM_PRODUCT, (int)i, im}, x_last, psets[m][i].last_eval});
                }
            }
            for (size_t i = 0; i < L; ++i) {                                      // lookup.rs:232-272
                Fr x_inv = domain.rotate_omega(x, -1), x_next = domain.rotate_omega(x, 1);
                queries.push_back({{K_LOOKUP, (int)(3 * i + 0), im}, x, lk[m][i].product});
                queries.push_back({{K_LOOKUP, (int)(3 * i + 1), im}, x, lk[m][i].input});
                queries.push_back({{K_LOOKUP, (int)(3 * i + 2), im}, x, lk[m][i].table});
                queries.push_back({{K_LOOKUP, (int)(3 * i + 1), im}, x_inv, lk[m][i].input_inv});
                queries.push_back({{K_LOOKUP, (int)(3 * i + 0), im}, x_next, lk[m][i].product_next});
            }
            for (size_t i = 0; i < Sh; ++i) {                                     // shuffle.rs:205-225
                Fr x_next = domain.rotate_omega(x, 1);
                queries.push_back({{K_SHUFFLE, (int)i, im}, x, sh[m][i].product});
                queries.push_back({{K_SHUFFLE, (int)i, im}, x_next, sh[m][i].product_next});
            }
        }
        for (size_t qi = 0; qi < cs.fixed_queries.size(); ++qi) {
            const Query& q = cs.fixed_queries[qi];
            queries.push_back({{K_FIXED, (int)q.column.index, 0}, domain.rotate_omega(x, q.rotation), fixed_evals[qi]});
        }
        for (size_t i = 0; i < P; ++i) queries.push_back({{K_PERM_COMMON, (int)i, 0}, x, perm_common[i]});  // permutation.rs:328-340
        queries.push_back({{K_H_MSM, 0, 0}, x, expected_h_eval});                 // vanishing.rs:124-136
        queries.push_back({{K_RANDOM, 0, 0}, x, random_eval});

        auto base_of = [&](const CommitRef& c) -> G1Affine {
            switch (c.kind) {
                case K_ADVICE: return advice_commitments[c.inst][c.idx];
                case K_PERM_PRODUCT: return perm_product[c.inst][c.idx];
                case K_LOOKUP: return (c.idx % 3 == 0) ? lk_product[c.inst][c.idx / 3] : (c.idx % 3 == 1 ? lk_input[c.inst][c.idx / 3] : lk_table[c.inst][c.idx / 3]);
                case K_SHUFFLE: return sh_product[c.inst][c.idx];
                case K_FIXED: return vk.fixed_commitments[c.idx];
                case K_PERM_COMMON: return vk.permutation_commitments[c.idx];
                default: return random_poly_commitment;
            }
        };

        in_opening = true;
        if (opts.multiopen == MO_GWC) {
            // ------------------------------------------------------------ gwc.rs:54-163
            Fr gv = tr.squeeze_challenge();
            std::vector<std::pair<Fr, std::vector<const VQuery*>>> by_point;   // first-appearance order of points (gwc.rs:138-163)
            for (const VQuery& q : queries) {
                bool found = false;
                for (auto& e : by_point) if (e.first == q.point) { e.second.push_back(&q); found = true; break; }
                if (!found) by_point.push_back({q.point, {&q}});
            }
            std::vector<G1Affine> w(by_point.size());
            for (auto& wi : w) wi = tr.read_point();
            Fr gu = tr.squeeze_challenge();
            MSMKZG commitment_multi, witness, witness_with_aux;
            Fr eval_multi = Fr::zero(), power_of_u = Fr::one();
            for (size_t i = 0; i < by_point.size(); ++i, power_of_u = gu * power_of_u) {
                MSMKZG commitment_batch; Fr eval_batch = Fr::zero(), power_of_v = Fr::one();
                for (const VQuery* q : by_point[i].second) {
                    MSMKZG msm;
                    if (q->c.kind == K_H_MSM) { msm = h_commitment; msm.scale(power_of_v); }
                    else msm.append_term(power_of_v, G1::from_affine(base_of(q->c)));
                    commitment_batch.add_msm(msm);
                    eval_batch += power_of_v * q->eval;
                    power_of_v = gv * power_of_v;
                }
                commitment_batch.scale(power_of_u);
                commitment_multi.add_msm(commitment_batch);
                eval_multi += power_of_u * eval_batch;
                witness_with_aux.append_term(power_of_u * by_point[i].first, G1::from_affine(w[i]));
                witness.append_term(power_of_u, G1::from_affine(w[i]));
            }
            acc.left.add_msm(witness);
            acc.right.add_msm(witness_with_aux);
            acc.right.add_msm(commitment_multi);
            acc.right.append_term(eval_multi, G1::from_affine(params.g).neg());
            if (trace) {
                trace->challenges = challenges;
                trace->theta = theta; trace->beta = beta; trace->gamma = gamma; trace->y = y; trace->x = x;
                trace->sh_y = Fr::zero(); trace->sh_v = gv; trace->sh_u = gu;
                trace->expected_h_eval = expected_h_eval; trace->expressions = exprs;
            }
            return OK;
        }
        // ---------------------------------------------------------------- shplonk.rs:175-267
        std::vector<RotationSet> rotation_sets; PointSet super;
        construct_intermediate_sets(queries, rotation_sets, super);
        Fr sy = tr.squeeze_challenge();
        Fr sv = tr.squeeze_challenge();
        G1Affine h1 = tr.read_point();
        Fr su = tr.squeeze_challenge();
        G1Affine h2 = tr.read_point();

        Fr z_0_diff_inverse = Fr::zero(), z_0 = Fr::zero();
        MSMKZG outer_msm; Fr r_outer_acc = Fr::zero();
        Fr power_of_v = Fr::one();
        for (size_t i = 0; i < rotation_sets.size(); ++i, power_of_v = sv * power_of_v) {
            const RotationSet& rs = rotation_sets[i];
            std::vector<Fr> diffs;
            for (const Fr& p : super.pts) if (std::find(rs.points.begin(), rs.points.end(), p) == rs.points.end()) diffs.push_back(p);
            Fr z_diff_i = evaluate_vanishing_polynomial(diffs, su);
            if (i == 0) {
                z_0 = evaluate_vanishing_polynomial(rs.points, su);
                if (z_diff_i.is_zero()) throw PanicEquiv{"called `Option::unwrap()` on a `None` value (shplonk.rs:215)"};
                z_0_diff_inverse = z_diff_i.inv();
                z_diff_i = Fr::one();
            } else {
                z_diff_i = z_diff_i * z_0_diff_inverse;
            }
            MSMKZG inner_msm; Fr r_inner_acc = Fr::zero();
            Fr power_of_y = Fr::one();
            for (size_t j = 0; j < rs.commitments.size(); ++j, power_of_y = sy * power_of_y) {
                const RotCommitment& cd = rs.commitments[j];
                std::vector<Fr> r_x = lagrange_interpolate(rs.points, cd.evals);
                Fr r_eval = power_of_y * eval_polynomial(r_x, su);
                MSMKZG msm;
                if (cd.c.kind == K_H_MSM) { msm = h_commitment; msm.scale(power_of_y); }
                else msm.append_term(power_of_y, G1::from_affine(base_of(cd.c)));
                inner_msm.add_msm(msm);
                r_inner_acc += r_eval;
            }
            inner_msm.scale(power_of_v * z_diff_i);
            outer_msm.add_msm(inner_msm);
            r_outer_acc += power_of_v * r_inner_acc * z_diff_i;
        }
        outer_msm.append_term(r_outer_acc.neg(), G1::from_affine(params.g));
        outer_msm.append_term(z_0.neg(), G1::from_affine(h1));
        outer_msm.append_term(su, G1::from_affine(h2));
        acc.left.append_term(Fr::one(), G1::from_affine(h2));
        acc.right.add_msm(outer_msm);

        if (trace) {
            trace->challenges = challenges;
            trace->theta = theta; trace->beta = beta; trace->gamma = gamma; trace->y = y; trace->x = x;
            trace->sh_y = sy; trace->sh_v = sv; trace->sh_u = su;
            trace->expected_h_eval = expected_h_eval;
            trace->expressions = exprs;
        }
        return OK;
    } catch (const TranscriptError& e) {
        if (err_msg) *err_msg = e.what;
        return in_opening ? Opening : Transcript;  // lib.rs:420-424 flattens multi-open errors to Opening
    } catch (const PanicEquiv& e) {
        if (err_msg) *err_msg = e.what;
        return ReferencePanic;
    }
}

// one circuit instance per transcript: instances.len() == 1, what every caller inside the reference passes
Error verify_proof(const ParamsKZG& params, const VerifyingKey& vk, const std::vector<std::vector<Fr>>& instances,
                   const uint8_t* proof, size_t proof_len, DualMSM& acc, VerifyTrace* trace, const char** err_msg, VerifyOptions opts) {
    return verify_proof_multi(params, vk, {instances}, proof, proof_len, acc, trace, err_msg, opts);
}

Error verify_single(const ParamsKZG& params, const VerifyingKey& vk, const std::vector<std::vector<Fr>>& instances,
                    const uint8_t* proof, size_t proof_len, VerifyOptions opts) {
    DualMSM msm;
    Error e = verify_proof(params, vk, instances, proof, proof_len, msm, nullptr, nullptr, opts);
    if (e != OK) return e;
    return msm.check(params) ? OK : ConstraintSystemFailure;
}

Error AccumulatorStrategy::process(const ParamsKZG& params, const VerifyingKey& vk, const std::vector<std::vector<Fr>>& instances,
                                   const uint8_t* proof, size_t proof_len, const Fr& rand, VerifyOptions opts) {
    acc.scale(rand);  // strategy.rs:129 — before the closure runs
    return verify_proof(params, vk, instances, proof, proof_len, acc, nullptr, nullptr, opts);
}

// ------------------------------------------------------------------ G2 compressed encoding
static bool fq2_sqrt(const Fq2& a, Fq2& out) {
    if (a.is_zero()) { out = Fq2::zero(); return true; }
    Fq two_inv = Fq::from_u64(2).inv();
    if (a.c1.is_zero()) {
        Fq s;
        if (fq_sqrt(a.c0, s)) { out = {s, Fq::zero()}; return true; }
        if (fq_sqrt(a.c0.neg(), s)) { out = {Fq::zero(), s}; return true; }
        return false;
    }
    Fq nrm;
    if (!fq_sqrt(a.norm(), nrm)) return false;
    Fq delta = (a.c0 + nrm) * two_inv, x0;
    if (!fq_sqrt(delta, x0)) {
        delta = (a.c0 - nrm) * two_inv;
        if (!fq_sqrt(delta, x0)) return false;
    }
    Fq x1 = a.c1 * (x0.dbl()).inv();
    out = {x0, x1};
    return out.sqr() == a;
}
// The sign convention for G2 is, like the G1 flag layout, not pinned by anything in
// /root/reference; here sign = parity of y.c0 (then of y.c1 if y.c0 == 0).
static bool fq2_sign(const Fq2& y) { return y.c0.is_zero() ? y.c1.is_odd() : y.c0.is_odd(); }

bool g2_from_bytes(const uint8_t in[64], G2Affine& out) {
    uint8_t tmp[64]; memcpy(tmp, in, 64);
    bool is_inf = tmp[63] & G1_FLAG_IDENTITY, sign = tmp[63] & G1_FLAG_SIGN;
    tmp[63] &= 0x3f;
    Fq2 x;
    if (!Fq::from_bytes(tmp, x.c0) || !Fq::from_bytes(tmp + 32, x.c1)) return false;
    if (is_inf) {
        if (!x.is_zero() || sign) return false;
        out.inf = true; out.x = Fq2::zero(); out.y = Fq2::zero();
        return true;
    }
    Fq2 y;
    if (!fq2_sqrt(x.sqr() * x + G2Affine::b(), y)) return false;
    if (fq2_sign(y) != sign) y = y.neg();
    out.x = x; out.y = y; out.inf = false;
    return true;
}
void g2_to_bytes(const G2Affine& p, uint8_t out[64]) {
    if (p.inf) { memset(out, 0, 64); out[63] = G1_FLAG_IDENTITY; return; }
    p.x.c0.to_bytes(out); p.x.c1.to_bytes(out + 32);
    if (fq2_sign(p.y)) out[63] |= G1_FLAG_SIGN;
}

}  // namespace h2o
