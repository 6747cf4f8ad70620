// ORACLE — TEST INFRASTRUCTURE ONLY (see bn254_fp.hpp header).
//
// VerifyingKey / ConstraintSystem / ParamsKZG data model and the reference byte formats.
//   VerifyingKey            plonk/vk.rs:16-26     read/write :41-115
//   ConstraintSystem        plonk/vk.rs:173-211   read/write :213-365
//   IndexedExpressionPoly   plonk/vk.rs:461-546   (SparsePolynomial<u16, SparseTerm>, multilinear.rs:16-21)
//   Column<Any> serde       plonk/circuit.rs:35-70
//   permutation Argument/VerifyingKey   plonk/permutation.rs:19-44,136-176
//   lookup / shuffle Argument serde     plonk/lookup.rs:36-68, plonk/shuffle.rs:70-102
//   ParamsKZG               poly/kzg/commitment.rs:22-29   read_custom/write_custom :142-207
//   integers big-endian (helpers.rs:120-166) except k in ParamsKZG (LE, commitment.rs:147,160-162)
//
// Where the reference's writer and reader disagree (instance/fixed query counts, vk.rs:243-251
// vs :310-322; lookup/shuffle expression order, lookup.rs:36-49 vs :51-68) this file follows
// the READER on both sides, because VerifyingKey::read is what the verifier consumes.
#pragma once
#include "transcript.hpp"
#include <stdexcept>

namespace h2o {

enum SerdeFormat { Processed = 0, RawBytes = 1, RawBytesUnchecked = 2 };

struct ByteReader {
    const uint8_t* d; size_t n, pos;
    ByteReader(const uint8_t* p, size_t len) : d(p), n(len), pos(0) {}
    const uint8_t* take(size_t k) {
        if (pos + k > n) throw std::runtime_error("failed to fill whole buffer");
        const uint8_t* r = d + pos; pos += k; return r;
    }
    uint8_t u8() { return *take(1); }
    uint16_t u16() { const uint8_t* b = take(2); return (uint16_t)((b[0] << 8) | b[1]); }
    uint32_t u32() { const uint8_t* b = take(4); return ((uint32_t)b[0] << 24) | ((uint32_t)b[1] << 16) | ((uint32_t)b[2] << 8) | b[3]; }
    int32_t i32() { return (int32_t)u32(); }
    uint32_t u32le() { const uint8_t* b = take(4); return ((uint32_t)b[3] << 24) | ((uint32_t)b[2] << 16) | ((uint32_t)b[1] << 8) | b[0]; }
};
struct ByteWriter {
    std::vector<uint8_t> out;
    void bytes(const uint8_t* p, size_t k) { out.insert(out.end(), p, p + k); }
    void u8(uint8_t v) { out.push_back(v); }
    void u16(uint16_t v) { out.push_back(v >> 8); out.push_back(v & 0xff); }
    void u32(uint32_t v) { for (int s = 24; s >= 0; s -= 8) out.push_back((v >> s) & 0xff); }
    void i32(int32_t v) { u32((uint32_t)v); }
    void u32le(uint32_t v) { for (int s = 0; s < 32; s += 8) out.push_back((v >> s) & 0xff); }
};

inline Fr read_fr(ByteReader& r, SerdeFormat f) {
    Fr x;
    if (f == Processed) { if (!Fr::from_bytes(r.take(32), x)) throw std::runtime_error("Invalid prime field point encoding"); }
    else { bool ok = Fr::from_raw(r.take(32), x); if (!ok && f == RawBytes) throw std::runtime_error("Invalid prime field point encoding"); }
    return x;
}
inline void write_fr(ByteWriter& w, const Fr& x, SerdeFormat f) {
    uint8_t b[32];
    if (f == Processed) x.to_bytes(b);
    else for (int i = 0; i < 4; ++i) for (int j = 0; j < 8; ++j) b[8 * i + j] = (uint8_t)(x.v[i] >> (8 * j));
    w.bytes(b, 32);
}
inline G1Affine read_g1(ByteReader& r, SerdeFormat f) {
    G1Affine p;
    if (f == Processed) {
        if (!g1_from_bytes(r.take(32), p)) throw std::runtime_error("Invalid point encoding in proof");
        return p;
    }
    bool okx = Fq::from_raw(r.take(32), p.x), oky = Fq::from_raw(r.take(32), p.y);
    p.inf = p.x.is_zero() && p.y.is_zero();
    if (f == RawBytes && (!okx || !oky || !p.on_curve())) throw std::runtime_error("invalid raw point");
    return p;
}
inline void write_fq_raw(ByteWriter& w, const Fq& x) {
    uint8_t b[32];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 8; ++j) b[8 * i + j] = (uint8_t)(x.v[i] >> (8 * j));
    w.bytes(b, 32);
}
inline void write_g1(ByteWriter& w, const G1Affine& p, SerdeFormat f) {
    if (f == Processed) { uint8_t b[32]; g1_to_bytes(p, b); w.bytes(b, 32); return; }
    G1Affine q = p; if (q.inf) { q.x = Fq::zero(); q.y = Fq::zero(); }
    write_fq_raw(w, q.x); write_fq_raw(w, q.y);
}
// G2 compressed (Processed): 64 bytes = x.c0 LE | x.c1 LE, flags in the last byte as for G1.
bool g2_from_bytes(const uint8_t in[64], G2Affine& out);
void g2_to_bytes(const G2Affine& p, uint8_t out[64]);
inline G2Affine read_g2(ByteReader& r, SerdeFormat f) {
    G2Affine p;
    if (f == Processed) {
        if (!g2_from_bytes(r.take(64), p)) throw std::runtime_error("Invalid point encoding in proof");
        return p;
    }
    bool ok = Fq::from_raw(r.take(32), p.x.c0); ok &= Fq::from_raw(r.take(32), p.x.c1);
    ok &= Fq::from_raw(r.take(32), p.y.c0); ok &= Fq::from_raw(r.take(32), p.y.c1);
    p.inf = p.x.is_zero() && p.y.is_zero();
    if (f == RawBytes && (!ok || !p.on_curve())) throw std::runtime_error("invalid raw point");
    return p;
}
inline void write_g2(ByteWriter& w, const G2Affine& p, SerdeFormat f) {
    if (f == Processed) { uint8_t b[64]; g2_to_bytes(p, b); w.bytes(b, 64); return; }
    write_fq_raw(w, p.x.c0); write_fq_raw(w, p.x.c1); write_fq_raw(w, p.y.c0); write_fq_raw(w, p.y.c1);
}

// -------------------------------------------------------------------------------- data model
static const uint8_t COL_INSTANCE = 254, COL_FIXED = 255;  // 0..2 = advice phase (circuit.rs:36-65)
struct Column {
    uint32_t index; uint8_t type;
    bool is_advice() const { return type <= 2; }
    bool operator==(const Column& o) const { return index == o.index && type == o.type; }
};
struct Query { Column column; int32_t rotation; };

struct ExprTerm { uint16_t coeff_idx; std::vector<std::pair<uint32_t, uint32_t>> factors; };  // (var, pow)
struct ExprPoly { uint32_t num_vars = 0; std::vector<ExprTerm> terms; };

struct LookupArg { std::vector<ExprPoly> input, table; };
struct ShuffleArg { std::vector<ExprPoly> input, shuffle; };

struct ConstraintSystem {
    uint32_t num_fixed_columns = 0, num_advice_columns = 0, num_instance_columns = 0, num_selectors = 0, num_challenges = 0;
    std::vector<uint8_t> advice_column_phase, challenge_phase;
    std::vector<ExprPoly> gates;
    std::vector<uint32_t> num_advice_queries;
    std::vector<Query> advice_queries, instance_queries, fixed_queries;
    std::vector<Column> permutation_columns;
    std::vector<LookupArg> lookups;
    std::vector<ShuffleArg> shuffles;
    std::vector<Fr> coeff_vals;

    // plonk/vk.rs:396-401
    size_t blinding_factors() const {
        size_t f = 1;
        if (!num_advice_queries.empty()) { f = 0; for (uint32_t q : num_advice_queries) if (q > f) f = q; }
        if (f < 3) f = 3;
        return f + 2;
    }
    // plonk/vk.rs:403-411
    uint8_t max_phase() const { uint8_t m = 0; for (uint8_t p : advice_column_phase) if (p > m) m = p; return m; }
    // plonk/vk.rs:413-455 (linear searches; panic -> exception)
    size_t get_any_query_index(const Column& c, int32_t rot) const {
        const std::vector<Query>& qs = c.is_advice() ? advice_queries : (c.type == COL_FIXED ? fixed_queries : instance_queries);
        for (size_t i = 0; i < qs.size(); ++i) if (qs[i].column == c && qs[i].rotation == rot) return i;
        throw std::runtime_error("get_query_index called for non-existent query");
    }
};

struct VerifyingKey {
    uint32_t k = 0;
    std::vector<G1Affine> fixed_commitments;
    std::vector<G1Affine> permutation_commitments;
    ConstraintSystem cs;
    uint32_t cs_degree = 0;
    Fr transcript_repr;
    std::vector<std::vector<uint8_t>> selectors;  // packed bits, ceil(2^k/8) bytes each; unused by verification
};

struct ParamsKZG {
    uint32_t k = 0;
    G1Affine g;
    G2Affine g2, s_g2;
    uint64_t n() const { return 1ULL << k; }
};

// -------------------------------------------------------------------------------- serde
inline ExprPoly read_expr(ByteReader& r) {
    ExprPoly e; e.num_vars = r.u32();
    uint32_t nt = r.u32(); e.terms.resize(nt);
    for (auto& t : e.terms) {
        t.coeff_idx = r.u16();
        uint32_t nf = r.u32(); t.factors.resize(nf);
        for (auto& f : t.factors) { f.first = r.u32(); f.second = r.u32(); }
    }
    return e;
}
inline void write_expr(ByteWriter& w, const ExprPoly& e) {
    w.u32(e.num_vars); w.u32((uint32_t)e.terms.size());
    for (const auto& t : e.terms) {
        w.u16(t.coeff_idx); w.u32((uint32_t)t.factors.size());
        for (const auto& f : t.factors) { w.u32(f.first); w.u32(f.second); }
    }
}
inline Column read_column(ByteReader& r) {
    Column c; c.index = r.u32(); c.type = r.u8();
    if (!(c.type <= 2 || c.type >= 254)) throw std::runtime_error("Invalid phase for advice column");
    return c;
}

inline ConstraintSystem read_cs(ByteReader& r, SerdeFormat f) {
    ConstraintSystem cs;
    cs.num_fixed_columns = r.u32(); cs.num_advice_columns = r.u32(); cs.num_instance_columns = r.u32();
    cs.num_selectors = r.u32(); cs.num_challenges = r.u32();
    uint32_t ng = r.u32(), nl = r.u32(), ns = r.u32(), nc = r.u32();
    for (uint32_t i = 0; i < cs.num_advice_columns; ++i) cs.advice_column_phase.push_back(r.u8());
    for (uint32_t i = 0; i < cs.num_challenges; ++i) cs.challenge_phase.push_back(r.u8());
    size_t total = 0;
    for (uint32_t i = 0; i < cs.num_advice_columns; ++i) { cs.num_advice_queries.push_back(r.u32()); total += cs.num_advice_queries.back(); }
    for (size_t i = 0; i < total; ++i) { Query q; q.column.index = r.u32(); q.column.type = r.u8(); q.rotation = r.i32(); cs.advice_queries.push_back(q); }
    for (uint32_t i = 0; i < cs.num_instance_columns; ++i) { Query q; q.column.index = r.u32(); q.column.type = COL_INSTANCE; q.rotation = r.i32(); cs.instance_queries.push_back(q); }
    for (uint32_t i = 0; i < cs.num_fixed_columns; ++i) { Query q; q.column.index = r.u32(); q.column.type = COL_FIXED; q.rotation = r.i32(); cs.fixed_queries.push_back(q); }
    uint32_t np = r.u32();
    for (uint32_t i = 0; i < np; ++i) cs.permutation_columns.push_back(read_column(r));
    for (uint32_t i = 0; i < ng; ++i) cs.gates.push_back(read_expr(r));
    for (uint32_t i = 0; i < nl; ++i) {
        LookupArg a; uint32_t m = r.u32();
        for (uint32_t j = 0; j < m; ++j) { a.input.push_back(read_expr(r)); a.table.push_back(read_expr(r)); }
        cs.lookups.push_back(a);
    }
    for (uint32_t i = 0; i < ns; ++i) {
        ShuffleArg a; uint32_t m = r.u32();
        for (uint32_t j = 0; j < m; ++j) { a.input.push_back(read_expr(r)); a.shuffle.push_back(read_expr(r)); }
        cs.shuffles.push_back(a);
    }
    for (uint32_t i = 0; i < nc; ++i) cs.coeff_vals.push_back(read_fr(r, f));
    return cs;
}
inline void write_cs(ByteWriter& w, const ConstraintSystem& cs, SerdeFormat f) {
    if (cs.instance_queries.size() != cs.num_instance_columns || cs.fixed_queries.size() != cs.num_fixed_columns)
        throw std::runtime_error("VK does not round-trip through the reference reader (vk.rs:310-322)");
    w.u32(cs.num_fixed_columns); w.u32(cs.num_advice_columns); w.u32(cs.num_instance_columns);
    w.u32(cs.num_selectors); w.u32(cs.num_challenges);
    w.u32((uint32_t)cs.gates.size()); w.u32((uint32_t)cs.lookups.size()); w.u32((uint32_t)cs.shuffles.size()); w.u32((uint32_t)cs.coeff_vals.size());
    for (uint8_t p : cs.advice_column_phase) w.u8(p);
    for (uint8_t p : cs.challenge_phase) w.u8(p);
    for (uint32_t q : cs.num_advice_queries) w.u32(q);
    for (const auto& q : cs.advice_queries) { w.u32(q.column.index); w.u8(q.column.type); w.i32(q.rotation); }
    for (const auto& q : cs.instance_queries) { w.u32(q.column.index); w.i32(q.rotation); }
    for (const auto& q : cs.fixed_queries) { w.u32(q.column.index); w.i32(q.rotation); }
    w.u32((uint32_t)cs.permutation_columns.size());
    for (const auto& c : cs.permutation_columns) { w.u32(c.index); w.u8(c.type); }
    for (const auto& g : cs.gates) write_expr(w, g);
    for (const auto& a : cs.lookups) {
        if (a.input.size() != a.table.size()) throw std::runtime_error("lookup arity mismatch");
        w.u32((uint32_t)a.input.size());
        for (size_t j = 0; j < a.input.size(); ++j) { write_expr(w, a.input[j]); write_expr(w, a.table[j]); }
    }
    for (const auto& a : cs.shuffles) {
        if (a.input.size() != a.shuffle.size()) throw std::runtime_error("shuffle arity mismatch");
        w.u32((uint32_t)a.input.size());
        for (size_t j = 0; j < a.input.size(); ++j) { write_expr(w, a.input[j]); write_expr(w, a.shuffle[j]); }
    }
    for (const auto& c : cs.coeff_vals) write_fr(w, c, f);
}

inline VerifyingKey read_vk(const uint8_t* data, size_t len, SerdeFormat f) {
    ByteReader r(data, len);
    VerifyingKey vk;
    vk.k = r.u32();
    uint32_t nf = r.u32();
    for (uint32_t i = 0; i < nf; ++i) vk.fixed_commitments.push_back(read_g1(r, f));
    vk.cs_degree = r.u32();
    vk.cs = read_cs(r, f);
    for (size_t i = 0; i < vk.cs.permutation_columns.size(); ++i) vk.permutation_commitments.push_back(read_g1(r, f));
    size_t sel_bytes = ((1ULL << vk.k) + 7) / 8;
    for (uint32_t i = 0; i < vk.cs.num_selectors; ++i) { const uint8_t* b = r.take(sel_bytes); vk.selectors.emplace_back(b, b + sel_bytes); }
    vk.transcript_repr = read_fr(r, f);
    return vk;
}
inline std::vector<uint8_t> write_vk(const VerifyingKey& vk, SerdeFormat f) {
    ByteWriter w;
    w.u32(vk.k); w.u32((uint32_t)vk.fixed_commitments.size());
    for (const auto& c : vk.fixed_commitments) write_g1(w, c, f);
    w.u32(vk.cs_degree);
    write_cs(w, vk.cs, f);
    for (const auto& c : vk.permutation_commitments) write_g1(w, c, f);
    for (const auto& s : vk.selectors) w.bytes(s.data(), s.size());
    write_fr(w, vk.transcript_repr, f);
    return w.out;
}

inline ParamsKZG read_params(const uint8_t* data, size_t len, SerdeFormat f) {
    ByteReader r(data, len);
    ParamsKZG p; p.k = r.u32le();
    p.g = read_g1(r, f); p.g2 = read_g2(r, f); p.s_g2 = read_g2(r, f);
    return p;
}
inline std::vector<uint8_t> write_params(const ParamsKZG& p, SerdeFormat f) {
    ByteWriter w; w.u32le(p.k);
    write_g1(w, p.g, f); write_g2(w, p.g2, f); write_g2(w, p.s_g2, f);
    return w.out;
}

}  // namespace h2o
