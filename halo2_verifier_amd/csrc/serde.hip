// Host side: the reference's byte formats WRITTEN (the readers live in vkplan.hip / params.hip).
//   VerifyingKey::write / to_bytes        plonk/vk.rs:41-64,118-123   ConstraintSystem::write :214-270
//   lookup / shuffle / permutation write   plonk/lookup.rs:36-49, plonk/shuffle.rs:70-84, plonk/permutation.rs:29-35,154-162
//   IndexedExpressionPoly::write           plonk/vk.rs:514-526
//   ParamsKZG::write_custom / to_bytes     poly/kzg/commitment.rs:142-152,215-224
//   SerdeFormat                            helpers.rs:7-19,48-58,88-97: Processed = compressed points + canonical scalars,
//                                          RawBytes / RawBytesUnchecked = uncompressed points, 4 x u64 LE Montgomery limbs (R = 2^256)
// No device code: conversions run wherever the library is loaded.
#include "../../include/h2v.h"
#include "vkplan.h"
#include <string.h>

namespace h2v {
namespace {

struct Writer {
    std::vector<uint8_t> b;
    void u8(uint8_t v) { b.push_back(v); }
    void u16(uint16_t v) { b.push_back((uint8_t)(v >> 8)); b.push_back((uint8_t)v); }                      // helpers.rs:144-149 (big endian)
    void u32(uint32_t v) { for (int i = 3; i >= 0; --i) b.push_back((uint8_t)(v >> (8 * i))); }
    void bytes(const uint8_t* p, size_t n) { b.insert(b.end(), p, p + n); }
};

// value -> halo2curves' in-memory form: the residue times 2^256, canonical, as 32 little-endian bytes.  The value is held as
// x * 2^261; one Montgomery product with the plain integer 2^256 (limb 8 = 2^24) gives x * 2^261 * 2^256 / 2^261 = x * 2^256.
template <class F> void field_to_mont_bytes(const F& x, uint8_t out[32]) {
    F k = F::zero();
    k.v[8] = 1u << 24;
    const F m = F::mul(x, k);
    uint32_t c[9], raw[8];
    m.canonical(c);
    F::unpack29(raw, c);
    for (int i = 0; i < 8; ++i) { out[4 * i] = (uint8_t)raw[i]; out[4 * i + 1] = (uint8_t)(raw[i] >> 8); out[4 * i + 2] = (uint8_t)(raw[i] >> 16); out[4 * i + 3] = (uint8_t)(raw[i] >> 24); }
}
void write_fr(Writer& w, int fmt, const Fr& x) {
    uint8_t b[32];
    if (fmt == H2V_SERDE_PROCESSED) x.to_bytes(b); else field_to_mont_bytes(x, b);
    w.bytes(b, 32);
}
// G1Affine::to_bytes: x little endian, byte 31 bit 6 = parity of y, bit 7 = identity (curve.hip.h: the inverse of g1_decompress)
void g1_compress(const G1A& p, uint8_t out[32]) {
    if (p.is_identity()) { memset(out, 0, 32); out[31] = G1_FLAG_IDENTITY; return; }
    p.x.to_bytes(out);
    if (p.y.is_odd()) out[31] |= G1_FLAG_SIGN;
}
void write_g1(Writer& w, int fmt, const G1A& p) {
    if (fmt == H2V_SERDE_PROCESSED) { uint8_t b[32]; g1_compress(p, b); w.bytes(b, 32); return; }
    uint8_t b[64];
    field_to_mont_bytes(p.x, b); field_to_mont_bytes(p.y, b + 32);     // the identity is (0, 0) in memory as well
    w.bytes(b, 64);
}
// compressed G2 (params.hip g2_decompress_host): x.c0 | x.c1 little endian, flags in byte 63; sign = parity of y.c0 (of y.c1 when y.c0 == 0)
void g2_compress(const G2A& p, uint8_t out[64]) {
    if (p.inf) { memset(out, 0, 64); out[63] = G1_FLAG_IDENTITY; return; }
    p.x.c0.to_bytes(out); p.x.c1.to_bytes(out + 32);
    const bool sign = p.y.c0.is_zero() ? p.y.c1.is_odd() : p.y.c0.is_odd();
    if (sign) out[63] |= G1_FLAG_SIGN;
}
void write_g2(Writer& w, int fmt, const G2A& p) {
    if (fmt == H2V_SERDE_PROCESSED) { uint8_t b[64]; g2_compress(p, b); w.bytes(b, 64); return; }
    uint8_t b[128];
    field_to_mont_bytes(p.x.c0, b); field_to_mont_bytes(p.x.c1, b + 32); field_to_mont_bytes(p.y.c0, b + 64); field_to_mont_bytes(p.y.c1, b + 96);
    w.bytes(b, 128);
}
void write_expr(Writer& w, const ExprH& e) {
    w.u32(e.num_vars); w.u32((uint32_t)e.terms.size());
    for (const TermH& t : e.terms) {
        w.u16(t.coeff_idx); w.u32((uint32_t)t.factors.size());
        for (const auto& f : t.factors) { w.u32(f.first); w.u32(f.second); }
    }
}
// a lookup / shuffle argument.  The reference's WRITER emits all first expressions, then all second ones (lookup.rs:36-49,
// shuffle.rs:70-84); its READER takes them in pairs (lookup.rs:51-68, shuffle.rs:86-102).  For arguments of one expression pair
// the two agree; for more, VerifyingKey::read does not invert VerifyingKey::write in the reference.  layout 0 = as the writer
// emits (the bytes VerifyingKey::write produces), 1 = as the reader consumes (bytes that read back as the same key).
void write_pairs(Writer& w, int layout, const std::vector<ExprH>& a, const std::vector<ExprH>& b) {
    w.u32((uint32_t)a.size());
    if (layout == H2V_VK_LAYOUT_READER) { for (size_t j = 0; j < a.size(); ++j) { write_expr(w, a[j]); write_expr(w, b[j]); } return; }
    for (const ExprH& e : a) write_expr(w, e);
    for (const ExprH& e : b) write_expr(w, e);
}

}  // namespace

void vk_to_bytes(const VkHost& vk, int fmt, int layout, std::vector<uint8_t>& out) {
    Writer w;
    w.u32(vk.k); w.u32((uint32_t)vk.fixed_commitments.size());
    for (const G1A& c : vk.fixed_commitments) write_g1(w, fmt, c);
    w.u32(vk.cs_degree);
    // ConstraintSystem::write (vk.rs:214-270)
    w.u32(vk.num_fixed_columns); w.u32(vk.num_advice_columns); w.u32(vk.num_instance_columns); w.u32(vk.num_selectors); w.u32(vk.num_challenges);
    w.u32((uint32_t)vk.gates.size()); w.u32((uint32_t)vk.lookups.size()); w.u32((uint32_t)vk.shuffles.size()); w.u32((uint32_t)vk.coeff_vals.size());
    for (uint8_t p : vk.advice_column_phase) w.u8(p);
    for (uint8_t p : vk.challenge_phase) w.u8(p);
    for (uint32_t q : vk.num_advice_queries) w.u32(q);
    for (const QueryH& q : vk.advice_queries) { w.u32(q.column.index); w.u8(q.column.type); w.u32((uint32_t)q.rotation); }
    // (a key that came through VerifyingKey::read holds exactly num_instance_columns / num_fixed_columns of these: vk.rs:310-322)
    for (const QueryH& q : vk.instance_queries) { w.u32(q.column.index); w.u32((uint32_t)q.rotation); }
    for (const QueryH& q : vk.fixed_queries) { w.u32(q.column.index); w.u32((uint32_t)q.rotation); }
    w.u32((uint32_t)vk.permutation_columns.size());
    for (const ColumnH& c : vk.permutation_columns) { w.u32(c.index); w.u8(c.type); }
    for (const ExprH& g : vk.gates) write_expr(w, g);
    for (const LookupH& a : vk.lookups) write_pairs(w, layout, a.input, a.table);
    for (const ShuffleH& a : vk.shuffles) write_pairs(w, layout, a.input, a.shuffle);
    for (const Fr& c : vk.coeff_vals) write_fr(w, fmt, c);
    // permutation::VerifyingKey::write, the selector bitmaps as they were read, transcript_repr
    for (const G1A& c : vk.permutation_commitments) write_g1(w, fmt, c);
    w.bytes(vk.selector_bytes.data(), vk.selector_bytes.size());
    write_fr(w, fmt, vk.transcript_repr);
    out.swap(w.b);
}

void params_to_bytes(const ParamsHost& p, int fmt, std::vector<uint8_t>& out) {
    Writer w;
    for (int i = 0; i < 4; ++i) w.u8((uint8_t)(p.k >> (8 * i)));     // k is little-endian here (commitment.rs:147)
    write_g1(w, fmt, p.g); write_g2(w, fmt, p.g2); write_g2(w, fmt, p.s_g2);
    out.swap(w.b);
}

}  // namespace h2v

using namespace h2v;

static int emit(const std::vector<uint8_t>& bytes, uint8_t* out, size_t* out_len, const char* who) {
    if (!out_len) { set_last_error(std::string(who) + ": out_len is null"); return H2V_ERR_BAD_ARGUMENT; }
    const size_t cap = *out_len;
    *out_len = bytes.size();
    if (!out) return 0;                                   // size query
    if (cap < bytes.size()) { set_last_error(std::string(who) + ": output buffer too small"); return H2V_ERR_BAD_ARGUMENT; }
    memcpy(out, bytes.data(), bytes.size());
    return 0;
}

extern "C" {

int h2v_vk_convert(const uint8_t* vk, size_t vk_len, int from_format, int to_format, int layout, uint8_t* out, size_t* out_len) {
    if (!vk) { set_last_error("h2v_vk_convert: null argument"); return H2V_ERR_BAD_ARGUMENT; }
    if (to_format < 0 || to_format > 2 || (layout != H2V_VK_LAYOUT_WRITER && layout != H2V_VK_LAYOUT_READER)) { set_last_error("h2v_vk_convert: unknown format / layout"); return H2V_ERR_BAD_ARGUMENT; }
    VkHost v;
    std::string err;
    if (!vk_from_bytes(vk, vk_len, from_format, v, err)) { set_last_error("VerifyingKey: " + err); return H2V_ERR_FORMAT; }
    std::vector<uint8_t> bytes;
    vk_to_bytes(v, to_format, layout, bytes);
    return emit(bytes, out, out_len, "h2v_vk_convert");
}

int h2v_params_convert(const uint8_t* params, size_t params_len, int from_format, int to_format, uint8_t* out, size_t* out_len) {
    if (!params) { set_last_error("h2v_params_convert: null argument"); return H2V_ERR_BAD_ARGUMENT; }
    if (to_format < 0 || to_format > 2) { set_last_error("h2v_params_convert: unknown format"); return H2V_ERR_BAD_ARGUMENT; }
    ParamsHost p;
    std::string err;
    if (!params_from_bytes(params, params_len, from_format, p, err)) { set_last_error("ParamsKZG: " + err); return H2V_ERR_FORMAT; }
    std::vector<uint8_t> bytes;
    params_to_bytes(p, to_format, bytes);
    return emit(bytes, out, out_len, "h2v_params_convert");
}

}  // extern "C"
