// Host side: ParamsKZG ingestion and the per-context pairing precomputation.
//   ParamsKZG::read_custom   poly/kzg/commitment.rs:155-207   (k is little-endian u32, :160-162)
//   SerdeFormat              helpers.rs:7-19, 33-65            (RawBytes = 4 x u64 LE Montgomery limbs)
//   G2Prepared::from         poly/kzg/msm.rs:186-187
#include "../../include/h2v.h"
#include "pairing_api.h"
#include <string.h>

namespace h2v {

static Fq fq_from_mont_bytes(const uint8_t* b, bool& ok) {
    uint32_t m[8];
    for (int i = 0; i < 8; ++i) m[i] = (uint32_t)b[4 * i] | ((uint32_t)b[4 * i + 1] << 8) | ((uint32_t)b[4 * i + 2] << 16) | ((uint32_t)b[4 * i + 3] << 24);
    if (Fq::geq_p(m)) ok = false;
    return Fq::from_mont256(m);
}

static Fq2 fq2_xi() { return {Fq::from_u32(9), Fq::from_u32(1)}; }

PairingConsts pairing_consts_host() {
    PairingConsts k;
    // (p - 1) / 6 by schoolbook long division over 32-bit limbs
    uint32_t e[8]; uint64_t rem = 0;
    uint32_t pm1[8];
    for (int i = 0; i < 8; ++i) pm1[i] = FqParams::P(i);
    pm1[0] -= 1;
    for (int i = 7; i >= 0; --i) { uint64_t cur = (rem << 32) | pm1[i]; e[i] = (uint32_t)(cur / 6); rem = cur % 6; }
    // g = xi^e in Fq2 (square and multiply)
    Fq2 g = Fq2::one(), xi = fq2_xi();
    for (int i = 255; i >= 0; --i) {
        g = g.sqr();
        if ((e[i / 32] >> (i % 32)) & 1) g = g * xi;
    }
    k.gamma1[0] = Fq2::one();
    for (int i = 1; i < 6; ++i) k.gamma1[i] = k.gamma1[i - 1] * g;
    for (int i = 0; i < 6; ++i) { k.gamma2[i] = k.gamma1[i] * k.gamma1[i].conj(); k.gamma3[i] = k.gamma1[i] * k.gamma2[i]; k.gamma4[i] = k.gamma2[i].sqr(); }
    k.two_inv = Fq::from_u32(2).inv();
    Fq2 three = {Fq::from_u32(3), Fq::zero()};
    k.twist_b = three * xi.inv();
    return k;
}

static bool fq_sqrt_host(const Fq& a, Fq& out) {
    Fq r = fq_sqrt_candidate(a);
    if (r.sqr() != a) return false;
    out = r;
    return true;
}
static bool fq2_sqrt_host(const Fq2& a, Fq2& out) {
    if (a.is_zero()) { out = Fq2::zero(); return true; }
    Fq two_inv = Fq::from_u32(2).inv();
    if (a.c1.is_zero()) {
        Fq s;
        if (fq_sqrt_host(a.c0, s)) { out = {s, Fq::zero()}; return true; }
        if (fq_sqrt_host(a.c0.neg(), s)) { out = {Fq::zero(), s}; return true; }
        return false;
    }
    Fq nrm;
    if (!fq_sqrt_host(a.norm(), nrm)) return false;
    Fq delta = (a.c0 + nrm) * two_inv, x0;
    if (!fq_sqrt_host(delta, x0)) {
        delta = (a.c0 - nrm) * two_inv;
        if (!fq_sqrt_host(delta, x0)) return false;
    }
    Fq x1 = a.c1 * x0.dbl().inv();
    out = {x0, x1};
    return out.sqr() == a;
}
static bool g2_on_curve(const G2A& p, const PairingConsts& k) { return p.inf || p.y.sqr() == p.x.sqr() * p.x + k.twist_b; }

// compressed G2: x.c0 | x.c1 little endian, flags in byte 63 as for G1; sign = parity of y.c0
// (of y.c1 when y.c0 == 0).  Like the G1 flag layout this is not pinned by anything in the
// reference tree.
static bool g2_decompress_host(const uint8_t in[64], const PairingConsts& k, G2A& out) {
    uint8_t tmp[64]; memcpy(tmp, in, 64);
    bool is_inf = tmp[63] & G1_FLAG_IDENTITY, sign = tmp[63] & G1_FLAG_SIGN;
    tmp[63] &= 0x3f;
    Fq2 x;
    if (!Fq::from_bytes(tmp, x.c0) || !Fq::from_bytes(tmp + 32, x.c1)) return false;
    if (is_inf) { if (!x.is_zero() || sign) return false; out.inf = true; out.x = Fq2::zero(); out.y = Fq2::zero(); return true; }
    Fq2 y;
    if (!fq2_sqrt_host(x.sqr() * x + k.twist_b, y)) return false;
    bool ysign = y.c0.is_zero() ? y.c1.is_odd() : y.c0.is_odd();
    if (ysign != sign) y = y.neg();
    out.x = x; out.y = y; out.inf = false;
    return true;
}

bool params_from_bytes(const uint8_t* d, size_t len, int format, ParamsHost& out, std::string& err) {
    PairingConsts k = pairing_consts_host();
    size_t need = format == H2V_SERDE_PROCESSED ? 4 + 32 + 64 + 64 : 4 + 64 + 128 + 128;
    if (format < 0 || format > 2) { err = "unknown serde format"; return false; }
    if (len < need) { err = "failed to fill whole buffer"; return false; }
    out.k = (uint32_t)d[0] | ((uint32_t)d[1] << 8) | ((uint32_t)d[2] << 16) | ((uint32_t)d[3] << 24);
    if (out.k > 28) { err = "k exceeds the 2-adicity of Fr"; return false; }
    const uint8_t* p = d + 4;
    if (format == H2V_SERDE_PROCESSED) {
        if (!g1_decompress(p, out.g)) { err = "invalid point encoding"; return false; }
        if (!g2_decompress_host(p + 32, k, out.g2) || !g2_decompress_host(p + 96, k, out.s_g2)) { err = "Invalid point encoding in proof"; return false; }
        return true;
    }
    bool ok = true;
    out.g.x = fq_from_mont_bytes(p, ok); out.g.y = fq_from_mont_bytes(p + 32, ok);
    auto g2 = [&](const uint8_t* q, G2A& o) {
        o.x.c0 = fq_from_mont_bytes(q, ok); o.x.c1 = fq_from_mont_bytes(q + 32, ok);
        o.y.c0 = fq_from_mont_bytes(q + 64, ok); o.y.c1 = fq_from_mont_bytes(q + 96, ok);
        o.inf = o.x.is_zero() && o.y.is_zero();
    };
    g2(p + 64, out.g2); g2(p + 192, out.s_g2);
    if (format == H2V_SERDE_RAW_BYTES) {
        if (!ok) { err = "field element not less than the modulus"; return false; }
        if (!out.g.on_curve() || !g2_on_curve(out.g2, k) || !g2_on_curve(out.s_g2, k)) { err = "point is not on the curve"; return false; }
    }
    return true;
}

int PairingDevice::upload(const ParamsHost& p) {
    PairingConsts k = pairing_consts_host();
    std::vector<LineCoeff> a(MAX_LINE_COEFFS), b(MAX_LINE_COEFFS);
    if (p.s_g2.inf || p.g2.inf) { set_last_error("params: g2 / s_g2 is the identity"); return H2V_ERR_FORMAT; }
    G2A ng2 = p.g2; ng2.y = ng2.y.neg();
    int na = g2_prepare(p.s_g2, k, a.data());
    int nb = g2_prepare(ng2, k, b.data());
    if (na != H2V_PAIRING_LINES || nb != H2V_PAIRING_LINES) { set_last_error("pairing: unexpected Miller loop length"); return H2V_ERR_DEVICE; }
    h_sg2 = p.s_g2; h_ng2 = ng2;
    H2V_HIP_CHECK(hipMalloc(&l_sg2, sizeof(LineCoeff) * na));
    H2V_HIP_CHECK(hipMalloc(&l_ng2, sizeof(LineCoeff) * na));
    H2V_HIP_CHECK(hipMalloc(&consts, sizeof(PairingConsts)));
    H2V_HIP_CHECK(hipMemcpy(l_sg2, a.data(), sizeof(LineCoeff) * na, hipMemcpyHostToDevice));
    H2V_HIP_CHECK(hipMemcpy(l_ng2, b.data(), sizeof(LineCoeff) * na, hipMemcpyHostToDevice));
    H2V_HIP_CHECK(hipMemcpy(consts, &k, sizeof(PairingConsts), hipMemcpyHostToDevice));
    const std::vector<uint32_t> ops = pairing_program(false), opsm = pairing_program(true);
    n_ops = (uint32_t)ops.size(); n_ops_merged = (uint32_t)opsm.size();
    H2V_HIP_CHECK(hipMalloc(&prog, 4 * ops.size()));
    H2V_HIP_CHECK(hipMemcpy(prog, ops.data(), 4 * ops.size(), hipMemcpyHostToDevice));
    H2V_HIP_CHECK(hipMalloc(&prog_merged, 4 * opsm.size()));
    H2V_HIP_CHECK(hipMemcpy(prog_merged, opsm.data(), 4 * opsm.size(), hipMemcpyHostToDevice));
    const std::vector<uint32_t> ops2 = pairing_program2();
    n_steps2 = (uint32_t)(ops2.size() / 2);
    if (n_steps2 > H2V_PAIR2_MAX_STEPS) { set_last_error("pairing: two-stream table too long"); return H2V_ERR_DEVICE; }
    H2V_HIP_CHECK(hipMalloc(&prog2, 4 * ops2.size()));
    H2V_HIP_CHECK(hipMemcpy(prog2, ops2.data(), 4 * ops2.size(), hipMemcpyHostToDevice));
    return 0;
}
// 2^shift * q by the Miller loop's own doubling step (homogeneous projective), back to affine
static G2A g2_times_pow2(const G2A& q, uint32_t shift, const PairingConsts& k) {
    if (!shift) return q;
    G2Hom r = {q.x, q.y, Fq2::one()};
    for (uint32_t i = 0; i < shift; ++i) (void)g2_dbl_step(r, k);
    const Fq2 zi = r.z.inv();
    return G2A{r.x * zi, r.y * zi, false};
}
int PairingDevice::split_lines(uint32_t shift, uint32_t parts, const LineCoeff** out) {
    std::lock_guard<std::mutex> lock(split_mu);
    for (const SplitTable& t : split) if (t.shift == shift && t.parts == parts) { *out = t.lines; return 0; }
    const PairingConsts k = pairing_consts_host();
    std::vector<LineCoeff> rows((size_t)2 * parts * H2V_PAIRING_LINES), one(MAX_LINE_COEFFS);
    G2A a = h_sg2, b = h_ng2;
    for (uint32_t j = 0; j < parts; ++j) {
        if (j) { a = g2_times_pow2(a, shift, k); b = g2_times_pow2(b, shift, k); }
        const G2A* side[2] = {&a, &b};
        for (int sd = 0; sd < 2; ++sd) {
            if (g2_prepare(*side[sd], k, one.data()) != H2V_PAIRING_LINES) { set_last_error("pairing: unexpected Miller loop length"); return H2V_ERR_DEVICE; }
            std::copy(one.begin(), one.begin() + H2V_PAIRING_LINES, rows.begin() + (size_t)(2 * j + sd) * H2V_PAIRING_LINES);
        }
    }
    LineCoeff* d = nullptr;
    H2V_HIP_CHECK(hipMalloc(&d, sizeof(LineCoeff) * rows.size()));
    hipError_t e = hipMemcpy(d, rows.data(), sizeof(LineCoeff) * rows.size(), hipMemcpyHostToDevice);
    if (e != hipSuccess) { hipFree(d); set_last_error(std::string("pairing: ") + hipGetErrorString(e)); return H2V_ERR_DEVICE; }
    split.push_back(SplitTable{shift, parts, d});
    *out = d;
    return 0;
}
void PairingDevice::release() {
    for (SplitTable& t : split) if (t.lines) hipFree(t.lines);
    split.clear();
    if (l_sg2) hipFree(l_sg2);
    if (l_ng2) hipFree(l_ng2);
    if (consts) hipFree(consts);
    if (prog) hipFree(prog);
    if (prog_merged) hipFree(prog_merged);
    if (prog2) hipFree(prog2);
    prog2 = nullptr; n_steps2 = 0;
    prog_merged = nullptr; n_ops_merged = 0;
    l_sg2 = l_ng2 = nullptr; consts = nullptr; prog = nullptr; n_ops = 0;
}

}  // namespace h2v
