// ORACLE — TEST INFRASTRUCTURE ONLY (see bn254_fp.hpp header).
//
// Optimal-ate pairing on BN254, the CPU restatement of what the reference obtains from
// halo2curves at poly/kzg/msm.rs:185-203:
//     E::multi_miller_loop(&[(left, s_g2_prepared), (right, n_g2_prepared)])
//         .final_exponentiation().is_identity()
// Miller loop over 6x+2 (x = 4965661367192848881) with homogeneous-projective line functions
// on the D-type twist, followed by the two Frobenius correction lines; final exponentiation
// = easy part (p^6-1)(p^2+1), hard part via the x-power chain of Fuentes-Castaneda et al.
#include "bn254_curve.hpp"

namespace h2o {

static const u64 BN_X = 4965661367192848881ULL;
// 6x+2 = 2^64 + ATE_LOW
static const u64 ATE_LOW = 0x9d797039be763ba8ULL;

static Fq fq_from_hex(const char* h) {  // 64 hex digits, big endian
    uint8_t le[32];
    for (int i = 0; i < 32; ++i) {
        auto nib = [](char c) -> int { return c <= '9' ? c - '0' : c - 'a' + 10; };
        le[31 - i] = (uint8_t)(nib(h[2 * i]) * 16 + nib(h[2 * i + 1]));
    }
    Fq r; Fq::from_bytes(le, r);
    return r;
}

G2Affine g2_generator() {
    G2Affine g; g.inf = false;
    g.x.c0 = fq_from_hex("1800deef121f1e76426a00665e5c4479674322d4f75edadd46debd5cd992f6ed");
    g.x.c1 = fq_from_hex("198e9393920d483a7260bfb731fb5d25f1aa493335a9e71297e485b7aef312c2");
    g.y.c0 = fq_from_hex("12c85ea5db8c6deb4aab71808dcb408fe3d1e7690c43d37b4ce6cc0166fa7daa");
    g.y.c1 = fq_from_hex("090689d0585ff075ec9e99ad690c3395bc4b313370b38ef355acdadcd122975b");
    return g;
}

namespace {
struct G2Hom { Fq2 x, y, z; };

G2Prepared::Coeff dbl_step(G2Hom& r, const Fq& two_inv, const Fq2& twist_b) {
    Fq2 a = (r.x * r.y).scale(two_inv);
    Fq2 b = r.y.sqr();
    Fq2 c = r.z.sqr();
    Fq2 e = twist_b * (c.dbl() + c);
    Fq2 f = e.dbl() + e;
    Fq2 g = (b + f).scale(two_inv);
    Fq2 h = (r.y + r.z).sqr() - (b + c);
    Fq2 i = e - b;
    Fq2 j = r.x.sqr();
    Fq2 e2 = e.sqr();
    r.x = a * (b - f);
    r.y = g.sqr() - (e2.dbl() + e2);
    r.z = b * h;
    return {h.neg(), j.dbl() + j, i};
}

G2Prepared::Coeff add_step(G2Hom& r, const G2Affine& q) {
    Fq2 theta = r.y - q.y * r.z;
    Fq2 lambda = r.x - q.x * r.z;
    Fq2 c = theta.sqr();
    Fq2 d = lambda.sqr();
    Fq2 e = lambda * d;
    Fq2 f = r.z * c;
    Fq2 g = r.x * d;
    Fq2 h = e + f - g.dbl();
    r.x = lambda * h;
    r.y = theta * (g - h) - e * r.y;
    r.z = r.z * e;
    Fq2 j = theta * q.x - lambda * q.y;
    return {lambda, theta.neg(), j};
}

// untwist-Frobenius-twist endomorphism on the twist: (x, y) -> (conj(x) xi^((p-1)/3), conj(y) xi^((p-1)/2))
G2Affine mul_by_char(const G2Affine& q) {
    const Fq12Consts& k = fq12_consts();
    G2Affine s; s.inf = q.inf;
    s.x = q.x.conj() * k.gamma1[2];
    s.y = q.y.conj() * k.gamma1[3];
    return s;
}

inline void ell(Fq12& f, const G2Prepared::Coeff& c, const G1Affine& p) {
    Fq12 l;
    l.c0 = {c.a.scale(p.y), Fq2::zero(), Fq2::zero()};
    l.c1 = {c.b.scale(p.x), c.c, Fq2::zero()};
    f = f * l;
}
}  // namespace

G2Prepared::G2Prepared(const G2Affine& q) : inf(q.inf) {
    if (q.inf) return;
    Fq two_inv = Fq::from_u64(2).inv();
    Fq2 tb = G2Affine::b();
    G2Hom r = {q.x, q.y, Fq2::one()};
    for (int i = 63; i >= 0; --i) {  // bit 64 is the leading one
        coeffs.push_back(dbl_step(r, two_inv, tb));
        if ((ATE_LOW >> i) & 1) coeffs.push_back(add_step(r, q));
    }
    G2Affine q1 = mul_by_char(q);
    G2Affine q2 = mul_by_char(q1);
    q2.y = q2.y.neg();
    coeffs.push_back(add_step(r, q1));
    coeffs.push_back(add_step(r, q2));
}

Fq12 multi_miller_loop(const G1Affine* ps, const G2Prepared* const* qs, size_t n) {
    Fq12 f = Fq12::one();
    std::vector<size_t> idx(n, 0);
    for (int i = 63; i >= 0; --i) {
        f = f.sqr();
        for (size_t k = 0; k < n; ++k) {
            if (ps[k].inf || qs[k]->inf) continue;
            ell(f, qs[k]->coeffs[idx[k]++], ps[k]);
        }
        if ((ATE_LOW >> i) & 1) {
            for (size_t k = 0; k < n; ++k) {
                if (ps[k].inf || qs[k]->inf) continue;
                ell(f, qs[k]->coeffs[idx[k]++], ps[k]);
            }
        }
    }
    for (int t = 0; t < 2; ++t)
        for (size_t k = 0; k < n; ++k) {
            if (ps[k].inf || qs[k]->inf) continue;
            ell(f, qs[k]->coeffs[idx[k]++], ps[k]);
        }
    return f;
}

Fq12 final_exponentiation(const Fq12& f) {
    // easy part
    Fq12 r = f.conj() * f.inv();   // f^(p^6-1)
    r = r.frob().frob() * r;       // ^(p^2+1)
    // hard part (x > 0 for BN254, so "exp by -x" = conj(r^x) in the cyclotomic subgroup)
    auto exp_neg_x = [](const Fq12& a) { return a.pow_u64(BN_X).conj(); };
    Fq12 y0 = exp_neg_x(r);
    Fq12 y1 = y0.sqr();
    Fq12 y2 = y1.sqr();
    Fq12 y3 = y2 * y1;
    Fq12 y4 = exp_neg_x(y3);
    Fq12 y5 = y4.sqr();
    Fq12 y6 = exp_neg_x(y5);
    y3 = y3.conj();
    y6 = y6.conj();
    Fq12 y7 = y6 * y4;
    Fq12 y8 = y7 * y3;
    Fq12 y9 = y8 * y1;
    Fq12 y10 = y8 * y4;
    Fq12 y11 = y10 * r;
    Fq12 y12 = y9.frob();
    Fq12 y13 = y12 * y11;
    y8 = y8.frob().frob();
    Fq12 y14 = y8 * y13;
    Fq12 rc = r.conj();
    Fq12 y15 = (rc * y9).frob().frob().frob();
    return y15 * y14;
}

bool pairing_product_is_one(const G1Affine& a1, const G2Affine& b1, const G1Affine& a2, const G2Affine& b2) {
    G2Prepared p1(b1), p2(b2);
    G1Affine ps[2] = {a1, a2};
    const G2Prepared* qs[2] = {&p1, &p2};
    return final_exponentiation(multi_miller_loop(ps, qs, 2)).is_one();
}

}  // namespace h2o
