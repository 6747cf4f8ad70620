// Extension tower Fq2 / Fq6 / Fq12 and the optimal-ate pairing pieces for BN254, usable from
// kernels and from the host side of the library.
//
//   Fq2  = Fq[u]  / (u^2 + 1)
//   Fq6  = Fq2[v] / (v^3 - xi),  xi = 9 + u
//   Fq12 = Fq6[w] / (w^2 - v)
//
// Replaces what the reference obtains from halo2curves at poly/kzg/msm.rs:185-203:
//   G2Prepared::from(s_g2), G2Prepared::from(-g2)            -> g2_prepare()   (host, once per context)
//   multi_miller_loop(&[(left, ..), (right, ..)]).final_exponentiation().is_identity()
//                                                             -> k_pairing_wave (pairing.hip), which works on the
//                                                                flat Fq2[w]/(w^6 - xi) view of Fq12 spread over a wave;
//                                                                the tower below serves g2_prepare and the one Fq12
//                                                                inversion of the final exponentiation
// Only the boolean is observable through the reference API, so the line-function
// normalisation and the exponent multiple used in the hard part are free choices.
#pragma once
#include "curve.hip.h"

namespace h2v {

struct Fq2 {
    Fq c0, c1;
    H2V_HD static Fq2 zero() { return {Fq::zero(), Fq::zero()}; }
    H2V_HD static Fq2 one() { return {Fq::one(), Fq::zero()}; }
    H2V_HD Fq2 operator+(const Fq2& o) const { return {c0 + o.c0, c1 + o.c1}; }
    H2V_HD Fq2 operator-(const Fq2& o) const { return {c0 - o.c0, c1 - o.c1}; }
    // Fq2 products are the call unit of the pairing: three (two) independent Fq products inlined side by side
    H2V_FN static Fq2 mul(Fq2 x, Fq2 o) {
        Fq a = Fq::mul_inl(x.c0, o.c0), b = Fq::mul_inl(x.c1, o.c1), m = Fq::mul_inl(x.c0 + x.c1, o.c0 + o.c1);
        return {a - b, m - a - b};
    }
    H2V_FN static Fq2 sqr_fn(Fq2 x) { return {Fq::mul_inl(x.c0 + x.c1, x.c0 - x.c1), Fq::mul_inl(x.c0, x.c1).dbl()}; }
    H2V_FN static Fq2 scale_fn(Fq2 x, Fq k) { return {Fq::mul_inl(x.c0, k), Fq::mul_inl(x.c1, k)}; }
    H2V_HD Fq2 operator*(const Fq2& o) const { return mul(*this, o); }
    H2V_HD Fq2 sqr() const { return sqr_fn(*this); }
    H2V_HD Fq2 dbl() const { return {c0.dbl(), c1.dbl()}; }
    H2V_HD Fq2 neg() const { return {c0.neg(), c1.neg()}; }
    H2V_HD Fq2 conj() const { return {c0, c1.neg()}; }
    H2V_HD Fq2 scale(const Fq& k) const { return scale_fn(*this, k); }
    H2V_HD Fq norm() const { return c0.sqr() + c1.sqr(); }
    H2V_FN Fq2 inv() const { Fq t = norm().inv(); return {c0 * t, (c1 * t).neg()}; }
    H2V_HD Fq2 mul_xi() const {  // * (9 + u)
        Fq t0 = c0.dbl().dbl().dbl() + c0, t1 = c1.dbl().dbl().dbl() + c1;
        return {t0 - c1, t1 + c0};
    }
    H2V_HD bool is_zero() const { return c0.is_zero() && c1.is_zero(); }
    H2V_HD bool operator==(const Fq2& o) const { return c0 == o.c0 && c1 == o.c1; }
};

struct Fq6 {
    Fq2 c0, c1, c2;
    H2V_HD static Fq6 zero() { return {Fq2::zero(), Fq2::zero(), Fq2::zero()}; }
    H2V_HD static Fq6 one() { return {Fq2::one(), Fq2::zero(), Fq2::zero()}; }
    H2V_HD Fq6 operator+(const Fq6& o) const { return {c0 + o.c0, c1 + o.c1, c2 + o.c2}; }
    H2V_HD Fq6 operator-(const Fq6& o) const { return {c0 - o.c0, c1 - o.c1, c2 - o.c2}; }
    H2V_HD Fq6 neg() const { return {c0.neg(), c1.neg(), c2.neg()}; }
    H2V_HD Fq6 operator*(const Fq6& o) const {
        Fq2 a = c0 * o.c0, b = c1 * o.c1, c = c2 * o.c2;
        Fq2 t0 = ((c1 + c2) * (o.c1 + o.c2) - b - c).mul_xi() + a;
        Fq2 t1 = (c0 + c1) * (o.c0 + o.c1) - a - b + c.mul_xi();
        Fq2 t2 = (c0 + c2) * (o.c0 + o.c2) - a - c + b;
        return {t0, t1, t2};
    }
    H2V_HD Fq6 mul_v() const { return {c2.mul_xi(), c0, c1}; }
    H2V_FN Fq6 inv() const {
        Fq2 A = c0.sqr() - (c1 * c2).mul_xi();
        Fq2 B = c2.sqr().mul_xi() - c0 * c1;
        Fq2 C = c1.sqr() - c0 * c2;
        Fq2 F = (c0 * A + (c2 * B + c1 * C).mul_xi()).inv();
        return {A * F, B * F, C * F};
    }
    H2V_HD bool operator==(const Fq6& o) const { return c0 == o.c0 && c1 == o.c1 && c2 == o.c2; }
};

// Frobenius coefficients xi^(i (p-1)/6), i = 1..5, computed once on the host: (sum a_k w^k)^p = sum conj(a_k) gamma1[k] w^k.  The higher
// powers as tables of their own (one step each in the pairing kernels' operation tables): ^(p^2): a_k gamma2[k], gamma2 = gamma1 conj(gamma1)
// in Fq;  ^(p^3): conj(a_k) gamma3[k], gamma3 = gamma1 gamma2;  ^(p^4): a_k gamma4[k], gamma4 = gamma2^2
struct PairingConsts {
    Fq2 gamma1[6];
    Fq2 gamma2[6], gamma3[6], gamma4[6];
    Fq two_inv;
    Fq2 twist_b;  // 3 / xi
};

struct Fq12 {
    Fq6 c0, c1;
    H2V_HD static Fq12 one() { return {Fq6::one(), Fq6::zero()}; }
    H2V_FN Fq12 operator*(const Fq12& o) const {
        Fq6 a = c0 * o.c0, b = c1 * o.c1;
        return {a + b.mul_v(), (c0 + c1) * (o.c0 + o.c1) - a - b};
    }
    H2V_HD Fq12 conj() const { return {c0, c1.neg()}; }
    H2V_FN Fq12 inv() const {
        Fq6 t = (c0 * c0 - (c1 * c1).mul_v()).inv();
        return {c0 * t, (c1 * t).neg()};
    }
};

struct G2A { Fq2 x, y; bool inf; };
struct LineCoeff { Fq2 a, b, c; };  // value at P = (xP, yP):  a * yP  +  (b * xP) w  +  c v w

static constexpr uint64_t BN_X = 4965661367192848881ULL;
static constexpr uint64_t ATE_LOW = 0x9d797039be763ba8ULL;  // 6x+2 = 2^64 + ATE_LOW
static constexpr int MAX_LINE_COEFFS = 64 + 64 + 2;  // >= 64 doublings + popcount(ATE_LOW) additions + 2

struct G2Hom { Fq2 x, y, z; };

H2V_FN LineCoeff g2_dbl_step(G2Hom& r, const PairingConsts& k) {
    Fq2 a = (r.x * r.y).scale(k.two_inv);
    Fq2 b = r.y.sqr(), c = r.z.sqr();
    Fq2 e = k.twist_b * (c.dbl() + c);
    Fq2 f = e.dbl() + e;
    Fq2 g = (b + f).scale(k.two_inv);
    Fq2 h = (r.y + r.z).sqr() - (b + c);
    Fq2 i = e - b, j = r.x.sqr(), e2 = e.sqr();
    r.x = a * (b - f);
    r.y = g.sqr() - (e2.dbl() + e2);
    r.z = b * h;
    return {h.neg(), j.dbl() + j, i};
}
H2V_FN LineCoeff g2_add_step(G2Hom& r, const Fq2& qx, const Fq2& qy) {
    Fq2 theta = r.y - qy * r.z, lambda = r.x - qx * r.z;
    Fq2 c = theta.sqr(), d = lambda.sqr();
    Fq2 e = lambda * d, f = r.z * c, g = r.x * d;
    Fq2 h = e + f - g.dbl();
    r.x = lambda * h;
    r.y = theta * (g - h) - e * r.y;
    r.z = r.z * e;
    return {lambda, theta.neg(), theta * qx - lambda * qy};
}
// Line coefficients of the whole Miller loop for a fixed Q (the role of G2Prepared).  Returns the count.
H2V_FN int g2_prepare(const G2A& q, const PairingConsts& k, LineCoeff* out) {
    int n = 0;
    G2Hom r = {q.x, q.y, Fq2::one()};
    for (int i = 63; i >= 0; --i) {
        out[n++] = g2_dbl_step(r, k);
        if ((ATE_LOW >> i) & 1) out[n++] = g2_add_step(r, q.x, q.y);
    }
    Fq2 q1x = q.x.conj() * k.gamma1[2], q1y = q.y.conj() * k.gamma1[3];
    Fq2 q2x = q1x.conj() * k.gamma1[2], q2y = (q1y.conj() * k.gamma1[3]).neg();
    out[n++] = g2_add_step(r, q1x, q1y);
    out[n++] = g2_add_step(r, q2x, q2y);
    return n;
}

}  // namespace h2v
