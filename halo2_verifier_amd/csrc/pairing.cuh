// Extension tower Fq2 / Fq6 / Fq12 and the optimal-ate pairing pieces for BN254, usable from
// kernels and from the host side of the library.
//
//   Fq2  = Fq[u]  / (u^2 + 1)
//   Fq6  = Fq2[v] / (v^3 - xi),  xi = 9 + u
//   Fq12 = Fq6[w] / (w^2 - v)
//
// Replaces what the reference obtains from halo2curves at poly/kzg/msm.rs:185-203:
//   G2Prepared::from(s_g2), G2Prepared::from(-g2)            -> g2_prepare()   (host, once per context)
//   multi_miller_loop(&[(left, ..), (right, ..)])             -> miller_loop_2()
//   .final_exponentiation().is_identity()                     -> final_exp_is_one()
// Only the boolean is observable through the reference API, so the line-function
// normalisation and the exponent multiple used in the hard part are free choices.
#pragma once
#include "curve.cuh"

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
    // (c0 + c1 v + c2 v^2) * (d0 + d1 v)
    H2V_HD Fq6 mul_by_01(const Fq2& d0, const Fq2& d1) const {
        Fq2 a = c0 * d0, b = c1 * d1;
        Fq2 t0 = ((c1 + c2) * d1 - b).mul_xi() + a;
        Fq2 t1 = (c0 + c1) * (d0 + d1) - a - b;
        Fq2 t2 = (c0 + c2) * d0 - a + b;
        return {t0, t1, t2};
    }
    H2V_HD Fq6 scale2(const Fq2& k) const { return {c0 * k, c1 * k, c2 * k}; }
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

// Frobenius coefficients xi^(i (p-1)/6), i = 1..5, computed once on the host
struct PairingConsts {
    Fq2 gamma1[6];
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
    H2V_FN Fq12 sqr() const {  // complex squaring: 2 Fq6 multiplications
        Fq6 ab = c0 * c1;
        Fq6 t = (c0 + c1) * (c0 + c1.mul_v()) - ab - ab.mul_v();
        return {t, ab + ab};
    }
    // Granger-Scott squaring, valid in the cyclotomic subgroup (after the easy part of the final exponentiation):
    // 6 Fq2 products instead of 12
    H2V_FN Fq12 cyclotomic_sqr() const {
        const Fq2 &r0 = c0.c0, &r4 = c0.c1, &r3 = c0.c2, &r2 = c1.c0, &r1 = c1.c1, &r5 = c1.c2;
        Fq2 tmp = r0 * r1;
        Fq2 t0 = (r0 + r1) * (r1.mul_xi() + r0) - tmp - tmp.mul_xi(), t1 = tmp.dbl();
        tmp = r2 * r3;
        Fq2 t2 = (r2 + r3) * (r3.mul_xi() + r2) - tmp - tmp.mul_xi(), t3 = tmp.dbl();
        tmp = r4 * r5;
        Fq2 t4 = (r4 + r5) * (r5.mul_xi() + r4) - tmp - tmp.mul_xi(), t5 = tmp.dbl();
        Fq12 r;
        r.c0.c0 = (t0 - r0).dbl() + t0;
        r.c1.c1 = (t1 + r1).dbl() + t1;
        Fq2 x5 = t5.mul_xi();
        r.c1.c0 = (x5 + r2).dbl() + x5;
        r.c0.c2 = (t4 - r3).dbl() + t4;
        r.c0.c1 = (t2 - r4).dbl() + t2;
        r.c1.c2 = (t3 + r5).dbl() + t3;
        return r;
    }
    H2V_HD Fq12 conj() const { return {c0, c1.neg()}; }
    H2V_FN Fq12 inv() const {
        Fq6 t = (c0 * c0 - (c1 * c1).mul_v()).inv();
        return {c0 * t, (c1 * t).neg()};
    }
    H2V_HD bool is_one() const { return c0 == Fq6::one() && c1 == Fq6::zero(); }
    // multiply by the sparse line value  l = a + (b + c v) w   (a, b, c in Fq2)
    H2V_FN Fq12 mul_by_034(const Fq2& a, const Fq2& b, const Fq2& c) const {
        Fq6 t0 = c0.scale2(a);
        Fq6 t1 = c1.mul_by_01(b, c);
        Fq6 s = (c0 + c1).mul_by_01(a + b, c);
        return {t0 + t1.mul_v(), s - t0 - t1};
    }
    H2V_FN Fq12 frob(const PairingConsts& k) const {
        Fq12 r;
        r.c0.c0 = c0.c0.conj();
        r.c1.c0 = c1.c0.conj() * k.gamma1[1];
        r.c0.c1 = c0.c1.conj() * k.gamma1[2];
        r.c1.c1 = c1.c1.conj() * k.gamma1[3];
        r.c0.c2 = c0.c2.conj() * k.gamma1[4];
        r.c1.c2 = c1.c2.conj() * k.gamma1[5];
        return r;
    }
};

struct G2A { Fq2 x, y; bool inf; };
struct LineCoeff { Fq2 a, b, c; };  // value at P = (xP, yP):  a * yP  +  (b * xP) w  +  c v w

static constexpr uint64_t BN_X = 4965661367192848881ULL;
static constexpr uint64_t ATE_LOW = 0x9d797039be763ba8ULL;  // 6x+2 = 2^64 + ATE_LOW
static constexpr int MAX_LINE_COEFFS = 64 + 64 + 2;

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

// f = prod over two (P_k, prepared Q_k) pairs.  An identity P_k contributes 1.
H2V_FN Fq12 miller_loop_2(const G1A& p0, const LineCoeff* l0, const G1A& p1, const LineCoeff* l1) {
    Fq12 f = Fq12::one();
    bool s0 = p0.is_identity(), s1 = p1.is_identity();
    int idx = 0;
    for (int i = 63; i >= 0; --i) {
        f = f.sqr();
        if (!s0) f = f.mul_by_034(l0[idx].a.scale(p0.y), l0[idx].b.scale(p0.x), l0[idx].c);
        if (!s1) f = f.mul_by_034(l1[idx].a.scale(p1.y), l1[idx].b.scale(p1.x), l1[idx].c);
        ++idx;
        if ((ATE_LOW >> i) & 1) {
            if (!s0) f = f.mul_by_034(l0[idx].a.scale(p0.y), l0[idx].b.scale(p0.x), l0[idx].c);
            if (!s1) f = f.mul_by_034(l1[idx].a.scale(p1.y), l1[idx].b.scale(p1.x), l1[idx].c);
            ++idx;
        }
    }
    for (int t = 0; t < 2; ++t, ++idx) {
        if (!s0) f = f.mul_by_034(l0[idx].a.scale(p0.y), l0[idx].b.scale(p0.x), l0[idx].c);
        if (!s1) f = f.mul_by_034(l1[idx].a.scale(p1.y), l1[idx].b.scale(p1.x), l1[idx].c);
    }
    return f;
}

H2V_FN Fq12 fq12_pow_x(const Fq12& a) {
    Fq12 r = a;  // BN_X has its top bit at position 62; only called on cyclotomic-subgroup elements
    for (int i = 61; i >= 0; --i) {
        r = r.cyclotomic_sqr();
        if ((BN_X >> i) & 1) r = r * a;
    }
    return r;
}

// final_exponentiation(f).is_identity()
H2V_FN bool final_exp_is_one(const Fq12& f, const PairingConsts& k) {
    Fq12 r = f.conj() * f.inv();
    r = r.frob(k).frob(k) * r;
    Fq12 y0 = fq12_pow_x(r).conj();
    Fq12 y1 = y0.cyclotomic_sqr();
    Fq12 y2 = y1.cyclotomic_sqr();
    Fq12 y3 = y2 * y1;
    Fq12 y4 = fq12_pow_x(y3).conj();
    Fq12 y5 = y4.cyclotomic_sqr();
    Fq12 y6 = fq12_pow_x(y5).conj();
    y3 = y3.conj();
    y6 = y6.conj();
    Fq12 y7 = y6 * y4;
    Fq12 y8 = y7 * y3;
    Fq12 y9 = y8 * y1;
    Fq12 y10 = y8 * y4;
    Fq12 y11 = y10 * r;
    Fq12 y12 = y9.frob(k);
    Fq12 y13 = y12 * y11;
    y8 = y8.frob(k).frob(k);
    Fq12 y14 = y8 * y13;
    Fq12 y15 = (r.conj() * y9).frob(k).frob(k).frob(k);
    return (y15 * y14).is_one();
}

}  // namespace h2v
