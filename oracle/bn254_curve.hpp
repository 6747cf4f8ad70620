// ORACLE — TEST INFRASTRUCTURE ONLY (see bn254_fp.hpp header).
//
// BN254 G1 (y^2 = x^3 + 3 over Fq), extension tower Fq2/Fq6/Fq12, G2 on the sextic twist
// y^2 = x^3 + 3/(9+u), optimal-ate pairing.  Restated from the public BN254 definition because
// the reference's implementation lives in the un-vendored halo2curves crate.
//
// Reference call sites:
//   G1 add/double/batch_normalize   poly/kzg/msm.rs:81-86, arithmetic.rs:7-95
//   G1Affine::from_bytes            transcript/mod.rs:158-166   (compressed 32-B encoding)
//   G2Prepared / multi_miller_loop / final_exponentiation / is_identity   poly/kzg/msm.rs:185-203
//
// Only group elements and booleans are observable through the reference API, so internal
// coordinates (Jacobian here) are free.
#pragma once
#include "bn254_fp.hpp"
#include <vector>

namespace h2o {

// ----------------------------------------------------------------------------------- G1
struct G1Affine {
    Fq x, y;
    bool inf;
    static G1Affine identity() { G1Affine p; p.x = Fq::zero(); p.y = Fq::zero(); p.inf = true; return p; }
    static G1Affine generator() { G1Affine p; p.x = Fq::from_u64(1); p.y = Fq::from_u64(2); p.inf = false; return p; }
    bool on_curve() const {
        if (inf) return true;
        return y.sqr() == x.sqr() * x + Fq::from_u64(3);
    }
    bool operator==(const G1Affine& o) const {
        if (inf || o.inf) return inf && o.inf;
        return x == o.x && y == o.y;
    }
    G1Affine neg() const { G1Affine r = *this; r.y = y.neg(); return r; }
};

// Compressed encoding of the halo2curves line the fork tracks (>= 0.4): 32 bytes, x little
// endian in the low 254 bits; byte 31 bit 7 = identity flag, bit 6 = sign = y.to_repr()[0] & 1.
// This is the one detail nothing in /root/reference pins (SURVEY.md §8c "Unpinned detail");
// it is a single pair of constants here and in the product (csrc/bn254.hip.h).
static const uint8_t G1_FLAG_IDENTITY = 0x80;
static const uint8_t G1_FLAG_SIGN = 0x40;

// Fq sqrt for p = 3 mod 4: a^((p+1)/4); returns false if a is a non-residue
inline bool fq_sqrt(const Fq& a, Fq& out) {
    u64 e[4]; memcpy(e, Fq::C().p, 32);
    // (p+1)/4
    u64 one[4] = {1, 0, 0, 0};
    add4(e, e, one);
    for (int i = 0; i < 4; ++i) e[i] = (e[i] >> 2) | (i < 3 ? (e[i + 1] << 62) : 0);
    Fq r = a.pow(e);
    if (r.sqr() != a) return false;
    out = r;
    return true;
}

inline bool g1_from_bytes(const uint8_t in[32], G1Affine& out) {
    uint8_t tmp[32]; memcpy(tmp, in, 32);
    bool is_inf = tmp[31] & G1_FLAG_IDENTITY;
    bool sign = tmp[31] & G1_FLAG_SIGN;
    tmp[31] &= 0x3f;
    Fq x;
    if (!Fq::from_bytes(tmp, x)) return false;
    if (is_inf) {
        if (!x.is_zero() || sign) return false;
        out = G1Affine::identity();
        return true;
    }
    Fq y;
    if (!fq_sqrt(x.sqr() * x + Fq::from_u64(3), y)) return false;
    if (y.is_odd() != sign) y = y.neg();
    out.x = x; out.y = y; out.inf = false;
    return true;
}
inline void g1_to_bytes(const G1Affine& p, uint8_t out[32]) {
    if (p.inf) { memset(out, 0, 32); out[31] = G1_FLAG_IDENTITY; return; }
    p.x.to_bytes(out);
    if (p.y.is_odd()) out[31] |= G1_FLAG_SIGN;
}

struct G1 {  // Jacobian: (X/Z^2, Y/Z^3); identity <=> Z == 0
    Fq X, Y, Z;
    static G1 identity() { G1 p; p.X = Fq::zero(); p.Y = Fq::one(); p.Z = Fq::zero(); return p; }
    static G1 from_affine(const G1Affine& a) {
        if (a.inf) return identity();
        G1 p; p.X = a.x; p.Y = a.y; p.Z = Fq::one(); return p;
    }
    bool is_identity() const { return Z.is_zero(); }
    G1 neg() const { G1 r = *this; r.Y = Y.neg(); return r; }
    G1 dbl() const {
        if (is_identity()) return *this;
        // a = 0: dbl-2009-l
        Fq A = X.sqr(), B = Y.sqr(), Cc = B.sqr();
        Fq D = ((X + B).sqr() - A - Cc).dbl();
        Fq E = A.dbl() + A, F = E.sqr();
        G1 r;
        r.X = F - D.dbl();
        r.Y = E * (D - r.X) - Cc.dbl().dbl().dbl();
        r.Z = (Y * Z).dbl();
        return r;
    }
    G1 add(const G1& o) const {
        if (is_identity()) return o;
        if (o.is_identity()) return *this;
        Fq Z1Z1 = Z.sqr(), Z2Z2 = o.Z.sqr();
        Fq U1 = X * Z2Z2, U2 = o.X * Z1Z1;
        Fq S1 = Y * o.Z * Z2Z2, S2 = o.Y * Z * Z1Z1;
        if (U1 == U2) {
            if (S1 == S2) return dbl();
            return identity();
        }
        Fq H = U2 - U1, I = H.dbl().sqr(), J = H * I, rr = (S2 - S1).dbl(), V = U1 * I;
        G1 r;
        r.X = rr.sqr() - J - V.dbl();
        r.Y = rr * (V - r.X) - (S1 * J).dbl();
        r.Z = ((Z + o.Z).sqr() - Z1Z1 - Z2Z2) * H;
        return r;
    }
    G1 add_affine(const G1Affine& o) const {
        if (o.inf) return *this;
        if (is_identity()) return from_affine(o);
        Fq Z1Z1 = Z.sqr();
        Fq U2 = o.x * Z1Z1, S2 = o.y * Z * Z1Z1;
        if (X == U2) {
            if (Y == S2) return dbl();
            return identity();
        }
        Fq H = U2 - X, HH = H.sqr(), I = HH.dbl().dbl(), J = H * I, rr = (S2 - Y).dbl(), V = X * I;
        G1 r;
        r.X = rr.sqr() - J - V.dbl();
        r.Y = rr * (V - r.X) - (Y * J).dbl();
        r.Z = (Z + H).sqr() - Z1Z1 - HH;
        return r;
    }
    G1Affine to_affine() const {
        if (is_identity()) return G1Affine::identity();
        Fq zi = Z.inv(), zi2 = zi.sqr();
        G1Affine a; a.x = X * zi2; a.y = Y * zi2 * zi; a.inf = false;
        return a;
    }
    // double-and-add over the canonical scalar (MSB first)
    G1 mul(const Fr& k) const {
        u64 e[4]; k.to_limbs(e);
        G1 r = identity();
        for (int i = 255; i >= 0; --i) {
            r = r.dbl();
            if ((e[i / 64] >> (i % 64)) & 1) r = r.add(*this);
        }
        return r;
    }
    bool eq(const G1& o) const { return to_affine() == o.to_affine(); }
};

// == group::Curve::batch_normalize (used by MSMKZG::eval, poly/kzg/msm.rs:84)
inline void g1_batch_normalize(const G1* in, G1Affine* out, size_t n) {
    std::vector<Fq> z(n);
    for (size_t i = 0; i < n; ++i) z[i] = in[i].Z;
    batch_invert(z.data(), n);
    for (size_t i = 0; i < n; ++i) {
        if (in[i].is_identity()) { out[i] = G1Affine::identity(); continue; }
        Fq zi2 = z[i].sqr();
        out[i].x = in[i].X * zi2; out[i].y = in[i].Y * zi2 * z[i]; out[i].inf = false;
    }
}

// ----------------------------------------------------------------------------------- Fq2
struct Fq2 {  // c0 + c1 u, u^2 = -1
    Fq c0, c1;
    static Fq2 zero() { return {Fq::zero(), Fq::zero()}; }
    static Fq2 one() { return {Fq::one(), Fq::zero()}; }
    Fq2 operator+(const Fq2& o) const { return {c0 + o.c0, c1 + o.c1}; }
    Fq2 operator-(const Fq2& o) const { return {c0 - o.c0, c1 - o.c1}; }
    Fq2 operator*(const Fq2& o) const {
        Fq a = c0 * o.c0, b = c1 * o.c1;
        return {a - b, (c0 + c1) * (o.c0 + o.c1) - a - b};
    }
    Fq2 sqr() const { return {(c0 + c1) * (c0 - c1), (c0 * c1).dbl()}; }
    Fq2 dbl() const { return {c0.dbl(), c1.dbl()}; }
    Fq2 neg() const { return {c0.neg(), c1.neg()}; }
    Fq2 conj() const { return {c0, c1.neg()}; }
    Fq2 scale(const Fq& k) const { return {c0 * k, c1 * k}; }
    Fq norm() const { return c0.sqr() + c1.sqr(); }
    Fq2 inv() const { Fq t = norm().inv(); return {c0 * t, (c1 * t).neg()}; }
    // multiply by xi = 9 + u
    Fq2 mul_xi() const {
        Fq t0 = c0.dbl().dbl().dbl() + c0, t1 = c1.dbl().dbl().dbl() + c1;
        return {t0 - c1, t1 + c0};
    }
    bool is_zero() const { return c0.is_zero() && c1.is_zero(); }
    bool operator==(const Fq2& o) const { return c0 == o.c0 && c1 == o.c1; }
    Fq2 pow(const u64* e, int nlimbs) const {
        Fq2 r = one();
        for (int i = nlimbs * 64 - 1; i >= 0; --i) {
            r = r.sqr();
            if ((e[i / 64] >> (i % 64)) & 1) r = r * *this;
        }
        return r;
    }
};

// ----------------------------------------------------------------------------------- Fq6
struct Fq6 {  // c0 + c1 v + c2 v^2, v^3 = xi
    Fq2 c0, c1, c2;
    static Fq6 zero() { return {Fq2::zero(), Fq2::zero(), Fq2::zero()}; }
    static Fq6 one() { return {Fq2::one(), Fq2::zero(), Fq2::zero()}; }
    Fq6 operator+(const Fq6& o) const { return {c0 + o.c0, c1 + o.c1, c2 + o.c2}; }
    Fq6 operator-(const Fq6& o) const { return {c0 - o.c0, c1 - o.c1, c2 - o.c2}; }
    Fq6 neg() const { return {c0.neg(), c1.neg(), c2.neg()}; }
    Fq6 operator*(const Fq6& o) const {
        Fq2 a = c0 * o.c0, b = c1 * o.c1, c = c2 * o.c2;
        Fq2 t0 = ((c1 + c2) * (o.c1 + o.c2) - b - c).mul_xi() + a;
        Fq2 t1 = (c0 + c1) * (o.c0 + o.c1) - a - b + c.mul_xi();
        Fq2 t2 = (c0 + c2) * (o.c0 + o.c2) - a - c + b;
        return {t0, t1, t2};
    }
    Fq6 sqr() const { return *this * *this; }
    Fq6 mul_v() const { return {c2.mul_xi(), c0, c1}; }
    Fq6 inv() const {
        Fq2 A = c0.sqr() - (c1 * c2).mul_xi();
        Fq2 B = c2.sqr().mul_xi() - c0 * c1;
        Fq2 Cc = c1.sqr() - c0 * c2;
        Fq2 F = (c0 * A + (c2 * B + c1 * Cc).mul_xi()).inv();
        return {A * F, B * F, Cc * F};
    }
    bool operator==(const Fq6& o) const { return c0 == o.c0 && c1 == o.c1 && c2 == o.c2; }
};

// ----------------------------------------------------------------------------------- Fq12
struct Fq12Consts {
    Fq2 gamma1[6];  // xi^(i (p-1)/6), i = 0..5   (Frobenius coefficients, basis 1,w,..,w^5)
};
const Fq12Consts& fq12_consts();

struct Fq12 {  // c0 + c1 w, w^2 = v
    Fq6 c0, c1;
    static Fq12 one() { return {Fq6::one(), Fq6::zero()}; }
    Fq12 operator*(const Fq12& o) const {
        Fq6 a = c0 * o.c0, b = c1 * o.c1;
        return {a + b.mul_v(), (c0 + c1) * (o.c0 + o.c1) - a - b};
    }
    Fq12 sqr() const { return *this * *this; }
    Fq12 conj() const { return {c0, c1.neg()}; }
    Fq12 inv() const {
        Fq6 t = (c0.sqr() - c1.sqr().mul_v()).inv();
        return {c0 * t, (c1 * t).neg()};
    }
    bool operator==(const Fq12& o) const { return c0 == o.c0 && c1 == o.c1; }
    bool is_one() const { return *this == one(); }
    // x -> x^p.  In the basis w^i (i=0..5 with coefficient order c0.c0,c1.c0,c0.c1,c1.c1,c0.c2,c1.c2)
    // Frobenius conjugates each Fq2 coefficient and scales the w^i one by gamma1[i].
    Fq12 frob() const {
        const Fq12Consts& k = fq12_consts();
        Fq12 r;
        r.c0.c0 = c0.c0.conj();
        r.c1.c0 = c1.c0.conj() * k.gamma1[1];
        r.c0.c1 = c0.c1.conj() * k.gamma1[2];
        r.c1.c1 = c1.c1.conj() * k.gamma1[3];
        r.c0.c2 = c0.c2.conj() * k.gamma1[4];
        r.c1.c2 = c1.c2.conj() * k.gamma1[5];
        return r;
    }
    Fq12 pow_u64(u64 e) const {
        Fq12 r = one();
        for (int i = 63; i >= 0; --i) {
            r = r.sqr();
            if ((e >> i) & 1) r = r * *this;
        }
        return r;
    }
};

inline Fq2 fq2_xi() { return {Fq::from_u64(9), Fq::from_u64(1)}; }

inline const Fq12Consts& fq12_consts() {
    static Fq12Consts c = [] {
        Fq12Consts k;
        // e = (p-1)/6 by long division
        u64 pm1[4]; memcpy(pm1, Fq::C().p, 32); pm1[0] -= 1;
        u64 e[4]; u64 rem = 0;
        for (int i = 3; i >= 0; --i) {
            u128 cur = ((u128)rem << 64) | pm1[i];
            e[i] = (u64)(cur / 6); rem = (u64)(cur % 6);
        }
        Fq2 g = fq2_xi().pow(e, 4);
        k.gamma1[0] = Fq2::one();
        for (int i = 1; i < 6; ++i) k.gamma1[i] = k.gamma1[i - 1] * g;
        return k;
    }();
    return c;
}

// ----------------------------------------------------------------------------------- G2
struct G2Affine {
    Fq2 x, y;
    bool inf;
    static Fq2 b() {  // 3 / xi
        Fq2 three = {Fq::from_u64(3), Fq::zero()};
        return three * fq2_xi().inv();
    }
    bool on_curve() const { return inf || y.sqr() == x.sqr() * x + b(); }
    G2Affine neg() const { G2Affine r = *this; r.y = y.neg(); return r; }
    bool operator==(const G2Affine& o) const {
        if (inf || o.inf) return inf && o.inf;
        return x == o.x && y == o.y;
    }
    // affine chord-and-tangent (slow; test helper only — the pairing uses projective steps)
    G2Affine add(const G2Affine& o) const {
        if (inf) return o;
        if (o.inf) return *this;
        Fq2 lam;
        if (x == o.x) {
            if (!(y == o.y) || y.is_zero()) { G2Affine r = *this; r.inf = true; return r; }
            Fq2 x2 = x.sqr();
            lam = (x2.dbl() + x2) * y.dbl().inv();
        } else {
            lam = (o.y - y) * (o.x - x).inv();
        }
        G2Affine r; r.inf = false;
        r.x = lam.sqr() - x - o.x;
        r.y = lam * (x - r.x) - y;
        return r;
    }
    G2Affine mul(const Fr& k) const {
        u64 e[4]; k.to_limbs(e);
        G2Affine r; r.inf = true; r.x = Fq2::zero(); r.y = Fq2::zero();
        for (int i = 255; i >= 0; --i) {
            r = r.add(r);
            if ((e[i / 64] >> (i % 64)) & 1) r = r.add(*this);
        }
        return r;
    }
};

// The standard BN254 G2 generator; pinned against the `g2` field of the reference SRS fixture
// (tests/test_oracle_srs_kat.py).
G2Affine g2_generator();

// ----------------------------------------------------------------------------------- pairing
// Line coefficients of the Miller loop for a fixed G2 point (the role of halo2curves'
// G2Prepared, poly/kzg/msm.rs:186-187).
struct G2Prepared {
    struct Coeff { Fq2 a, b, c; };  // line = a * yP  +  b * xP * w  +  c * v w   (sparse 0,3,4)
    std::vector<Coeff> coeffs;
    bool inf;
    explicit G2Prepared(const G2Affine& q);
};

Fq12 multi_miller_loop(const G1Affine* ps, const G2Prepared* const* qs, size_t n);
Fq12 final_exponentiation(const Fq12& f);
// e(a1,b1) * e(a2,b2) == 1 ?
bool pairing_product_is_one(const G1Affine& a1, const G2Affine& b1, const G1Affine& a2, const G2Affine& b2);

}  // namespace h2o
