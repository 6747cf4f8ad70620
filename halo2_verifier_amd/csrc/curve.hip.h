// BN254 G1 group law for gfx950 kernels (and the host side of the library).
// y^2 = x^3 + 3 over Fq; Jacobian coordinates (X/Z^2, Y/Z^3), identity <=> Z == 0.
// Affine points are stored as (x, y) in Montgomery form; (0, 0) is not on the curve and marks
// the identity.
//
// Replaces the halo2curves G1 arithmetic the reference reaches through
// MSMKZG::eval -> best_multiexp (poly/kzg/msm.rs:81-86, arithmetic.rs:7-108) and through
// G1Affine::from_bytes (transcript/mod.rs:158-166).
#pragma once
#include "bn254.hip.h"

namespace h2v {

// compressed-point flag bits (SURVEY.md §8c "Unpinned detail"): byte 31 bit 7 = identity, bit 6 = sign(y)
static constexpr uint8_t G1_FLAG_IDENTITY = 0x80;
static constexpr uint8_t G1_FLAG_SIGN = 0x40;

struct G1A {
    Fq x, y;
    H2V_HD bool is_identity() const { return x.is_zero() && y.is_zero(); }
    H2V_HD static G1A identity() { G1A p; p.x = Fq::zero(); p.y = Fq::zero(); return p; }
    H2V_HD bool on_curve() const {
        if (is_identity()) return true;
        Fq three = Fq::from_u32(3);
        return y.sqr() == x.sqr() * x + three;
    }
};

struct G1J {
    Fq X, Y, Z;
    H2V_HD static G1J identity() { G1J p; p.X = Fq::zero(); p.Y = Fq::one(); p.Z = Fq::zero(); return p; }
    H2V_HD bool is_identity() const { return Z.is_zero(); }
    H2V_HD static G1J from_affine(const G1A& a) {
        if (a.is_identity()) return identity();
        G1J p; p.X = a.x; p.Y = a.y; p.Z = Fq::one(); return p;
    }
    H2V_HD G1J neg() const { G1J r = *this; r.Y = Y.neg(); return r; }
};

// The group law routines come in two forms: *_inl bodies that hot kernels inline so that the accumulator stays in
// registers, and plain functions (real calls) for everything else.  Field products are inlined in both (see Fp::mul).
H2V_FN G1J g1_dbl(const G1J& p);
#define H2V_M(a, b) Fq::mul_inl((a), (b))
#define H2V_S(a) (a).sqr_inl()
// -8 in Montgomery form (29-bit limbs, R = 2^261): the constant of Y3 = E (D - X3) - 8 C as the second term of a dot2
__host__ __device__ __forceinline__ Fq g1_minus_eight() {
    const Fq k = {{0x1d9096cdu, 0x022be75eu, 0x12c66350u, 0x1e1f47a4u, 0x0b760e9fu, 0x1e7fbbcfu, 0x134a4383u, 0x01be3557u, 0x0022eb3au}};
    return k;
}
// dbl-2009-l for a = 0 on the lazy linear forms (bn254.hip.h): A = X^2, B = Y^2, C = B^2, D = 4 X B (a product instead of
// 2((X + B)^2 - A - C): 36 multiply-adds more, three corrected additions fewer), E = 3A, X3 = E^2 - 2D through ONE carry sweep
// (from_wide), Y3 = E (D + 2p - X3) - 8C as one dot2, Z3 = (2Y) Z.  Bounds for coordinates below 2p, in multiples of p: A, B, C, XB < 1.03;
// E < 3.1; D < 4.2; E^2 + 9p - 2D in (0.7, 10.1) -> X3 < 1.0001; D + 2p - X3 < 6.2; Y3 < 1.13; Z3 < 1.05.  13 corrected additions
// (~65 instructions each) became 5 lazy ones and a sweep: 2295 -> ~1700 instructions.  The identity (Z = 0) stays the identity.
__host__ __device__ __forceinline__ G1J g1_dbl_inl(const G1J& p) {
    if (p.is_identity()) return p;
    const Fq A = H2V_S(p.X), B = H2V_S(p.Y), XB = H2V_M(p.X, B), C = H2V_S(B);
    const Fq E = Fq::lazy_add2(A, A), F = H2V_S(E), D = Fq::lazy_dbl(Fq::lazy_dbl(XB));
    int64_t acc[9];
#pragma unroll
    for (int l = 0; l < 9; ++l) acc[l] = (int64_t)F.v[l] + (int64_t)Fq::KP29(9, l) - 2 * (int64_t)D.v[l];
    G1J r;
    r.X = Fq::from_wide(acc);
    r.Y = Fq::dot2_inl(E, Fq::lazy_sub(D, r.X), C, g1_minus_eight());
    r.Z = H2V_M(Fq::lazy_dbl(p.Y), p.Z);
    return r;
}

// In-place forms for the hot loops of the MSM kernels.  They return false — leaving the accumulator untouched — in
// the two degenerate cases (equal or opposite points), which the caller handles on a slow path OUTSIDE its loop: a
// call inside the loop would take the accumulator's address and force it out of registers into scratch memory.
// Both use the lazy linear forms of bn254.hip.h and fold every difference of products into one reduction (dot2):
//   H = U2 - U1, I = (2H)^2, r = 2(S2 - S1),  X3 = r^2 - I (H + 2 U1),  Y3 = I (r U1 - 2 S1 H) - r X3,  Z3 = 2 Z1 Z2 H
// (add-2007-bl with J = H I and V = U1 I multiplied out).  Bounds, in multiples of p, for coordinates below 2: U, S < 1.03;
// H < 3.03; 2H, r < 6.1; I < 1.23 (so "I is zero" is the limb string 0 or p, and I = 0 <=> the x coordinates agree);
// 8p - H - 2 U1 in (0.9, 8); 4p - 2 S1 <= 4; X3 < 1.28; r U1 - 2 S1 H < 1.15; Y3 < 1.09; Z3 < 1.08 — the results are ordinary
// representatives below 2p again, nothing downstream sees the lazy values.
__host__ __device__ __forceinline__ bool g1_madd_fast(G1J& acc, const G1A& q) {
    if (q.is_identity()) return true;
    if (acc.is_identity()) { acc.X = q.x; acc.Y = q.y; acc.Z = Fq::one(); return true; }
    const Fq Z1Z1 = H2V_S(acc.Z);
    const Fq U2 = H2V_M(q.x, Z1Z1), S2 = H2V_M(H2V_M(q.y, acc.Z), Z1Z1);
    const Fq H = Fq::lazy_sub(U2, acc.X), H2 = Fq::lazy_dbl(H), I = H2V_S(H2);
    if (I.is_zero()) return false;
    const Fq B = Fq::lazy_neg2(acc.Y), rr = Fq::lazy_add2(B, S2);
    const Fq X3 = Fq::dot2_inl(rr, rr, I, Fq::template lazy_lin<8, 1, 2>(H, acc.X));
    const Fq W = Fq::dot2_inl(rr, acc.X, H, B);
    acc.Y = Fq::dot2_inl(I, W, rr, Fq::lazy_neg(X3));
    acc.Z = H2V_M(acc.Z, H2);
    acc.X = X3;
    return true;
}
__host__ __device__ __forceinline__ bool g1_add_fast(G1J& acc, const G1J& q) {
    if (q.is_identity()) return true;
    if (acc.is_identity()) { acc.X = q.X; acc.Y = q.Y; acc.Z = q.Z; return true; }
    const Fq Z1Z1 = H2V_S(acc.Z), Z2Z2 = H2V_S(q.Z);
    const Fq U1 = H2V_M(acc.X, Z2Z2), U2 = H2V_M(q.X, Z1Z1);
    const Fq H = Fq::lazy_sub(U2, U1), H2 = Fq::lazy_dbl(H), I = H2V_S(H2);
    if (I.is_zero()) return false;
    const Fq S1 = H2V_M(H2V_M(acc.Y, q.Z), Z2Z2), S2 = H2V_M(H2V_M(q.Y, acc.Z), Z1Z1);
    const Fq B = Fq::lazy_neg2(S1), rr = Fq::lazy_add2(B, S2);
    const Fq X3 = Fq::dot2_inl(rr, rr, I, Fq::template lazy_lin<8, 1, 2>(H, U1));
    const Fq W = Fq::dot2_inl(rr, U1, H, B);
    acc.Y = Fq::dot2_inl(I, W, rr, Fq::lazy_neg(X3));
    acc.Z = H2V_M(H2V_M(acc.Z, q.Z), H2);
    acc.X = X3;
    return true;
}
// The complete additions: the fast forms above, and behind their `false` the two cases they leave out (rare: the sums of an MSM
// meet equal or opposite points only in adversarial inputs — which the tests construct).
__host__ __device__ __forceinline__ G1J g1_add_inl(const G1J& p, const G1J& q) {
    G1J r = p;
    if (g1_add_fast(r, q)) return r;
    // the x coordinates agree: the same point (double it) or opposite points
    const Fq Z1Z1 = H2V_S(p.Z), Z2Z2 = H2V_S(q.Z);
    const Fq S1 = H2V_M(H2V_M(p.Y, q.Z), Z2Z2), S2 = H2V_M(H2V_M(q.Y, p.Z), Z1Z1);
    if (S1 == S2) return g1_dbl(r);   // (a copy is passed so that the caller's accumulator never has its address taken: it stays in registers)
    return G1J::identity();
}

__host__ __device__ __forceinline__ G1J g1_add_affine_inl(const G1J& p, const G1A& q) {
    G1J r = p;
    if (g1_madd_fast(r, q)) return r;
    const Fq Z1Z1 = H2V_S(p.Z);
    const Fq S2 = H2V_M(H2V_M(q.y, p.Z), Z1Z1);
    if (p.Y == S2) return g1_dbl(r);
    return G1J::identity();
}

H2V_FN G1J g1_dbl(const G1J& p) { return g1_dbl_inl(p); }
// x -> beta x, the x coordinate of the GLV endomorphism phi(x, y) = (beta x, y).  beta = the cube root of unity
// 0x30644e72e131a0295e6dd9e7e0acccb0c28f069fbb966e3de4bd44e5607cfd48 in Montgomery form (29-bit limbs, R = 2^261); multiplied inline:
// through the out-of-line Fp::mul the constant travelled as a stack argument (36 bytes of scratch written and read back per call).
__host__ __device__ __forceinline__ Fq g1_beta_times(const Fq& x) {
    const Fq beta = {{0x18ccb791u, 0x175b1c3au, 0x0b83d6e2u, 0x0e8ed071u, 0x1282bee2u, 0x04220e84u, 0x1fe4017fu, 0x15084d4au, 0x00169119u}};
    return Fq::mul_inl(x, beta);
}
__host__ __device__ __forceinline__ G1A g1_phi(const G1A& p) { G1A r; r.x = g1_beta_times(p.x); r.y = p.y; return r; }   // (the identity (0, 0) stays the identity)
H2V_FN G1J g1_add(const G1J& p, const G1J& q) { return g1_add_inl(p, q); }
H2V_FN G1J g1_add_affine(const G1J& p, const G1A& q) { return g1_add_affine_inl(p, q); }

H2V_FN G1A g1_to_affine(const G1J& p) {
    if (p.is_identity()) return G1A::identity();
    Fq zi = p.Z.inv(), zi2 = zi.sqr();
    G1A a; a.x = p.X * zi2; a.y = p.Y * zi2 * zi;
    return a;
}

// sqrt in Fq for p = 3 mod 4: a^((p+1)/4); caller checks r^2 == a.
// The exponent is a constant, so the whole sliding-window schedule (which odd power multiplies in after which run of
// squarings) is computed at compile time and the loop below unrolls into straight-line calls: the table of odd powers
// is indexed by constants only and stays in registers — a run-time-indexed table would live in scratch memory.
struct FqSqrtSchedule { uint8_t op[400]; int n; };  // 0: square; k > 0: multiply by a^(2k-1)
template <int W> constexpr FqSqrtSchedule fq_sqrt_schedule() {   // sliding windows of at most W bits
    // (p+1)/4 = 0x0c19139cb84c680a6e14116da060561765e05aa45a1c72a34f082305b61f3f52
    constexpr uint32_t e[8] = {0xb61f3f52u, 0x4f082305u, 0x5a1c72a3u, 0x65e05aa4u, 0xa0605617u, 0x6e14116du, 0xb84c680au, 0x0c19139cu};
    FqSqrtSchedule s{};
    s.n = 0;
    int i = 255;
    while (i >= 0 && !((e[i >> 5] >> (i & 31)) & 1)) --i;
    while (i >= 0) {
        if (!((e[i >> 5] >> (i & 31)) & 1)) { s.op[s.n++] = 0; --i; continue; }
        int l = i - (W - 1) < 0 ? 0 : i - (W - 1);           // window of at most W bits ending in a set bit
        while (!((e[l >> 5] >> (l & 31)) & 1)) ++l;
        uint32_t v = 0;
        for (int k = i; k >= l; --k) { v = (v << 1) | ((e[k >> 5] >> (k & 31)) & 1); s.op[s.n++] = 0; }
        s.op[s.n++] = (uint8_t)((v + 1) / 2);
        i = l - 1;
    }
    return s;
}
// (Host code and the odd device caller use this straight-line form; k_decompress runs fq_sqrt_candidate_loop_w3 below — W = 3: a table
// of four odd powers, 36 registers instead of 72, ten more products in ~320; the 4-bit table does not fit four waves per SIMD.)
template <int W> __host__ __device__ inline __attribute__((noinline)) Fq fq_sqrt_candidate_w(const Fq& a) {
    constexpr FqSqrtSchedule S = fq_sqrt_schedule<W>();
    constexpr int T = 1 << (W - 1);
    Fq t[T];
    t[0] = a;
    const Fq a2 = a.sqr();
#pragma unroll
    for (int k = 1; k < T; ++k) t[k] = t[k - 1] * a2;
    Fq r = Fq::one();
    bool started = false;  // squarings of the leading 1 are skipped (resolved at compile time once unrolled)
#pragma unroll
    for (int k = 0; k < S.n; ++k) {
        if (S.op[k] == 0) { if (started) r = r.sqr(); }
        else { r = started ? r * t[S.op[k] - 1] : t[S.op[k] - 1]; started = true; }
    }
    return r;
}
H2V_FN Fq fq_sqrt_candidate(const Fq& a) { return fq_sqrt_candidate_w<4>(a); }

// The same exponentiation as a LOOP over runs of the schedule — "s squarings, then multiply by table entry i" — with ONE inlined squaring
// and ONE inlined product in its body (device only; k_decompress).  The straight-line form above calls the out-of-line product 312 times
// and the compiler spends 43 instructions per call site on moving the nine limbs in and out of the argument registers: 13 500 of the
// ~79 000 instructions of a square root.  Here the value stays in its registers; the table entry is picked by selects (the run table is
// wave-uniform: scalar loads).
struct FqSqrtRuns { uint8_t nsq[160]; uint8_t idx[160]; int n; int first; };   // idx 255: no multiplication behind the run's squarings
template <int W> constexpr FqSqrtRuns fq_sqrt_runs() {
    constexpr FqSqrtSchedule S = fq_sqrt_schedule<W>();
    FqSqrtRuns r{};
    r.n = 0; r.first = -1;
    int pending = 0;
    for (int k = 0; k < S.n; ++k) {
        if (S.op[k] == 0) { if (r.first >= 0) ++pending; continue; }       // squarings of the leading 1 are skipped
        if (r.first < 0) { r.first = S.op[k] - 1; continue; }              // the first window: r = t[first]
        r.nsq[r.n] = (uint8_t)pending; r.idx[r.n] = (uint8_t)(S.op[k] - 1); ++r.n; pending = 0;
    }
    if (pending) { r.nsq[r.n] = (uint8_t)pending; r.idx[r.n] = 255; ++r.n; }
    return r;
}
#if defined(__HIPCC__)
__device__ __constant__ const FqSqrtRuns fq_sqrt_runs_w3 = fq_sqrt_runs<3>();
__device__ __forceinline__ Fq fq_sqrt_candidate_loop_w3(const Fq& a) {
    Fq t0 = a;
    const Fq a2 = a.sqr_inl();
    const Fq t1 = Fq::mul_inl(t0, a2), t2 = Fq::mul_inl(t1, a2), t3 = Fq::mul_inl(t2, a2);   // a, a^3, a^5, a^7
    auto pick = [&](uint32_t i) -> Fq {
        Fq s;
#pragma unroll
        for (int l = 0; l < H2V_LIMBS; ++l) s.v[l] = i == 0 ? t0.v[l] : (i == 1 ? t1.v[l] : (i == 2 ? t2.v[l] : t3.v[l]));
        return s;
    };
    Fq r = pick((uint32_t)fq_sqrt_runs_w3.first);
    const int n = fq_sqrt_runs_w3.n;
#pragma unroll 1
    for (int k = 0; k < n; ++k) {
        const uint32_t nsq = fq_sqrt_runs_w3.nsq[k], idx = fq_sqrt_runs_w3.idx[k];
#pragma unroll 1
        for (uint32_t i = 0; i < nsq; ++i) r = r.sqr_inl();
        // (idx is wave-uniform: four copies of the product behind scalar branches instead of 27 selects in front of one)
        if (idx == 0u) r = Fq::mul_inl(r, t0);
        else if (idx == 1u) r = Fq::mul_inl(r, t1);
        else if (idx == 2u) r = Fq::mul_inl(r, t2);
        else if (idx == 3u) r = Fq::mul_inl(r, t3);
    }
    return r;
}
#endif

// G1Affine::from_bytes (compressed).  Returns false for an invalid encoding.
H2V_FN bool g1_decompress(const uint8_t in[32], G1A& out) {
    uint8_t tmp[32];
    for (int i = 0; i < 32; ++i) tmp[i] = in[i];
    bool is_inf = tmp[31] & G1_FLAG_IDENTITY, sign = tmp[31] & G1_FLAG_SIGN;
    tmp[31] &= 0x3f;
    Fq x;
    if (!Fq::from_bytes(tmp, x)) return false;
    if (is_inf) {
        if (!x.is_zero() || sign) return false;
        out = G1A::identity();
        return true;
    }
    Fq rhs = x.sqr() * x + Fq::from_u32(3);
    Fq y = fq_sqrt_candidate(rhs);
    if (y.sqr() != rhs) return false;
    if (y.is_odd() != sign) y = y.neg();
    out.x = x; out.y = y;
    return true;
}

}  // namespace h2v
