// BN254 prime-field arithmetic for gfx950 (and the host side of the same library).
//
// Representation: 9 limbs of 29 bits (value = sum v[i] * 2^(29 i)), Montgomery form with R = 2^261, kept in [0, 2p)
// ("lazily" reduced: the canonical representative is produced only where bytes or comparisons need it).
//
// Why 29-bit limbs: CDNA4 has no 64x64->128 multiply and no multiply-add with carry-in; the widest step is
// v_mad_u64_u32 (32x32 + 64 -> 64).  With 32-bit limbs every such step needs its addend zero-extended and its carry
// split off again — the compiled 8 x 32 CIOS product is 128 multiply-adds buried in ~400 moves and 64-bit adds.  With
// 29-bit limbs a whole column of the product, a_i*b_j and m_i*p_j alike (18 terms < 2^58), fits a 64-bit accumulator,
// so the product is 162 back-to-back v_mad_u64_u32 into one register pair plus a shift and a mask per column: ~235
// instructions instead of ~530, 2.2x faster on a lone wave and 2x at full occupancy (tools/limb29_microbench.hip).
// R = 2^261 > 32p also removes the final conditional subtraction: inputs < 2p give outputs < 1.04p.
// Additions cost a little more (no hardware carry chain across 29-bit limbs), but the kernels are product-bound.
//
// Both Fq (curve coordinates) and Fr (scalars, challenges, evaluations) use this template.
//
// This is the product's own arithmetic; it shares no code with oracle/ (which uses 4 x 64-bit
// limbs and __int128 on the CPU).  The functions are __host__ __device__ so that the host
// side of the library (VK ingestion, plan compilation, G2 line precomputation) uses exactly
// the arithmetic the kernels use.
//
// Reference call sites this replaces (the reference gets them from the un-vendored halo2curves
// crate): Fr mul/add/sub/invert/pow throughout lib.rs and plonk/*.rs;
// Fr::from_uniform_bytes (transcript/mod.rs:500-514); Fq arithmetic under G1 add/double
// (poly/kzg/msm.rs:81-86) and under the pairing (poly/kzg/msm.rs:185-203).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// Small helpers are plain inline; the multi-hundred-instruction bodies (Montgomery product,
// exponentiation, group law, tower products) are real function calls on the device: inlining
// them everywhere makes kernels like the pairing explode in code size and compile time, while a
// call costs a few dozen cycles against a ~250-instruction body.
#define H2V_HD __host__ __device__ inline
#define H2V_FN __host__ __device__ inline __attribute__((noinline))

namespace h2v {

#define H2V_LIMBS 9
#define H2V_LIMB_BITS 29
#define H2V_LIMB_MASK 0x1fffffffu

// P: the modulus as 8 x 32-bit words (byte-level canonicity checks, exponents); everything else in 29-bit limbs:
// P29 = p, ONE = R mod p, R2 = R^2 mod p, M256 = 2^256 * R mod p (the Montgomery form of 2^256),
// K266 = 2^266 mod p (turns a 2^256-Montgomery residue, halo2curves' RawBytes limbs, into this representation).
struct FqParams {
    static constexpr uint32_t INV29 = 0x04866389u;  // -p^{-1} mod 2^29
    H2V_HD static constexpr uint32_t P(int i) {
        constexpr uint32_t p[8] = {0xd87cfd47u, 0x3c208c16u, 0x6871ca8du, 0x97816a91u, 0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u};
        return p[i];
    }
    H2V_HD static constexpr uint32_t P29(int i) {
        constexpr uint32_t v[9] = {0x187cfd47u, 0x010460b6u, 0x1c72a34fu, 0x02d522d0u, 0x1585d978u, 0x02db40c0u, 0x00a6e141u, 0x0e5c2634u, 0x0030644eu};
        return v[i];
    }
    H2V_HD static constexpr uint32_t ONE(int i) {
        constexpr uint32_t v[9] = {0x157ccc21u, 0x141c2758u, 0x185230d3u, 0x014c0419u, 0x0aa36fb9u, 0x1d4240ceu, 0x11d54c07u, 0x052ac7a8u, 0x000dc836u};
        return v[i];
    }
    H2V_HD static constexpr uint32_t R2(int i) {
        constexpr uint32_t v[9] = {0x059bac10u, 0x0d1503a3u, 0x018016b8u, 0x10ab0ca8u, 0x02632639u, 0x02c0169fu, 0x169bfd53u, 0x11869d4cu, 0x002a11a6u};
        return v[i];
    }
    H2V_HD static constexpr uint32_t M256(int i) {
        constexpr uint32_t v[9] = {0x0f6b5c04u, 0x08ead878u, 0x1645525du, 0x1aefe9cdu, 0x09d605edu, 0x0483a115u, 0x0d08508bu, 0x0dba4804u, 0x001982b4u};
        return v[i];
    }
    H2V_HD static constexpr uint32_t K266(int i) {
        constexpr uint32_t v[9] = {0x13349ca1u, 0x1a5d84a8u, 0x0a3e5cacu, 0x100249e0u, 0x12b951e8u, 0x0e92d304u, 0x14cb95b3u, 0x041b9d3du, 0x00058003u};
        return v[i];
    }
};
struct FrParams {
    static constexpr uint32_t INV29 = 0x0fffffffu;
    H2V_HD static constexpr uint32_t P(int i) {
        constexpr uint32_t p[8] = {0xf0000001u, 0x43e1f593u, 0x79b97091u, 0x2833e848u, 0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u};
        return p[i];
    }
    H2V_HD static constexpr uint32_t P29(int i) {
        constexpr uint32_t v[9] = {0x10000001u, 0x1f0fac9fu, 0x0e5c2450u, 0x07d090f3u, 0x1585d283u, 0x02db40c0u, 0x00a6e141u, 0x0e5c2634u, 0x0030644eu};
        return v[i];
    }
    H2V_HD static constexpr uint32_t ONE(int i) {
        constexpr uint32_t v[9] = {0x0fffff57u, 0x1ea70ab4u, 0x052c068bu, 0x17504f49u, 0x0aa8075bu, 0x1d4240ceu, 0x11d54c07u, 0x052ac7a8u, 0x000dc836u};
        return v[i];
    }
    H2V_HD static constexpr uint32_t R2(int i) {
        constexpr uint32_t v[9] = {0x05b69bd4u, 0x06170a5au, 0x020cddceu, 0x1db6310bu, 0x0e54d0ffu, 0x1cf855e3u, 0x1c15e103u, 0x07d09161u, 0x000a054au};
        return v[i];
    }
    H2V_HD static constexpr uint32_t M256(int i) {
        constexpr uint32_t v[9] = {0x142db4dfu, 0x19d6990eu, 0x1472f48cu, 0x06dbe7e3u, 0x0b84d579u, 0x10f9faf7u, 0x121f4380u, 0x17a112deu, 0x001275c7u};
        return v[i];
    }
    H2V_HD static constexpr uint32_t K266(int i) {
        constexpr uint32_t v[9] = {0x0fffead7u, 0x1d5444f4u, 0x04438aa5u, 0x03b4d096u, 0x134c84dau, 0x0e92d304u, 0x14cb95b3u, 0x041b9d3du, 0x00058003u};
        return v[i];
    }
};

// ---------------------------------------------------------------------------------------------------------------------
// Modular inversion by "safegcd" divsteps (Bernstein-Yang 2019, in the constant-time formulation with signed 30-bit limbs
// that Wuille's proof-carrying implementation made standard): d, e, f, g start as 0, 1, p, x; one batch of 30 divsteps looks
// only at the low 30 bits of f and g and yields a 2 x 2 integer matrix t with [f, g] <- t [f, g] / 2^30 (exact) and
// [d, e] <- t [d, e] / 2^30 mod p; 590 divsteps bring any 256-bit g to 0, f to +-1 and d to +-x^-1 (20 batches = 600 here).
// Every step is branch-free and identical for all inputs — exactly what a wave wants: ~15 000 integer instructions for one
// inverse, against ~75 000 for the Fermat chain x^(p-2) (253 squarings + 64 products).  Inverses sit on the critical path of
// every latency-bound kernel (the Fr program's one batched inversion, the pairing's final exponentiation, the affine
// conversion of the accumulators).  inv(0) = 0.
struct Signed30 { int32_t v[9]; };   // value = sum v[i] 2^(30 i), limbs in (-2^30, 2^30)
template <class PR> struct ModInv30 {
    static constexpr int32_t M30 = (int32_t)(0xffffffffu >> 2);
    // p as 30-bit limbs, from the 32-bit words
    H2V_HD static constexpr int32_t P30(int i) {
        const int bit = 30 * i, w = bit >> 5, sh = bit & 31;
        if (w >= 8) return 0;
        uint64_t x = (uint64_t)PR::P(w) >> sh;
        if (w + 1 < 8) x |= (uint64_t)PR::P(w + 1) << (32 - sh);
        return (int32_t)(x & (uint64_t)M30);
    }
    // p^-1 mod 2^30 (Newton iteration on the low word; p is odd)
    H2V_HD static constexpr uint32_t PINV30() {
        uint32_t p0 = PR::P(0), x = p0;            // correct to 3 bits
        for (int i = 0; i < 5; ++i) x *= 2u - p0 * x;
        return x & (uint32_t)M30;
    }
    struct Trans { int32_t u, v, q, r; };
    // 30 divsteps on the low limbs; zeta = -(delta + 1/2)
    H2V_HD static int32_t divsteps_30(int32_t zeta, uint32_t f0, uint32_t g0, Trans& t) {
        uint32_t u = 1, v = 0, q = 0, r = 1, f = f0, g = g0;
#pragma unroll
        for (int i = 0; i < 30; ++i) {
            uint32_t mask1 = (uint32_t)(zeta >> 31);          // zeta < 0
            const uint32_t mask2 = 0u - (g & 1u);             // g odd
            const uint32_t x = (f ^ mask1) - mask1, y = (u ^ mask1) - mask1, z = (v ^ mask1) - mask1;   // conditionally negated f, u, v
            g += x & mask2; q += y & mask2; r += z & mask2;
            mask1 &= mask2;
            zeta = (int32_t)((uint32_t)zeta ^ mask1) - 1;     // -zeta - 2 or zeta - 1
            f += g & mask1; u += q & mask1; v += r & mask1;
            g >>= 1; u <<= 1; v <<= 1;
        }
        t.u = (int32_t)u; t.v = (int32_t)v; t.q = (int32_t)q; t.r = (int32_t)r;
        return zeta;
    }
    // [d, e] <- t [d, e] / 2^30 mod p, keeping both in (-2p, p)
    H2V_HD static void update_de(Signed30& d, Signed30& e, const Trans& t) {
        const int32_t u = t.u, v = t.v, q = t.q, r = t.r;
        const int32_t sd = d.v[8] >> 31, se = e.v[8] >> 31;
        int32_t md = (u & sd) + (v & se), me = (q & sd) + (r & se);
        int32_t di = d.v[0], ei = e.v[0];
        int64_t cd = (int64_t)u * di + (int64_t)v * ei, ce = (int64_t)q * di + (int64_t)r * ei;
        // choose md, me so that t [d, e] + p [md, me] has 30 zero bottom bits
        md -= (int32_t)((PINV30() * (uint32_t)cd + (uint32_t)md) & (uint32_t)M30);
        me -= (int32_t)((PINV30() * (uint32_t)ce + (uint32_t)me) & (uint32_t)M30);
        cd += (int64_t)P30(0) * md; ce += (int64_t)P30(0) * me;
        cd >>= 30; ce >>= 30;
#pragma unroll
        for (int i = 1; i < 9; ++i) {
            di = d.v[i]; ei = e.v[i];
            cd += (int64_t)u * di + (int64_t)v * ei; ce += (int64_t)q * di + (int64_t)r * ei;
            cd += (int64_t)P30(i) * md; ce += (int64_t)P30(i) * me;
            d.v[i - 1] = (int32_t)cd & M30; cd >>= 30;
            e.v[i - 1] = (int32_t)ce & M30; ce >>= 30;
        }
        d.v[8] = (int32_t)cd; e.v[8] = (int32_t)ce;
    }
    // [f, g] <- t [f, g] / 2^30 (the bottom 30 bits are zero by construction of t)
    H2V_HD static void update_fg(Signed30& f, Signed30& g, const Trans& t) {
        const int32_t u = t.u, v = t.v, q = t.q, r = t.r;
        int32_t fi = f.v[0], gi = g.v[0];
        int64_t cf = (int64_t)u * fi + (int64_t)v * gi, cg = (int64_t)q * fi + (int64_t)r * gi;
        cf >>= 30; cg >>= 30;
#pragma unroll
        for (int i = 1; i < 9; ++i) {
            fi = f.v[i]; gi = g.v[i];
            cf += (int64_t)u * fi + (int64_t)v * gi; cg += (int64_t)q * fi + (int64_t)r * gi;
            f.v[i - 1] = (int32_t)cf & M30; cf >>= 30;
            g.v[i - 1] = (int32_t)cg & M30; cg >>= 30;
        }
        f.v[8] = (int32_t)cf; g.v[8] = (int32_t)cg;
    }
    // r in (-2p, p), negated if sign < 0, brought to [0, p)
    H2V_HD static void normalize(Signed30& r, int32_t sign) {
        int32_t cond_add = r.v[8] >> 31;
        const int32_t cond_negate = sign >> 31;
#pragma unroll
        for (int i = 0; i < 9; ++i) { r.v[i] += P30(i) & cond_add; r.v[i] = (r.v[i] ^ cond_negate) - cond_negate; }
#pragma unroll
        for (int i = 0; i < 8; ++i) { r.v[i + 1] += r.v[i] >> 30; r.v[i] &= M30; }
        cond_add = r.v[8] >> 31;
#pragma unroll
        for (int i = 0; i < 9; ++i) r.v[i] += P30(i) & cond_add;
#pragma unroll
        for (int i = 0; i < 8; ++i) { r.v[i + 1] += r.v[i] >> 30; r.v[i] &= M30; }
    }
    // x (canonical integer < p, 8 words) -> x^-1 mod p (8 words); 0 -> 0
    H2V_FN static void invert(const uint32_t x[8], uint32_t out[8]) {
        Signed30 d, e, f, g;
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            const int bit = 30 * i, w = bit >> 5, sh = bit & 31;
            uint64_t t = w < 8 ? (uint64_t)x[w] >> sh : 0;
            if (w + 1 < 8) t |= (uint64_t)x[w + 1] << (32 - sh);
            g.v[i] = (int32_t)(t & (uint64_t)M30);
            f.v[i] = P30(i); d.v[i] = 0; e.v[i] = i == 0 ? 1 : 0;
        }
        int32_t zeta = -1;
        for (int it = 0; it < 20; ++it) {
            Trans t;
            zeta = divsteps_30(zeta, (uint32_t)f.v[0], (uint32_t)g.v[0], t);
            update_de(d, e, t);
            update_fg(f, g, t);
        }
        normalize(d, f.v[8]);
#pragma unroll
        for (int w = 0; w < 8; ++w) {
            const int bit = 32 * w, i = bit / 30, sh = bit % 30;
            uint64_t t = (uint64_t)(uint32_t)d.v[i] >> sh;
            if (i + 1 < 9) t |= (uint64_t)(uint32_t)d.v[i + 1] << (30 - sh);
            if (i + 2 < 9 && 60 - sh < 32) t |= (uint64_t)(uint32_t)d.v[i + 2] << (60 - sh);
            out[w] = (uint32_t)t;
        }
    }
};

template <class PR> struct Fp {
    uint32_t v[H2V_LIMBS];  // 29-bit limbs of a representative in [0, 2p) of (value * R) mod p

    H2V_HD static Fp zero() { Fp r; for (int i = 0; i < 9; ++i) r.v[i] = 0; return r; }
    H2V_HD static Fp one() { Fp r; for (int i = 0; i < 9; ++i) r.v[i] = PR::ONE(i); return r; }
    H2V_HD static Fp r2() { Fp r; for (int i = 0; i < 9; ++i) r.v[i] = PR::R2(i); return r; }

    // [0, 2p): zero is the limb string 0 or the limb string p
    H2V_HD bool is_zero() const {
        uint32_t o = 0, q = 0;
#pragma unroll
        for (int i = 0; i < 9; ++i) { o |= v[i]; q |= v[i] ^ PR::P29(i); }
        return o == 0 || q == 0;
    }
    H2V_HD bool operator==(const Fp& b) const { return (*this - b).is_zero(); }
    H2V_HD bool operator!=(const Fp& b) const { return !(*this == b); }

    // raw (non-Montgomery) 8 x 32-bit word comparison a >= p
    H2V_HD static bool geq_p(const uint32_t a[8]) {
        for (int i = 7; i >= 0; --i) {
            uint32_t p = PR::P(i);
            if (a[i] > p) return true;
            if (a[i] < p) return false;
        }
        return true;
    }
    // 256-bit integer as 8 words <-> 9 limbs (plain repacking, no arithmetic)
    H2V_HD static void pack29(uint32_t out[9], const uint32_t raw[8]) {
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            const int bit = 29 * i, w = bit >> 5, sh = bit & 31;
            uint32_t x = raw[w] >> sh;
            if (sh > 3 && w + 1 < 8) x |= raw[w + 1] << (32 - sh);
            out[i] = x & H2V_LIMB_MASK;
        }
    }
    H2V_HD static void unpack29(uint32_t raw[8], const uint32_t in[9]) {  // in < 2^256
#pragma unroll
        for (int w = 0; w < 8; ++w) {
            const int bit = 32 * w, i = bit / 29, sh = bit % 29;
            uint32_t x = in[i] >> sh;                       // 29 - sh bits
            x |= in[i + 1] << (29 - sh);                    // i + 1 <= 8 always: 32*7 / 29 = 7
            if (29 - sh + 29 < 32 && i + 2 < 9) x |= in[i + 2] << (58 - sh);
            raw[w] = x;
        }
    }
    // the canonical representative's limbs
    H2V_HD void canonical(uint32_t out[9]) const {
        int32_t u[9]; int32_t c = 0;
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            int32_t w = (int32_t)v[i] - (int32_t)PR::P29(i) + c;
            if (i < 8) { u[i] = w & (int32_t)H2V_LIMB_MASK; c = w >> 29; } else u[i] = w;
        }
        const bool neg = u[8] < 0;
#pragma unroll
        for (int i = 0; i < 9; ++i) out[i] = neg ? v[i] : (uint32_t)u[i];
    }

    // a + b and a - b for representatives < 2p, result < 2p: limb-wise with signed carries, one conditional +-2p
    H2V_HD Fp operator+(const Fp& b) const {
        uint32_t t[9]; int32_t u[9];
        uint32_t ct = 0; int32_t cu = 0;
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            const uint32_t s0 = v[i] + b.v[i];
            const uint32_t s = s0 + ct;
            const int32_t w = (int32_t)s0 - (int32_t)(2u * PR::P29(i)) + cu;
            if (i < 8) { t[i] = s & H2V_LIMB_MASK; ct = s >> 29; u[i] = w & (int32_t)H2V_LIMB_MASK; cu = w >> 29; }
            else { t[i] = s; u[i] = w; }
        }
        const bool neg = u[8] < 0;  // a + b < 2p
        Fp r;
#pragma unroll
        for (int i = 0; i < 9; ++i) r.v[i] = neg ? t[i] : (uint32_t)u[i];
        return r;
    }
    H2V_HD Fp operator-(const Fp& b) const {
        int32_t t[9], u[9];
        int32_t ct = 0, cu = 0;
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            const int32_t d = (int32_t)v[i] - (int32_t)b.v[i];
            const int32_t s = d + ct;
            const int32_t w = d + (int32_t)(2u * PR::P29(i)) + cu;
            if (i < 8) { t[i] = s & (int32_t)H2V_LIMB_MASK; ct = s >> 29; u[i] = w & (int32_t)H2V_LIMB_MASK; cu = w >> 29; }
            else { t[i] = s; u[i] = w; }
        }
        const bool neg = t[8] < 0;  // a < b: take a - b + 2p
        Fp r;
#pragma unroll
        for (int i = 0; i < 9; ++i) r.v[i] = (uint32_t)(neg ? u[i] : t[i]);
        return r;
    }
    H2V_HD Fp neg() const { return zero() - *this; }
    H2V_HD Fp dbl() const { return *this + *this; }

    // ---- Linear forms WITHOUT the correction step ("lazy").  The results are limb-normalised integers that may exceed 2p; they are
    // only ever handed to products: mul / sqr / dot2 take any limb-normalised operands (their 64-bit columns hold 27 limb products), and
    // operands below A p and B p give a product below (1 + A B p/R) p with p/R < 0.0060 — e.g. 8p times 8p still comes out below 1.4p.
    // A field addition or subtraction with its compare-and-select costs ≈ 65 instructions, these 27-45: in the group law below the
    // linear operations were 40 % of the instructions.  Every caller states the bounds of what it passes.
    // limb i of k p (k <= 64), normalised
    H2V_HD static constexpr uint32_t KP29(uint32_t k, int i) {
        uint64_t c = 0; uint32_t out = 0;
        for (int j = 0; j <= i; ++j) { const uint64_t t = (uint64_t)k * PR::P29(j) + c; out = j < 8 ? (uint32_t)(t & H2V_LIMB_MASK) : (uint32_t)t; c = t >> 29; }
        return out;
    }
    // k p - sa a - sb b as an integer (must be >= 0; sa, sb in {-2..2}: a negative coefficient adds), signed carries, no reduction
    template <uint32_t K, int SA, int SB> __host__ __device__ __forceinline__ static Fp lazy_lin(const Fp& a, const Fp& b) {
        static_assert(K <= 16 && SA >= -2 && SA <= 2 && SB >= -2 && SB <= 2, "limb sums must stay inside 32 bits (the limbs of k p are normalised), the value below 2^261");
        Fp r; int32_t c = 0;
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            const int32_t w = (int32_t)KP29(K, i) - SA * (int32_t)a.v[i] - SB * (int32_t)b.v[i] + c;
            if (i < 8) { r.v[i] = (uint32_t)w & H2V_LIMB_MASK; c = w >> 29; } else r.v[i] = (uint32_t)w;
        }
        return r;
    }
    __host__ __device__ __forceinline__ static Fp lazy_sub(const Fp& a, const Fp& b) { return lazy_lin<2, -1, 1>(a, b); }      // a + 2p - b, b <= 2p
    __host__ __device__ __forceinline__ static Fp lazy_neg(const Fp& b) { Fp z = zero(); return lazy_lin<2, 0, 1>(z, b); }      // 2p - b,     b <= 2p
    __host__ __device__ __forceinline__ static Fp lazy_neg2(const Fp& b) { Fp z = zero(); return lazy_lin<4, 0, 2>(z, b); }     // 4p - 2b,    b <= 2p
    __host__ __device__ __forceinline__ static Fp lazy_dbl(const Fp& a) { Fp z = zero(); return lazy_lin<0, -2, 0>(a, z); }     // 2a
    __host__ __device__ __forceinline__ static Fp lazy_add2(const Fp& a, const Fp& b) { return lazy_lin<0, -1, -2>(a, b); }     // a + 2b

    // Montgomery product a*b/R mod p by product scanning: column k collects a_i*b_(k-i) and m_i*p_(k-i) in one 64-bit
    // accumulator (<= 18 terms < 2^58 each, plus a carry < 2^35), m_k is chosen to clear the column's low 29 bits.
    // Any limb-normalised inputs are safe against overflow; representatives < 2p give a result < 1.04p.
    // (The compiler's reassociation starts every column from zero and adds the carry of the previous column last: a 64-bit add per
    // column, 143 per mixed addition beside its 1593 multiply-adds.  Forcing the chain to start FROM the carry (an empty asm statement
    // after each step) removes them, but every multiply-add that follows an asm statement gets an s_nop from the hazard recogniser and
    // the products no longer interleave: 4.69 instead of 4.63 us per list entry at three waves per SIMD, 4.79 instead of 5.13 at two
    // (tools/affine_microbench.hip, round 3).  Left to the compiler.)
    __host__ __device__ __forceinline__ static Fp mul_inl(const Fp& a, const Fp& b) {
        uint64_t acc = 0;
        uint32_t m[9];
        Fp r;
#pragma unroll
        for (int k = 0; k < 9; ++k) {
#pragma unroll
            for (int i = 0; i <= k; ++i) acc += (uint64_t)a.v[i] * b.v[k - i];
#pragma unroll
            for (int i = 0; i < k; ++i) acc += (uint64_t)m[i] * PR::P29(k - i);
            m[k] = ((uint32_t)acc * PR::INV29) & H2V_LIMB_MASK;
            acc += (uint64_t)m[k] * PR::P29(0);
            acc >>= 29;
        }
#pragma unroll
        for (int k = 9; k < 17; ++k) {
#pragma unroll
            for (int i = k - 8; i <= 8; ++i) acc += (uint64_t)a.v[i] * b.v[k - i];
#pragma unroll
            for (int i = k - 8; i <= 8; ++i) acc += (uint64_t)m[i] * PR::P29(k - i);
            r.v[k - 9] = (uint32_t)acc & H2V_LIMB_MASK;
            acc >>= 29;
        }
        r.v[8] = (uint32_t)acc;
        return r;
    }
    // A value given as signed per-limb accumulators, V = sum acc[l] * 2^(29 l) with 0 <= V < 2^261 (any integer linear
    // combination of a few residues, e.g. 9a + (2p - b)), brought to [0, 1.0001p) WITHOUT a Montgomery pass: one carry
    // sweep, then V - q p with q = floor(top limb / (top limb of p + 1)) — an under-estimate of floor(V / p) by at most
    // one (the top limbs are 29 and 22 bits: V - q p < p + (q + 1) 2^232), so the result is non-negative and below 2p.
    // q <= 2^29 / 2^21.6 < 170.  The division is a multiplication by floor(2^40 / d) with one correction step: exact.
    H2V_HD static Fp from_wide(const int64_t acc[9]) {
        uint32_t v[9];
        int64_t carry = 0;
#pragma unroll
        for (int l = 0; l < 9; ++l) {
            const int64_t t = acc[l] + carry;
            if (l < 8) { v[l] = (uint32_t)(t & (int64_t)H2V_LIMB_MASK); carry = t >> 29; } else v[l] = (uint32_t)t;
        }
        constexpr uint32_t d = PR::P29(8) + 1;
        constexpr uint64_t M = (uint64_t(1) << 40) / d;
        uint32_t q = (uint32_t)(((uint64_t)v[8] * M) >> 40);
        if (v[8] - q * d >= d) ++q;
        Fp r;
        carry = 0;
#pragma unroll
        for (int l = 0; l < 9; ++l) {
            const int64_t t = (int64_t)v[l] - (int64_t)q * (int64_t)PR::P29(l) + carry;
            if (l < 8) { r.v[l] = (uint32_t)(t & (int64_t)H2V_LIMB_MASK); carry = t >> 29; } else r.v[l] = (uint32_t)t;
        }
        return r;
    }
    // (a0*b0 + a1*b1)/R mod p in ONE reduction pass: 27 terms per column still fit the 64-bit accumulator (27 * 2^58 + carry
    // < 2^63).  Representatives < 2p give a result < 1.05p.  This is what makes an Fq2 product two passes and no additions.
    __host__ __device__ __forceinline__ static Fp dot2_inl(const Fp& a0, const Fp& b0, const Fp& a1, const Fp& b1) {
        uint64_t acc = 0;
        uint32_t m[9];
        Fp r;
#pragma unroll
        for (int k = 0; k < 9; ++k) {
#pragma unroll
            for (int i = 0; i <= k; ++i) { acc += (uint64_t)a0.v[i] * b0.v[k - i]; acc += (uint64_t)a1.v[i] * b1.v[k - i]; }
#pragma unroll
            for (int i = 0; i < k; ++i) acc += (uint64_t)m[i] * PR::P29(k - i);
            m[k] = ((uint32_t)acc * PR::INV29) & H2V_LIMB_MASK;
            acc += (uint64_t)m[k] * PR::P29(0);
            acc >>= 29;
        }
#pragma unroll
        for (int k = 9; k < 17; ++k) {
#pragma unroll
            for (int i = k - 8; i <= 8; ++i) { acc += (uint64_t)a0.v[i] * b0.v[k - i]; acc += (uint64_t)a1.v[i] * b1.v[k - i]; }
#pragma unroll
            for (int i = k - 8; i <= 8; ++i) acc += (uint64_t)m[i] * PR::P29(k - i);
            r.v[k - 9] = (uint32_t)acc & H2V_LIMB_MASK;
            acc >>= 29;
        }
        r.v[8] = (uint32_t)acc;
        return r;
    }
    // a^2/R mod p: the 36 off-diagonal products are taken once against doubled limbs (45 + 81 multiply-adds instead of 162)
    __host__ __device__ __forceinline__ Fp sqr_inl() const {
        uint64_t acc = 0;
        uint32_t m[9], d[9];
        Fp r;
#pragma unroll
        for (int i = 0; i < 9; ++i) d[i] = v[i] << 1;
#pragma unroll
        for (int k = 0; k < 9; ++k) {
#pragma unroll
            for (int i = 0; 2 * i < k; ++i) acc += (uint64_t)d[i] * v[k - i];
            if (k % 2 == 0) acc += (uint64_t)v[k / 2] * v[k / 2];
#pragma unroll
            for (int i = 0; i < k; ++i) acc += (uint64_t)m[i] * PR::P29(k - i);
            m[k] = ((uint32_t)acc * PR::INV29) & H2V_LIMB_MASK;
            acc += (uint64_t)m[k] * PR::P29(0);
            acc >>= 29;
        }
#pragma unroll
        for (int k = 9; k < 17; ++k) {
#pragma unroll
            for (int i = k - 8; 2 * i < k; ++i) acc += (uint64_t)d[i] * v[k - i];
            if (k % 2 == 0) acc += (uint64_t)v[k / 2] * v[k / 2];
#pragma unroll
            for (int i = k - 8; i <= 8; ++i) acc += (uint64_t)m[i] * PR::P29(k - i);
            r.v[k - 9] = (uint32_t)acc & H2V_LIMB_MASK;
            acc >>= 29;
        }
        r.v[8] = (uint32_t)acc;
        return r;
    }
    // The same product as a real function call (operands and result by value, i.e. in VGPRs).  Mid-level routines
    // (Fq2 products, the G1 group law) inline mul_inl so that their independent products can be interleaved by the
    // scheduler, while everything else calls this one copy to keep code size and compile time bounded.
    H2V_FN static Fp mul(Fp a, Fp b) { return mul_inl(a, b); }
    H2V_FN static Fp sqr_fn(Fp a) { return a.sqr_inl(); }
    H2V_HD Fp operator*(const Fp& b) const { return mul(*this, b); }
    H2V_HD Fp sqr() const { return sqr_fn(*this); }

    // integer < 2^256 (as words) -> Montgomery form of its residue
    H2V_HD static Fp from_raw(const uint32_t raw[8]) { Fp t; pack29(t.v, raw); return mul(t, r2()); }
    H2V_HD static Fp from_raw_unreduced(const uint32_t raw[8]) { return from_raw(raw); }  // 2^256 < 8p: a valid product operand as is
    // residue in 2^256-Montgomery form (words, < p) -> this representation: m * 2^266 / 2^261 = m * 2^5 = a * 2^261
    H2V_HD static Fp from_mont256(const uint32_t m[8]) { Fp t, k; pack29(t.v, m); for (int i = 0; i < 9; ++i) k.v[i] = PR::K266(i); return mul(t, k); }
    // canonical integer of the value, as words
    H2V_HD void to_raw(uint32_t out[8]) const {
        Fp o = zero(); o.v[0] = 1;
        Fp r = mul(*this, o);
        uint32_t c[9]; r.canonical(c);
        unpack29(out, c);
    }
    H2V_HD static Fp from_u32(uint32_t x) { uint32_t raw[8] = {x, 0, 0, 0, 0, 0, 0, 0}; return from_raw(raw); }

    // little-endian canonical bytes; returns false when the value is >= p (ff::PrimeField::from_repr)
    H2V_HD static bool from_bytes(const uint8_t b[32], Fp& out) {
        uint32_t raw[8];
        for (int i = 0; i < 8; ++i) raw[i] = (uint32_t)b[4 * i] | ((uint32_t)b[4 * i + 1] << 8) | ((uint32_t)b[4 * i + 2] << 16) | ((uint32_t)b[4 * i + 3] << 24);
        if (geq_p(raw)) return false;
        out = from_raw(raw);
        return true;
    }
    H2V_HD void to_bytes(uint8_t b[32]) const {
        uint32_t raw[8]; to_raw(raw);
        for (int i = 0; i < 8; ++i) { b[4 * i] = (uint8_t)raw[i]; b[4 * i + 1] = (uint8_t)(raw[i] >> 8); b[4 * i + 2] = (uint8_t)(raw[i] >> 16); b[4 * i + 3] = (uint8_t)(raw[i] >> 24); }
    }
    // 512-bit little-endian integer mod p (ff::FromUniformBytes<64>); w[0..15] = LE words
    H2V_HD static Fp from_uniform_words(const uint32_t w[16]) {
        Fp lo = from_raw(w), hi = from_raw(w + 8), m256;
        for (int i = 0; i < 9; ++i) m256.v[i] = PR::M256(i);
        return lo + hi * m256;
    }

    // x^e for a 32-bit exponent, MSB first.  The exponent is wave-uniform wherever the kernels use
    // it (plan constants), so there is no divergence.
    H2V_FN Fp pow_u32(uint32_t e) const {
        Fp r = one();
        bool started = false;
        for (int i = 31; i >= 0; --i) {
            if (started) r = r.sqr();
            if ((e >> i) & 1) { r = started ? r * *this : *this; started = true; }
        }
        return r;
    }
    // x^e for a 256-bit exponent given as limbs (uniform), 4-bit fixed window
    H2V_FN Fp pow_limbs(const uint32_t e[8]) const {
        Fp tbl[16];
        tbl[0] = one(); tbl[1] = *this;
        for (int i = 2; i < 16; ++i) tbl[i] = tbl[i - 1] * *this;
        Fp r = one();
        for (int i = 63; i >= 0; --i) {
            r = r.sqr().sqr().sqr().sqr();
            uint32_t d = (e[i / 8] >> (4 * (i % 8))) & 15u;
            if (d) r = r * tbl[d];
        }
        return r;
    }
    // Inverse by Fermat: x^(p-2), 4-bit fixed window (253 squarings + 64 products ~ 0.12 ms on a lone lane); inv(0) = 0.
    // (The first arithmetic of this library, 8 x 32-bit limbs, used a binary extended-GCD here because it beat ~320 products
    // of 1 us each; with the 29-bit product the exponentiation is the faster one — 0.12 ms against 0.29 ms measured in
    // k_point_to_bytes — and it is uniform across a wave.)
    H2V_FN Fp inv_fermat() const {
        uint32_t e[8];
        for (int i = 0; i < 8; ++i) e[i] = PR::P(i);
        e[0] -= 2;  // p is odd and its low limb is >= 2 for both fields
        return pow_limbs(e);
    }
    // Inverse by safegcd divsteps (ModInv30 above).  The value held is a = x R; a^-1 = x^-1 R^-1 as a plain integer; one Montgomery
    // product with R^3 brings it back: x^-1 R^-1 R^3 / R = x^-1 R.
    H2V_FN Fp inv_safegcd() const {
        uint32_t c[9], raw[8], o[8];
        canonical(c);
        unpack29(raw, c);
        ModInv30<PR>::invert(raw, o);
        Fp t; pack29(t.v, o);
        return mul(t, mul(r2(), r2()));
    }
    H2V_HD Fp inv() const { return inv_safegcd(); }
    H2V_HD bool is_odd() const { uint32_t raw[8]; to_raw(raw); return raw[0] & 1; }
};

typedef Fp<FqParams> Fq;
typedef Fp<FrParams> Fr;

}  // namespace h2v
