// BN254 prime-field arithmetic for gfx950 (and the host side of the same library).
//
// 8 x 32-bit limbs, Montgomery form (R = 2^256), fully reduced after every operation.
// CDNA4 has no 64x64->128 VALU multiply; the inner step is v_mad_u64_u32 (32x32 + 64).
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
// call costs a few dozen cycles against a ~400-instruction body.
#define H2V_HD __host__ __device__ inline
#define H2V_FN __host__ __device__ inline __attribute__((noinline))

namespace h2v {

struct FqParams {
    static constexpr uint32_t INV = 0xe4866389u;  // -p^{-1} mod 2^32
    H2V_HD static constexpr uint32_t P(int i) {
        constexpr uint32_t p[8] = {0xd87cfd47u, 0x3c208c16u, 0x6871ca8du, 0x97816a91u, 0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u};
        return p[i];
    }
    H2V_HD static constexpr uint32_t ONE(int i) {  // R mod p
        constexpr uint32_t v[8] = {0xc58f0d9du, 0xd35d438du, 0xf5c70b3du, 0x0a78eb28u, 0x7879462cu, 0x666ea36fu, 0x9a07df2fu, 0x0e0a77c1u};
        return v[i];
    }
    H2V_HD static constexpr uint32_t R2(int i) {  // R^2 mod p
        constexpr uint32_t v[8] = {0x538afa89u, 0xf32cfc5bu, 0xd44501fbu, 0xb5e71911u, 0x0a417ff6u, 0x47ab1effu, 0xcab8351fu, 0x06d89f71u};
        return v[i];
    }
    H2V_HD static constexpr uint32_t R3(int i) {  // R^3 mod p
        constexpr uint32_t v[8] = {0xda1530dfu, 0xb1cd6dafu, 0xa7283db6u, 0x62f210e6u, 0x0ada0afbu, 0xef7f0b0cu, 0x2d592544u, 0x20fd6e90u};
        return v[i];
    }
};
struct FrParams {
    static constexpr uint32_t INV = 0xefffffffu;
    H2V_HD static constexpr uint32_t P(int i) {
        constexpr uint32_t p[8] = {0xf0000001u, 0x43e1f593u, 0x79b97091u, 0x2833e848u, 0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u};
        return p[i];
    }
    H2V_HD static constexpr uint32_t ONE(int i) {
        constexpr uint32_t v[8] = {0x4ffffffbu, 0xac96341cu, 0x9f60cd29u, 0x36fc7695u, 0x7879462eu, 0x666ea36fu, 0x9a07df2fu, 0x0e0a77c1u};
        return v[i];
    }
    H2V_HD static constexpr uint32_t R2(int i) {
        constexpr uint32_t v[8] = {0xae216da7u, 0x1bb8e645u, 0xe35c59e3u, 0x53fe3ab1u, 0x53bb8085u, 0x8c49833du, 0x7f4e44a5u, 0x0216d0b1u};
        return v[i];
    }
    H2V_HD static constexpr uint32_t R3(int i) {
        constexpr uint32_t v[8] = {0xb4bf0040u, 0x5e94d8e1u, 0x1cfbb6b8u, 0x2a489cbeu, 0xa19fcfedu, 0x893cc664u, 0x7fcc657cu, 0x0cf8594bu};
        return v[i];
    }
};

template <class PR> struct Fp {
    uint32_t v[8];

    H2V_HD static Fp zero() { Fp r; for (int i = 0; i < 8; ++i) r.v[i] = 0; return r; }
    H2V_HD static Fp one() { Fp r; for (int i = 0; i < 8; ++i) r.v[i] = PR::ONE(i); return r; }
    H2V_HD static Fp r2() { Fp r; for (int i = 0; i < 8; ++i) r.v[i] = PR::R2(i); return r; }

    H2V_HD bool is_zero() const { uint32_t o = 0; for (int i = 0; i < 8; ++i) o |= v[i]; return o == 0; }
    H2V_HD bool operator==(const Fp& b) const { uint32_t o = 0; for (int i = 0; i < 8; ++i) o |= v[i] ^ b.v[i]; return o == 0; }
    H2V_HD bool operator!=(const Fp& b) const { return !(*this == b); }

    // raw (non-Montgomery) limb comparison a >= p
    H2V_HD static bool geq_p(const uint32_t a[8]) {
        for (int i = 7; i >= 0; --i) {
            uint32_t p = PR::P(i);
            if (a[i] > p) return true;
            if (a[i] < p) return false;
        }
        return true;
    }
    // carry chains written with __builtin_addc / __builtin_subc so that they lower to v_add_co / v_addc_co (one
    // instruction per limb) instead of 64-bit adds and shifts: an Fq addition is ~24 instructions, not ~130
    H2V_HD static uint32_t sub_p(uint32_t r[8], const uint32_t a[8]) {
        unsigned borrow = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) r[i] = __builtin_subc(a[i], PR::P(i), borrow, &borrow);
        return borrow;
    }

    H2V_HD Fp operator+(const Fp& b) const {
        uint32_t t[8]; unsigned carry = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) t[i] = __builtin_addc(v[i], b.v[i], carry, &carry);
        // p < 2^254: a + b < 2^255, no carry out of limb 7
        uint32_t u[8]; uint32_t borrow = sub_p(u, t);
        Fp r;
#pragma unroll
        for (int i = 0; i < 8; ++i) r.v[i] = borrow ? t[i] : u[i];
        return r;
    }
    H2V_HD Fp operator-(const Fp& b) const {
        uint32_t t[8]; unsigned borrow = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) t[i] = __builtin_subc(v[i], b.v[i], borrow, &borrow);
        unsigned carry = 0; Fp r;
        const uint32_t mask = borrow ? 0xffffffffu : 0u;
#pragma unroll
        for (int i = 0; i < 8; ++i) r.v[i] = __builtin_addc(t[i], PR::P(i) & mask, carry, &carry);
        return r;
    }
    H2V_HD Fp neg() const { return zero() - *this; }
    H2V_HD Fp dbl() const { return *this + *this; }

    // Montgomery product a*b/R mod p (CIOS).  Inputs < p  =>  output < p.
    // Also correct for a < 2^256 (unreduced) with b < p: the running value stays < 2^256 + p.
    __host__ __device__ __forceinline__ static Fp mul_inl(const Fp& a, const Fp& b) {
        uint32_t t[9];
#pragma unroll
        for (int i = 0; i < 9; ++i) t[i] = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            uint64_t carry = 0;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                uint64_t s = (uint64_t)a.v[j] * b.v[i] + t[j] + carry;
                t[j] = (uint32_t)s; carry = s >> 32;
            }
            uint64_t top = (uint64_t)t[8] + carry;
            uint32_t m = t[0] * PR::INV;
            carry = ((uint64_t)m * PR::P(0) + t[0]) >> 32;
#pragma unroll
            for (int j = 1; j < 8; ++j) {
                uint64_t s = (uint64_t)m * PR::P(j) + t[j] + carry;
                t[j - 1] = (uint32_t)s; carry = s >> 32;
            }
            top += carry;
            t[7] = (uint32_t)top; t[8] = (uint32_t)(top >> 32);
        }
        uint32_t u[8]; uint32_t borrow = sub_p(u, t);
        bool take_sub = t[8] != 0 || !borrow;
        Fp r;
#pragma unroll
        for (int i = 0; i < 8; ++i) r.v[i] = take_sub ? u[i] : t[i];
        return r;
    }
    // The same product as a real function call (operands and result by value, i.e. in VGPRs).  Mid-level routines
    // (Fq2 products, the G1 group law) inline mul_inl so that their independent products can be interleaved by the
    // scheduler — a lone wave is latency-bound on the carry chains of a single product — while everything else
    // calls this one copy to keep code size and compile time bounded.
    H2V_FN static Fp mul(Fp a, Fp b) { return mul_inl(a, b); }
    __host__ __device__ __forceinline__ Fp sqr_inl() const { return mul_inl(*this, *this); }
    H2V_HD Fp operator*(const Fp& b) const { return mul(*this, b); }
    H2V_HD Fp sqr() const { return mul(*this, *this); }

    // canonical integer (as limbs) -> Montgomery.  Requires raw < p.
    H2V_HD static Fp from_raw(const uint32_t raw[8]) { Fp t; for (int i = 0; i < 8; ++i) t.v[i] = raw[i]; return mul(t, r2()); }
    // any 256-bit integer -> Montgomery (reduces first; 2^256 < 6p)
    H2V_HD static Fp from_raw_unreduced(const uint32_t raw[8]) {
        uint32_t t[8]; for (int i = 0; i < 8; ++i) t[i] = raw[i];
        for (int k = 0; k < 5; ++k) { uint32_t u[8]; uint32_t borrow = sub_p(u, t); if (!borrow) for (int i = 0; i < 8; ++i) t[i] = u[i]; }
        return from_raw(t);
    }
    H2V_HD void to_raw(uint32_t out[8]) const {
        Fp o = zero(); o.v[0] = 1;
        Fp r = mul(*this, o);
        for (int i = 0; i < 8; ++i) out[i] = r.v[i];
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
        Fp lo = from_raw_unreduced(w), hi = from_raw_unreduced(w + 8);
        return lo + hi * r2();  // the element with Montgomery limbs R^2 is the value R = 2^256
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
    // Fermat inverse x^(p-2); inv(0) = 0.  Kept as the cross-check of inv().
    H2V_FN Fp inv_fermat() const {
        uint32_t e[8];
        for (int i = 0; i < 8; ++i) e[i] = PR::P(i);
        e[0] -= 2;  // p is odd and its low limb is >= 2 for both fields
        return pow_limbs(e);
    }
    // Inverse by the binary extended Euclidean algorithm on the raw limbs: <= 2*254 shift/subtract steps of ~70
    // instructions instead of 320 Montgomery products of ~600 — about 5x fewer instructions on the lanes' critical
    // path (the Fr program's one inversion per proof, the affine conversions, the Fq12 inversion of the pairing).
    // Works on the Montgomery limbs m = aR as a plain integer: m^-1 = a^-1 R^-1, and one product with R^3 gives a^-1 R.
    // inv(0) = 0.  The loop is bounded, so every lane leaves it.
    H2V_FN Fp inv() const {
        if (is_zero()) return zero();
        uint32_t u[8], w[8], x1[8], x2[8];
        for (int i = 0; i < 8; ++i) { u[i] = v[i]; w[i] = PR::P(i); x1[i] = i == 0 ? 1u : 0u; x2[i] = 0u; }
        auto is_one = [](const uint32_t a[8]) { uint32_t o = a[0] ^ 1u; for (int i = 1; i < 8; ++i) o |= a[i]; return o == 0; };
        auto shr1 = [](uint32_t a[8], uint32_t top) { for (int i = 0; i < 7; ++i) a[i] = (a[i] >> 1) | (a[i + 1] << 31); a[7] = (a[7] >> 1) | (top << 31); };
        auto halve_mod = [&](uint32_t x[8]) {  // x <- x / 2 mod p
            uint32_t carry = 0;
            if (x[0] & 1u) { for (int i = 0; i < 8; ++i) { uint64_t t = (uint64_t)x[i] + PR::P(i) + carry; x[i] = (uint32_t)t; carry = (uint32_t)(t >> 32); } }
            shr1(x, carry);
        };
        auto sub_raw = [](uint32_t a[8], const uint32_t b[8]) {  // a <- a - b, returns borrow
            uint32_t borrow = 0;
            for (int i = 0; i < 8; ++i) { uint64_t t = (uint64_t)a[i] - b[i] - borrow; a[i] = (uint32_t)t; borrow = (uint32_t)(t >> 32) & 1u; }
            return borrow;
        };
        auto sub_mod = [&](uint32_t a[8], const uint32_t b[8]) {  // a <- a - b mod p
            if (sub_raw(a, b)) { uint32_t carry = 0; for (int i = 0; i < 8; ++i) { uint64_t t = (uint64_t)a[i] + PR::P(i) + carry; a[i] = (uint32_t)t; carry = (uint32_t)(t >> 32); } }
        };
        auto geq = [](const uint32_t a[8], const uint32_t b[8]) { for (int i = 7; i >= 0; --i) { if (a[i] > b[i]) return true; if (a[i] < b[i]) return false; } return true; };
        for (int iter = 0; iter < 1024 && !is_one(u) && !is_one(w); ++iter) {
            if (!(u[0] & 1u)) { shr1(u, 0); halve_mod(x1); }
            else if (!(w[0] & 1u)) { shr1(w, 0); halve_mod(x2); }
            else if (geq(u, w)) { sub_raw(u, w); sub_mod(x1, x2); shr1(u, 0); halve_mod(x1); }   // both odd: the difference is even
            else { sub_raw(w, u); sub_mod(x2, x1); shr1(w, 0); halve_mod(x2); }
        }
        Fp t, r3;
        const bool from_u = is_one(u);
        for (int i = 0; i < 8; ++i) { t.v[i] = from_u ? x1[i] : x2[i]; r3.v[i] = PR::R3(i); }
        return mul(t, r3);
    }
    H2V_HD bool is_odd() const { uint32_t raw[8]; to_raw(raw); return raw[0] & 1; }
};

typedef Fp<FqParams> Fq;
typedef Fp<FrParams> Fr;

}  // namespace h2v
