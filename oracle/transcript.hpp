// ORACLE — TEST INFRASTRUCTURE ONLY (see bn254_fp.hpp header).
//
// BLAKE2b-512 (RFC 7693) with the personalisation the reference uses, and the Blake2b
// Fiat-Shamir transcript reader/writer.  Follows transcript/mod.rs:
//   init: Blake2bParams::new().hash_length(64).personal(b"Halo2-Transcript")   :124-133
//   squeeze_challenge: absorb 0x00, clone, finalize clone -> 64 B -> from_uniform_bytes  :209-214,500-514
//   common_point: absorb 0x01 | x.to_repr() | y.to_repr(); identity is an error          :216-224
//   common_scalar: absorb 0x02 | repr                                                     :226-231
//   read_point / read_scalar                                                              :158-176
// blake2b_simd (the reference's hash dependency) is not vendored; pinned here against
// hashlib.blake2b(person=..., digest_size=64) in tests/test_oracle_transcript.py.
#pragma once
#include "bn254_curve.hpp"
#include <string>
#include <vector>

namespace h2o {

struct Blake2b {
    u64 h[8];
    u64 t0, t1;
    uint8_t buf[128];
    size_t buflen;

    static inline u64 rotr(u64 x, int n) { return (x >> n) | (x << (64 - n)); }
    static const u64* IV() {
        static const u64 iv[8] = {0x6a09e667f3bcc908ULL, 0xbb67ae8584caa73bULL, 0x3c6ef372fe94f82bULL, 0xa54ff53a5f1d36f1ULL,
                                  0x510e527fade682d1ULL, 0x9b05688c2b3e6c1fULL, 0x1f83d9abfb41bd6bULL, 0x5be0cd19137e2179ULL};
        return iv;
    }
    // digest_length = 64, key_length = 0, fanout = depth = 1, 16-byte personalisation
    explicit Blake2b(const char personal[16]) {
        uint8_t P[64]; memset(P, 0, 64);
        P[0] = 64; P[1] = 0; P[2] = 1; P[3] = 1;
        memcpy(P + 48, personal, 16);
        for (int i = 0; i < 8; ++i) {
            u64 w = 0;
            for (int j = 0; j < 8; ++j) w |= (u64)P[8 * i + j] << (8 * j);
            h[i] = IV()[i] ^ w;
        }
        t0 = t1 = 0; buflen = 0;
    }
    void compress(const uint8_t block[128], bool last) {
        static const uint8_t S[12][16] = {
            {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15}, {14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3},
            {11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4}, {7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8},
            {9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13}, {2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9},
            {12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11}, {13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10},
            {6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5}, {10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0},
            {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15}, {14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3}};
        u64 m[16], v[16];
        for (int i = 0; i < 16; ++i) {
            u64 w = 0;
            for (int j = 0; j < 8; ++j) w |= (u64)block[8 * i + j] << (8 * j);
            m[i] = w;
        }
        for (int i = 0; i < 8; ++i) { v[i] = h[i]; v[i + 8] = IV()[i]; }
        v[12] ^= t0; v[13] ^= t1;
        if (last) v[14] = ~v[14];
#define H2O_G(a, b, c, d, x, y)                \
    v[a] = v[a] + v[b] + (x); v[d] = rotr(v[d] ^ v[a], 32); \
    v[c] = v[c] + v[d];       v[b] = rotr(v[b] ^ v[c], 24); \
    v[a] = v[a] + v[b] + (y); v[d] = rotr(v[d] ^ v[a], 16); \
    v[c] = v[c] + v[d];       v[b] = rotr(v[b] ^ v[c], 63);
        for (int r = 0; r < 12; ++r) {
            const uint8_t* s = S[r];
            H2O_G(0, 4, 8, 12, m[s[0]], m[s[1]]);
            H2O_G(1, 5, 9, 13, m[s[2]], m[s[3]]);
            H2O_G(2, 6, 10, 14, m[s[4]], m[s[5]]);
            H2O_G(3, 7, 11, 15, m[s[6]], m[s[7]]);
            H2O_G(0, 5, 10, 15, m[s[8]], m[s[9]]);
            H2O_G(1, 6, 11, 12, m[s[10]], m[s[11]]);
            H2O_G(2, 7, 8, 13, m[s[12]], m[s[13]]);
            H2O_G(3, 4, 9, 14, m[s[14]], m[s[15]]);
        }
#undef H2O_G
        for (int i = 0; i < 8; ++i) h[i] ^= v[i] ^ v[i + 8];
    }
    void update(const uint8_t* in, size_t n) {
        while (n > 0) {
            if (buflen == 128) {  // buffer full and more input follows: not the last block
                t0 += 128; if (t0 < 128) t1++;
                compress(buf, false);
                buflen = 0;
            }
            size_t take = 128 - buflen; if (take > n) take = n;
            memcpy(buf + buflen, in, take);
            buflen += take; in += take; n -= take;
        }
    }
    void finalize(uint8_t out[64]) const {  // const: works on a copy (== state.clone().finalize())
        Blake2b c = *this;
        c.t0 += c.buflen; if (c.t0 < c.buflen) c.t1++;
        memset(c.buf + c.buflen, 0, 128 - c.buflen);
        c.compress(c.buf, true);
        for (int i = 0; i < 8; ++i) for (int j = 0; j < 8; ++j) out[8 * i + j] = (uint8_t)(c.h[i] >> (8 * j));
    }
};

// Error strings mirror the reference's io::Error = &'static str values.
struct TranscriptError { const char* what; };

struct TranscriptBase {
    Blake2b state;
    TranscriptBase() : state("Halo2-Transcript") {}
    Fr squeeze_challenge() {
        uint8_t z = 0; state.update(&z, 1);
        uint8_t out[64]; state.finalize(out);
        return Fr::from_uniform_bytes(out);
    }
    void common_point(const G1Affine& p) {
        if (p.inf) throw TranscriptError{"cannot write points at infinity to the transcript"};
        uint8_t b[65]; b[0] = 1; p.x.to_bytes(b + 1); p.y.to_bytes(b + 33);
        state.update(b, 65);
    }
    void common_scalar(const Fr& s) {
        uint8_t b[33]; b[0] = 2; s.to_bytes(b + 1);
        state.update(b, 33);
    }
};

// == Blake2bRead<&[u8], G1Affine, Challenge255<_>>
struct TranscriptRead : TranscriptBase {
    const uint8_t* data; size_t len, pos;
    TranscriptRead(const uint8_t* d, size_t n) : data(d), len(n), pos(0) {}
    G1Affine read_point() {
        if (pos + 32 > len) throw TranscriptError{"failed to fill whole buffer"};
        G1Affine p;
        if (!g1_from_bytes(data + pos, p)) throw TranscriptError{"invalid point encoding in proof"};
        pos += 32;
        common_point(p);
        return p;
    }
    Fr read_scalar() {
        if (pos + 32 > len) throw TranscriptError{"failed to fill whole buffer"};
        Fr s;
        if (!Fr::from_bytes(data + pos, s)) throw TranscriptError{"invalid field element encoding in proof"};
        pos += 32;
        common_scalar(s);
        return s;
    }
};

// == Blake2bWrite (transcript/mod.rs:336-398); used only by the test-only prover
struct TranscriptWrite : TranscriptBase {
    std::vector<uint8_t> out;
    void write_point(const G1Affine& p) {
        common_point(p);
        uint8_t b[32]; g1_to_bytes(p, b);
        out.insert(out.end(), b, b + 32);
    }
    void write_scalar(const Fr& s) {
        common_scalar(s);
        uint8_t b[32]; s.to_bytes(b);
        out.insert(out.end(), b, b + 32);
    }
};

}  // namespace h2o
