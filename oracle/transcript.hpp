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

// Legacy Keccak-256 (pad 0x01 .. 0x80, rate 136), as the `sha3 0.9.1` crate's Keccak256 the reference uses for its
// EVM-style transcript (transcript/mod.rs:110-116,136-151,234-272).  hashlib only has SHA3-256 (different padding),
// so this is pinned against a pure-Python Keccak in tests/test_oracle_transcript.py.
struct Keccak256 {
    u64 st[25];
    uint8_t buf[136];
    size_t buflen;
    Keccak256() { memset(st, 0, sizeof st); buflen = 0; }
    static inline u64 rotl(u64 x, int n) { return n ? (x << n) | (x >> (64 - n)) : x; }
    static void permute(u64 a[25]) {
        static const u64 RC[24] = {0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808aULL, 0x8000000080008000ULL, 0x000000000000808bULL, 0x0000000080000001ULL,
                                   0x8000000080008081ULL, 0x8000000000008009ULL, 0x000000000000008aULL, 0x0000000000000088ULL, 0x0000000080008009ULL, 0x000000008000000aULL,
                                   0x000000008000808bULL, 0x800000000000008bULL, 0x8000000000008089ULL, 0x8000000000008003ULL, 0x8000000000008002ULL, 0x8000000000000080ULL,
                                   0x000000000000800aULL, 0x800000008000000aULL, 0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};
        static const int ROT[25] = {0, 1, 62, 28, 27, 36, 44, 6, 55, 20, 3, 10, 43, 25, 39, 41, 45, 15, 21, 8, 18, 2, 61, 56, 14};
        for (int r = 0; r < 24; ++r) {
            u64 c[5], d[5], b[25];
            for (int x = 0; x < 5; ++x) c[x] = a[x] ^ a[x + 5] ^ a[x + 10] ^ a[x + 15] ^ a[x + 20];
            for (int x = 0; x < 5; ++x) d[x] = c[(x + 4) % 5] ^ rotl(c[(x + 1) % 5], 1);
            for (int i = 0; i < 25; ++i) a[i] ^= d[i % 5];
            for (int x = 0; x < 5; ++x) for (int y = 0; y < 5; ++y) b[y + 5 * ((2 * x + 3 * y) % 5)] = rotl(a[x + 5 * y], ROT[x + 5 * y]);
            for (int x = 0; x < 5; ++x) for (int y = 0; y < 5; ++y) a[x + 5 * y] = b[x + 5 * y] ^ (~b[(x + 1) % 5 + 5 * y] & b[(x + 2) % 5 + 5 * y]);
            a[0] ^= RC[r];
        }
    }
    void absorb_block(const uint8_t* b) {
        for (int i = 0; i < 17; ++i) { u64 w = 0; for (int j = 0; j < 8; ++j) w |= (u64)b[8 * i + j] << (8 * j); st[i] ^= w; }
        permute(st);
    }
    void update(const uint8_t* in, size_t n) {
        while (n > 0) {
            size_t take = 136 - buflen; if (take > n) take = n;
            memcpy(buf + buflen, in, take); buflen += take; in += take; n -= take;
            if (buflen == 136) { absorb_block(buf); buflen = 0; }
        }
    }
    void finalize(uint8_t out[32]) const {
        Keccak256 c = *this;
        memset(c.buf + c.buflen, 0, 136 - c.buflen);
        c.buf[c.buflen] ^= 0x01; c.buf[135] ^= 0x80;
        c.absorb_block(c.buf);
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 8; ++j) out[8 * i + j] = (uint8_t)(c.st[i] >> (8 * j));
    }
};

enum TranscriptKind { TR_BLAKE2B = 0, TR_KECCAK256 = 1 };

// Error strings mirror the reference's io::Error = &'static str values.
struct TranscriptError { const char* what; };

struct TranscriptBase {
    int kind;
    Blake2b state;
    Keccak256 kstate;
    explicit TranscriptBase(int k = TR_BLAKE2B) : kind(k), state("Halo2-Transcript") {
        if (kind == TR_KECCAK256) kstate.update((const uint8_t*)"Halo2-Transcript", 16);   // transcript/mod.rs:143-145
    }
    void absorb(const uint8_t* b, size_t n) { if (kind == TR_KECCAK256) kstate.update(b, n); else state.update(b, n); }
    Fr squeeze_challenge() {
        uint8_t z = 0; absorb(&z, 1);
        uint8_t out[64];
        if (kind == TR_KECCAK256) {   // transcript/mod.rs:239-254: two clones with suffix 10 / 11, 32 bytes each
            Keccak256 lo = kstate, hi = kstate;
            uint8_t a = 10, b = 11; lo.update(&a, 1); hi.update(&b, 1);
            lo.finalize(out); hi.finalize(out + 32);
        } else state.finalize(out);
        return Fr::from_uniform_bytes(out);
    }
    void common_point(const G1Affine& p) {
        if (p.inf) throw TranscriptError{"cannot write points at infinity to the transcript"};
        uint8_t b[65]; b[0] = 1; p.x.to_bytes(b + 1); p.y.to_bytes(b + 33);
        absorb(b, 65);
    }
    void common_scalar(const Fr& s) {
        uint8_t b[33]; b[0] = 2; s.to_bytes(b + 1);
        absorb(b, 33);
    }
};

// == Blake2bRead<&[u8], G1Affine, Challenge255<_>>
struct TranscriptRead : TranscriptBase {
    const uint8_t* data; size_t len, pos;
    TranscriptRead(const uint8_t* d, size_t n, int kind = TR_BLAKE2B) : TranscriptBase(kind), data(d), len(n), pos(0) {}
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
    explicit TranscriptWrite(int kind = TR_BLAKE2B) : TranscriptBase(kind) {}
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
