// Per-proof kernels of the batch verifier: everything verify_proof (lib.rs:33-425) and
// VerifierSHPLONK::verify_proof (shplonk.rs:175-267) do before the MSMs are evaluated.
//
//   k_decompress      one lane per (proof, point)   G1Affine::from_bytes               transcript/mod.rs:158-166
//   k_check_scalars   one lane per (proof, scalar)  Fr::from_repr canonicity           transcript/mod.rs:168-176
//   k_stream_build    one lane per (proof, 8-byte word of the absorbed stream)         transcript/mod.rs:216-231
//   k_transcript      four lanes per proof          Blake2b-512 + challenges           transcript/mod.rs:124-133,209-214,500-514
//   k_multipliers     one workgroup                 suffix products of the batch draws kzg/strategy.rs:129, msm.rs:173-176
//   k_instance_eval   one workgroup per proof       Lagrange sum over a wide instance column   lib.rs:173-218, poly/domain.rs:187-212
//   k_frvm            one lane per proof            the compiled Fr program            lib.rs:173-346, shplonk.rs:202-264
//   k_fold_shared     one workgroup per shared base sum over proofs of the scalars of VK-wide bases
//
// Layouts are chosen so that lanes (= proofs) are contiguous in the fastest dimension for
// everything the per-proof kernels re-read: stream words [word][proof], challenges [c][proof],
// program slots [slot][proof], shared scalars [j][proof].
#include "../../include/h2v.h"
#include "batch.h"
#include <atomic>

namespace h2v {

// 32 B in (two 16-byte loads: a point sits at a multiple of 32 bytes inside a proof whose length is a multiple of 32), 72 + 32 B
// out as 8-byte / 16-byte stores.  Round 1 moved every byte on its own: 13x the algorithmic traffic (profiles/r01_pmc_fetch_write.csv).
__global__ void __launch_bounds__(64, 4) k_decompress(const uint8_t* __restrict__ proofs, uint32_t proof_len, const uint32_t* __restrict__ point_offsets,
                                                   uint32_t np, uint32_t n_main_points, uint32_t n, G1A* __restrict__ pts, G1A* __restrict__ phi, uint8_t* __restrict__ ycanon,
                                                   int* __restrict__ status) {
    const uint32_t t_raw = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t t = min(t_raw, n * np - 1);   // lanes past the end shadow the last point (all lanes reach the barriers below); their output is not stored
    uint32_t p = t / np, slot = t % np;
    const uint4* src = reinterpret_cast<const uint4*>(proofs + (size_t)p * proof_len + point_offsets[slot]);
    const uint4 lo = src[0], hi = src[1];
    uint32_t w[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
    // G1Affine::from_bytes (compressed): flags in the top byte, x below
    const bool is_inf = (w[7] >> 24) & G1_FLAG_IDENTITY, sign = (w[7] >> 24) & G1_FLAG_SIGN;
    w[7] &= 0x3fffffffu;
    G1A a = G1A::identity();
    bool ok = !Fq::geq_p(w);
    uint32_t yraw[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (ok) {
        const Fq x = Fq::from_raw(w);
        if (is_inf) ok = false;   // the identity decodes (x = 0, no sign) but cannot be absorbed ("cannot write points at infinity to the transcript"): an error either way
        else {
            const Fq three = {{FqParams::ONE(0), FqParams::ONE(1), FqParams::ONE(2), FqParams::ONE(3), FqParams::ONE(4), FqParams::ONE(5), FqParams::ONE(6), FqParams::ONE(7), FqParams::ONE(8)}};
            const Fq rhs = x.sqr() * x + (three + three + three);
            Fq y = fq_sqrt_candidate_loop_w3(rhs);
            if (y.sqr() != rhs) ok = false;
            else {
                y.to_raw(yraw);                       // canonical y: its parity decides the sign, its bytes are absorbed
                uint32_t nz = 0;
#pragma unroll
                for (int i = 0; i < 8; ++i) nz |= yraw[i];
                if ((bool)(yraw[0] & 1u) != sign && nz) {   // the other root: p - y (y = 0 is its own negative)
                    y = y.neg();
                    uint64_t borrow = 0;
#pragma unroll
                    for (int i = 0; i < 8; ++i) { const uint64_t d = (uint64_t)FqParams::P(i) - yraw[i] - borrow; yraw[i] = (uint32_t)d; borrow = (d >> 32) & 1u; }
                }
                a.x = x; a.y = y;
            }
        }
    }
    if (!ok) { status_set(status, p, slot < n_main_points ? H2V_DEV_ST_TRANSCRIPT : H2V_DEV_ST_OPENING); a = G1A::identity(); for (int i = 0; i < 8; ++i) yraw[i] = 0; }
    // Outputs through LDS: consecutive lanes own consecutive points (t = proof * np + slot), so a wave's 64 points are 4608
    // contiguous bytes and its 64 canonical y are 2048.  Written lane by lane (72-byte stride) every store instruction touched 36
    // lines partially, and the L2 fetched each of them to merge the bytes: 0.2 GB of reads and 0.14 GB of writes for 26 MB of
    // output (profiles/r02_pmc_fetch_write_steps20.csv, first collection).  Transposed in LDS, every store instruction writes
    // 256 contiguous bytes.
    __shared__ uint32_t out_lds[64 * 18];
    const uint32_t lane = threadIdx.x, t0 = blockIdx.x * blockDim.x, live = min(64u, n * np - t0);
    const uint32_t* av = reinterpret_cast<const uint32_t*>(&a);
#pragma unroll
    for (int i = 0; i < 18; ++i) out_lds[lane * 18 + i] = av[i];
    __syncthreads();
    uint32_t* dp = reinterpret_cast<uint32_t*>(pts + t0);
#pragma unroll
    for (int i = 0; i < 18; ++i) { const uint32_t k = i * 64 + lane; if (k < live * 18) dp[k] = out_lds[k]; }
    __syncthreads();
    // phi(P) = (beta x, y) beside every point: the MSM's second GLV half gathers it instead of multiplying per list entry (msm.hip: msm_entry_load)
    a.x = g1_beta_times(a.x);
#pragma unroll
    for (int i = 0; i < 9; ++i) out_lds[lane * 18 + i] = av[i];     // (y is still there)
    __syncthreads();
    dp = reinterpret_cast<uint32_t*>(phi + t0);
#pragma unroll
    for (int i = 0; i < 18; ++i) { const uint32_t k = i * 64 + lane; if (k < live * 18) dp[k] = out_lds[k]; }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 8; ++i) out_lds[lane * 8 + i] = yraw[i];
    __syncthreads();
    uint32_t* dy = reinterpret_cast<uint32_t*>(ycanon + (size_t)t0 * 32);
#pragma unroll
    for (int i = 0; i < 8; ++i) { const uint32_t k = i * 64 + lane; if (k < live * 8) dy[k] = out_lds[k]; }
}

__global__ void __launch_bounds__(256) k_check_scalars(const uint8_t* __restrict__ proofs, uint32_t proof_len, const uint32_t* __restrict__ scalar_offsets,
                                                       uint32_t ns, const uint8_t* __restrict__ inst, uint32_t ninst, uint32_t n, int* __restrict__ status) {
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t per = ns + ninst;
    if (t >= n * per) return;
    uint32_t p = t / per, i = t % per;
    const uint8_t* b = i < ns ? proofs + (size_t)p * proof_len + scalar_offsets[i] : inst + ((size_t)p * ninst + (i - ns)) * 32;
    uint32_t raw[8];
    for (int j = 0; j < 8; ++j) raw[j] = (uint32_t)b[4 * j] | ((uint32_t)b[4 * j + 1] << 8) | ((uint32_t)b[4 * j + 2] << 16) | ((uint32_t)b[4 * j + 3] << 24);
    // a non-canonical public input cannot be represented as an Fr on the reference side at all: bad argument for that proof
    if (Fr::geq_p(raw)) status_set(status, p, i < ns ? H2V_DEV_ST_TRANSCRIPT : H2V_DEV_ST_INVALID_INSTANCES);
}

// One 8-byte word of the absorbed stream per thread, PROOF-MAJOR: words[proof][word].  Consecutive lanes build consecutive words
// of one proof, so their source bytes are (mostly) consecutive bytes of that proof and the stores are contiguous; the hash kernels
// then read a proof's 128-byte block as one line.  Round 1 had lanes = proofs: every byte load of a wave touched 64 different
// lines (0.13 ms per 20-step launch for 37 MB of stream).
#define STREAM_PROOFS_PER_BLOCK 8u   // a workgroup per proof was launch-rate bound: 20 480 workgroups of ~1 us of work each took 0.15 ms
__global__ void __launch_bounds__(256) k_stream_build(const TranscriptSrc* __restrict__ stream, uint32_t stream_len, const uint8_t* __restrict__ proofs,
                                                      uint32_t proof_len, const uint8_t* __restrict__ ycanon, uint32_t np, const uint8_t* __restrict__ inst,
                                                      uint32_t ninst, uint32_t n, uint32_t stream_words, unsigned long long* __restrict__ words) {
    const uint32_t p0 = blockIdx.x * STREAM_PROOFS_PER_BLOCK, p1 = min(n, p0 + STREAM_PROOFS_PER_BLOCK);
    for (uint32_t w = threadIdx.x; w < stream_words; w += blockDim.x) {
        // the word's eight table entries are the same for every proof: read once, used for the block's proofs
        uint32_t kind[8], value[8], off[8];
#pragma unroll
        for (uint32_t b = 0; b < 8; ++b) {
            const uint32_t pos = w * 8 + b, q = pos < stream_len ? pos : stream_len - 1;
            const TranscriptSrc sd = stream[q];
            kind[b] = pos < stream_len ? sd.kind : (uint32_t)TranscriptSrc::CONST;
            value[b] = pos < stream_len ? sd.value : 0u;
            off[b] = kind[b] == TranscriptSrc::CONST ? 0u : sd.offset;
        }
        // A word is [tail of item A][constant bytes (prefixes, markers)][head of item B] with either part possibly empty; items are 32
        // bytes (a point's x or y, a scalar, a public input) at odd addresses (every item is preceded by a one-byte prefix).  Each part
        // is ONE unaligned 8-byte load per proof — the eight bytes that END with A's last byte, shifted down; the eight that START with
        // B's first byte, shifted up: both stay inside their 32-byte items — instead of eight byte loads per word (58 -> 38 us per
        // 20-step launch).  A flag byte inside the word is masked afterwards.  A word of any other shape takes the byte path below.
        auto is_data = [](uint32_t k) { return k != TranscriptSrc::CONST; };
        auto src_of = [](uint32_t k) { return k == TranscriptSrc::PROOF_MASKED ? (uint32_t)TranscriptSrc::PROOF : k; };
        uint32_t la = 0;                                    // bytes of run A: positions [0, la)
        while (la < 8 && is_data(kind[la]) && src_of(kind[la]) == src_of(kind[0]) && off[la] == off[0] + la) ++la;
        uint32_t sb = la;                                   // run B: positions [sb, 8)
        unsigned long long cbytes = 0;
        while (sb < 8 && !is_data(kind[sb])) { cbytes |= (unsigned long long)value[sb] << (8 * sb); ++sb; }
        bool fits = true;
        for (uint32_t b2 = sb; b2 < 8; ++b2) fits = fits && is_data(kind[b2]) && src_of(kind[b2]) == src_of(kind[sb]) && off[b2] == off[sb] + (b2 - sb);
        // (A shorter than the word is the TAIL of its item — what follows it is a constant or another source — so the bytes in front of
        // it, which the shifted load also reads, belong to the same item; B is the HEAD of its item for the same reason.  Items are 32 bytes.)
        if (la > 0 && la < 8 && off[0] < 8 - la) fits = false;
        uint32_t mask_byte = 8;
#pragma unroll
        for (uint32_t b2 = 0; b2 < 8; ++b2) if (kind[b2] == TranscriptSrc::PROOF_MASKED) mask_byte = b2;
        if (fits) {
            typedef unsigned long long u64_unaligned __attribute__((aligned(1)));
            const unsigned long long keep = mask_byte < 8 ? ~(0xc0ULL << (8 * mask_byte)) : ~0ULL;
            auto base_of = [&](uint32_t k) -> const uint8_t* { return k == TranscriptSrc::YCOORD ? ycanon : (k == TranscriptSrc::INSTANCE ? inst : proofs); };
            auto per_of = [&](uint32_t k) -> size_t { return k == TranscriptSrc::YCOORD ? (size_t)np * 32 : (k == TranscriptSrc::INSTANCE ? (size_t)ninst * 32 : (size_t)proof_len); };
            const uint8_t* const baseA = la ? base_of(kind[0]) : proofs;
            const size_t perA = la ? per_of(kind[0]) : 0, offA = la ? (size_t)off[0] + la - 8 : 0;       // the load ends with A's last byte
            const uint8_t* const baseB = sb < 8 ? base_of(kind[sb]) : proofs;
            const size_t perB = sb < 8 ? per_of(kind[sb]) : 0, offB = sb < 8 ? off[sb] : 0;                 // the load starts with B's first byte
            for (uint32_t p = p0; p < p1; ++p) {
                unsigned long long v = cbytes;
                if (la) v |= *reinterpret_cast<const u64_unaligned*>(baseA + (size_t)p * perA + offA) >> (8 * (8 - la));
                if (sb < 8) v |= *reinterpret_cast<const u64_unaligned*>(baseB + (size_t)p * perB + offB) << (8 * sb);
                words[(size_t)p * stream_words + w] = v & keep;
            }
            continue;
        }
        for (uint32_t p = p0; p < p1; ++p) {
            const uint8_t* const pb = proofs + (size_t)p * proof_len;
            const uint8_t* const yb = ycanon + (size_t)p * np * 32;
            const uint8_t* const ib = inst + (size_t)p * ninst * 32;
            // eight source addresses (selects, no branches: the kinds differ from lane to lane), eight byte loads back to back,
            // constants and the flag mask merged afterwards
            uint32_t raw[8];
#pragma unroll
            for (uint32_t b = 0; b < 8; ++b) {
                const uint8_t* base = kind[b] == TranscriptSrc::YCOORD ? yb : (kind[b] == TranscriptSrc::INSTANCE ? ib : pb);
                raw[b] = base[off[b]];
            }
            unsigned long long v = 0;
#pragma unroll
            for (uint32_t b = 0; b < 8; ++b) {
                uint32_t byte = raw[b];
                if (kind[b] == TranscriptSrc::CONST) byte = value[b];
                else if (kind[b] == TranscriptSrc::PROOF_MASKED) byte &= 0x3f;
                v |= (unsigned long long)byte << (8 * b);
            }
            words[(size_t)p * stream_words + w] = v;
        }
    }
}

// ------------------------------------------------------------------ BLAKE2b (RFC 7693)
// a 64-bit rotation by a constant is two v_alignbit_b32 (the shift / shift / or form the compiler makes of the C expression is four instructions)
__device__ __forceinline__ unsigned long long rotr64(unsigned long long x, int c) {
    const uint32_t lo = (uint32_t)x, hi = (uint32_t)(x >> 32);
    if (c == 32) return ((unsigned long long)lo << 32) | hi;
    const uint32_t a = c < 32 ? hi : lo, b = c < 32 ? lo : hi;      // rotating by c >= 32 = swapping the halves, then by c - 32
    const uint32_t k = (uint32_t)(c & 31);
    const uint32_t nl = __builtin_amdgcn_alignbit(a, b, k), nh = __builtin_amdgcn_alignbit(b, a, k);
    return ((unsigned long long)nh << 32) | nl;
}
__constant__ unsigned long long BLAKE_IV[8] = {0x6a09e667f3bcc908ULL, 0xbb67ae8584caa73bULL, 0x3c6ef372fe94f82bULL, 0xa54ff53a5f1d36f1ULL,
                                               0x510e527fade682d1ULL, 0x9b05688c2b3e6c1fULL, 0x1f83d9abfb41bd6bULL, 0x5be0cd19137e2179ULL};
__constant__ uint8_t BLAKE_SIGMA[12][16] = {
    {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15}, {14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3},
    {11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4}, {7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8},
    {9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13}, {2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9},
    {12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11}, {13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10},
    {6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5}, {10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0},
    {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15}, {14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3}};

// ---- BLAKE2b across a quad of lanes.  The compression function works on a 4 x 4 matrix of 64-bit words: a column step (four
// independent G functions) and a diagonal step (four more).  Lane r of a quad holds column r (v[r], v[4+r], v[8+r], v[12+r]);
// the diagonal step is the column step after rotating rows 1..3 by 1..3 lanes (DPP quad permutes, no LDS).  One proof per quad:
// the ~21 dependent compressions of a proof's transcript take a quarter of the instructions per lane — the transcript is a
// latency chain (16 waves per 1024 proofs), so that is a quarter of its time.
__device__ __forceinline__ unsigned long long quad_rot64(unsigned long long x, const int ctrl_sel) {
    uint32_t lo = (uint32_t)x, hi = (uint32_t)(x >> 32);
    // ctrl_sel: 1 -> lane r reads lane (r + 1) & 3, 2 -> (r + 2) & 3, 3 -> (r + 3) & 3
    if (ctrl_sel == 1) { lo = (uint32_t)__builtin_amdgcn_mov_dpp((int)lo, 0x39, 0xf, 0xf, true); hi = (uint32_t)__builtin_amdgcn_mov_dpp((int)hi, 0x39, 0xf, 0xf, true); }
    else if (ctrl_sel == 2) { lo = (uint32_t)__builtin_amdgcn_mov_dpp((int)lo, 0x4e, 0xf, 0xf, true); hi = (uint32_t)__builtin_amdgcn_mov_dpp((int)hi, 0x4e, 0xf, 0xf, true); }
    else { lo = (uint32_t)__builtin_amdgcn_mov_dpp((int)lo, 0x93, 0xf, 0xf, true); hi = (uint32_t)__builtin_amdgcn_mov_dpp((int)hi, 0x93, 0xf, 0xf, true); }
    return ((unsigned long long)hi << 32) | lo;
}
#define H2V_G4(a, b, c, d, x, y)                     \
    a = a + b + (x); d = rotr64(d ^ a, 32);          \
    c = c + d;       b = rotr64(b ^ c, 24);          \
    a = a + b + (y); d = rotr64(d ^ a, 16);          \
    c = c + d;       b = rotr64(b ^ c, 63);
// h0 = h[r], h1 = h[4 + r]; msg = this proof's 16 message words in LDS; sig[rd] packs the lane's four message indices of round rd
__device__ __forceinline__ void blake2b_compress_quad(unsigned long long& h0, unsigned long long& h1, const unsigned long long* msg, const uint32_t* sig,
                                                       unsigned long long t, bool last, uint32_t r) {
    unsigned long long a = h0, b = h1, c = BLAKE_IV[r], d = BLAKE_IV[4 + r];
    if (r == 0) d ^= t;           // v12 ^= t0 (t1 = 0: streams are far below 2^64 bytes)
    if (last && r == 2) d = ~d;   // v14
#pragma unroll
    for (int rd = 0; rd < 12; ++rd) {
        const uint32_t s4 = sig[rd];
        H2V_G4(a, b, c, d, msg[s4 & 15u], msg[(s4 >> 4) & 15u]);
        b = quad_rot64(b, 1); c = quad_rot64(c, 2); d = quad_rot64(d, 3);
        H2V_G4(a, b, c, d, msg[(s4 >> 8) & 15u], msg[(s4 >> 12) & 15u]);
        b = quad_rot64(b, 3); c = quad_rot64(c, 2); d = quad_rot64(d, 1);
    }
    h0 ^= a ^ c; h1 ^= b ^ d;
}
#undef H2V_G4

#define TR4_MSG_STRIDE 17   // 16 message words + 1: the sixteen proofs of a wave hit sixteen different LDS banks
// squeeze_at[q] = absorbed length at which challenge q is produced: digest(stream[0..L)) with the state
// cloned (transcript/mod.rs:209-214), 64 bytes -> Fr::from_uniform_bytes (:500-514).  16 proofs per wave.
__global__ void __launch_bounds__(64) k_transcript(const unsigned long long* __restrict__ words_all, uint32_t stream_words, const uint32_t* __restrict__ squeeze_at, uint32_t n_squeeze,
                                                   uint32_t n, Fr* __restrict__ chal) {
    extern __shared__ unsigned long long tr_lds[];   // [16][TR4_MSG_STRIDE] message block, then [16][n_squeeze][8] digests
    const uint32_t tid = threadIdx.x, r = tid & 3u, pq = tid >> 2;
    const uint32_t p_raw = blockIdx.x * 16 + pq, p = p_raw < n ? p_raw : n - 1;   // lanes beyond n shadow the last proof (uniform control flow)
    const unsigned long long* words = words_all + (size_t)p * stream_words;   // this proof's stream
    unsigned long long* msg = tr_lds + pq * TR4_MSG_STRIDE;
    unsigned long long* dig = tr_lds + 16 * TR4_MSG_STRIDE + (size_t)pq * n_squeeze * 8;
    uint32_t sig[12];
#pragma unroll
    for (int rd = 0; rd < 12; ++rd)
        sig[rd] = (uint32_t)BLAKE_SIGMA[rd][2 * r] | ((uint32_t)BLAKE_SIGMA[rd][2 * r + 1] << 4) | ((uint32_t)BLAKE_SIGMA[rd][8 + 2 * r] << 8) | ((uint32_t)BLAKE_SIGMA[rd][9 + 2 * r] << 12);
    // parameter block: digest 64, no key, fanout = depth = 1, personal "Halo2-Transcript" (transcript/mod.rs:126-129)
    const unsigned long long pers0 = 0x72542d326f6c6148ULL, pers1 = 0x7470697263736e61ULL;  // "Halo2-Tr" "anscript" little endian
    unsigned long long h0 = BLAKE_IV[r] ^ (r == 0 ? 0x01010040ULL : 0ULL), h1 = BLAKE_IV[4 + r] ^ (r == 2 ? pers0 : (r == 3 ? pers1 : 0ULL));
    uint32_t blk = 0;
    // the lane's four words of the block that is needed next, fetched while the block before it is compressed (a load per block on the
    // critical path was ~1 us of every one of the ~18 compressions of a proof)
    const uint32_t n_blocks = stream_words / 16;
    unsigned long long pf[4];
    uint32_t pf_blk = 0;
#pragma unroll
    for (uint32_t j = 0; j < 4; ++j) pf[j] = words[4 * r + j];
    for (uint32_t q = 0; q < n_squeeze; ++q) {
        const uint32_t len = squeeze_at[q];  // >= 1
        const uint32_t last_blk = (len - 1) / 128;
        for (; blk < last_blk; ++blk) {
            __syncthreads();
#pragma unroll
            for (uint32_t j = 0; j < 4; ++j) msg[4 * r + j] = pf_blk == blk ? pf[j] : words[(size_t)blk * 16 + 4 * r + j];
            pf_blk = blk + 1 < n_blocks ? blk + 1 : blk;
#pragma unroll
            for (uint32_t j = 0; j < 4; ++j) pf[j] = words[(size_t)pf_blk * 16 + 4 * r + j];
            __syncthreads();
            blake2b_compress_quad(h0, h1, msg, sig, (unsigned long long)(blk + 1) * 128, false, r);
        }
        const uint32_t rem = len - last_blk * 128;  // 1..128 bytes of the final block
        __syncthreads();
#pragma unroll
        for (uint32_t j = 0; j < 4; ++j) {
            unsigned long long w = pf_blk == last_blk ? pf[j] : words[(size_t)last_blk * 16 + 4 * r + j];   // (stays prefetched: the stream goes on inside this block)
            const uint32_t lo = 8 * (4 * r + j);
            if (lo >= rem) w = 0;
            else if (lo + 8 > rem) w &= (1ULL << (8 * (rem - lo))) - 1;
            msg[4 * r + j] = w;
        }
        __syncthreads();
        unsigned long long c0 = h0, c1 = h1;   // the clone
        blake2b_compress_quad(c0, c1, msg, sig, len, true, r);
        dig[q * 8 + r] = c0; dig[q * 8 + 4 + r] = c1;
    }
    __syncthreads();
    // 64 digest bytes -> Fr; the quad's lanes share the challenges
    for (uint32_t q = r; q < n_squeeze; q += 4) {
        uint32_t w32[16];
#pragma unroll
        for (int i = 0; i < 8; ++i) { const unsigned long long v = dig[q * 8 + i]; w32[2 * i] = (uint32_t)v; w32[2 * i + 1] = (uint32_t)(v >> 32); }
        const Fr c = Fr::from_uniform_words(w32);
        if (p_raw < n) chal[(size_t)q * n + p] = c;
    }
}

// ------------------------------------------------------------------ legacy Keccak-256 transcript (transcript/mod.rs:234-272)
__device__ __forceinline__ unsigned long long rotl64(unsigned long long x, int c) { return c ? (x << c) | (x >> (64 - c)) : x; }
__constant__ unsigned long long KECCAK_RC[24] = {
    0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808aULL, 0x8000000080008000ULL, 0x000000000000808bULL, 0x0000000080000001ULL,
    0x8000000080008081ULL, 0x8000000000008009ULL, 0x000000000000008aULL, 0x0000000000000088ULL, 0x0000000080008009ULL, 0x000000008000000aULL,
    0x000000008000808bULL, 0x800000000000008bULL, 0x8000000000008089ULL, 0x8000000000008003ULL, 0x8000000000008002ULL, 0x8000000000000080ULL,
    0x000000000000800aULL, 0x800000008000000aULL, 0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};
__device__ __noinline__ void keccak_f1600(unsigned long long a[25]) {
    const int ROT[25] = {0, 1, 62, 28, 27, 36, 44, 6, 55, 20, 3, 10, 43, 25, 39, 41, 45, 15, 21, 8, 18, 2, 61, 56, 14};
    for (int r = 0; r < 24; ++r) {
        unsigned long long c[5], b[25];
#pragma unroll
        for (int x = 0; x < 5; ++x) c[x] = a[x] ^ a[x + 5] ^ a[x + 10] ^ a[x + 15] ^ a[x + 20];
#pragma unroll
        for (int x = 0; x < 5; ++x) {
            unsigned long long d = c[(x + 4) % 5] ^ rotl64(c[(x + 1) % 5], 1);
#pragma unroll
            for (int y = 0; y < 5; ++y) a[x + 5 * y] ^= d;
        }
#pragma unroll
        for (int x = 0; x < 5; ++x)
#pragma unroll
            for (int y = 0; y < 5; ++y) b[y + 5 * ((2 * x + 3 * y) % 5)] = rotl64(a[x + 5 * y], ROT[x + 5 * y]);
#pragma unroll
        for (int x = 0; x < 5; ++x)
#pragma unroll
            for (int y = 0; y < 5; ++y) a[x + 5 * y] = b[x + 5 * y] ^ (~b[(x + 1) % 5 + 5 * y] & b[(x + 2) % 5 + 5 * y]);
        a[0] ^= KECCAK_RC[r];
    }
}
// digest of  stream[0..len) | suffix  given the running state `st` that has absorbed the first `blk` full 136-byte blocks
__device__ __forceinline__ void keccak_tail(const unsigned long long st[25], const unsigned long long* __restrict__ words, uint32_t stream_words, uint32_t p, uint32_t blk,
                                            uint32_t len, uint32_t suffix, unsigned long long out[4]) {
    unsigned long long a[25];
    for (int i = 0; i < 25; ++i) a[i] = st[i];
    uint32_t rem = len - blk * 136;  // 0..135 bytes of the stream remain, then the suffix byte, then the padding
    for (int j = 0; j < 17; ++j) {
        unsigned long long w = 0;
        uint32_t lo = 8 * j;
        if (lo < rem) {
            w = words[(size_t)p * stream_words + (size_t)blk * 17 + j];
            if (lo + 8 > rem) w &= (1ULL << (8 * (rem - lo))) - 1;
        }
        if (rem >= lo && rem < lo + 8) w |= (unsigned long long)suffix << (8 * (rem - lo));
        a[j] ^= w;
    }
    uint32_t pad_at = rem + 1;  // first padding byte
    if (pad_at == 136) {        // the suffix filled the block: the padding gets a block of its own
        keccak_f1600(a);
        a[0] ^= 0x01ULL; a[16] ^= 0x8000000000000000ULL;
    } else {
        a[pad_at / 8] ^= 0x01ULL << (8 * (pad_at % 8));
        a[16] ^= 0x8000000000000000ULL;
    }
    keccak_f1600(a);
    for (int i = 0; i < 4; ++i) out[i] = a[i];
}
// stream words are laid out in 136-byte blocks here: word (blk * 17 + j)
__global__ void __launch_bounds__(64) k_transcript_keccak(const unsigned long long* __restrict__ words, uint32_t stream_words, const uint32_t* __restrict__ squeeze_at, uint32_t n_squeeze,
                                                          uint32_t n, Fr* __restrict__ chal) {
    uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    unsigned long long st[25];
    for (int i = 0; i < 25; ++i) st[i] = 0;
    uint32_t blk = 0;
    for (uint32_t q = 0; q < n_squeeze; ++q) {
        uint32_t len = squeeze_at[q];
        uint32_t full = len / 136;
        for (; blk < full; ++blk) {
            for (int j = 0; j < 17; ++j) st[j] ^= words[(size_t)p * stream_words + (size_t)blk * 17 + j];
            keccak_f1600(st);
        }
        unsigned long long lo[4], hi[4];
        keccak_tail(st, words, stream_words, p, blk, len, 10, lo);   // KECCAK256_PREFIX_CHALLENGE_LO
        keccak_tail(st, words, stream_words, p, blk, len, 11, hi);   // KECCAK256_PREFIX_CHALLENGE_HI
        uint32_t w32[16];
        for (int i = 0; i < 4; ++i) { w32[2 * i] = (uint32_t)lo[i]; w32[2 * i + 1] = (uint32_t)(lo[i] >> 32); w32[8 + 2 * i] = (uint32_t)hi[i]; w32[8 + 2 * i + 1] = (uint32_t)(hi[i] >> 32); }
        chal[(size_t)q * n + p] = Fr::from_uniform_words(w32);
    }
}

// mult[p] = prod_{j > first+p} r_j over the tail of draws; tail[0] is the draw of this shard's proof 0.
// One workgroup per group of a grouped batch (blockIdx.x = group; n_tail and n are per group).
__global__ void __launch_bounds__(1024) k_multipliers(const uint8_t* __restrict__ tail, uint32_t n_tail, uint32_t n, Fr* __restrict__ mult) {
    __shared__ Fr part[1024];
    uint32_t t = threadIdx.x;
    tail += (size_t)blockIdx.x * n_tail * 32; mult += (size_t)blockIdx.x * n;
    uint32_t chunk = (n_tail + 1023) / 1024;
    uint32_t lo = min(n_tail, t * chunk), hi = min(n_tail, lo + chunk);
    Fr prod = Fr::one();
    for (uint32_t j = lo; j < hi; ++j) { Fr r; Fr::from_bytes(tail + 32 * (size_t)j, r); prod = prod * r; }
    part[t] = prod;
    __syncthreads();
    // inclusive suffix scan: part[t] = prod_{t' >= t} P_t'
    for (uint32_t d = 1; d < 1024; d <<= 1) {
        Fr v = t + d < 1024 ? part[t + d] : Fr::one();
        __syncthreads();
        part[t] = part[t] * v;
        __syncthreads();
    }
    Fr run = t + 1 < 1024 ? part[t + 1] : Fr::one();  // product of everything after this lane's chunk
    for (uint32_t j = hi; j > lo; --j) {
        if (j - 1 < n) mult[j - 1] = run;
        Fr r; Fr::from_bytes(tail + 32 * (size_t)(j - 1), r);
        run = run * r;
    }
}

// Instance evaluation for wide instance vectors (lib.rs:173-218; l_i_range, poly/domain.rs:187-212):
//   E = sum_j a_j * l_{j-rot}(x),   l_i(x) = omega^i (x^n - 1) / (n (x - omega^i))
// One workgroup per proof.  Thread t owns j = t, t + 256, ...: omega^(j-rot) advances by omega^256 per step.  The
// denominators are inverted in chunks of INSTEVAL_CHUNK per thread with Montgomery's trick (prefix products in LDS, one
// Fermat inversion per chunk, the denominators recomputed on the way back instead of stored); partial sums meet in an LDS
// tree.  A zero denominator (x on the domain) is reported like the program's zero inversions (H2V_ERR_REFERENCE_PANIC).
#define INSTEVAL_THREADS 256
#define INSTEVAL_CHUNK 16
__global__ void __launch_bounds__(INSTEVAL_THREADS) k_instance_eval(InstEvalArgs a) {
    extern __shared__ Fr insteval_lds[];                         // [INSTEVAL_CHUNK][256] prefix products, then [256] partial sums
    Fr* pre = insteval_lds;
    Fr* red = insteval_lds + (size_t)INSTEVAL_CHUNK * INSTEVAL_THREADS;
    const uint32_t p = blockIdx.x, t = threadIdx.x;
    const Fr x = a.chal[(size_t)a.x_chal * a.n + p];
    Fr w = a.w_start * a.omega.pow_u32(t);                       // omega^(t - rot)
    Fr acc = Fr::zero();
    bool zero_den = false;
    const uint8_t* vals = a.inst + ((size_t)p * a.ninst + a.base) * 32;
    for (uint32_t j0 = t; j0 < a.len; j0 += INSTEVAL_THREADS * INSTEVAL_CHUNK) {
        // forward: prefix products of the chunk's denominators
        Fr run = Fr::one(), wc = w;
        uint32_t cnt = 0;
        for (uint32_t c = 0; c < INSTEVAL_CHUNK && j0 + c * INSTEVAL_THREADS < a.len; ++c, ++cnt) {
            pre[(size_t)c * INSTEVAL_THREADS + t] = run;
            run = run * (x - wc);
            wc = wc * a.omega_step;
        }
        w = wc;                                                  // start of the thread's next chunk
        if (run.is_zero()) { zero_den = true; continue; }
        Fr inv = run.inv();
        // backward: 1/d_c = inv * prefix_c, then inv *= d_c; omega^(j-rot) steps back by omega^-256
        for (uint32_t c = cnt; c-- > 0;) {
            wc = wc * a.omega_step_inv;
            const Fr inv_d = inv * pre[(size_t)c * INSTEVAL_THREADS + t];
            inv = inv * (x - wc);
            uint8_t tmp[32];
            const uint8_t* src = vals + (size_t)(j0 + c * INSTEVAL_THREADS) * 32;
            for (int i = 0; i < 32; ++i) tmp[i] = src[i];
            Fr v;
            if (!Fr::from_bytes(tmp, v)) v = Fr::zero();        // non-canonical values were already reported by k_check_scalars
            acc = acc + v * wc * inv_d;
        }
    }
    red[t] = acc;
    __syncthreads();
    for (uint32_t d = INSTEVAL_THREADS / 2; d > 0; d >>= 1) {
        if (t < d) red[t] = red[t] + red[t + d];
        __syncthreads();
    }
    if (zero_den) status_set(a.status, p, H2V_DEV_ST_PANIC);
    if (t == 0) {
        Fr xn = x;
        for (uint32_t i = 0; i < a.k; ++i) xn = xn.sqr();
        a.out[p] = red[0] * (xn - Fr::one()) * a.n_inv;
    }
}

// Program slots.  The hottest ones (the lowest numbers: the allocator hands out freed slots last-in first-out) live in LDS,
// [slot][limb][lane]; the rest are limb-planar in global memory: limb l of slot s of proof p is word (s * 9 + l) * n + p, so that
// a wave's 64 proofs read 64 consecutive words per limb.  Round 1 kept every slot in global memory: each of the ~600
// instructions of a program waited for two L2 round trips before its ~0.5 us of arithmetic.
// a 32-byte little-endian scalar -> Fr (zero when not canonical).  Proof scalars and instance values sit at multiples of 32
// bytes inside records whose length is a multiple of 32, in hipMalloc'ed buffers: two 16-byte loads instead of 32 byte loads.
__device__ __forceinline__ Fr fr_from_le32(const uint8_t* b) {
    const uint4 lo = reinterpret_cast<const uint4*>(b)[0], hi = reinterpret_cast<const uint4*>(b)[1];
    const uint32_t raw[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
    if (Fr::geq_p(raw)) return Fr::zero();
    return Fr::from_raw(raw);
}

struct SlotFile {
    uint32_t* lds; uint32_t lds_slots;
    Fr* glob; uint32_t n, p, lane;
    __device__ __forceinline__ Fr load(uint32_t s) const {
        Fr r;
        if (s < lds_slots) {
            const uint32_t* q = lds + (size_t)s * H2V_LIMBS * 64 + lane;
#pragma unroll
            for (int l = 0; l < H2V_LIMBS; ++l) r.v[l] = q[l * 64];
        } else {
            const uint32_t* q = reinterpret_cast<const uint32_t*>(glob) + (size_t)s * H2V_LIMBS * n + p;
#pragma unroll
            for (int l = 0; l < H2V_LIMBS; ++l) r.v[l] = q[(size_t)l * n];
        }
        return r;
    }
    __device__ __forceinline__ void store(uint32_t s, const Fr& v) const {
        if (s < lds_slots) {
            uint32_t* q = lds + (size_t)s * H2V_LIMBS * 64 + lane;
#pragma unroll
            for (int l = 0; l < H2V_LIMBS; ++l) q[l * 64] = v.v[l];
        } else {
            uint32_t* q = reinterpret_cast<uint32_t*>(glob) + (size_t)s * H2V_LIMBS * n + p;
#pragma unroll
            for (int l = 0; l < H2V_LIMBS; ++l) q[(size_t)l * n] = v.v[l];
        }
    }
};

// one instruction stream over the slot file `sf` of proof p; `live` = p is a real proof (lanes past the end of a two-stream
// workgroup keep running for the barriers but touch no memory of their own)
template <bool TWO> __device__ __forceinline__ void frvm_run(const FrvmArgs& a, const VmInstr* __restrict__ code, uint32_t n_code, const SlotFile& sf, uint32_t p, bool live) {
    const uint32_t n = a.n;
    // an operand of MUL / ADD / SUB is a slot or (VM_CONST_OPERAND) a program constant, read with uniform loads
    // (an operand that is the previous instruction's result is taken from registers, not read back from the slot file: chains —
    // Horner folds, prefix products — skip a dependent LDS round trip per instruction.  The test is on wave-uniform instruction words.)
    Fr last = Fr::zero(); uint32_t last_d = 0xffffffffu;
    auto slot = [&](uint32_t x) -> Fr { return x == last_d ? last : sf.load(x); };
    auto opnd = [&](uint32_t x) -> Fr { return (x & VM_CONST_OPERAND) ? a.consts[x & ~VM_CONST_OPERAND] : slot(x); };
    auto put = [&](uint32_t d, const Fr& v) { sf.store(d, v); last = v; last_d = d; };
    VmInstr nx = code[0];
    for (uint32_t pc = 0; pc < n_code; ++pc) {
        const VmInstr in = nx;  // wave-uniform; the next instruction is fetched while this one executes
        nx = code[pc + 1 < n_code ? pc + 1 : pc];
        switch (in.op) {
            case OP_BARRIER: if (TWO) __syncthreads(); last_d = 0xffffffffu; break;   // (behind a barrier another stream may own the slot `last` mirrored)
            case OP_CONST: put(in.d, a.consts[in.a]); break;
            case OP_MUL: put(in.d, Fr::mul_inl(opnd(in.a), opnd(in.b))); break;
            case OP_ADD: put(in.d, opnd(in.a) + opnd(in.b)); break;
            case OP_SUB: put(in.d, opnd(in.a) - opnd(in.b)); break;
            case OP_NEG: put(in.d, slot(in.a).neg()); break;
            case OP_INV: {
                Fr v = slot(in.a);
                if (live && v.is_zero()) status_set(a.status, p, H2V_DEV_ST_PANIC);
                // all lanes invert at once: the divsteps of Fp::inv are branch-free, so the wave stays uniform
                put(in.d, v.inv());
                break;
            }
            case OP_POW: put(in.d, slot(in.a).pow_u32(in.b)); break;
            case OP_SQRN: {
                Fr v = slot(in.a);
                for (uint32_t i = 0; i < in.b; ++i) v = v.sqr();
                put(in.d, v);
                break;
            }
            case OP_LOAD_SCALAR: put(in.d, fr_from_le32(a.proofs + (size_t)p * a.proof_len + a.scalar_offsets[in.a])); break;  // status already set by k_check_scalars
            case OP_LOAD_INST: put(in.d, fr_from_le32(a.inst + ((size_t)p * a.ninst + in.a) * 32)); break;
            case OP_LOAD_CHAL: put(in.d, a.chal[(size_t)in.a * n + p]); break;
            case OP_LOAD_INSTEVAL: put(in.d, a.insteval[(size_t)in.a * n + p]); break;
            case OP_LOAD_MULT: put(in.d, a.mult[p]); break;
            case OP_STORE_MSM: {
                uint32_t raw[8];
                slot(in.a).to_raw(raw);
                bool bad = status_get(a.status, p) != 0;
                uint32_t* dst = a.msm_scal + ((size_t)p * a.np + in.b) * 8;
                if (live) for (int i = 0; i < 8; ++i) dst[i] = bad ? 0u : raw[i];
                break;
            }
            case OP_STORE_GUARD: {
                uint32_t raw[8];
                slot(in.a).to_raw(raw);
                uint32_t* dst = a.guard_scal + ((size_t)p * a.n_guard + in.b) * 8;
                if (live) for (int i = 0; i < 8; ++i) dst[i] = raw[i];
                break;
            }
            case OP_STORE_SHARED: {
                Fr v = slot(in.a);
                if (status_get(a.status, p) != 0) v = Fr::zero();
                if (live) a.shared[(size_t)in.b * n + p] = v;
                break;
            }
            case OP_STORE_LEFT: {
                uint32_t raw[8];
                slot(in.a).to_raw(raw);
                bool bad = status_get(a.status, p) != 0;
                uint32_t* dst = a.left_scal + ((size_t)p * a.np + in.b) * 8;
                if (live) for (int i = 0; i < 8; ++i) dst[i] = bad ? 0u : raw[i];
                break;
            }
            default: break;
        }
    }
}
__global__ void __launch_bounds__(64) k_frvm(FrvmArgs a, uint32_t lds_slots) {
    __builtin_amdgcn_s_setprio(3);   // a latency chain: its waves win the issue arbitration over the throughput kernels of other launches in flight
    extern __shared__ uint32_t frvm_lds[];
    uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= a.n) return;
    const SlotFile sf{frvm_lds, lds_slots, a.slots, a.n, p, threadIdx.x};
    frvm_run<false>(a, a.code, a.n_code, sf, p, true);
}
// Several instruction streams per proof (vkplan.hip: Builder::emit_streams): the K waves of a workgroup work on the SAME 64 proofs —
// wave w runs stream w — and share the slot file; values cross between them over OP_BARRIER only.  While one wave sits in the batched
// inversion's single inverse (a sixth of the program's work, indivisible), the others evaluate the expressions that do not need it.
__global__ void __launch_bounds__(64 * FRVM_MAX_STREAMS) k_frvm2(FrvmArgs a, uint32_t lds_slots) {
    __builtin_amdgcn_s_setprio(3);
    extern __shared__ uint32_t frvm_lds[];
    const uint32_t w = threadIdx.x >> 6, lane = threadIdx.x & 63u, k = a.streams - 2;
    const uint32_t p_raw = blockIdx.x * 64 + lane;
    const bool live = p_raw < a.n;
    const uint32_t p = live ? p_raw : a.n - 1;   // lanes past the end shadow the last proof: they read its inputs, write nothing, and keep the barriers whole
    // (a shadow lane has its own LDS lane but shares the last proof's global slots: it writes there exactly what that proof's lane writes)
    const SlotFile sf{frvm_lds, lds_slots, a.slots, a.n, p, lane};
    frvm_run<true>(a, a.code_k[k][w], a.n_code_k[k][w], sf, p, live);
}

// msm_scal[n*np + g*n_shared + j] = canonical( sum over the proofs p of group g of shared[j][p] )
__global__ void __launch_bounds__(256) k_fold_shared(const Fr* __restrict__ shared, uint32_t n, uint32_t np, uint32_t gs, uint32_t* __restrict__ msm_scal) {
    __shared__ Fr red[256];
    uint32_t j = blockIdx.x, g = blockIdx.y, t = threadIdx.x;
    Fr acc = Fr::zero();
    for (uint32_t p = g * gs + t; p < (g + 1) * gs; p += 256) acc = acc + shared[(size_t)j * n + p];
    red[t] = acc;
    __syncthreads();
    for (uint32_t d = 128; d > 0; d >>= 1) {
        if (t < d) red[t] = red[t] + red[t + d];
        __syncthreads();
    }
    if (t == 0) {
        uint32_t raw[8]; red[0].to_raw(raw);
        uint32_t* dst = msm_scal + ((size_t)n * np + (size_t)g * gridDim.x + j) * 8;
        for (int i = 0; i < 8; ++i) dst[i] = raw[i];
    }
}


// ------------------------------------------------------------------ launchers
int decompress_stage_enqueue(hipStream_t s, const StageArgs& g) {
    const uint32_t n = g.n;
    if (!n) return 0;
    const Plan& pl = *g.plan;
    H2V_HIP_CHECK(hipMemsetAsync(g.status, 0, sizeof(int) * n, s));
    uint32_t tp = n * pl.n_points;
    hipLaunchKernelGGL(k_decompress, dim3((tp + 63) / 64), dim3(64), 0, s, g.proofs, pl.proof_len, g.pd->point_offsets, pl.n_points, pl.n_main_points, n, g.pts, g.phi, g.ycanon, g.status);
    uint32_t ts = n * (pl.n_scalars + pl.n_instance_values);
    if (ts) hipLaunchKernelGGL(k_check_scalars, dim3((ts + 255) / 256), dim3(256), 0, s, g.proofs, pl.proof_len, g.pd->scalar_offsets, pl.n_scalars, g.inst, pl.n_instance_values, n, g.status);
    H2V_HIP_CHECK(hipGetLastError());
    return 0;
}
int decompress_begin_enqueue(hipStream_t s, const StageArgs& g) {
    if (g.n) H2V_HIP_CHECK(hipMemsetAsync(g.status, 0, sizeof(int) * g.n, s));
    return 0;
}
int decompress_range_enqueue(hipStream_t s, const StageArgs& g, uint32_t p0, uint32_t p1, const uint8_t* src, uint32_t src_stride) {
    if (p1 <= p0) return 0;
    const Plan& pl = *g.plan;
    const uint32_t m = p1 - p0, tp = m * pl.n_points;
    if (!tp) return 0;
    if (!src) { src = g.proofs; src_stride = pl.proof_len; }
    hipLaunchKernelGGL(k_decompress, dim3((tp + 63) / 64), dim3(64), 0, s, src + (size_t)p0 * src_stride, src_stride, g.pd->point_offsets, pl.n_points, pl.n_main_points, m,
                       g.pts + (size_t)p0 * pl.n_points, g.phi + (size_t)p0 * pl.n_points, g.ycanon + (size_t)p0 * pl.n_points * 32, g.status + p0);
    H2V_HIP_CHECK(hipGetLastError());
    return 0;
}
int decompress_finish_enqueue(hipStream_t s, const StageArgs& g) {
    const Plan& pl = *g.plan;
    const uint32_t ts = g.n * (pl.n_scalars + pl.n_instance_values);
    if (ts) hipLaunchKernelGGL(k_check_scalars, dim3((ts + 255) / 256), dim3(256), 0, s, g.proofs, pl.proof_len, g.pd->scalar_offsets, pl.n_scalars, g.inst, pl.n_instance_values, g.n, g.status);
    H2V_HIP_CHECK(hipGetLastError());
    return 0;
}
int transcript_stage_enqueue(hipStream_t s, const StageArgs& g) {
    const uint32_t n = g.n;
    if (!n) return 0;
    const Plan& pl = *g.plan;
    uint32_t stream_len = (uint32_t)pl.stream.size();
    uint32_t n_words = g.stream_words;
    hipLaunchKernelGGL(k_stream_build, dim3((n + STREAM_PROOFS_PER_BLOCK - 1) / STREAM_PROOFS_PER_BLOCK), dim3(256), 0, s, g.pd->stream, stream_len, g.proofs, pl.proof_len, g.ycanon, pl.n_points, g.inst,
                       pl.n_instance_values, n, n_words, g.words);
    if (pl.opts.transcript == H2V_TRANSCRIPT_KECCAK256)
        hipLaunchKernelGGL(k_transcript_keccak, dim3((n + 63) / 64), dim3(64), 0, s, g.words, n_words, g.pd->squeeze_at, (uint32_t)pl.squeeze_at.size(), n, g.chal);
    else
    {
        const uint32_t nsq = (uint32_t)pl.squeeze_at.size();
        const size_t lds = ((size_t)16 * TR4_MSG_STRIDE + (size_t)16 * nsq * 8) * 8;
        if (lds > 60 * 1024) { set_last_error("transcript: too many challenges for one workgroup's LDS"); return H2V_ERR_UNSUPPORTED; }
        hipLaunchKernelGGL(k_transcript, dim3((n + 15) / 16), dim3(64), lds, s, g.words, n_words, g.pd->squeeze_at, nsq, n, g.chal);
    }
    H2V_HIP_CHECK(hipGetLastError());
    return 0;
}
int multipliers_enqueue(hipStream_t s, const uint8_t* d_tail, uint32_t n_tail, uint32_t n, uint32_t groups, Fr* d_mult) {
    hipLaunchKernelGGL(k_multipliers, dim3(groups), dim3(1024), 0, s, d_tail, n_tail / groups, n / groups, d_mult);
    H2V_HIP_CHECK(hipGetLastError());
    return 0;
}
__global__ void __launch_bounds__(256) k_gather_multipliers(const Fr* __restrict__ src, const uint32_t* __restrict__ idx, uint32_t n, Fr* __restrict__ out) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = src[idx[i]];
}
int gather_multipliers_enqueue(hipStream_t s, const Fr* d_src, const uint32_t* d_idx, uint32_t n, Fr* d_out) {
    if (!n) return 0;
    hipLaunchKernelGGL(k_gather_multipliers, dim3((n + 255) / 256), dim3(256), 0, s, d_src, d_idx, n, d_out);
    H2V_HIP_CHECK(hipGetLastError());
    return 0;
}
int instance_eval_enqueue(hipStream_t s, const InstEvalArgs& a) {
    if (!a.n) return 0;
    const size_t lds = ((size_t)INSTEVAL_CHUNK + 1) * INSTEVAL_THREADS * sizeof(Fr);   // 153 KB: one workgroup per CU
    H2V_HIP_CHECK(hipFuncSetAttribute((const void*)k_instance_eval, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));   // per device; cheap
    hipLaunchKernelGGL(k_instance_eval, dim3(a.n), dim3(INSTEVAL_THREADS), lds, s, a);
    H2V_HIP_CHECK(hipGetLastError());
    return 0;
}
int frvm_enqueue(hipStream_t s, const FrvmArgs& a, uint32_t n_slots) {
    if (!a.n) return 0;
    // LDS for the hottest slots: 2304 bytes per slot and 64 proofs.  Every wave should still get a SIMD of its own (a program is a
    // latency chain), so the budget follows the number of workgroups: the whole 160 KB of a CU while there are at most 256,
    // half of it up to 384.
    const uint32_t waves = (a.n + 63) / 64;
    // streams per proof: as many as leave every wave of the launch a SIMD of its own (1024 SIMDs)
    uint32_t K = waves <= 341 ? 4u : 2u;   // measured at 320 workgroups: 260 / 255 / 242 us with 2 / 3 / 4 streams (four still win with 1280 waves)
    if (a.force_streams > 0) K = (uint32_t)a.force_streams;   // h2v_tuning.frvm_streams (1 = the single-stream interpreter k_frvm)
    const bool two = K >= 2 && K <= FRVM_MAX_STREAMS && a.code_k[K - 2][0] && a.n_code_k[K - 2][0];
    FrvmArgs ak = a;
    if (two) { n_slots = a.n_slots_k[K - 2]; ak.streams = K; }
    // (a launch with more workgroups than that is a throughput launch — several of them are in flight — and an LDS-hungry kernel keeps
    // the other kernels' workgroups off its CUs: a small slice then)
    uint32_t budget = waves <= 256 ? 156 * 1024 : (waves <= 384 ? 78 * 1024 : 36 * 1024);
    if (a.force_lds_kb > 0) budget = (uint32_t)std::min(a.force_lds_kb, 156) * 1024;   // h2v_tuning.frvm_lds_kb
    const uint32_t lds_slots = std::min<uint32_t>(n_slots, budget / (H2V_LIMBS * 64 * 4));
    const size_t lds = (size_t)lds_slots * H2V_LIMBS * 64 * 4;
    static std::atomic<size_t> granted{0};   // raising the limit is per function and sticky; do it once per size class
    if (lds > 64 * 1024 && granted.load() < lds) {
        H2V_HIP_CHECK(hipFuncSetAttribute((const void*)k_frvm, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        H2V_HIP_CHECK(hipFuncSetAttribute((const void*)k_frvm2, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        granted.store(160 * 1024);
    }
    if (two) hipLaunchKernelGGL(k_frvm2, dim3(waves), dim3(64 * K), lds, s, ak, lds_slots);
    else hipLaunchKernelGGL(k_frvm, dim3(waves), dim3(64), lds, s, a, lds_slots);
    H2V_HIP_CHECK(hipGetLastError());
    return 0;
}
int fold_shared_enqueue(hipStream_t s, const Fr* d_shared, uint32_t n, uint32_t np, uint32_t n_shared, uint32_t groups, uint32_t* d_msm_scal) {
    if (!n_shared) return 0;
    hipLaunchKernelGGL(k_fold_shared, dim3(n_shared, groups), dim3(256), 0, s, d_shared, n, np, n / groups, d_msm_scal);
    H2V_HIP_CHECK(hipGetLastError());
    return 0;
}

}  // namespace h2v
