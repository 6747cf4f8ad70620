// Final pairing check kernel:  e(left, s_g2) * e(right, -g2) == 1  (DualMSM::check,
// poly/kzg/msm.rs:185-203).
//
// A pairing is a strictly sequential chain of ~470 Fq12 operations (Miller loop over 6x+2, then the final
// exponentiation); a batch ends with exactly one of them, so what counts is the LATENCY of one Fq12 operation.
//
// The whole pairing is a TABLE of ~470 Fq12 operations (built once per context on the host: Miller loop, easy part, the x-power chain
// of the hard part) interpreted by one loop; an Fq12 element lives in LDS as six Fq2 coefficients of Fq2[w]/(w^6 - xi).
//   k_pairing    one operation stream, two waves per check;  k_pairing2: two streams side by side, four waves per check (the checks
//                over split accumulators: every launch of at most 64 groups).  Both run the same step:
//   pair_step6   a product in ONE phase: eight lanes per output coordinate, each an unreduced 18-limb a0 B0 + a1 B1, the limbs added
//                across the lanes by DPP, one Montgomery reduction per output; every register coefficient is kept in six forms
//                (c0, c1, -c1, xi c and its negated imaginary part) so that no term needs a subtraction or a multiplication by xi;
//                one barrier per step, products never in place (pair_rename_registers).  Details above k_pairing2.
//   k_pair_lines the line products of every Miller iteration, all iterations at once, before the sequential part (it still uses the
//                older two-phase product: dot2 lanes into LDS, then fold lanes — fq12_fold below).
// History: round 1 applied xi to an operand inside every product lane and ran through ~100 call sites with spills around each; round 2
// made the table and the two-phase product (72 dot2 lanes, 18 fold outputs); round 3 measured that step at 2390 + 2650 cycles plus two
// barriers and replaced it.
//
// The G2 side is constant per context, so its Miller-loop line coefficients are precomputed once on the host
// (g2_prepare); the lanes evaluate them at the two G1 points up front, all lines in parallel, into LDS.
// The G1 points are used projectively (line values scaled by Z^3 in Fq*, which the final exponentiation kills),
// so no field inversion is needed for them.  The inverse of the final exponentiation (f^-1 for f^(p^6 - 1)) descends by norms:
// N = f conj(f) lies in Fq6, N^-1 = N^(p^2) N^(p^4) / Norm(N) with Norm(N) in Fq2, 1 / Norm(N) = conj(Norm(N)) / nu with nu in Fq —
// four wave products, and the division by nu is NOT done: the scalar rides through the rest of the table and the final test
// is "the value lies in Fq*" instead of "the value is 1" (pairing_program explains why that is the same verdict).
//
// SingleStrategy (one check per proof) launches one such workgroup per proof.
#include "../../include/h2v.h"
#include "internal.h"
#include "pairing.hip.h"
#include "pairing_api.h"

namespace h2v {

#define N_LINES H2V_PAIRING_LINES
#define PAIR_ITERS 66        // line products of the merged program: 64 Miller iterations + the two Frobenius corrections
#define PAIR_THREADS 128
#define PAIR_REGS 17         // logical registers of the single-stream table
#define PAIR1_REGS 19        // physical registers of k_pairing (a product writes a fresh one: pair_rename_registers)
#define PAIR_MAX_OPS 512

enum PairOpCode : uint32_t { P_SQR = 1, P_MUL = 2, P_MULL = 3, P_CONJ = 4, P_FROB = 5, P_CONJ0 = 6, P_COPY = 7, P_CHECK = 8, P_FROB2 = 9, P_FROB3 = 10, P_FROB4 = 11 };   // P_FROBn: x^(p^n)
static inline uint32_t pair_op(uint32_t op, uint32_t d, uint32_t a, uint32_t b) { return op | (d << 8) | (a << 16) | (b << 24); }

// Physical registers.  The kernels run a product in ONE phase — operands read at its start, result written at its end, one barrier per
// step — so a product's destination must not be a register that anything in the same step still reads (f <- f^2 in place would race).
// Every product (and every Frobenius: its lanes span two waves) therefore writes a fresh physical register; the one its logical
// register held before is free again from the next step on.  `cols`: the operation columns of a step (one, or two for k_pairing2);
// at most one renamed operation per column and step, so n_phys - n_logical >= the number of columns.  Conjugations and copies stay
// in place: each lane reads its coefficient before it writes it, all within one wave.
static void pair_rename_registers(std::vector<std::vector<uint32_t>*> cols, uint32_t n_logical, uint32_t n_phys) {
    std::vector<uint32_t> map(n_logical), spare;
    for (uint32_t r = 0; r < n_logical; ++r) map[r] = r;
    for (uint32_t r = n_phys; r-- > n_logical;) spare.push_back(r);
    const size_t steps = cols[0]->size();
    for (size_t i = 0; i < steps; ++i) {
        std::vector<std::pair<uint32_t, uint32_t>> renamed;   // (logical, new physical)
        std::vector<uint32_t> release;
        for (auto* col : cols) {   // sources through the map as it stands BEFORE this step, for every column
            uint32_t& x = (*col)[i];
            const uint32_t op = x & 255u, d = (x >> 8) & 255u, a = (x >> 16) & 255u, b = x >> 24;
            if (!op) continue;
            const bool fresh = (op >= P_SQR && op <= P_MULL) || op == P_FROB || op >= P_FROB2;
            const uint32_t pa = map[a], pb = op == P_MUL ? map[b] : b;   // P_MULL: b is a line index
            uint32_t pd2 = op == P_CHECK ? 0u : map[d];
            if (fresh) { pd2 = spare.back(); spare.pop_back(); release.push_back(map[d]); renamed.push_back({d, pd2}); }
            x = pair_op(op, pd2, pa, pb);
        }
        for (auto& r : renamed) map[r.first] = r.second;
        for (uint32_t r : release) spare.push_back(r);
    }
}

// The operation table of one pairing check (host, once per context).  Registers: 0 = f, 1 = r, 2.. = temporaries.
std::vector<uint32_t> pairing_program(bool merged) {
    std::vector<uint32_t> p;
    enum { F = 0, R = 1, T0 = 2, T1 = 3, T2 = 4, T3 = 5, T4 = 6, T5 = 7, T6 = 8 };
    auto sqr = [&](uint32_t d, uint32_t a) { p.push_back(pair_op(P_SQR, d, a, 0)); };
    auto mul = [&](uint32_t d, uint32_t a, uint32_t b) { p.push_back(pair_op(P_MUL, d, a, b)); };
    auto conj = [&](uint32_t d, uint32_t a) { p.push_back(pair_op(P_CONJ, d, a, 0)); };
    auto frob = [&](uint32_t d, uint32_t a, uint32_t n = 1) { p.push_back(pair_op(n == 1 ? P_FROB : (n == 2 ? P_FROB2 : (n == 3 ? P_FROB3 : P_FROB4)), d, a, 0)); };   // ^(p^n)
    auto copy = [&](uint32_t d, uint32_t a) { p.push_back(pair_op(P_COPY, d, a, 0)); };
    // Miller loop: one squaring and one line product per doubling step, one more line product per addition step
    // (merged: the doubling and addition lines of one iteration arrive already multiplied together, k_pair_lines: one line
    // product per iteration, 36 fewer dependent operations)
    uint32_t idx = 0;
    for (int i = 63; i >= 0; --i) {
        sqr(F, F);
        p.push_back(pair_op(P_MULL, F, F, idx++));
        if (!merged && ((ATE_LOW >> i) & 1)) p.push_back(pair_op(P_MULL, F, F, idx++));
    }
    p.push_back(pair_op(P_MULL, F, F, idx++));
    p.push_back(pair_op(P_MULL, F, F, idx++));
    // final exponentiation, easy part: r = f^((p^6 - 1)(p^2 + 1))
    conj(T1, F);                       // conj(f) = f^(p^6)
    mul(T2, F, T1);                    // N = f conj(f), in Fq6 (even powers of w)
    frob(T3, T2, 2);                   // N^(p^2)
    frob(T4, T2, 4);                   // N^(p^4)
    mul(T3, T3, T4);                   // T = N^(p^2) N^(p^4)
    mul(T4, T2, T3);                   // Norm(N) = N T, in Fq2 (coefficient 0)
    // No inversion: 1 / Norm(N) = conj(Norm(N)) / nu with nu = |Norm(N)|^2 in Fq, and the scalar is carried instead of divided out.  From
    // here on every value is lambda^e times what the textbook chain holds, lambda in Fq*: products add the exponents, squarings double
    // them, conjugations (the hard part's inverses) and Frobenius maps leave lambda alone, so the table ends with lambda^E y, y the true
    // result.  y lies in the group of r-th roots of unity, which meets Fq* in {1} (r does not divide p - 1): lambda^E y is in Fq* exactly
    // when y = 1 — which is what P_CHECK tests.  (The inversion was 116 000 cycles, 48 us, of every check: 8 % of k_pairing2.)
    p.push_back(pair_op(P_CONJ0, T4, T4, 0));   // conj(Norm(N)) = nu / Norm(N)
    mul(T3, T3, T4);                   // nu N^-1
    mul(T0, T1, T3);                   // nu f^-1
    mul(R, T1, T0);                    // nu f^(p^6 - 1)
    frob(T0, R, 2);
    mul(R, T0, R);                     // ^(p^2 + 1)
    // hard part: the x-power chain of Fuentes-Castaneda et al. (y0 .. y16); inverses in the cyclotomic subgroup are conjugates
    // d = a^BN_X for a in the cyclotomic subgroup (a^-1 = conj(a)), d != a: width-4 signed windows over the exponent — the odd
    // powers a, a^3, a^5, a^7 and their conjugates are tabulated (4 products + 4 conjugations), then 62 squarings and 13 products
    // instead of the 27 of plain square-and-multiply: 10 fewer dependent Fq12 operations per x-power, 30 per pairing
    const uint32_t TAB = 9;   // registers TAB + 2 j: a^(2j+1), TAB + 2 j + 1: its conjugate
    auto pow_x = [&](uint32_t d, uint32_t a) {
        std::vector<int> dig;   // width-4 non-adjacent form, least significant first
        for (unsigned long long n = BN_X; n;) {
            int z = 0;
            if (n & 1) { z = (int)(n & 15); if (z >= 8) z -= 16; n -= (unsigned long long)(long long)z; }
            dig.push_back(z);
            n >>= 1;
        }
        copy(TAB, a);
        sqr(d, a);                                   // a^2 (d is free until the main loop starts)
        for (uint32_t j = 1; j < 4; ++j) mul(TAB + 2 * j, TAB + 2 * (j - 1), d);   // a^3, a^5, a^7
        for (uint32_t j = 0; j < 4; ++j) conj(TAB + 2 * j + 1, TAB + 2 * j);
        bool started = false;
        for (size_t i = dig.size(); i-- > 0;) {
            if (started) sqr(d, d);
            const int z = dig[i];
            if (!z) continue;
            const uint32_t reg = TAB + 2 * (uint32_t)((z < 0 ? -z : z) >> 1) + (z < 0 ? 1 : 0);
            if (started) mul(d, d, reg); else { copy(d, reg); started = true; }
        }
    };
    const uint32_t y0 = T0, y1 = T1, y3 = T2, y4 = T3, y6 = T4, u = T5, v = T6;
    pow_x(y0, R); conj(y0, y0);        // y0 = r^-x
    sqr(y1, y0);                       // y1 = y0^2
    sqr(u, y1);                        // y2 = y1^2
    mul(y3, u, y1);                    // y3 = y2 y1
    pow_x(y4, y3); conj(y4, y4);       // y4 = y3^-x
    sqr(u, y4);                        // y5 = y4^2
    pow_x(y6, u); conj(y6, y6);        // y6 = y5^-x
    conj(y3, y3);
    conj(y6, y6);
    mul(u, y6, y4);                    // y7 = y6 y4
    mul(u, u, y3);                     // y8 = y7 y3            (u = y8)
    mul(v, u, y1);                     // y9 = y8 y1            (v = y9)
    mul(y0, u, y4);                    // y10 = y8 y4
    mul(y0, y0, R);                    // y11 = y10 r           (y0 = y11)
    frob(y1, v);                       // y12 = y9^p
    mul(y0, y1, y0);                   // y13 = y12 y11
    frob(u, u, 2);                     // y8^(p^2)
    mul(y0, u, y0);                    // y14
    conj(y1, R);
    mul(y1, y1, v);                    // conj(r) y9
    frob(y1, y1, 3);                   // y15
    mul(y0, y1, y0);                   // y16
    p.push_back(pair_op(P_CHECK, 0, y0, 0));
    pair_rename_registers({&p}, PAIR_REGS, PAIR1_REGS);
    return p;
}

// ---- The same check as TWO operation streams run side by side by two groups of two waves (k_pairing2; checks over split
// accumulators, i.e. the latency-critical ones).  A pairing is a chain, but its long stretches are Horner schemes in the exponent and
// can be cut like the MSM's:
//   Miller loop   f = prod_i L_i^(2^(63 - i)): group A runs iterations 0 .. 21 and then squares its value 42 more times, group B runs
//                 iterations 22 .. 63 and the two Frobenius corrections, one product joins them: 86 steps instead of 130;
//   x-powers      a^x = prod over the set bits k of a^(2^k), right to left: A runs the 62 squarings, B multiplies the squares in as they
//                 appear — 63 steps per x-power (a left-to-right chain pays its multiplications in sequence: 70 with signed windows
//                 cut between the streams, 75 on one stream);
//   the tail of the hard part pairs what is independent (y9 | y10, y12 | y11, ...).
// Both groups step together (one barrier per step); a step's two operations never write, or write and read, the same register.
// One step = uint2 (operation of A, operation of B), 0 = nothing to do.
#define PAIR2_LOGICAL_REGS 19   // the table below names registers 0 .. 18
#define PAIR2_REGS 22           // physical registers (pair_rename_registers)
#define PAIR2_MAX_STEPS H2V_PAIR2_MAX_STEPS
std::vector<uint32_t> pairing_program2() {
    std::vector<uint32_t> A, B;
    enum { FA = 0, FB = 1, R = 2, T0 = 3, T1 = 4, T2 = 5, T3 = 6, T4 = 7, T5 = 8, T6 = 9, TAB = 10, D2 = 18 };
    auto sync = [&]() { while (A.size() < B.size()) A.push_back(0); while (B.size() < A.size()) B.push_back(0); };
    auto a1 = [&](uint32_t w) { sync(); A.push_back(w); B.push_back(0); };                 // a step of A alone
    auto ab = [&](uint32_t wa, uint32_t wb) { sync(); A.push_back(wa); B.push_back(wb); };  // a step of both
    auto SQR = [](uint32_t d, uint32_t a) { return pair_op(P_SQR, d, a, 0); };
    auto MUL = [](uint32_t d, uint32_t a, uint32_t b) { return pair_op(P_MUL, d, a, b); };
    auto MULL = [](uint32_t d, uint32_t a, uint32_t l) { return pair_op(P_MULL, d, a, l); };
    auto CONJ = [](uint32_t d, uint32_t a) { return pair_op(P_CONJ, d, a, 0); };
    auto FROB = [](uint32_t d, uint32_t a) { return pair_op(P_FROB, d, a, 0); };
    auto FROB2 = [](uint32_t d, uint32_t a) { return pair_op(P_FROB2, d, a, 0); };
    auto FROB3 = [](uint32_t d, uint32_t a) { return pair_op(P_FROB3, d, a, 0); };
    auto FROB4 = [](uint32_t d, uint32_t a) { return pair_op(P_FROB4, d, a, 0); };
    auto COPY = [](uint32_t d, uint32_t a) { return pair_op(P_COPY, d, a, 0); };
    // Miller loop over the merged line table (entry i: iteration i; 64, 65: the Frobenius corrections); FA = FB = 1 at the start
    const int K = 22;
    for (int i = 0; i < K; ++i) { if (i) A.push_back(SQR(FA, FA)); A.push_back(MULL(FA, FA, (uint32_t)i)); }
    for (int i = K; i < 64; ++i) A.push_back(SQR(FA, FA));
    for (int i = K; i < 64; ++i) { if (i > K) B.push_back(SQR(FB, FB)); B.push_back(MULL(FB, FB, (uint32_t)i)); }
    B.push_back(MULL(FB, FB, 64)); B.push_back(MULL(FB, FB, 65));
    a1(MUL(FA, FA, FB));
    // final exponentiation, easy part: r = f^((p^6 - 1)(p^2 + 1))
    const uint32_t F = FA;
    a1(CONJ(T1, F));                       // conj(f) = f^(p^6)
    ab(MUL(T2, F, T1), SQR(T5, T1));       // N = f conj(f), in Fq6 (even powers of w)   | conj(f)^2
    ab(FROB2(T3, T2), FROB4(T4, T2));      // N^(p^2)               | N^(p^4)
    a1(MUL(T3, T3, T4));                   // T = N^(p^2) N^(p^4)
    a1(MUL(T4, T2, T3));                   // Norm(N) = N T, in Fq2 (coefficient 0)
    a1(pair_op(P_CONJ0, T4, T4, 0));       // conj(Norm(N)) = nu / Norm(N), nu in Fq: the scalar is carried, not divided out (pairing_program)
    a1(MUL(T3, T3, T4));                   // nu N^-1
    a1(MUL(R, T5, T3));                    // nu f^(p^6 - 1) = conj(f)^2 nu N^-1
    a1(FROB2(T0, R));
    a1(MUL(R, T0, R));                     // ^(p^2 + 1)
    // d = a^BN_X, d != a, RIGHT TO LEFT over the bits of x: A squares s <- s^2 (s_k = a^(2^k) in D2), B multiplies d by s_k wherever bit k is
    // set — one step behind A, in the same steps — so an x-power is as deep as its 62 squarings + 1 (round 3; until then left to right
    // with signed windows, the digits cut between the streams: 5 steps of table + 64 + 1 to join = 70).  Bit 0 of x is set: d starts as a.
    int top = 63;
    while (!((BN_X >> top) & 1ull)) --top;
    // `side`: operations that touch neither a, d nor D2, run in order in B's free steps (the zero bits of x)
    auto pow_x = [&](uint32_t d, uint32_t a, std::vector<uint32_t> side = {}) {
        ab(SQR(D2, a), COPY(d, a));
        size_t done = 0;
        for (int k = 1; k <= top; ++k) {
            uint32_t wb = ((BN_X >> k) & 1ull) ? MUL(d, d, D2) : 0u;
            if (!wb && done < side.size()) wb = side[done++];
            ab(k < top ? SQR(D2, D2) : 0u, wb);
        }
        for (; done < side.size(); ++done) a1(side[done]);   // (x has zero bits to spare)
    };
    static_assert(BN_X & 1ull, "pow_x starts from bit 0");
    // The chain of Fuentes-Castaneda et al. wants y0 = r^-x, y4 = y3^-x, y6 = y5^-x (inverses = conjugates in the cyclotomic subgroup).
    // With z = r^x: y1 = conj(z^2), y3 = conj(z^6), so y4 = conj(y3^x) = (z^6)^x and conj(y3) = z^6 — the two values the tail uses —
    // need no conjugation at all; y1's runs in B's column beside a squaring (into a register of its own: A reads z^2 in the same step).
    const uint32_t y0 = T0, y1 = TAB, z2 = T1, y3 = T2, y4 = T3, y6 = T4, u = T5, v = T6;
    pow_x(y0, R);                          // z = r^x
    a1(SQR(z2, y0));                       // z^2
    ab(SQR(u, z2), CONJ(y1, z2));          // z^4                   | y1 = conj(z^2)
    a1(MUL(y3, u, z2));                    // z^6 = conj(y3)        (y3 holds the conjugate, which is what the tail multiplies by)
    pow_x(y4, y3);                         // y4 = y3^-x = (z^6)^x
    a1(SQR(u, y4));                        // y5 = y4^2
    const uint32_t cr = TAB + 1, yr = TAB + 2, w = TAB + 3, y43 = TAB + 4;
    pow_x(y6, u, {CONJ(cr, R), MUL(yr, y4, R), MUL(y43, y4, y3)});   // y5^x = conj(y6) (likewise)      | beside it: conj(r), y4 r, y4 y3
    a1(MUL(u, y6, y43));                   // y8 = y7 y3 = y6 (y4 y3)      (u = y8)
    ab(MUL(v, u, y1), MUL(y0, u, yr));     // y9 = y8 y1 (v)        | y11 = y8 y4 r
    ab(FROB(z2, v), MUL(w, cr, v));        // y12 = y9^p            | conj(r) y9
    ab(MUL(y0, z2, y0), FROB3(w, w));      // y13 = y12 y11         | y15 = (conj(r) y9)^(p^3)
    ab(MUL(y0, y0, w), FROB2(u, u));       // y13 y15               | y8^(p^2)
    a1(MUL(y0, y0, u));                    // y16 = y8^(p^2) y13 y15
    a1(pair_op(P_CHECK, 0, y0, 0));
    sync();
    pair_rename_registers({&A, &B}, PAIR2_LOGICAL_REGS, PAIR2_REGS);
    std::vector<uint32_t> steps;
    for (size_t i = 0; i < A.size(); ++i) { steps.push_back(A[i]); steps.push_back(B[i]); }
    return steps;
}

// ---- k_pair_lines' product (the two-phase form: dot2 lanes into LDS, then these fold lanes); the sequential kernels use pair_step6 below
struct Coef { Fq c0, c1, n1; };   // an Fq2 coefficient and the negated imaginary part: n1 = -c1

// k-fold multiples of p on the limbs (offsets that keep the fold's integer combinations non-negative)
__device__ __forceinline__ int64_t p_times(int l, int64_t k) { return k * (int64_t)FqParams::P29(l); }

// The fold of one Fq12 product: prod[6 i + j] = a_i * b_j -> the six coefficients of the result, w^6 = xi applied.  Lanes t < 72
// (whole quads): 18 outputs (coefficient k, kind 0 = re / 1 = im / 2 = -im), a quad of lanes each.  Lane 0 of the quad sums
// the partial products with i + j = k ("low"), lane 1 those with i + j = k + 6 in the output's own coordinate, lane 2
// the same entries in the other coordinate; lane 0 then collects the three sums (DPP quad broadcasts), applies
// w^6 = xi = 9 + u as integer weights and reduces once.  One lane per output did all three sums one after the other —
// twice the instructions on the critical path.
__device__ __forceinline__ void fq12_fold(const Fq2* prod, Coef* dst, uint32_t t) {
    const uint32_t o = t >> 2, role = t & 3u, k = o % 6, kind = o / 6;
    const bool im = kind != 0;
    // a_i b_j with i + j = k sits at entry k + 5 i (i <= k); with i + j = k + 6 at entry k + 6 + 5 i (i > k)
    const bool same = role != 2;                           // which coordinate of the entry this lane adds up
    const uint32_t first = role == 0 ? 0 : k + 1, last = role == 0 ? k + 1 : (role == 3 ? 0 : 6);
    const Fq2* e = &prod[k + 5 * first + (role == 0 ? 0 : 6)];
    const Fq* q = (im == same) ? &e->c1 : &e->c0;
    // a fixed six steps with the loads of all of them in flight at once (a loop over [first, last) waited for LDS every round;
    // the lanes of a quad differ in their counts, so the wave ran the maximum anyway): steps outside the lane's range read a zero entry
    uint32_t sum[9];
#pragma unroll
    for (int l = 0; l < 9; ++l) sum[l] = 0;
    const uint32_t cnt = last > first ? last - first : 0;
    const Fq* zero_entry = &prod[36].c0;               // entry 36 of every product array is kept at zero: steps beyond the lane's count read it
#pragma unroll
    for (uint32_t ii = 0; ii < 6; ++ii) {
        const Fq* qq = ii < cnt ? q + 10 * ii : zero_entry;   // 10 Fq = 5 entries
#pragma unroll
        for (int l = 0; l < 9; ++l) sum[l] += qq->v[l];
    }
    // partial products are < 1.05p: re = lo + 9 hs - ho + 6p in (0, 60p); im = lo + 9 hs + ho < 59p; -im = 60p - im
    int64_t acc[9];
#pragma unroll
    for (int l = 0; l < 9; ++l) {
        const uint32_t hs = (uint32_t)__builtin_amdgcn_mov_dpp((int)sum[l], 0x55, 0xf, 0xf, true);   // quad_perm [1,1,1,1]
        const uint32_t ho = (uint32_t)__builtin_amdgcn_mov_dpp((int)sum[l], 0xaa, 0xf, 0xf, true);   // quad_perm [2,2,2,2]
        const int64_t pos = (int64_t)sum[l] + 9 * (int64_t)hs;
        if (kind == 0) acc[l] = pos - (int64_t)ho + p_times(l, 6);
        else if (kind == 1) acc[l] = pos + (int64_t)ho;
        else acc[l] = p_times(l, 60) - pos - (int64_t)ho;
    }
    if (role == 0) {
        const Fq r = Fq::from_wide(acc);
        Coef* out = &dst[k];
        if (kind == 0) out->c0 = r; else if (kind == 1) out->c1 = r; else out->n1 = r;
    }
}

// ---- k_pairing2: two operation streams per check (pairing_program2), 256 threads, group g = t / 128 executes column g of every step;
// the line products always come from k_pair_lines.
//
// A product is ONE phase (round 3; until then: 72 dot2 lanes, LDS, barrier, fold lanes, barrier — measured inside this kernel at
// 2390 + 80 + 2650 + 70 cycles per step, the fold the larger half; k_pair_lines' tree still has that form).  The result's coordinate
// (k, re / im) is a sum of six Fq2-product coordinates, a_i * b_j over i + j = k and xi a_i * b_j over i + j = k + 6, i.e. of twelve Fq
// products.  EIGHT LANES own one output: lane i of the group computes term i as the 18-limb integer a_i0 B0 + a_i1 B1 WITHOUT a
// reduction (162 multiply-adds, one carry sweep), the eight lanes add their limbs with three DPP steps (quad_perm, quad_perm,
// row_half_mirror: 32-bit adds, six terms of 29-bit limbs fit), and ONE Montgomery reduction per output follows — 12 instead of 72,
// no partial products through LDS, no barrier between "products" and "fold".  No subtraction anywhere: a register keeps SIX forms of
// every coefficient c = c0 + c1 u — c0, c1, -c1, and xi c = (9 c0 - c1) + (9 c1 + c0) u with its negated imaginary part — so B0, B1
// are always stored values: re: (c0, -c1) or (xi c)_0, -(xi c)_1; im: (c1, c0) or (xi c)_1, (xi c)_0.  The groups of (k, re) and
// (k, im) share a 16-lane row: they swap their reduced values with one row rotation and lanes 0 .. 5 of the re group each derive and
// store one of the six forms (integer weights on the limbs, one from_wide: every stored form is below 2p, so the twelve products of
// an output sum to less than 48 p^2 and its reduction to less than 1.3 p).
struct Coef6 { Fq f[6]; };   // c0, c1, -c1, 9 c0 - c1, 9 c1 + c0, -(9 c1 + c0): non-negative representatives below 2p
// form `which` of the coefficient re + im u, both limb-normalised and below 4p
__device__ __forceinline__ Fq coef_form(const Fq& re, const Fq& im, uint32_t which) {
    const int32_t alpha = which == 0 ? 1 : (which == 3 ? 9 : (which == 4 ? 1 : (which == 5 ? -1 : 0)));
    const int32_t beta = which == 1 ? 1 : (which == 2 ? -1 : (which == 3 ? -1 : (which == 4 ? 9 : (which == 5 ? -9 : 0))));
    const int32_t kappa = which == 2 ? 4 : (which == 3 ? 4 : (which == 5 ? 40 : 0));   // keeps alpha re + beta im + kappa p >= 0 (and below 80p < 2^261)
    int64_t acc[9];
#pragma unroll
    for (int l = 0; l < 9; ++l) acc[l] = (int64_t)alpha * (int64_t)re.v[l] + (int64_t)beta * (int64_t)im.v[l] + (int64_t)kappa * (int64_t)FqParams::P29(l);
    return Fq::from_wide(acc);
}
template <int CTRL> __device__ __forceinline__ uint32_t dpp_u32(uint32_t x) { return (uint32_t)__builtin_amdgcn_mov_dpp((int)x, CTRL, 0xf, 0xf, true); }
// one product step of a group: lane tl = 8 * (2 k + coordinate) + term
__device__ __forceinline__ void pair_step6(uint32_t op, uint32_t rd, uint32_t ra, uint32_t rb, const Fq2 (*line)[6], Coef6 (*reg)[6], uint32_t tl) {
    const uint32_t out = tl >> 3, term = tl & 7u;
    const bool active = out < 12 && term < 6;
    const uint32_t k = active ? out >> 1 : 0, i = active ? term : 0, coord = out & 1u;
    const bool high = i > k;                          // i + j = k + 6: the term carries xi
    const uint32_t j = high ? k + 6 - i : k - i;
    // A: the factor taken as it is (a line, or coefficient i of register ra); B: coefficient j of the other factor, in the stored form the term needs
    const Fq* pa = op == P_MULL ? &line[rb][i].c0 : &reg[ra][i].f[0];   // (c0, c1) are adjacent in both
    const Coef6* pb = &reg[op == P_MULL ? ra : (op == P_SQR ? ra : rb)][j];
    const uint32_t s0 = coord ? (high ? 4u : 1u) : (high ? 3u : 0u), s1 = coord ? (high ? 3u : 0u) : (high ? 5u : 2u);
    Fq A0 = pa[0], A1 = pa[1];
    const Fq B0 = pb->f[s0], B1 = pb->f[s1];
    if (!active) { A0 = Fq::zero(); A1 = Fq::zero(); }
    // the term as an 18-limb integer (below 2 * 2p * 2p)
    uint32_t T[18];
    {
        uint64_t acc = 0;
#pragma unroll
        for (int c = 0; c < 17; ++c) {
#pragma unroll
            for (int x = (c > 8 ? c - 8 : 0); x <= (c < 8 ? c : 8); ++x) { acc += (uint64_t)A0.v[x] * B0.v[c - x]; acc += (uint64_t)A1.v[x] * B1.v[c - x]; }
            T[c] = (uint32_t)acc & H2V_LIMB_MASK;
            acc >>= 29;
        }
        T[17] = (uint32_t)acc;
    }
    // the six terms of the output, limb by limb (every lane of the eight ends with the sum)
#pragma unroll
    for (int c = 0; c < 18; ++c) {
        uint32_t x = T[c];
        x += dpp_u32<0xB1>(x);    // quad_perm [1, 0, 3, 2]
        x += dpp_u32<0x4E>(x);    // quad_perm [2, 3, 0, 1]
        x += dpp_u32<0x141>(x);   // row_half_mirror: the other quad of the eight
        T[c] = x;
    }
    // one Montgomery reduction (R = 2^261) of the 18 limbs (each below 6 * 2^29)
    Fq r;
    {
        uint64_t acc = 0;
        uint32_t m[9];
#pragma unroll
        for (int c = 0; c < 9; ++c) {
            acc += T[c];
#pragma unroll
            for (int x = 0; x < c; ++x) acc += (uint64_t)m[x] * FqParams::P29(c - x);
            m[c] = ((uint32_t)acc * FqParams::INV29) & H2V_LIMB_MASK;
            acc += (uint64_t)m[c] * FqParams::P29(0);
            acc >>= 29;
        }
#pragma unroll
        for (int c = 9; c < 17; ++c) {
            acc += T[c];
#pragma unroll
            for (int x = c - 8; x <= 8; ++x) acc += (uint64_t)m[x] * FqParams::P29(c - x);
            r.v[c - 9] = (uint32_t)acc & H2V_LIMB_MASK;
            acc >>= 29;
        }
        r.v[8] = (uint32_t)acc + T[17];
    }
    // the other coordinate of the coefficient sits eight lanes away in the same row
    Fq o;
#pragma unroll
    for (int l = 0; l < 9; ++l) o.v[l] = dpp_u32<0x128>(r.v[l]);   // row_ror:8
    const Fq f = coef_form(coord ? o : r, coord ? r : o, i);
    if (active && coord == 0) reg[rd][k].f[i] = f;
}
// coefficient-wise operations on the six-form registers.  Conjugations / copy: lane (coefficient k, form) for t < 36, all
// in the group's first wave (a lane reads its coefficient before any lane writes it: these may run in place).  Frobenius (never in
// place: pairing_program2 gives it a fresh destination): lane (k, form, coordinate) for t < 72 — conj(c) gamma^k as two dot2 over stored forms, re = c0 g0 + c1 g1, im = c0 g1 +
// (-c1) g0, one per lane of a pair; the pair swaps them (as a call to Fq2::mul on every lane this step cost 8700 cycles, a product step 3700).
__device__ __forceinline__ void pair_coefficients6(uint32_t op, uint32_t rd, uint32_t ra, Coef6 (*reg)[6], const PairingConsts* __restrict__ consts, uint32_t t) {
    if (op == P_FROB || op >= P_FROB2) {
        if (t >= 72) return;
        const uint32_t k = t / 12, which = (t % 12) >> 1, coord = t & 1u;
        const Coef6& x = reg[ra][k];
        // x^(p^n): odd n conjugates the coefficient — conj(c) g: re = c0 g0 + c1 g1, im = c0 g1 + (-c1) g0; even n: c g: re = c0 g0 + (-c1) g1,
        // im = c0 g1 + c1 g0 (g in Fq then, g1 = 0: the same two dot2 either way)
        const bool cj = op == P_FROB || op == P_FROB3;
        const Fq2 gm = (op == P_FROB ? consts->gamma1 : (op == P_FROB2 ? consts->gamma2 : (op == P_FROB3 ? consts->gamma3 : consts->gamma4)))[k];   // gamma^0 = 1
        const Fq mine = Fq::dot2_inl(x.f[0], coord ? gm.c1 : gm.c0, (coord != 0) == cj ? x.f[2] : x.f[1], coord ? gm.c0 : gm.c1);
        Fq other;
#pragma unroll
        for (int l = 0; l < 9; ++l) other.v[l] = dpp_u32<0xB1>(mine.v[l]);   // quad_perm [1, 0, 3, 2]: the pair's other coordinate
        const Fq f = coef_form(coord ? other : mine, coord ? mine : other, which);
        if (coord == 0) reg[rd][k].f[which] = f;
        return;
    }
    if (t >= 36) return;
    const uint32_t k = t / 6, which = t % 6;
    const Coef6& x = reg[ra][k];
    Fq re = x.f[0], im = x.f[1];
    if (op == P_CONJ) {            // x^(p^6): w -> -w
        if (k & 1u) { re = Fq::lazy_neg(re); im = x.f[2]; }
    } else if (op == P_CONJ0) {    // an element of Fq2 (coefficient 0): its conjugate; the other coefficients <- 0
        if (k == 0) im = x.f[2];
        else { re = Fq::zero(); im = Fq::zero(); }
    }
    reg[rd][k].f[which] = coef_form(re, im, which);   // P_COPY: as it was
}
__device__ __forceinline__ bool pair_in_fq_star6(const Coef6* x) {
    bool in = !x[0].f[0].is_zero() && x[0].f[1].is_zero();
    for (int k2 = 1; k2 < 6; ++k2) in = in && x[k2].f[0].is_zero() && x[k2].f[1].is_zero();
    return in;
}
struct alignas(16) PairShared2 {
    Fq2 line[PAIR_ITERS][6];
    Coef6 reg[PAIR2_REGS][6];
    uint2 prog[PAIR2_MAX_STEPS];
};
static_assert(sizeof(PairShared2) <= 64 * 1024, "k_pairing2's static LDS");
static_assert(sizeof(PairShared2) + H2V_AUX_LDS_RESERVE > 160 * 1024, "the auxiliary-stream kernels' LDS request must not fit beside a k_pairing2 workgroup (internal.h)");
__global__ void __launch_bounds__(2 * PAIR_THREADS, 1) k_pairing2(uint32_t n, const PairingConsts* __restrict__ consts, const uint2* __restrict__ prog, uint32_t n_steps,
                                                               const Fq2* __restrict__ pre, uint32_t* __restrict__ ok) {
    __builtin_amdgcn_s_setprio(3);
    __shared__ PairShared2 s;
    const uint32_t chk = blockIdx.x, t = threadIdx.x, g = t / PAIR_THREADS, tl = t % PAIR_THREADS;
    if (chk >= n) return;
    {
        const uint4* src = reinterpret_cast<const uint4*>(pre + (size_t)chk * PAIR_ITERS * 6);
        uint4* dst = reinterpret_cast<uint4*>(&s.line[0][0]);
        for (uint32_t k = t; k < PAIR_ITERS * 6 * sizeof(Fq2) / 16; k += 2 * PAIR_THREADS) dst[k] = src[k];
    }
    if (t < 72) {   // registers 0 and 1 (the two halves of the Miller value) start at one: lane (register, coefficient, form)
        const uint32_t k = (t % 36) / 6;
        s.reg[t / 36][k].f[t % 6] = coef_form(k == 0 ? Fq::one() : Fq::zero(), Fq::zero(), t % 6);
    }
    for (uint32_t k = t; k < n_steps; k += 2 * PAIR_THREADS) s.prog[k] = prog[k];
    __syncthreads();
    uint2 w2_next = s.prog[0];
    for (uint32_t pc = 0; pc < n_steps; ++pc) {
        const uint2 w2 = w2_next;
        w2_next = s.prog[pc + 1 < n_steps ? pc + 1 : pc];   // the next step's words are read while this one runs: no LDS round trip between a barrier and the decode
        const uint32_t w = (uint32_t)__builtin_amdgcn_readfirstlane((int)(g ? w2.y : w2.x));   // uniform per wave: decoded on the scalar unit
        const uint32_t op = w & 255u, rd = (w >> 8) & 255u, ra = (w >> 16) & 255u, rb = w >> 24;
        if (op >= P_SQR && op <= P_MULL) pair_step6(op, rd, ra, rb, s.line, s.reg, tl);
        else if (op == P_CHECK) { if (tl == 0) ok[chk] = pair_in_fq_star6(s.reg[ra]) ? 1u : 0u; }
        else if (op) pair_coefficients6(op, rd, ra, s.reg, consts, tl);
        __syncthreads();   // the one barrier of a step: a product's destination is a register nothing in its step reads (pair_rename_registers)
    }
}

// ---- k_pairing: ONE operation stream per check, two waves (a launch of more than 64 groups, SingleStrategy's one check per proof,
// h2v_pairing_check, the single-stream form of a split check): the same one-phase product step and six-form registers as k_pairing2.
// The lines come ready from k_pair_lines (`pre`, one merged line product per iteration) or are evaluated here at the two whole points
// (all H2V_PAIRING_LINES of them, two sparse values multiplied per line).
struct alignas(16) PairShared1 {
    Fq2 line[N_LINES][6];
    Coef6 reg[PAIR1_REGS][6];
    uint32_t prog[PAIR_MAX_OPS];
};
__global__ void __launch_bounds__(PAIR_THREADS) k_pairing(const G1J* __restrict__ pairs, uint32_t n, const LineCoeff* __restrict__ l_sg2,
                                                          const LineCoeff* __restrict__ l_ng2, const PairingConsts* __restrict__ consts,
                                                          const uint32_t* __restrict__ prog, uint32_t n_ops, const Fq2* __restrict__ pre, uint32_t* __restrict__ ok) {
    __builtin_amdgcn_s_setprio(3);   // a latency chain: its waves win the issue arbitration over the throughput kernels of other launches in flight
    extern __shared__ uint4 pair1_lds[];
    PairShared1& s = *reinterpret_cast<PairShared1*>(pair1_lds);
    const uint32_t chk = blockIdx.x, t = threadIdx.x;
    if (chk >= n) return;
    if (pre) {
        // the steps' line products come ready from k_pair_lines (a check over split accumulators: 2 * parts lines per step)
        const uint4* src = reinterpret_cast<const uint4*>(pre + (size_t)chk * PAIR_ITERS * 6);
        uint4* dst = reinterpret_cast<uint4*>(&s.line[0][0]);
        for (uint32_t k = t; k < PAIR_ITERS * 6 * sizeof(Fq2) / 16; k += PAIR_THREADS) dst[k] = src[k];
    } else {
        const G1J P0 = pairs[2 * chk], P1 = pairs[2 * chk + 1];
        const bool skip0 = P0.is_identity(), skip1 = P1.is_identity();
        // line l(P) = a*y + b*x*w + c*w^3 with (x, y) = (X/Z^2, Y/Z^3); scaled by Z^3: a*Y + b*X*Z*w + c*Z^3*w^3
        const Fq xz0 = P0.X * P0.Z, z30 = P0.Z.sqr() * P0.Z, xz1 = P1.X * P1.Z, z31 = P1.Z.sqr() * P1.Z;
        for (uint32_t li = t; li < N_LINES; li += PAIR_THREADS) {
            const LineCoeff q0 = l_sg2[li], q1 = l_ng2[li];
            Fq2 a0 = q0.a.scale(P0.Y), b0 = q0.b.scale(xz0), c0 = q0.c.scale(z30);
            Fq2 a1 = q1.a.scale(P1.Y), b1 = q1.b.scale(xz1), c1 = q1.c.scale(z31);
            // an identity point contributes the line value 1
            if (skip0) { a0 = Fq2::one(); b0 = Fq2::zero(); c0 = Fq2::zero(); }
            if (skip1) { a1 = Fq2::one(); b1 = Fq2::zero(); c1 = Fq2::zero(); }
            // (a0 + b0 w + c0 w^3)(a1 + b1 w + c1 w^3) = (a0a1 + xi c0c1) + (a0b1 + b0a1) w + b0b1 w^2 + (a0c1 + c0a1) w^3 + (b0c1 + c0b1) w^4
            Fq2 a0a1 = a0 * a1, b0b1 = b0 * b1, c0c1 = c0 * c1;
            Fq2 ab = (a0 + b0) * (a1 + b1) - a0a1 - b0b1;
            Fq2 ac = (a0 + c0) * (a1 + c1) - a0a1 - c0c1;
            Fq2 bc = (b0 + c0) * (b1 + c1) - b0b1 - c0c1;
            s.line[li][0] = a0a1 + c0c1.mul_xi(); s.line[li][1] = ab; s.line[li][2] = b0b1; s.line[li][3] = ac; s.line[li][4] = bc;
            s.line[li][5] = Fq2::zero();
        }
    }
    if (t < 36) s.reg[0][t / 6].f[t % 6] = coef_form(t < 6 ? Fq::one() : Fq::zero(), Fq::zero(), t % 6);   // register 0 (f) starts at one
    for (uint32_t k = t; k < n_ops; k += PAIR_THREADS) s.prog[k] = prog[k];
    __syncthreads();
    uint32_t w_next = s.prog[0];
    for (uint32_t pc = 0; pc < n_ops; ++pc) {
        const uint32_t w = (uint32_t)__builtin_amdgcn_readfirstlane((int)w_next);   // uniform: decoded on the scalar unit
        w_next = s.prog[pc + 1 < n_ops ? pc + 1 : pc];   // read while this operation runs
        const uint32_t op = w & 255u, rd = (w >> 8) & 255u, ra = (w >> 16) & 255u, rb = w >> 24;
        if (op <= P_MULL) pair_step6(op, rd, ra, rb, s.line, s.reg, t);
        else if (op == P_CHECK) { if (t == 0) ok[chk] = pair_in_fq_star6(s.reg[ra]) ? 1u : 0u; }
        else pair_coefficients6(op, rd, ra, s.reg, consts, t);
        __syncthreads();
    }
}

// ---- line products for a check over SPLIT accumulators (msm.hip: msm_final_parts).  Check `chk` is
//   prod_j e(L_j, 2^(shift j) s_g2) * e(R_j, -2^(shift j) g2) == 1,    L_j = piece (2 chk) S + j,  R_j = piece (2 chk + 1) S + j:
// 2 S Miller loops over the same 6x+2, i.e. 2 S line values per step, multiplied together here — every ITERATION (its doubling
// step and its addition step if it has one; the two Frobenius corrections count as iterations) of every check in its own
// workgroup, all at once — so that the sequential part (k_pairing) does ONE line product per iteration whatever S is.
// Per workgroup: the 2 S sparse values (a Y + b X Z w + c Z^3 w^3; the pieces arrive as (X Z, Y, Z^3)), the S sparse x sparse
// products, then a binary tree of general products over Fq2[w]/(w^6 - xi) — the same dot2 lanes and fold as k_pairing.
#define PL_THREADS 192          // two general products per pass (72 lanes each)
#define PL_MAX_PARTS MSM_MAX_PARTS
#define PL_MAX_EL (2 * PL_MAX_PARTS)   // sparse pairs per iteration: parts x (doubling, addition)
struct PairIters { uint8_t first[PAIR_ITERS], cnt[PAIR_ITERS]; };   // iteration -> its lines in the context's tables
struct PairLinesShared {
    Coef el0[PL_MAX_EL][6];                         // the tree's elements, ping-pong with u.t.el1
    union {
        struct { Coef ev[2 * PL_MAX_EL][3]; Fq2 sp[PL_MAX_EL][9]; } l;   // line values (w^0, w^1, w^3) and the sparse products A_u * B_v at [3 u + v]
        struct { Coef el1[PL_MAX_EL][6]; Fq2 prod[2][37]; } t;           // the tree: partial products of the general products in flight; [36] stays zero (fq12_fold)
    } u;
};
// (the tree's buffers share the leaves' space: 23.3 KB instead of 28.6, six workgroups per CU instead of five — the 1320 workgroups of a
// 20-group launch are all resident at once.  Measured: 50 us either way; the kernel is bound by its half-empty waves — 72 lanes per product —
// sharing SIMDs, not by a second round.)
static_assert(sizeof(PairLinesShared) * 6 <= 160 * 1024, "six k_pair_lines workgroups per CU");
__global__ void __launch_bounds__(PL_THREADS) k_pair_lines(const G1JSlot* __restrict__ ready, uint32_t S, const LineCoeff* __restrict__ tab, PairIters its, Fq2* __restrict__ out) {
    __builtin_amdgcn_s_setprio(3);   // a latency chain: its waves win the issue arbitration over the throughput kernels of other launches in flight
    __shared__ PairLinesShared s;
    const uint32_t it = blockIdx.x, chk = blockIdx.y, t = threadIdx.x;
    const uint32_t first = its.first[it], NP = S * its.cnt[it];   // sparse pair q = li * S + j: the lines (both sides) of piece j at line first + li
    // line leaf l = 2 q + side at the point ready[(2 chk + side) S + j] = (X Z, Y, Z^3): a Y + b (X Z) w + c Z^3 w^3.  Lane (l, k, component).
    for (uint32_t idx = t; idx < 2 * NP * 6; idx += PL_THREADS) {
        const uint32_t l = idx / 6, k = (idx % 6) >> 1, c = idx & 1u, q = l >> 1, side = l & 1u, j = q % S, li = q / S;
        const G1J& P = ready[(size_t)(2 * chk + side) * S + j].p;
        const LineCoeff& lc = tab[(size_t)(2 * j + side) * N_LINES + first + li];
        const Fq2& co = k == 0 ? lc.a : (k == 1 ? lc.b : lc.c);
        const Fq& f = k == 0 ? P.Y : (k == 1 ? P.X : P.Z);
        Fq r = Fq::mul_inl(c ? co.c1 : co.c0, f);
        if (P.is_identity()) r = (k == 0 && c == 0) ? Fq::one() : Fq::zero();   // an identity point contributes the line value 1
        Coef& e = s.u.l.ev[l][k];
        if (c) { e.c1 = r; e.n1 = r.neg(); } else e.c0 = r;
    }
    __syncthreads();
    // the NP sparse x sparse products, nine Fq2 products each: lane (j, u, v, coordinate), one dot2
    for (uint32_t idx = t; idx < NP * 18; idx += PL_THREADS) {
        const uint32_t j = idx / 18, pr = (idx % 18) >> 1, coord = idx & 1u;
        const Coef& a = s.u.l.ev[2 * j][pr / 3];
        const Coef& b = s.u.l.ev[2 * j + 1][pr % 3];
        const Fq r = Fq::dot2_inl(a.c0, coord ? b.c1 : b.c0, a.c1, coord ? b.c0 : b.n1);
        if (coord) s.u.l.sp[j][pr].c1 = r; else s.u.l.sp[j][pr].c0 = r;
    }
    __syncthreads();
    // (a0 + b0 w + c0 w^3)(a1 + b1 w + c1 w^3) = (a0a1 + xi c0c1) + (a0b1 + b0a1) w + b0b1 w^2 + (a0c1 + c0a1) w^3 + (b0c1 + c0b1) w^4:
    // lane (j, coefficient k, kind re / im / -im): at most two entries, xi as integer weights, one from_wide
    for (uint32_t idx = t; idx < NP * 18; idx += PL_THREADS) {
        const uint32_t j = idx / 18, k = (idx % 18) / 3, kind = idx % 3;
        Fq r = Fq::zero();
        if (k < 5) {
            const uint32_t i1 = k == 0 ? 0 : (k == 1 ? 1 : (k == 2 ? 4 : (k == 3 ? 2 : 5)));
            const uint32_t i2 = k == 0 ? 8 : (k == 1 ? 3 : (k == 3 ? 6 : 7));
            const Fq2& x = s.u.l.sp[j][i1];
            const Fq2& y = s.u.l.sp[j][i2];
            const bool im = kind != 0;
            const Fq& xs = im ? x.c1 : x.c0;            // the output's own coordinate
            const Fq& ys = im ? y.c1 : y.c0;
            const Fq& yo = im ? y.c0 : y.c1;            // the other one (xi couples them)
            const int64_t wy = k == 2 ? 0 : (k == 0 ? 9 : 1), wo = k == 0 ? (im ? 1 : -1) : 0;
            int64_t acc[9];
#pragma unroll
            for (int lm = 0; lm < 9; ++lm) {
                const int64_t v = (int64_t)xs.v[lm] + wy * (int64_t)ys.v[lm] + wo * (int64_t)yo.v[lm];   // in (-1.05p, 12p)
                acc[lm] = kind == 2 ? p_times(lm, 14) - v : v + p_times(lm, 2);
            }
            r = Fq::from_wide(acc);
        }
        Coef& e = s.el0[j][k];
        if (kind == 0) e.c0 = r; else if (kind == 1) e.c1 = r; else e.n1 = r;
    }
    __syncthreads();
    // binary tree of general products, two per pass (its buffers take the place of the leaves')
    if (t < 2) s.u.t.prod[t][36] = Fq2::zero();   // read by fq12_fold; a barrier below comes before any fold
    uint32_t m = NP, cur = 0;
    auto el = [](uint32_t which) -> Coef (*)[6] { return which ? s.u.t.el1 : s.el0; };
    while (m > 1) {   // uniform
        const uint32_t np = m >> 1;
        for (uint32_t p0 = 0; p0 < np; p0 += 2) {
            const uint32_t slot = t / 72, lt = t % 72, p = p0 + slot;
            const bool on = slot < 2 && p < np;
            if (on) {
                const uint32_t pr = lt >> 1, coord = lt & 1u, i = pr / 6, j = pr % 6;
                const Coef& a = el(cur)[2 * p][i];
                const Coef& b = el(cur)[2 * p + 1][j];
                const Fq r = Fq::dot2_inl(a.c0, coord ? b.c1 : b.c0, a.c1, coord ? b.c0 : b.n1);
                if (coord) s.u.t.prod[slot][pr].c1 = r; else s.u.t.prod[slot][pr].c0 = r;
            }
            __syncthreads();
            if (on) fq12_fold(s.u.t.prod[slot], el(cur ^ 1)[p], lt);
            __syncthreads();
        }
        if ((m & 1u) && t < 6) el(cur ^ 1)[np][t] = el(cur)[m - 1][t];   // the odd one out moves up as it is
        __syncthreads();
        m = np + (m & 1u); cur ^= 1;
    }
    if (t < 6) out[((size_t)chk * PAIR_ITERS + it) * 6 + t] = Fq2{el(cur)[0][t].c0, el(cur)[0][t].c1};
}

// k_pairing's LDS (70 KB: 102 lines, 19 six-form registers, the table) is above the 64 KB a kernel gets without asking; per device, once
static int pair1_lds_grant() {
    static_assert(sizeof(PairShared1) <= 80 * 1024, "two workgroups of k_pairing per CU");
    static_assert(sizeof(PairShared1) + H2V_AUX_LDS_RESERVE > 160 * 1024, "the auxiliary-stream kernels' LDS request must not fit beside a k_pairing workgroup");
    H2V_HIP_CHECK(hipFuncSetAttribute((const void*)k_pairing, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(PairShared1)));
    return 0;
}
int pairing_check_enqueue(hipStream_t s, const PairingDevice& pd, const G1J* d_pairs, uint32_t n, uint32_t* d_ok) {
    if (!n) return 0;
    if (!pd.prog || pd.n_ops > PAIR_MAX_OPS) { set_last_error("pairing: operation table missing or too long"); return H2V_ERR_BAD_ARGUMENT; }
    int rc = pair1_lds_grant();
    if (rc) return rc;
    hipLaunchKernelGGL(k_pairing, dim3(n), dim3(PAIR_THREADS), sizeof(PairShared1), s, d_pairs, n, pd.l_sg2, pd.l_ng2, pd.consts, pd.prog, pd.n_ops, (const Fq2*)nullptr, d_ok);
    H2V_HIP_CHECK(hipGetLastError());
    return 0;
}

int pairing_check_split_enqueue(hipStream_t s, PairingDevice& pd, const G1JSlot* d_ready, uint32_t n, uint32_t parts, uint32_t shift, void* d_line_ws, uint32_t* d_ok, bool one_stream) {
    if (!n) return 0;
    if (!pd.prog_merged || pd.n_ops_merged > PAIR_MAX_OPS) { set_last_error("pairing: operation table missing or too long"); return H2V_ERR_BAD_ARGUMENT; }
    if (!parts || parts > PL_MAX_PARTS || !d_line_ws) { set_last_error("pairing: bad split"); return H2V_ERR_BAD_ARGUMENT; }
    const LineCoeff* tab = nullptr;
    int rc = pd.split_lines(shift, parts, &tab);
    if (rc) return rc;
    Fq2* lines = reinterpret_cast<Fq2*>(d_line_ws);
    PairIters its;
    uint32_t line = 0;
    for (int i = 63; i >= 0; --i) {
        const uint32_t cnt = 1u + (uint32_t)((ATE_LOW >> i) & 1);
        its.first[63 - i] = (uint8_t)line; its.cnt[63 - i] = (uint8_t)cnt;
        line += cnt;
    }
    for (int i = 64; i < PAIR_ITERS; ++i) { its.first[i] = (uint8_t)line++; its.cnt[i] = 1; }
    hipLaunchKernelGGL(k_pair_lines, dim3(PAIR_ITERS, n), dim3(PL_THREADS), 0, s, d_ready, parts, tab, its, lines);
    // (one_stream: h2v_tuning.pairing_one_stream — the single-stream table over the same lines)
    if (pd.prog2 && !one_stream) hipLaunchKernelGGL(k_pairing2, dim3(n), dim3(2 * PAIR_THREADS), 0, s, n, pd.consts, reinterpret_cast<const uint2*>(pd.prog2), pd.n_steps2, (const Fq2*)lines, d_ok);
    else {
        if ((rc = pair1_lds_grant())) return rc;
        hipLaunchKernelGGL(k_pairing, dim3(n), dim3(PAIR_THREADS), sizeof(PairShared1), s, (const G1J*)nullptr, n, pd.l_sg2, pd.l_ng2, pd.consts, pd.prog_merged, pd.n_ops_merged, (const Fq2*)lines, d_ok);
    }
    H2V_HIP_CHECK(hipGetLastError());
    return 0;
}

}  // namespace h2v
