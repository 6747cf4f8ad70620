// Host-side exerciser of the product's own field arithmetic (halo2_verifier_amd/csrc/bn254.hip.h is __host__ __device__:
// the plan compiler and the G2 line precomputation run it on the CPU).  Reads lines "op a b" with a, b 64-hex-digit
// big-endian residues (op in: mul sqr add sub neg inv dot2 [a b c d] ...) from stdin for the field named by argv[1]
// (fq | fr) and prints the canonical result as 64 hex digits.  tests/test_field_host.py drives it against Python big ints.
#include <cstdio>
#include <cstring>
#include <string>
#include "../../halo2_verifier_amd/csrc/bn254.hip.h"
using namespace h2v;

template <class F> struct FP29;
template <> struct FP29<Fq> { static uint32_t limb(int l) { return FqParams::P29(l); } };
template <> struct FP29<Fr> { static uint32_t limb(int l) { return FrParams::P29(l); } };
static bool parse(const char* hex, uint32_t raw[8]) {
    if (strlen(hex) != 64) return false;
    for (int w = 0; w < 8; ++w) {
        unsigned v = 0;
        if (sscanf(hex + 64 - 8 * (w + 1), "%8x", &v) != 1) return false;
        raw[w] = v;
    }
    return true;
}
template <class F> static void print(const F& x) {
    uint32_t raw[8]; x.to_raw(raw);
    for (int w = 7; w >= 0; --w) printf("%08x", raw[w]);
    printf("\n");
}
template <class F> static int run() {
    char op[16], a[80], b[80], c[80], d[80];
    char line[512];
    while (fgets(line, sizeof line, stdin)) {
        int k = sscanf(line, "%15s %79s %79s %79s %79s", op, a, b, c, d);
        if (k < 2) continue;
        uint32_t ra[8], rb[8], rc[8], rd[8];
        if (!parse(a, ra)) return 2;
        F x = F::from_raw(ra), y = F::zero(), z = F::zero(), w = F::zero();
        if (k >= 3) { if (!parse(b, rb)) return 2; y = F::from_raw(rb); }
        if (k >= 4) { if (!parse(c, rc)) return 2; z = F::from_raw(rc); }
        if (k >= 5) { if (!parse(d, rd)) return 2; w = F::from_raw(rd); }
        std::string o = op;
        if (o == "mul") print(x * y);
        else if (o == "sqr") print(x.sqr());
        else if (o == "add") print(x + y);
        else if (o == "sub") print(x - y);
        else if (o == "neg") print(x.neg());
        else if (o == "dbl") print(x.dbl());
        else if (o == "inv") print(x.inv());
        else if (o == "invf") print(x.inv_fermat());          // the fixed-exponent chain, kept as the cross-check of inv()
        else if (o == "invl") print((x + y - y).inv());       // a lazily reduced operand (representative in [p, 2p) half of the time)
        else if (o == "dot2") print(F::dot2_inl(x, y, z, w));
        else if (o == "lazy") {             // the lazy linear forms: raw limbs out (the test checks the INTEGER identities and the limb bounds), then products of lazy operands
            const F a = (x + y) - y, b = (z + w) - w;      // representatives in [p, 2p) half of the time
            const F s1 = F::lazy_sub(a, b), s2 = F::lazy_dbl(s1), s3 = F::template lazy_lin<8, 1, 2>(s1, b), s4 = F::lazy_neg2(b), s5 = F::lazy_add2(s4, a), s6 = F::lazy_neg(b);
            const F* all[8] = {&a, &b, &s1, &s2, &s3, &s4, &s5, &s6};
            for (const F* f : all) { for (int i = 0; i < 9; ++i) printf("%x%c", f->v[i], i == 8 ? ' ' : ','); }
            printf("\n");
            print(F::mul_inl(s3, s5)); print(s2.sqr_inl()); print(F::dot2_inl(s2, s5, s3, s4));
        }
        else if (o == "chain") {            // (((x*y - x) + y)^2 - y) * x : lazily reduced intermediates feed every kind of operation
            F t = x * y - x; t = t + y; t = t.sqr() - y; print(t * x);
        } else if (o == "wide") {           // 9x + (2p - y) + 9(2p - z) + w + x + y as ONE integer combination of the limbs, then from_wide
            int64_t acc[9];
            for (int l = 0; l < 9; ++l) {
                const int64_t p2 = 2 * (int64_t)FP29<F>::limb(l);
                acc[l] = 9 * (int64_t)x.v[l] + (p2 - (int64_t)y.v[l]) + 9 * (p2 - (int64_t)z.v[l]) + (int64_t)w.v[l] + (int64_t)x.v[l] + (int64_t)y.v[l];
            }
            print(F::from_wide(acc));
        } else if (o == "eq") printf("%d\n", (x == y) ? 1 : 0);
        else if (o == "iszero") printf("%d\n", (x - y).is_zero() ? 1 : 0);
        else if (o == "mont256") print(F::from_mont256(ra));   // a = residue * 2^256 mod p as words
        else return 3;
    }
    return 0;
}
int main(int argc, char** argv) {
    if (argc != 2) return 1;
    return std::string(argv[1]) == "fq" ? run<Fq>() : run<Fr>();
}
