// Host-side exerciser of the product's G1 group law (halo2_verifier_amd/csrc/curve.hip.h), including the in-place fast
// forms that REPORT the degenerate cases instead of handling them.  Lines on stdin:
//   mchain k x y | achain k x y | dchain k x y (k hex) | add x1 y1 x2 y2 | madd x1 y1 x2 y2 | dbl x1 y1 | fast_madd x1 y1 x2 y2 | fast_add x1 y1 x2 y2 | scaled k x1 y1 x2 y2
// coordinates as 64 hex digits ("0"*64, "0"*64 = the identity).  `scaled k`: P1 is first mapped to Jacobian coordinates with
// Z = k (so that equal points meet with different representations), then added to P2.
// Output: "x y" canonical hex (identity = zeros), fast_* prefix the line with the returned flag.
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>
#include "../../halo2_verifier_amd/csrc/curve.hip.h"
using namespace h2v;
namespace h2v { void set_last_error(const std::string&) {} }

static bool parse(const char* hex, Fq& out) {
    if (strlen(hex) != 64) return false;
    uint32_t raw[8];
    for (int w = 0; w < 8; ++w) { unsigned v = 0; if (sscanf(hex + 64 - 8 * (w + 1), "%8x", &v) != 1) return false; raw[w] = v; }
    out = Fq::from_raw(raw);
    return true;
}
static void hex(const Fq& x) { uint32_t raw[8]; x.to_raw(raw); for (int w = 7; w >= 0; --w) printf("%08x", raw[w]); }
static void print(const G1J& p) {
    G1A a = g1_to_affine(p);
    if (a.is_identity()) { printf("%064d %064d\n", 0, 0); return; }
    hex(a.x); printf(" "); hex(a.y); printf("\n");
}
static G1A aff(const Fq& x, const Fq& y) { if (x.is_zero() && y.is_zero()) return G1A::identity(); G1A a; a.x = x; a.y = y; return a; }
static G1J rescale(const G1A& a, const Fq& k) {   // (x, y) -> (x k^2, y k^3, k)
    if (a.is_identity()) return G1J::identity();
    G1J j; Fq k2 = k.sqr(); j.X = a.x * k2; j.Y = a.y * k2 * k; j.Z = k; return j;
}
// the representation contract of every coordinate that leaves a group-law routine: limbs normalised, value below 2p
static bool below_2p(const Fq& a) {
    for (int i = 0; i < 8; ++i) if (a.v[i] > H2V_LIMB_MASK) return false;
    for (int i = 8; i >= 0; --i) { const uint32_t t = Fq::KP29(2, i); if (a.v[i] < t) return true; if (a.v[i] > t) return false; }
    return false;
}
static bool contract(const G1J& p) { return below_2p(p.X) && below_2p(p.Y) && below_2p(p.Z); }
// `mchain k x y` / `achain k x y`: sum of (i + 2) P for i < k through the in-place fast forms (mixed / full addition; the full form
// meets operands rescaled by changing Z), every intermediate checked against the contract above and against the complete routine.
static int chain(bool mixed, uint32_t k, const G1A& P) {
    G1J acc = G1J::identity(), ref = G1J::identity(), run = g1_dbl(G1J::from_affine(P));
    for (uint32_t i = 0; i < k; ++i) {
        const G1A t = g1_to_affine(run);
        bool ok;
        if (mixed) ok = g1_madd_fast(acc, t);
        else { G1J q = rescale(t, Fq::from_u32(3 + i) * run.Z); ok = g1_add_fast(acc, q); if (!contract(q)) return 4; }
        if (!ok) { acc = g1_add_affine(acc, t); }   // (i + 2) P met the running sum: the caller's slow path
        ref = g1_add_affine(ref, t);
        if (!contract(acc)) { fprintf(stderr, "coordinate out of [0, 2p) after step %u\n", i); return 5; }
        const G1A a = g1_to_affine(acc), b = g1_to_affine(ref);
        if (!(a.x == b.x) || !(a.y == b.y)) { fprintf(stderr, "chain diverged at step %u\n", i); return 6; }
        run = g1_add_affine(run, P);
    }
    print(acc);
    return 0;
}
int main() {
    char line[1024], op[16], t[5][80];
    while (fgets(line, sizeof line, stdin)) {
        int n = sscanf(line, "%15s %79s %79s %79s %79s %79s", op, t[0], t[1], t[2], t[3], t[4]);
        if (n < 3) continue;
        std::string o = op;
        Fq v[5];
        if (o == "dchain") {      // k doublings in a row, each fed by the last: contract after every one
            unsigned k = 0; if (sscanf(t[0], "%x", &k) != 1) return 2;
            Fq x, y; if (!parse(t[1], x) || !parse(t[2], y)) return 2;
            G1J acc = rescale(aff(x, y), Fq::from_u32(9));
            for (unsigned i = 0; i < k; ++i) { acc = g1_dbl(acc); if (!contract(acc)) { fprintf(stderr, "coordinate out of [0, 2p) after doubling %u\n", i); return 5; } }
            print(acc);
            continue;
        }
        if (o == "mchain" || o == "achain") {
            unsigned k = 0; if (sscanf(t[0], "%x", &k) != 1) return 2;
            Fq x, y; if (!parse(t[1], x) || !parse(t[2], y)) return 2;
            if (int rc = chain(o == "mchain", k, aff(x, y))) return rc;
            continue;
        }
        for (int i = 0; i < n - 1; ++i) if (!parse(t[i], v[i])) return 2;
        if (o == "dbl") { print(g1_dbl(G1J::from_affine(aff(v[0], v[1])))); continue; }
        if (o == "scaled") { print(g1_add(rescale(aff(v[1], v[2]), v[0]), rescale(aff(v[3], v[4]), v[0] + Fq::one()))); continue; }
        G1A p = aff(v[0], v[1]), q = aff(v[2], v[3]);
        if (o == "add") print(g1_add(G1J::from_affine(p), G1J::from_affine(q)));
        else if (o == "madd") print(g1_add_affine(G1J::from_affine(p), q));
        else if (o == "fast_madd") { G1J acc = rescale(p, Fq::from_u32(7)); bool ok = g1_madd_fast(acc, q); printf("%d ", ok ? 1 : 0); print(acc); }
        else if (o == "fast_add") { G1J acc = rescale(p, Fq::from_u32(5)); bool ok = g1_add_fast(acc, rescale(q, Fq::from_u32(11))); printf("%d ", ok ? 1 : 0); print(acc); }
        else return 3;
    }
    return 0;
}
