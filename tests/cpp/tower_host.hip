// Host-side exerciser of the extension tower of csrc/pairing.hip.h (Fq2 = Fq[u]/(u^2+1), Fq6 = Fq2[v]/(v^3 - xi),
// Fq12 = Fq6[w]/(w^2 - v), xi = 9 + u), which the library runs on the CPU for the G2 line precomputation and on one lane
// for the Fq6 inversion of the final exponentiation.  stdin: "mul|inv|frob <12 x 64-hex> [<12 x 64-hex>]" with the twelve
// Fq coefficients in the order c0.c0.c0 c0.c0.c1 c0.c1.c0 ... (Fq6 c0 then c1; inside Fq6: c0, c1, c2; inside Fq2: c0, c1).
// stdout: the twelve coefficients of the result in the same order.  tests/test_field_host.py maps them to the flat
// representation Fq[w]/(w^12 - 18 w^6 + 82) of the Python restatement and compares.
#include <cstdio>
#include <cstring>
#include <string>
#include "../../halo2_verifier_amd/csrc/pairing.hip.h"
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
static Fq* coeff(Fq12& f, int i) {
    Fq6& h = i < 6 ? f.c0 : f.c1; i %= 6;
    Fq2& q = i < 2 ? h.c0 : (i < 4 ? h.c1 : h.c2);
    return (i & 1) ? &q.c1 : &q.c0;
}
static void print(Fq12 f) { for (int i = 0; i < 12; ++i) { if (i) printf(" "); hex(*coeff(f, i)); } printf("\n"); }
int main() {
    static char line[4096];
    while (fgets(line, sizeof line, stdin)) {
        char* tok = strtok(line, " \n");
        if (!tok) continue;
        std::string op = tok;
        Fq12 a, b;
        for (int i = 0; i < 12; ++i) { tok = strtok(nullptr, " \n"); if (!tok || !parse(tok, *coeff(a, i))) return 2; }
        if (op == "mul") {
            for (int i = 0; i < 12; ++i) { tok = strtok(nullptr, " \n"); if (!tok || !parse(tok, *coeff(b, i))) return 2; }
            print(a * b);
        } else if (op == "inv") print(a.inv());
        else if (op == "conj") print(a.conj());
        else return 3;
    }
    return 0;
}
