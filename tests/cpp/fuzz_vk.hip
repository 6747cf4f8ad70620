// Host-side mutation fuzz of the parsers of UNTRUSTED bytes under AddressSanitizer (CPU only, no GPU): VerifyingKey / ParamsKZG readers, the
// writers behind h2v_vk_convert / h2v_params_convert, and the plan compiler on every key that still parses.  Every mutated input lives in an
// exact-size heap block, so a read past its end is caught.  Usage: fuzz_vk <vk file> <params file> <iterations>
// Build (tests/test_vk_fuzz_asan.py): hipcc -O1 -g -std=c++17 --offload-arch=gfx950 -fsanitize=address -fno-gpu-sanitize fuzz_vk.hip ../../halo2_verifier_amd/csrc/{vkplan,params,serde}.hip
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <random>
#include <string>
#include <vector>
#include "../../include/h2v.h"
#include "../../halo2_verifier_amd/csrc/pairing_api.h"
#include "../../halo2_verifier_amd/csrc/vkplan.h"
namespace h2v {
static thread_local std::string g_err;
void set_last_error(const std::string& s) { g_err = s; }
std::vector<uint32_t> pairing_program(bool) { return {}; }
std::vector<uint32_t> pairing_program2() { return {}; }
}
using namespace h2v;
static std::vector<uint8_t> slurp(const char* p) { std::ifstream f(p, std::ios::binary); return std::vector<uint8_t>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>()); }
int main(int argc, char** argv) {
    std::vector<uint8_t> vk = slurp(argv[1]), params = slurp(argv[2]);
    const int iters = atoi(argv[3]);
    std::mt19937_64 rng(12345);
    size_t parsed = 0, planned = 0;
    ParamsHost ph; std::string err;
    if (!params_from_bytes(params.data(), params.size(), 1, ph, err)) { printf("params: %s\n", err.c_str()); return 1; }
    for (int it = 0; it < iters; ++it) {
        std::vector<uint8_t> m = vk;
        const int kind = rng() % 4;
        if (kind == 0) m.resize(rng() % (m.size() + 1));                                   // truncation
        else if (kind == 1) { for (int k = 0; k < 1 + (int)(rng() % 4); ++k) m[rng() % m.size()] ^= (uint8_t)(1u << (rng() % 8)); }   // bit flips
        else if (kind == 2) { size_t o = rng() % (m.size() - 4); m[o] = 0xff; m[o + 1] = 0xff; m[o + 2] = (uint8_t)rng(); m[o + 3] = (uint8_t)rng(); }   // a huge count somewhere
        else { size_t o = rng() % std::min<size_t>(m.size() - 4, 200); m[o + 3] = (uint8_t)(rng() % 64); }   // a small change in the header counts
        // exact-size heap copy so that any read past the end is caught
        uint8_t* heap = (uint8_t*)malloc(m.size() ? m.size() : 1);
        memcpy(heap, m.data(), m.size());
        for (int fmt = 0; fmt < 3; ++fmt) {
            size_t n = 0;
            int rc = h2v_vk_convert(heap, m.size(), fmt == 0 ? 1 : fmt, fmt, (int)(rng() & 1), nullptr, &n);
            if (rc == 0) { std::vector<uint8_t> out(n); size_t cap = n; h2v_vk_convert(heap, m.size(), fmt == 0 ? 1 : fmt, fmt, 0, out.data(), &cap); }
        }
        VkHost v;
        if (vk_from_bytes(heap, m.size(), 1, v, err)) {
            ++parsed;
            if (v.k == ph.k) {
                Plan plan;
                std::vector<size_t> lens(v.num_instance_columns, 3);
                PlanOptions po;
                po.multiopen = (int)(rng() & 1);
                if (v.advice_queries.size() < 4000 && v.gates.size() < 4000 && compile_plan(v, ph, lens, po, plan, err) == 0) ++planned;
            }
        }
        free(heap);
        // params
        std::vector<uint8_t> pm = params;
        if (rng() & 1) pm.resize(rng() % (pm.size() + 1)); else pm[rng() % pm.size()] ^= (uint8_t)(1u << (rng() % 8));
        uint8_t* ph2 = (uint8_t*)malloc(pm.size() ? pm.size() : 1); memcpy(ph2, pm.data(), pm.size());
        for (int fmt = 0; fmt < 3; ++fmt) { size_t n = 0; h2v_params_convert(ph2, pm.size(), 1, fmt, nullptr, &n); }
        free(ph2);
    }
    printf("fuzzed %d VKs: %zu still parsed, %zu compiled to a plan\n", iters, parsed, planned);
    return 0;
}
