// Host-side checker of the plan compiler's multi-stream Fr programs (no GPU needed): compiles a plan from VK + params bytes and
// verifies, for K = 2, 3, 4 instruction streams,
//   (1) race freedom — inside one barrier epoch no stream writes a slot that another stream reads or writes (the streams of a proof
//       run in different waves: only a barrier orders them), and the streams hold the same number of barriers;
//   (2) equivalence — evaluated symbolically (hash-consed expression ids), every STORE of the multi-stream program writes the same
//       expression to the same place as the single-stream program, and nothing is stored twice or left out;
//   (3) status order — every store that reads the proof's status word (STORE_MSM / STORE_SHARED / STORE_LEFT zero a failed proof's
//       scalars) runs after every OP_INV (which sets it when the reference would panic): later in the same stream, or behind a barrier.
// Usage: plan_host <vk file> <params file> <multiopen> <transcript> <circuit_instances> <guard 0|1> <col_len>...
// Build (tests/test_plan_streams.py): hipcc -O1 -std=c++17 --offload-arch=gfx950 plan_host.hip ../../halo2_verifier_amd/csrc/vkplan.hip ../../halo2_verifier_amd/csrc/params.hip
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <map>
#include <set>
#include <string>
#include <tuple>
#include <vector>
#include "../../include/h2v.h"
#include "../../halo2_verifier_amd/csrc/pairing_api.h"
#include "../../halo2_verifier_amd/csrc/vkplan.h"

namespace h2v {
void set_last_error(const std::string&) {}
// params.hip also holds the pairing tables' upload, which refers to the (device-side) operation tables: not used here
std::vector<uint32_t> pairing_program(bool) { return {}; }
std::vector<uint32_t> pairing_program2() { return {}; }
}
using namespace h2v;

static std::vector<uint8_t> slurp(const char* path) {
    std::ifstream f(path, std::ios::binary);
    return std::vector<uint8_t>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}

struct Access { std::vector<uint32_t> reads; int write; };   // slots an instruction reads / the slot it writes (-1: none)
static Access access_of(const VmInstr& in) {
    Access a{{}, -1};
    auto slot_operand = [&](uint32_t x) { if (!(x & VM_CONST_OPERAND)) a.reads.push_back(x); };
    switch (in.op) {
        case OP_MUL: case OP_ADD: case OP_SUB: slot_operand(in.a); slot_operand(in.b); a.write = (int)in.d; break;
        case OP_NEG: case OP_INV: case OP_POW: case OP_SQRN: a.reads.push_back(in.a); a.write = (int)in.d; break;
        case OP_CONST: case OP_LOAD_SCALAR: case OP_LOAD_INST: case OP_LOAD_CHAL: case OP_LOAD_MULT: case OP_LOAD_INSTEVAL: a.write = (int)in.d; break;
        case OP_STORE_MSM: case OP_STORE_SHARED: case OP_STORE_LEFT: case OP_STORE_GUARD: a.reads.push_back(in.a); break;
        default: break;
    }
    return a;
}

// symbolic evaluation: expression ids, hash-consed
struct Sym {
    std::map<std::tuple<uint32_t, uint64_t, uint64_t, uint32_t>, uint64_t> ids;
    uint64_t id(uint32_t op, uint64_t a, uint64_t b, uint32_t imm) {
        auto k = std::make_tuple(op, a, b, imm);
        auto it = ids.find(k);
        if (it != ids.end()) return it->second;
        const uint64_t v = ids.size() + 1;
        ids[k] = v;
        return v;
    }
};
typedef std::map<std::pair<uint32_t, uint32_t>, uint64_t> Outputs;   // (store op, index) -> expression

static bool step(Sym& sym, const VmInstr& in, std::map<uint32_t, uint64_t>& slots, Outputs& out, std::string& err) {
    auto rd = [&](uint32_t s, uint64_t& v) { auto it = slots.find(s); if (it == slots.end()) { err = "read of a slot nobody wrote: " + std::to_string(s); return false; } v = it->second; return true; };
    auto operand = [&](uint32_t x, uint64_t& v) { if (x & VM_CONST_OPERAND) { v = sym.id(OP_CONST, 0, 0, x & ~VM_CONST_OPERAND); return true; } return rd(x, v); };
    uint64_t a = 0, b = 0;
    switch (in.op) {
        case OP_MUL: case OP_ADD: case OP_SUB: if (!operand(in.a, a) || !operand(in.b, b)) return false; slots[in.d] = sym.id(in.op, a, b, 0); break;
        case OP_NEG: case OP_INV: if (!rd(in.a, a)) return false; slots[in.d] = sym.id(in.op, a, 0, 0); break;
        case OP_POW: case OP_SQRN: if (!rd(in.a, a)) return false; slots[in.d] = sym.id(in.op, a, 0, in.b); break;
        case OP_CONST: slots[in.d] = sym.id(OP_CONST, 0, 0, in.a); break;
        case OP_LOAD_SCALAR: case OP_LOAD_INST: case OP_LOAD_CHAL: case OP_LOAD_MULT: case OP_LOAD_INSTEVAL: slots[in.d] = sym.id(in.op, 0, 0, in.a); break;
        case OP_STORE_MSM: case OP_STORE_SHARED: case OP_STORE_LEFT: case OP_STORE_GUARD: {
            if (!rd(in.a, a)) return false;
            const auto key = std::make_pair(in.op, in.b);
            if (out.count(key)) { err = "stored twice: op " + std::to_string(in.op) + " index " + std::to_string(in.b); return false; }
            out[key] = a;
            break;
        }
        default: break;
    }
    return true;
}

int main(int argc, char** argv) {
    if (argc < 7) { fprintf(stderr, "usage: plan_host vk params multiopen transcript circuit_instances guard col_len...\n"); return 2; }
    const std::vector<uint8_t> vkb = slurp(argv[1]), pb = slurp(argv[2]);
    PlanOptions opts; opts.multiopen = atoi(argv[3]); opts.transcript = atoi(argv[4]); opts.circuit_instances = atoi(argv[5]); opts.guard_terms = atoi(argv[6]) != 0;
    std::vector<size_t> col_lens;
    for (int i = 7; i < argc; ++i) col_lens.push_back((size_t)atoll(argv[i]));
    std::string err;
    VkHost vk; ParamsHost params;
    if (!vk_from_bytes(vkb.data(), vkb.size(), H2V_SERDE_RAW_BYTES, vk, err)) { fprintf(stderr, "vk: %s\n", err.c_str()); return 1; }
    if (!params_from_bytes(pb.data(), pb.size(), H2V_SERDE_RAW_BYTES, params, err)) { fprintf(stderr, "params: %s\n", err.c_str()); return 1; }
    Plan plan;
    const int rc = compile_plan(vk, params, col_lens, opts, plan, err);
    if (rc) { fprintf(stderr, "compile_plan: %d %s\n", rc, err.c_str()); return 1; }
    // the single-stream program is the reference
    Sym sym;
    Outputs want;
    {
        std::map<uint32_t, uint64_t> slots;
        for (const VmInstr& in : plan.code) if (!step(sym, in, slots, want, err)) { fprintf(stderr, "single stream: %s\n", err.c_str()); return 1; }
    }
    // self-test of this checker (tests/test_plan_streams.py): with the barriers taken out the streams must be found to race
    if (getenv("PLAN_HOST_DROP_BARRIERS"))
        for (int K = 2; K <= FRVM_MAX_STREAMS; ++K) for (int q = 0; q < K; ++q) {
            std::vector<VmInstr> kept;
            for (const VmInstr& in : plan.code_k[K - 2][q]) if (in.op != OP_BARRIER) kept.push_back(in);
            plan.code_k[K - 2][q] = kept;
        }
    for (int K = 2; K <= FRVM_MAX_STREAMS; ++K) {
        const std::vector<VmInstr>* code = plan.code_k[K - 2];
        // epochs: instructions between consecutive barriers
        std::vector<std::vector<std::vector<VmInstr>>> ep(K);   // [stream][epoch][instruction]
        size_t n_epochs = 0;
        for (int q = 0; q < K; ++q) {
            ep[q].push_back({});
            for (const VmInstr& in : code[q]) { if (in.op == OP_BARRIER) ep[q].push_back({}); else ep[q].back().push_back(in); }
            if (q == 0) n_epochs = ep[q].size();
            else if (ep[q].size() != n_epochs) { fprintf(stderr, "K=%d: stream %d has %zu barriers, stream 0 has %zu\n", K, q, ep[q].size() - 1, n_epochs - 1); return 1; }
        }
        std::map<uint32_t, uint64_t> slots;
        Outputs got;
        size_t n_instr = 0, max_slot = 0;
        for (size_t e = 0; e < n_epochs; ++e) {
            // (1) inside the epoch: no slot written by one stream and touched by another
            std::vector<std::set<uint32_t>> R(K), W(K);
            for (int q = 0; q < K; ++q) for (const VmInstr& in : ep[q][e]) {
                const Access a = access_of(in);
                for (uint32_t s : a.reads) { R[q].insert(s); if (s > max_slot) max_slot = s; }
                if (a.write >= 0) { W[q].insert((uint32_t)a.write); if ((size_t)a.write > max_slot) max_slot = (size_t)a.write; }
                ++n_instr;
            }
            for (int q = 0; q < K; ++q) for (int r = 0; r < K; ++r) if (q != r) for (uint32_t s : W[q]) if (R[r].count(s) || W[r].count(s)) {
                fprintf(stderr, "K=%d epoch %zu: slot %u is written by stream %d and %s by stream %d without a barrier between them\n", K, e, s, q, W[r].count(s) ? "written" : "read", r);
                return 1;
            }
            // (2) race free, so any interleaving of the epoch's streams gives the same values: one stream after the other
            for (int q = 0; q < K; ++q) for (const VmInstr& in : ep[q][e]) if (!step(sym, in, slots, got, err)) { fprintf(stderr, "K=%d stream %d epoch %zu: %s\n", K, q, e, err.c_str()); return 1; }
        }
        // (3) every status-reading store behind every inversion
        {
            std::vector<std::pair<int, size_t>> inv_at;   // (stream, epoch) of every OP_INV, with its index inside the epoch
            std::vector<size_t> inv_idx;
            for (int q = 0; q < K; ++q) for (size_t e = 0; e < n_epochs; ++e) for (size_t i = 0; i < ep[q][e].size(); ++i) if (ep[q][e][i].op == OP_INV) { inv_at.push_back({q, e}); inv_idx.push_back(i); }
            for (int q = 0; q < K; ++q) for (size_t e = 0; e < n_epochs; ++e) for (size_t i = 0; i < ep[q][e].size(); ++i) {
                const uint32_t op = ep[q][e][i].op;
                if (op != OP_STORE_MSM && op != OP_STORE_SHARED && op != OP_STORE_LEFT) continue;
                for (size_t k = 0; k < inv_at.size(); ++k) {
                    const bool ordered = inv_at[k].second < e || (inv_at[k].second == e && inv_at[k].first == q && inv_idx[k] < i);
                    if (!ordered) { fprintf(stderr, "K=%d: a status-reading store (stream %d, epoch %zu) is not ordered behind the inversion of stream %d, epoch %zu\n", K, q, e, inv_at[k].first, inv_at[k].second); return 1; }
                }
            }
        }
        if (got != want) {
            fprintf(stderr, "K=%d: the stores differ from the single-stream program's (%zu vs %zu)\n", K, got.size(), want.size());
            return 1;
        }
        if (max_slot + 1 > plan.n_slots_k[K - 2]) { fprintf(stderr, "K=%d: slot %zu beyond n_slots %u\n", K, max_slot, plan.n_slots_k[K - 2]); return 1; }
        printf("K=%d ok: %zu instructions, %zu barriers, %u slots, %zu stores\n", K, n_instr, n_epochs - 1, plan.n_slots_k[K - 2], got.size());
    }
    printf("single ok: %zu instructions, %u slots\n", plan.code.size(), plan.n_slots);
    printf("dag: work %.1f products, critical path %.1f; estimated makespan with 2 / 3 / 4 streams: %.1f / %.1f / %.1f\n", plan.dag_work, plan.dag_critical_path, plan.makespan_k[0], plan.makespan_k[1], plan.makespan_k[2]);
    return 0;
}
