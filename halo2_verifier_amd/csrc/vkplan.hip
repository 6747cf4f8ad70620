// Host side: VerifyingKey ingestion and compilation of the per-VK verification plan
// (see vkplan.h).  Pure host code; compiled by hipcc because it shares bn254.hip.h with the kernels.
#include "../../include/h2v.h"
#include "ctx.h"
#include "vkplan.h"
#include <algorithm>
#include <array>
#include <set>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

namespace h2v {

// =============================================================================== VK parsing
namespace {
struct Reader {
    const uint8_t* d; size_t n, pos = 0; bool ok = true;
    Reader(const uint8_t* p, size_t len) : d(p), n(len) {}
    const uint8_t* take(size_t k) {
        static const uint8_t zeros[128] = {0};
        if (pos + k > n) { ok = false; return zeros; }
        const uint8_t* r = d + pos; pos += k; return r;
    }
    uint8_t u8() { return *take(1); }
    uint16_t u16() { const uint8_t* b = take(2); return (uint16_t)((b[0] << 8) | b[1]); }            // helpers.rs:121-125 (big endian)
    uint32_t u32() { const uint8_t* b = take(4); return ((uint32_t)b[0] << 24) | ((uint32_t)b[1] << 16) | ((uint32_t)b[2] << 8) | b[3]; }
    int32_t i32() { return (int32_t)u32(); }
};

template <class F> bool field_from_mont_bytes(const uint8_t* b, F& x) {
    uint32_t m[8];  // halo2curves' in-memory form: residue * 2^256 mod p
    for (int i = 0; i < 8; ++i) m[i] = (uint32_t)b[4 * i] | ((uint32_t)b[4 * i + 1] << 8) | ((uint32_t)b[4 * i + 2] << 16) | ((uint32_t)b[4 * i + 3] << 24);
    x = F::from_mont256(m);
    return !F::geq_p(m);
}
bool read_fr(Reader& r, int fmt, Fr& x, std::string& err) {
    const uint8_t* b = r.take(32);
    if (fmt == H2V_SERDE_PROCESSED) { if (!Fr::from_bytes(b, x)) { err = "Invalid prime field point encoding"; return false; } return true; }
    bool ok = field_from_mont_bytes(b, x);
    if (!ok && fmt == H2V_SERDE_RAW_BYTES) { err = "Invalid prime field point encoding"; return false; }
    return true;
}
bool read_g1(Reader& r, int fmt, G1A& p, std::string& err) {
    if (fmt == H2V_SERDE_PROCESSED) {
        if (!g1_decompress(r.take(32), p)) { err = "Invalid point encoding in proof"; return false; }
        return true;
    }
    const uint8_t* b = r.take(64);
    const bool okx = field_from_mont_bytes(b, p.x), oky = field_from_mont_bytes(b + 32, p.y);
    bool ok = okx && oky;
    if (fmt == H2V_SERDE_RAW_BYTES && (!ok || !p.on_curve())) { err = "invalid uncompressed point"; return false; }
    return true;
}
bool read_expr(Reader& r, ExprH& e) {
    e.num_vars = r.u32();
    uint32_t nt = r.u32();
    if (!r.ok || nt > (1u << 24)) return false;
    e.terms.resize(nt);
    for (auto& t : e.terms) {
        t.coeff_idx = r.u16();
        uint32_t nf = r.u32();
        if (!r.ok || nf > (1u << 20)) return false;
        t.factors.resize(nf);
        for (auto& f : t.factors) { f.first = r.u32(); f.second = r.u32(); }
    }
    return r.ok;
}
}  // namespace

size_t VkHost::blinding_factors() const {
    size_t f = 1;
    if (!num_advice_queries.empty()) { f = 0; for (uint32_t q : num_advice_queries) if (q > f) f = q; }
    if (f < 3) f = 3;
    return f + 2;
}

bool vk_from_bytes(const uint8_t* data, size_t len, int fmt, VkHost& vk, std::string& err) {
    if (fmt < 0 || fmt > 2) { err = "unknown serde format"; return false; }
    Reader r(data, len);
    const uint32_t LIM = 1u << 20;
    vk.k = r.u32();
    uint32_t nfix = r.u32();
    if (!r.ok || vk.k > 28 || nfix > LIM) { err = "failed to fill whole buffer"; return false; }
    vk.fixed_commitments.resize(nfix);
    for (auto& c : vk.fixed_commitments) if (!read_g1(r, fmt, c, err)) return false;
    vk.cs_degree = r.u32();
    vk.num_fixed_columns = r.u32(); vk.num_advice_columns = r.u32(); vk.num_instance_columns = r.u32();
    vk.num_selectors = r.u32(); vk.num_challenges = r.u32();
    uint32_t ng = r.u32(), nl = r.u32(), ns = r.u32(), nc = r.u32();
    if (!r.ok || vk.num_fixed_columns > LIM || vk.num_advice_columns > LIM || vk.num_instance_columns > LIM || vk.num_challenges > LIM || ng > LIM || nl > LIM || ns > LIM || nc > 65536 * 16) { err = "failed to fill whole buffer"; return false; }
    for (uint32_t i = 0; i < vk.num_advice_columns; ++i) vk.advice_column_phase.push_back(r.u8());
    for (uint32_t i = 0; i < vk.num_challenges; ++i) vk.challenge_phase.push_back(r.u8());
    size_t total = 0;
    for (uint32_t i = 0; i < vk.num_advice_columns; ++i) { vk.num_advice_queries.push_back(r.u32()); total += vk.num_advice_queries.back(); }
    if (!r.ok || total > LIM) { err = "failed to fill whole buffer"; return false; }
    for (size_t i = 0; i < total; ++i) { QueryH q; q.column.index = r.u32(); q.column.type = r.u8(); q.rotation = r.i32(); vk.advice_queries.push_back(q); }
    // the reader takes the instance / fixed query counts from the column counts (plonk/vk.rs:310-322)
    for (uint32_t i = 0; i < vk.num_instance_columns; ++i) { QueryH q; q.column.index = r.u32(); q.column.type = COL_INSTANCE; q.rotation = r.i32(); vk.instance_queries.push_back(q); }
    for (uint32_t i = 0; i < vk.num_fixed_columns; ++i) { QueryH q; q.column.index = r.u32(); q.column.type = COL_FIXED; q.rotation = r.i32(); vk.fixed_queries.push_back(q); }
    uint32_t np = r.u32();
    if (!r.ok || np > LIM) { err = "failed to fill whole buffer"; return false; }
    for (uint32_t i = 0; i < np; ++i) {
        ColumnH c; c.index = r.u32(); c.type = r.u8();
        if (!(c.type <= 2 || c.type >= 254)) { err = "Invalid phase for advice column"; return false; }  // plonk/circuit.rs:52-61
        vk.permutation_columns.push_back(c);
    }
    vk.gates.resize(ng);
    for (auto& g : vk.gates) if (!read_expr(r, g)) { err = "failed to fill whole buffer"; return false; }
    vk.lookups.resize(nl);
    for (auto& a : vk.lookups) {  // reader order: input, table interleaved (plonk/lookup.rs:51-68)
        uint32_t m = r.u32();
        if (!r.ok || m > LIM) { err = "failed to fill whole buffer"; return false; }
        a.input.resize(m); a.table.resize(m);
        for (uint32_t j = 0; j < m; ++j) if (!read_expr(r, a.input[j]) || !read_expr(r, a.table[j])) { err = "failed to fill whole buffer"; return false; }
    }
    vk.shuffles.resize(ns);
    for (auto& a : vk.shuffles) {  // plonk/shuffle.rs:86-102
        uint32_t m = r.u32();
        if (!r.ok || m > LIM) { err = "failed to fill whole buffer"; return false; }
        a.input.resize(m); a.shuffle.resize(m);
        for (uint32_t j = 0; j < m; ++j) if (!read_expr(r, a.input[j]) || !read_expr(r, a.shuffle[j])) { err = "failed to fill whole buffer"; return false; }
    }
    vk.coeff_vals.resize(nc);
    for (auto& c : vk.coeff_vals) if (!read_fr(r, fmt, c, err)) return false;
    vk.permutation_commitments.resize(vk.permutation_columns.size());
    for (auto& c : vk.permutation_commitments) if (!read_g1(r, fmt, c, err)) return false;
    size_t sel_bytes = (((size_t)1 << vk.k) + 7) / 8;
    if ((size_t)vk.num_selectors * sel_bytes > len) { err = "failed to fill whole buffer"; return false; }
    for (uint32_t i = 0; i < vk.num_selectors; ++i) { const uint8_t* sb = r.take(sel_bytes); if (r.ok) vk.selector_bytes.insert(vk.selector_bytes.end(), sb, sb + sel_bytes); }  // selector bitmaps: unused by verification
    if (!read_fr(r, fmt, vk.transcript_repr, err)) return false;
    if (!r.ok) { err = "failed to fill whole buffer"; return false; }
    if (vk.cs_degree < 3) { err = "cs_degree below the permutation argument's minimum of 3"; return false; }
    // VerifyingKey::read builds EvaluationDomain::new(cs_degree, k), which asserts that the extended domain 2^k (cs_degree - 1) fits the
    // 2-adicity of Fr (poly/domain.rs:44-50: extended_k <= S = 28).  cs_degree is the one count of the format that consumes no bytes,
    // so nothing else bounds it (found by fuzzing the parser under AddressSanitizer: a flipped bit made the plan compiler lay out
    // 2^31 quotient commitments).
    if (((uint64_t)(vk.cs_degree - 1) << vk.k) > (1ull << 28)) { err = "cs_degree: the extended domain exceeds the 2-adicity of Fr (EvaluationDomain::new asserts extended_k <= 28)"; return false; }
    return true;
}

// =============================================================================== program builder
namespace {
typedef uint32_t Val;  // SSA value id
struct Node { uint32_t op; Val a, b; uint32_t imm; bool has_result; };

struct Builder {
    std::vector<Node> nodes;
    std::vector<Fr> consts;
    std::map<std::array<uint32_t, 9>, Val> const_node;
    std::vector<int64_t> const_of;  // node id -> const index or -1
    std::map<uint32_t, Val> scalar_node, inst_node, chal_node;
    Val mult_node = (Val)-1;

    Val push(uint32_t op, Val a, Val b, uint32_t imm, bool res, int64_t cidx = -1) {
        nodes.push_back({op, a, b, imm, res}); const_of.push_back(cidx);
        return (Val)nodes.size() - 1;
    }
    bool is_const(Val v) const { return const_of[v] >= 0; }
    const Fr& cval(Val v) const { return consts[(size_t)const_of[v]]; }
    Val cst(const Fr& f) {
        std::array<uint32_t, 9> key; f.canonical(key.data());
        auto it = const_node.find(key);
        if (it != const_node.end()) return it->second;
        consts.push_back(f);
        Val v = push(OP_CONST, 0, 0, (uint32_t)consts.size() - 1, true, (int64_t)consts.size() - 1);
        const_node[key] = v;
        return v;
    }
    Val zero() { return cst(Fr::zero()); }
    Val one() { return cst(Fr::one()); }
    Val mul(Val a, Val b) {
        if (is_const(a) && is_const(b)) return cst(cval(a) * cval(b));
        if (is_const(a) && cval(a) == Fr::one()) return b;
        if (is_const(b) && cval(b) == Fr::one()) return a;
        if ((is_const(a) && cval(a).is_zero()) || (is_const(b) && cval(b).is_zero())) return zero();
        return push(OP_MUL, a, b, 0, true);
    }
    Val add(Val a, Val b) {
        if (is_const(a) && is_const(b)) return cst(cval(a) + cval(b));
        if (is_const(a) && cval(a).is_zero()) return b;
        if (is_const(b) && cval(b).is_zero()) return a;
        return push(OP_ADD, a, b, 0, true);
    }
    Val sub(Val a, Val b) {
        if (is_const(a) && is_const(b)) return cst(cval(a) - cval(b));
        if (is_const(b) && cval(b).is_zero()) return a;
        return push(OP_SUB, a, b, 0, true);
    }
    Val neg(Val a) { if (is_const(a)) return cst(cval(a).neg()); return push(OP_NEG, a, 0, 0, true); }
    Val sqr(Val a) { return mul(a, a); }
    Val inv(Val a) { return push(OP_INV, a, 0, 0, true); }
    Val pow(Val a, uint32_t e) {
        if (e == 0) return one();
        if (e == 1) return a;
        if (is_const(a)) return cst(cval(a).pow_u32(e));
        if (e == 2) return mul(a, a);
        return push(OP_POW, a, 0, e, true);
    }
    Val sqrn(Val a, uint32_t k) { if (k == 0) return a; return push(OP_SQRN, a, 0, k, true); }
    Val load_scalar(uint32_t i) { auto it = scalar_node.find(i); if (it != scalar_node.end()) return it->second; return scalar_node[i] = push(OP_LOAD_SCALAR, 0, 0, i, true); }
    Val load_inst(uint32_t i) { auto it = inst_node.find(i); if (it != inst_node.end()) return it->second; return inst_node[i] = push(OP_LOAD_INST, 0, 0, i, true); }
    Val load_insteval(uint32_t i) { return push(OP_LOAD_INSTEVAL, 0, 0, i, true); }
    Val load_chal(uint32_t i) { auto it = chal_node.find(i); if (it != chal_node.end()) return it->second; return chal_node[i] = push(OP_LOAD_CHAL, 0, 0, i, true); }
    Val load_mult() { if (mult_node == (Val)-1) mult_node = push(OP_LOAD_MULT, 0, 0, 0, true); return mult_node; }
    void store_msm(Val a, uint32_t slot) { push(OP_STORE_MSM, a, 0, slot, false); }
    void store_shared(Val a, uint32_t j) { push(OP_STORE_SHARED, a, 0, j, false); }
    void store_left(Val a, uint32_t slot) { push(OP_STORE_LEFT, a, 0, slot, false); }
    void store_guard(Val a, uint32_t term) { push(OP_STORE_GUARD, a, 0, term, false); }

    // values[i] <- 1 / values[i] for all i with ONE inversion (Montgomery's trick)
    void batch_invert(std::vector<Val>& vals) {
        if (vals.empty()) return;
        std::vector<Val> prefix(vals.size());
        prefix[0] = vals[0];
        for (size_t i = 1; i < vals.size(); ++i) prefix[i] = mul(prefix[i - 1], vals[i]);
        Val run = inv(prefix.back());
        for (size_t i = vals.size(); i-- > 1;) {
            Val r = mul(run, prefix[i - 1]);
            run = mul(run, vals[i]);
            vals[i] = r;
        }
        vals[0] = run;
    }

    // Emits the final instruction stream: emission order, then liveness-based slot assignment.
    //  * constants are OPERANDS of MUL / ADD / SUB (VM_CONST_OPERAND | index): no instruction, no slot; a constant that feeds
    //    anything else is materialised by an OP_CONST right before that use;
    //  * LOAD_* nodes are emitted at their first use, not where the builder created them (it creates all of a proof's
    //    evaluations up front: twenty values that would otherwise sit in slots for most of the program).
    // Fewer live values = fewer slots = all of them in LDS (k_frvm).
    static bool is_load(uint32_t op) { return op == OP_LOAD_SCALAR || op == OP_LOAD_INST || op == OP_LOAD_CHAL || op == OP_LOAD_MULT || op == OP_LOAD_INSTEVAL; }
    static bool takes_const_operands(uint32_t op) { return op == OP_MUL || op == OP_ADD || op == OP_SUB; }
    void emit(std::vector<VmInstr>& code, uint32_t& n_slots) const {
        const size_t n = nodes.size();
        auto n_operands = [&](const Node& nd) -> int {
            switch (nd.op) {
                case OP_MUL: case OP_ADD: case OP_SUB: return 2;
                case OP_NEG: case OP_INV: case OP_POW: case OP_SQRN: case OP_STORE_MSM: case OP_STORE_SHARED: case OP_STORE_LEFT: case OP_STORE_GUARD: return 1;
                default: return 0;
            }
        };
        std::vector<Val> order;
        std::vector<char> emitted(n, 0);
        for (size_t i = 0; i < n; ++i) {
            const Node& nd = nodes[i];
            if (nd.op == OP_CONST || is_load(nd.op)) continue;   // on demand
            const int k = n_operands(nd);
            for (int j = 0; j < k; ++j) {
                const Val v = j == 0 ? nd.a : nd.b;
                if (emitted[v]) continue;
                const bool c = nodes[v].op == OP_CONST;
                if (is_load(nodes[v].op) || (c && !takes_const_operands(nd.op))) { order.push_back(v); emitted[v] = 1; }
            }
            order.push_back((Val)i); emitted[i] = 1;
        }
        const size_t m = order.size();
        std::vector<size_t> pos(n, (size_t)-1), last_use(n, 0);
        for (size_t q = 0; q < m; ++q) pos[order[q]] = q;
        for (size_t q = 0; q < m; ++q) {
            const Node& nd = nodes[order[q]];
            const int k = n_operands(nd);
            for (int j = 0; j < k; ++j) { const Val v = j == 0 ? nd.a : nd.b; if (emitted[v]) last_use[v] = std::max(last_use[v], q); }
        }
        std::vector<uint32_t> slot(n, 0), free_list;
        std::vector<std::vector<Val>> dying(m);
        for (size_t q = 0; q < m; ++q) if (nodes[order[q]].has_result) dying[std::max(last_use[order[q]], q)].push_back(order[q]);
        uint32_t next = 0;
        auto operand = [&](uint32_t consumer_op, Val v) -> uint32_t {
            if (nodes[v].op == OP_CONST && takes_const_operands(consumer_op)) return VM_CONST_OPERAND | nodes[v].imm;
            return slot[v];
        };
        for (size_t q = 0; q < m; ++q) {
            const Val i = order[q];
            const Node& nd = nodes[i];
            VmInstr in{nd.op, 0, 0, 0};
            switch (nd.op) {
                case OP_MUL: case OP_ADD: case OP_SUB: in.a = operand(nd.op, nd.a); in.b = operand(nd.op, nd.b); break;
                case OP_NEG: case OP_INV: in.a = slot[nd.a]; break;
                case OP_POW: case OP_SQRN: in.a = slot[nd.a]; in.b = nd.imm; break;
                case OP_STORE_MSM: case OP_STORE_SHARED: case OP_STORE_LEFT: case OP_STORE_GUARD: in.a = slot[nd.a]; in.b = nd.imm; break;
                default: in.a = nd.imm; break;  // CONST / LOAD_*
            }
            if (nd.has_result) {
                // the interpreter reads both operands before it writes, so the destination may reuse the slot of an operand
                // that dies here; lowest free slot first: the low numbers are the ones kept in LDS
                uint32_t sl;
                if (!free_list.empty()) { auto it = std::min_element(free_list.begin(), free_list.end()); sl = *it; free_list.erase(it); } else sl = next++;
                slot[i] = sl; in.d = sl;
            }
            code.push_back(in);
            for (Val v : dying[q]) free_list.push_back(slot[v]);
        }
        n_slots = next ? next : 1;
    }

    // ---- The same program as TWO instruction streams per proof (k_frvm2: two waves of a workgroup work on the same 64 proofs and
    // share their slot file).  A proof's program is a latency chain with one long indivisible block in the middle — the batched
    // inversion's single inverse, ~60 products' worth — and most of the expression evaluation does not depend on it.  List
    // scheduling on two processors: the stream that is behind takes the ready node it can start soonest (ties: longest path to the
    // end first); a value crosses between the streams only over an OP_BARRIER, which both streams execute (it aligns their clocks, so
    // the scheduler places one only when the consumer cannot start earlier anyway).  Loads and materialised constants are private:
    // a stream (re)loads what it needs right before the first use.
    // Slots: a value touched by one stream only is recycled in that stream's program order; a shared one becomes free at the first
    // barrier after its last use, for either stream.
    static double op_weight(const Node& nd) {
        switch (nd.op) {
            case OP_MUL: return 1.0;
            case OP_ADD: case OP_SUB: case OP_NEG: return 0.15;
            case OP_INV: return 60.0;
            case OP_POW: { double w = 0; for (uint32_t e = nd.imm; e > 1; e >>= 1) w += 1.5; return w; }
            case OP_SQRN: return (double)nd.imm;
            case OP_CONST: return 0.1;
            default: return 0.5;   // loads (from_raw) and stores (to_raw)
        }
    }
    mutable double makespan_k = 0;   // the scheduler's estimate for emit_streams' last program, in products
    // K instruction streams (K <= FRVM_MAX_STREAMS): see the comment above; a barrier is executed by ALL streams.
    void emit_streams(const int K, std::vector<VmInstr>* code, uint32_t& n_slots) const {
        const size_t n = nodes.size();
        const double BAR = 0.3;   // what the scheduler charges for a barrier, in products (swept in round 2: 0.1 .. 2.0, flat around 0.3)
        auto n_operands = [&](const Node& nd) -> int {
            switch (nd.op) {
                case OP_MUL: case OP_ADD: case OP_SUB: return 2;
                case OP_NEG: case OP_INV: case OP_POW: case OP_SQRN: case OP_STORE_MSM: case OP_STORE_SHARED: case OP_STORE_LEFT: case OP_STORE_GUARD: return 1;
                default: return 0;
            }
        };
        auto is_compute = [&](Val v) { return nodes[v].op != OP_CONST && !is_load(nodes[v].op); };
        auto inline_const = [&](const Node& consumer, Val v) { return nodes[v].op == OP_CONST && takes_const_operands(consumer.op); };
        auto operand = [&](Val v, int j) -> Val { return j ? nodes[v].b : nodes[v].a; };
        auto distinct = [&](Val v, int j) { return !(j == 1 && nodes[v].b == nodes[v].a); };
        // The per-proof status is a value too: OP_INV sets it (a zero where the reference panics) and the stores of the two MSM channels
        // and of the shared scalars read it (a failed proof's scalars are zeroed).  The scheduler orders VALUE dependencies only, so
        // with several streams a store could run before the inversion of another stream — whether a panicking proof was zeroed out of
        // its accumulators depended on wave timing (round-2 review).  Every status-reading store therefore also depends on every
        // OP_INV ("effect dependency": ordering only — no slot is read, no load is emitted for it).
        auto reads_status = [&](Val v) { const uint32_t op = nodes[v].op; return op == OP_STORE_MSM || op == OP_STORE_SHARED || op == OP_STORE_LEFT; };
        std::vector<Val> inv_nodes;
        for (size_t i = 0; i < n; ++i) if (nodes[i].op == OP_INV) inv_nodes.push_back((Val)i);
        // the compute nodes v has to wait for: its operands, and for a status-reading store the inversions
        auto deps = [&](Val v) -> std::vector<Val> {
            std::vector<Val> d;
            for (int j = 0; j < n_operands(nodes[v]); ++j) { const Val u = operand(v, j); if (is_compute(u) && distinct(v, j)) d.push_back(u); }
            if (reads_status(v)) for (Val u : inv_nodes) if (std::find(d.begin(), d.end(), u) == d.end()) d.push_back(u);
            return d;
        };
        // longest path to the end (compute nodes)
        std::vector<double> bl(n, 0.0);
        std::vector<std::vector<Val>> succ(n);
        for (size_t i = 0; i < n; ++i) if (is_compute((Val)i)) for (Val v : deps((Val)i)) succ[v].push_back((Val)i);
        for (size_t i = n; i-- > 0;) if (is_compute((Val)i)) { double m = 0; for (Val sc : succ[i]) m = std::max(m, bl[sc]); bl[i] = m + op_weight(nodes[i]); }
        struct Item { int kind; Val v; double fin; };   // 0: compute node, 1: private load / constant, 2: barrier; fin: scheduled finish time
        std::vector<std::vector<Item>> items(K);
        std::vector<int> stream_of(n, -1);
        std::vector<uint32_t> bar_at(n, 0);      // barriers of its stream that precede the node
        std::vector<size_t> pos(n, 0);           // index of the node's item in its stream
        std::vector<std::vector<char>> loaded(K, std::vector<char>(n, 0));
        std::vector<int> missing(n, 0);
        std::vector<Val> ready;
        size_t left = 0;
        auto is_store = [&](Val v) { const uint32_t op = nodes[v].op; return op == OP_STORE_MSM || op == OP_STORE_SHARED || op == OP_STORE_LEFT || op == OP_STORE_GUARD; };
        for (size_t i = 0; i < n; ++i) if (is_compute((Val)i)) {
            ++left;
            const int m = (int)deps((Val)i).size();
            missing[i] = m;
            if (!m) ready.push_back((Val)i);
        }
        std::vector<double> clk(K, 0.0);
        std::vector<size_t> last_bar(K, 0);      // index of the stream's last barrier item + 1 (0: none): a new barrier goes behind it
        uint32_t bars = 0;
        Val last_node = (Val)-1;
        const double SLACK = 0.0;   // extra products a stream may idle before a node is moved to another one (swept in round 2: no gain)
        // Where stream q could start v: at its own clock — or, if an operand sits in another stream behind no barrier yet, after a
        // NEW barrier.  A barrier is one item in every stream: at the end of q; in a stream that holds such an operand, behind the
        // last of them; in every other stream anywhere behind its previous barrier — and in both cases not before the first item
        // that ends after q's clock, so that the stream does not wait for q.  All streams leave it at the latest arrival + BAR.
        struct Place { bool bar; std::vector<size_t> p; double start; };   // p[r]: the barrier goes behind item p[r] - 1 of stream r (p[r] = insertion index)
        auto place = [&](Val v, int q) -> Place {
            Place pl{false, {}, clk[q]};
            std::vector<size_t> lo(K, 0);
            for (Val u : deps(v))
                if (stream_of[u] != q && bar_at[u] == bars) { pl.bar = true; lo[stream_of[u]] = std::max(lo[stream_of[u]], pos[u] + 1); }
            if (!pl.bar) return pl;
            pl.p.assign(K, 0);
            double tb = clk[q];
            for (int r = 0; r < K; ++r) {
                if (r == q) continue;
                size_t ins = std::max(lo[r], last_bar[r]);     // insertion index: items [0, ins) stay in front of the barrier
                while (ins < items[r].size() && (ins == 0 ? 0.0 : items[r][ins - 1].fin) < clk[q]) ++ins;
                pl.p[r] = ins;
                tb = std::max(tb, ins == 0 ? 0.0 : items[r][ins - 1].fin);
            }
            pl.start = tb + BAR;
            return pl;
        };
        auto append = [&](int q, Val v) {
            for (int j = 0; j < n_operands(nodes[v]); ++j) {
                const Val u = operand(v, j);
                if (is_compute(u) || inline_const(nodes[v], u) || loaded[q][u]) continue;
                clk[q] += op_weight(nodes[u]); items[q].push_back({1, u, clk[q]}); loaded[q][u] = 1;
            }
            clk[q] += op_weight(nodes[v]);
            pos[v] = items[q].size(); items[q].push_back({0, v, clk[q]});
            stream_of[v] = q; bar_at[v] = bars; --left;
        };
        while (left) {
            // the ready node with the longest tail; the stream where it finishes first
            size_t best = 0;
            for (size_t r = 1; r < ready.size(); ++r) if (bl[ready[r]] > bl[ready[best]] + 1e-9) best = r;
            // (SLACK > 0: within that slack of the longest tail a consumer of the node scheduled last goes first — fewer live slots,
            // but measured slower with four streams: 213 / 230 / 254 us for a slack of 0 / 1 / 3 products; the default is 0)
            if (last_node != (Val)-1) {
                const double top = bl[ready[best]];
                size_t loc = (size_t)-1;
                for (size_t r = 0; r < ready.size(); ++r) {
                    if (bl[ready[r]] < top - SLACK) continue;
                    const Node& nd = nodes[ready[r]];
                    bool uses = false;
                    for (int j = 0; j < n_operands(nd); ++j) if ((j ? nd.b : nd.a) == last_node) uses = true;
                    if (uses && (loc == (size_t)-1 || bl[ready[r]] > bl[ready[loc]])) loc = r;
                }
                if (loc != (size_t)-1) best = loc;
            }
            const Val v = ready[best];
            last_node = v;
            ready.erase(ready.begin() + (ptrdiff_t)best);
            int q = 0;
            Place chosen = place(v, 0);
            for (int r = 1; r < K; ++r) {
                Place c = place(v, r);
                if (c.start < chosen.start - 1e-9 || (c.start < chosen.start + 1e-9 && chosen.bar && !c.bar)) { chosen = c; q = r; }
            }
            if (chosen.bar) {
                const double tb = chosen.start;                       // every stream leaves the barrier at tb
                for (int r = 0; r < K; ++r) {
                    if (r == q) continue;
                    const size_t ins = chosen.p[r];
                    const double before = ins == 0 ? 0.0 : items[r][ins - 1].fin;
                    const double shift = tb - before;                 // what the items of r behind the barrier are delayed by
                    items[r].insert(items[r].begin() + (ptrdiff_t)ins, Item{2, 0, tb});
                    for (size_t k = ins + 1; k < items[r].size(); ++k) {
                        items[r][k].fin += shift;
                        if (items[r][k].kind == 0) { pos[items[r][k].v] = k; bar_at[items[r][k].v] = bars + 1; }
                    }
                    clk[r] = items[r].back().fin;
                    last_bar[r] = ins + 1;
                }
                items[q].push_back(Item{2, 0, tb});
                clk[q] = tb;
                last_bar[q] = items[q].size();
                ++bars;
            }
            append(q, v);
            // its stores at once, on the same stream: they free the slot and need no barrier — unless something else the store waits for
            // (its value when it was the inversion that completed, or the inversion) sits in another stream with no barrier behind it yet
            std::vector<Val> now_ready;
            for (Val sc : succ[v]) if (--missing[sc] == 0) now_ready.push_back(sc);
            for (Val sc : now_ready) {
                bool direct = is_store(sc);
                if (direct) for (Val u : deps(sc)) if (stream_of[u] != q && bar_at[u] == bars) direct = false;
                if (direct) append(q, sc); else ready.push_back(sc);
            }
        }
        makespan_k = *std::max_element(clk.begin(), clk.end());
        // ---- slots.  Value ids: compute node v -> v; private load / constant u of stream q -> n * (1 + q) + u
        auto value_of = [&](Val u, int q) -> size_t { return is_compute(u) ? (size_t)u : n * (size_t)(1 + q) + u; };
        const size_t NV = (size_t)(K + 1) * n;
        std::vector<uint32_t> last_epoch(NV, 0), slot(NV, 0);
        std::vector<char> shared(NV, 0), has_def(NV, 0);
        std::vector<int> owner(NV, -1);
        std::vector<size_t> last_pos(NV, 0);   // position of the last use in the owner's stream (private values)
        for (int q = 0; q < K; ++q) {
            uint32_t ep = 0;
            for (size_t k = 0; k < items[q].size(); ++k) {
                const Item& it = items[q][k];
                if (it.kind == 2) { ++ep; continue; }
                const Node& nd = nodes[it.v];
                auto touch = [&](size_t id) {
                    if (owner[id] == -1) owner[id] = q; else if (owner[id] != q) shared[id] = 1;
                    last_epoch[id] = std::max(last_epoch[id], ep);
                    if (owner[id] == q) last_pos[id] = k;
                };
                if (it.kind == 0) for (int j = 0; j < n_operands(nd); ++j) { const Val u = j ? nd.b : nd.a; if (!inline_const(nd, u)) touch(value_of(u, q)); }
                if (nd.has_result) { const size_t id = it.kind == 1 ? value_of(it.v, q) : (size_t)it.v; has_def[id] = 1; touch(id); }
            }
        }
        // a compute node belongs to the stream that defines it: a use seen first (another stream, a later epoch) makes it shared
        for (size_t i = 0; i < n; ++i) if (is_compute((Val)i) && stream_of[i] >= 0 && owner[i] != stream_of[i]) { owner[i] = stream_of[i]; shared[i] = 1; }
        uint32_t next = 0;
        std::vector<std::vector<uint32_t>> local_free(K);
        std::vector<std::pair<uint32_t, uint32_t>> pool;   // (slot, first epoch it may be reused in)
        std::vector<std::vector<size_t>> release_after(1);  // shared values by last epoch
        for (size_t id = 0; id < NV; ++id) if (has_def[id] && shared[id]) { if (release_after.size() <= last_epoch[id]) release_after.resize(last_epoch[id] + 1); release_after[last_epoch[id]].push_back(id); }
        std::vector<size_t> cur(K, 0);
        uint32_t ep = 0;
        auto take = [&](int q) -> uint32_t {
            if (!local_free[q].empty()) { auto itf = std::min_element(local_free[q].begin(), local_free[q].end()); const uint32_t sl = *itf; local_free[q].erase(itf); return sl; }
            size_t bi = (size_t)-1;
            for (size_t k = 0; k < pool.size(); ++k) if (pool[k].second <= ep && (bi == (size_t)-1 || pool[k].first < pool[bi].first)) bi = k;
            if (bi != (size_t)-1) { const uint32_t sl = pool[bi].first; pool.erase(pool.begin() + (ptrdiff_t)bi); return sl; }
            return next++;
        };
        for (;;) {
            for (int q = 0; q < K; ++q) {
                size_t& k = cur[q];
                for (; k < items[q].size() && items[q][k].kind != 2; ++k) {
                    const Item& it = items[q][k];
                    const Node& nd = nodes[it.v];
                    VmInstr in{nd.op, 0, 0, 0};
                    auto opnd = [&](Val u) -> uint32_t { return inline_const(nd, u) ? (VM_CONST_OPERAND | nodes[u].imm) : slot[value_of(u, q)]; };
                    if (it.kind == 1) in.a = nd.imm;
                    else switch (nd.op) {
                        case OP_MUL: case OP_ADD: case OP_SUB: in.a = opnd(nd.a); in.b = opnd(nd.b); break;
                        case OP_NEG: case OP_INV: in.a = opnd(nd.a); break;
                        default: in.a = opnd(nd.a); in.b = nd.imm; break;   // POW / SQRN / STORE_*
                    }
                    // private operands that die here free their slots first: the interpreter reads before it writes
                    if (it.kind == 0) for (int j = 0; j < n_operands(nd); ++j) {
                        const Val u = j ? nd.b : nd.a;
                        if (inline_const(nd, u) || (j == 1 && nd.b == nd.a)) continue;
                        const size_t id = value_of(u, q);
                        if (!shared[id] && last_pos[id] == k) local_free[q].push_back(slot[id]);
                    }
                    if (nd.has_result) {
                        const size_t id = it.kind == 1 ? value_of(it.v, q) : (size_t)it.v;
                        slot[id] = take(q); in.d = slot[id];
                        if (!shared[id] && last_pos[id] == k) local_free[q].push_back(slot[id]);   // never used
                    }
                    code[q].push_back(in);
                }
            }
            bool done = true;
            for (int q = 0; q < K; ++q) if (cur[q] < items[q].size()) done = false;
            if (done) break;
            // all streams stand at the same barrier
            for (int q = 0; q < K; ++q) { code[q].push_back(VmInstr{OP_BARRIER, 0, 0, 0}); ++cur[q]; }
            if (ep < release_after.size()) for (size_t id : release_after[ep]) pool.push_back({slot[id], ep + 1});
            ++ep;
        }
        n_slots = next ? next : 1;
    }
};

enum CommitKind { K_ADVICE, K_PERM_PRODUCT, K_LOOKUP, K_SHUFFLE, K_FIXED, K_PERM_COMMON, K_H_MSM, K_RANDOM };
struct CommitRef {
    int kind, idx, inst;   // inst: the circuit instance the commitment belongs to (0 for the VK-wide ones)
    bool operator==(const CommitRef& o) const { return kind == o.kind && idx == o.idx && inst == o.inst; }
};
struct SymQuery { CommitRef c; int64_t rot; Val eval; };  // point = x * omega^rot (rot normalised mod n)
}  // namespace

// =============================================================================== plan compiler
int compile_plan(const VkHost& vk, const ParamsHost& params, const std::vector<size_t>& col_lens, PlanOptions opts, Plan& plan, std::string& err) {
    plan.opts = opts;
    const bool gwc = opts.multiopen == H2V_MULTIOPEN_GWC;
    if (opts.multiopen < 0 || opts.multiopen > 1 || opts.transcript < 0 || opts.transcript > 1) { err = "unknown multiopen / transcript option"; return H2V_ERR_BAD_ARGUMENT; }
    // M circuit instances share the transcript (`instances: &[&[&[Fr]]]`, lib.rs:33-55): col_lens is instance-major, M x columns
    const size_t M = opts.circuit_instances > 0 ? (size_t)opts.circuit_instances : 1;
    if (col_lens.size() != M * vk.num_instance_columns) { err = "instances do not match the VK's instance column count"; return H2V_ERR_INVALID_INSTANCES; }
    const size_t NIC = vk.num_instance_columns;
    if (params.k != vk.k) { err = "params.k differs from vk.k"; return H2V_ERR_BAD_ARGUMENT; }
    const uint64_t n = 1ULL << vk.k;
    size_t total_inst = 0;
    for (size_t l : col_lens) total_inst += l;
    if (total_inst > (1u << 20)) { err = "more than 2^20 instance values per proof are not supported by this build"; return H2V_ERR_INSTANCE_TOO_LARGE; }
    for (size_t l : col_lens) if (l > n) { err = "instance column longer than the domain"; return H2V_ERR_INSTANCE_TOO_LARGE; }
    plan.col_lens = col_lens; plan.n_instance_values = (uint32_t)total_inst;
    {   // h2v_options.instance_kernel_threshold overrides the bound (tests force the kernel path on small circuits with 1)
        const size_t threshold = opts.instance_kernel_threshold > 0 ? (size_t)opts.instance_kernel_threshold - 1 : 1024;
        plan.wide_instances = total_inst > threshold;
    }

    const size_t A = vk.num_advice_columns, L = vk.lookups.size(), Sh = vk.shuffles.size(), P = vk.permutation_columns.size();
    const size_t chunk = vk.cs_degree - 2, nsets = P == 0 ? 0 : (P + chunk - 1) / chunk, H = vk.cs_degree - 1;
    const size_t Qa = vk.advice_queries.size(), Qf = vk.fixed_queries.size(), Ch = vk.num_challenges;
    const size_t bf = vk.blinding_factors();
    uint8_t max_phase = 0;
    for (uint8_t p : vk.advice_column_phase) max_phase = std::max(max_phase, p);
    if (vk.advice_column_phase.size() != A || vk.challenge_phase.size() != Ch || vk.fixed_commitments.size() < vk.num_fixed_columns) { err = "inconsistent VK"; return H2V_ERR_FORMAT; }

    // ---------------- domain constants (poly/domain.rs:34-140)
    Fr omega;
    {
        // ROOT_OF_UNITY = 7^((r-1)/2^28); omega = ROOT_OF_UNITY^(2^(28-k))
        uint32_t e[8]; for (int i = 0; i < 8; ++i) e[i] = FrParams::P(i);
        e[0] -= 1;
        for (int i = 0; i < 8; ++i) e[i] = (e[i] >> 28) | (i < 7 ? (e[i + 1] << 4) : 0);
        omega = Fr::from_u32(7).pow_limbs(e);
        for (uint32_t i = vk.k; i < 28; ++i) omega = omega.sqr();
    }
    const Fr omega_inv = omega.inv();
    const Fr n_inv = Fr::from_u32((uint32_t)n).inv();  // k <= 28
    Fr delta = Fr::from_u32(7);
    for (int i = 0; i < 28; ++i) delta = delta.sqr();  // DELTA = 7^(2^28)
    auto omega_pow = [&](int64_t r) {
        Fr base = r >= 0 ? omega : omega_inv;
        uint64_t e = (uint64_t)(r >= 0 ? r : -r);
        Fr acc = Fr::one();
        for (int i = 63; i >= 0; --i) { acc = acc.sqr(); if ((e >> i) & 1) acc = acc * base; }
        return acc;
    };
    auto norm_rot = [&](int64_t r) { int64_t m = (int64_t)n; return ((r % m) + m) % m; };

    // a proof of this VK: Np points, Ns scalars.  The layout tables below are linear in them; a key that asks for more
    // than 2^16 of either is refused rather than laid out (real keys: tens to hundreds)
    {
        const uint64_t np_total = (uint64_t)M * (A + 3 * L + Sh + nsets) + 1 + H + 2 + 64;
        const uint64_t ns_total = (uint64_t)M * (Qa + 3 * nsets + 5 * L + 2 * Sh) + Qf + 1 + P;
        if (np_total > 65536 || ns_total > 65536) { err = "a proof of this VerifyingKey has more than 65536 points or scalars: not supported by this build"; return H2V_ERR_UNSUPPORTED; }
    }
    // ---------------- proof layout + transcript stream
    // point slots; the per-instance ones are indexed [m * count + i]
    std::vector<uint32_t> advice_slot(M * A, 0), lk_input_slot(M * L), lk_table_slot(M * L), lk_product_slot(M * L), sh_slot(M * Sh), perm_slot(M * nsets), h_slot(H);
    uint32_t random_slot = 0;
    uint32_t np = 0, nsc = 0, off = 0;
    std::vector<uint32_t> squeeze_order;  // challenge id of each squeeze
    auto emit_const = [&](uint8_t b) { plan.stream.push_back({TranscriptSrc::CONST, b, 0}); };
    auto absorb_point = [&]() -> uint32_t {
        uint32_t slot = np++;
        plan.point_offsets.push_back(off);
        emit_const(1);
        for (uint32_t i = 0; i < 32; ++i) plan.stream.push_back({(uint8_t)(i == 31 ? TranscriptSrc::PROOF_MASKED : TranscriptSrc::PROOF), 0, off + i});
        for (uint32_t i = 0; i < 32; ++i) plan.stream.push_back({TranscriptSrc::YCOORD, 0, slot * 32 + i});
        off += 32;
        return slot;
    };
    auto absorb_scalar = [&]() -> uint32_t {
        uint32_t idx = nsc++;
        plan.scalar_offsets.push_back(off);
        emit_const(2);
        for (uint32_t i = 0; i < 32; ++i) plan.stream.push_back({TranscriptSrc::PROOF, 0, off + i});
        off += 32;
        return idx;
    };
    auto squeeze = [&](uint32_t chal_id) {
        emit_const(0);
        plan.squeeze_at.push_back((uint32_t)plan.stream.size());
        squeeze_order.push_back(chal_id);
    };
    // SHPLONK squeezes y', v, u (shplonk.rs:195-199); GWC squeezes v, u (gwc.rs:73-83)
    const uint32_t C_THETA = (uint32_t)Ch, C_BETA = C_THETA + 1, C_GAMMA = C_THETA + 2, C_Y = C_THETA + 3, C_X = C_THETA + 4;
    const uint32_t C_SY = C_THETA + 5, C_SV = gwc ? C_THETA + 5 : C_THETA + 6, C_SU = C_SV + 1;
    plan.n_user_challenges = (uint32_t)Ch; plan.n_challenges = C_SU + 1;
    if (opts.transcript == H2V_TRANSCRIPT_KECCAK256)   // Keccak256Read::init absorbs the label (transcript/mod.rs:143-145)
        for (const char* c = "Halo2-Transcript"; *c; ++c) emit_const((uint8_t)*c);
    {   // vk.hash_into + instances (plonk/vk.rs:145-152, lib.rs:76-82)
        uint8_t repr[32]; vk.transcript_repr.to_bytes(repr);
        emit_const(2);
        for (int i = 0; i < 32; ++i) emit_const(repr[i]);
        for (uint32_t v = 0; v < total_inst; ++v) { emit_const(2); for (uint32_t i = 0; i < 32; ++i) plan.stream.push_back({TranscriptSrc::INSTANCE, 0, v * 32 + i}); }
    }
    for (unsigned phase = 0; phase <= max_phase; ++phase) {  // lib.rs:91-109: every instance's advice of the phase, then its challenges
        for (size_t m = 0; m < M; ++m)
            for (size_t i = 0; i < A; ++i) if (vk.advice_column_phase[i] == phase) advice_slot[m * A + i] = absorb_point();
        for (size_t i = 0; i < Ch; ++i) if (vk.challenge_phase[i] == phase) squeeze((uint32_t)i);
    }
    squeeze(C_THETA);
    for (size_t m = 0; m < M; ++m) for (size_t i = 0; i < L; ++i) { lk_input_slot[m * L + i] = absorb_point(); lk_table_slot[m * L + i] = absorb_point(); }   // lib.rs:117-126
    squeeze(C_BETA); squeeze(C_GAMMA);
    for (size_t m = 0; m < M; ++m) for (size_t i = 0; i < nsets; ++i) perm_slot[m * nsets + i] = absorb_point();     // lib.rs:134-139
    for (size_t m = 0; m < M; ++m) for (size_t i = 0; i < L; ++i) lk_product_slot[m * L + i] = absorb_point();         // lib.rs:141-150
    for (size_t m = 0; m < M; ++m) for (size_t i = 0; i < Sh; ++i) sh_slot[m * Sh + i] = absorb_point();               // lib.rs:152-161
    random_slot = absorb_point();
    squeeze(C_Y);
    for (size_t i = 0; i < H; ++i) h_slot[i] = absorb_point();
    squeeze(C_X);
    // evaluations (lib.rs:220-253)
    std::vector<uint32_t> s_adv(M * Qa), s_fix(Qf), s_sigma(P);   // advice evaluations: instance by instance (lib.rs:220-222)
    for (auto& s : s_adv) s = absorb_scalar();
    for (auto& s : s_fix) s = absorb_scalar();
    uint32_t s_random = absorb_scalar();
    for (auto& s : s_sigma) s = absorb_scalar();
    struct PS { uint32_t eval, next, last; bool has_last; };
    std::vector<PS> s_perm(M * nsets);
    for (size_t m = 0; m < M; ++m)
        for (size_t i = 0; i < nsets; ++i) {
            PS& ps = s_perm[m * nsets + i];
            ps.eval = absorb_scalar(); ps.next = absorb_scalar();
            ps.has_last = i + 1 < nsets;
            ps.last = ps.has_last ? absorb_scalar() : 0;
        }
    struct LS { uint32_t product, product_next, input, input_inv, table; };
    std::vector<LS> s_lk(M * L);
    for (auto& s : s_lk) { s.product = absorb_scalar(); s.product_next = absorb_scalar(); s.input = absorb_scalar(); s.input_inv = absorb_scalar(); s.table = absorb_scalar(); }
    struct SS { uint32_t product, product_next; };
    std::vector<SS> s_sh(M * Sh);
    for (auto& s : s_sh) { s.product = absorb_scalar(); s.product_next = absorb_scalar(); }
    plan.n_main_points = np;
    plan.opening_offset = off;
    // distinct opening points in first-appearance order of the query list (lib.rs:349-414) — GWC reads one witness
    // point per distinct point (gwc.rs:138-163); computed here because the number of points it reads depends on it
    std::vector<int64_t> gwc_points;
    std::vector<uint32_t> gwc_w_slot;
    if (gwc) {
        auto seen = [&](int64_t r) { int64_t k2 = norm_rot(r); for (int64_t e : gwc_points) if (e == k2) return; gwc_points.push_back(k2); };
        // (every instance contributes the same rotations in the same order, so the first-appearance order is instance 0's)
        for (const QueryH& q : vk.advice_queries) seen(q.rotation);
        if (nsets) { seen(0); seen(1); if (nsets > 1) seen(-(int64_t)(bf + 1)); }
        if (L) { seen(0); seen(-1); seen(1); }
        if (Sh) { seen(0); seen(1); }
        for (const QueryH& q : vk.fixed_queries) seen(q.rotation);
        seen(0);
        squeeze(C_SV);
        for (size_t i = 0; i < gwc_points.size(); ++i) gwc_w_slot.push_back(absorb_point());
        squeeze(C_SU);
    } else {
        squeeze(C_SY); squeeze(C_SV);
        plan.slot_h1 = absorb_point();
        squeeze(C_SU);
        plan.slot_h2 = absorb_point();
    }
    plan.n_points = np; plan.n_scalars = nsc; plan.proof_len = off;
    // challenge id -> position in squeeze order
    std::vector<uint32_t> sq_of(plan.n_challenges, 0);
    for (size_t q = 0; q < squeeze_order.size(); ++q) sq_of[squeeze_order[q]] = (uint32_t)q;

    // ---------------- the Fr program
    Builder b;
    auto chal = [&](uint32_t id) { return b.load_chal(sq_of[id]); };
    std::vector<Val> user_ch(Ch);
    for (size_t i = 0; i < Ch; ++i) user_ch[i] = chal((uint32_t)i);
    Val theta = chal(C_THETA), beta = chal(C_BETA), gamma = chal(C_GAMMA), y = chal(C_Y), x = chal(C_X), sv = chal(C_SV), su = chal(C_SU);
    plan.x_chal = sq_of[C_X]; plan.domain_k = vk.k; plan.omega = omega; plan.n_inv = n_inv;
    Val sy = gwc ? sv : chal(C_SY);
    Val xn = b.sqrn(x, vk.k);  // x^n, n = 2^k   (lib.rs:180,259)
    Val xn_m1 = b.sub(xn, b.one());

    // every inversion of the proof goes through one batch inversion
    //   [0] xn - 1 (vanishing.rs:100)  [1] x (interpolation denominators)  [2] z_diff_0 (shplonk.rs:215)  [3..] x - omega^i
    std::map<int64_t, size_t> l_index;  // normalised rotation -> position in `dens`
    std::vector<int64_t> l_rots;
    auto need_l = [&](int64_t r) { int64_t k2 = norm_rot(r); if (!l_index.count(k2)) { l_index[k2] = l_rots.size(); l_rots.push_back(r); } };
    for (int64_t r = -(int64_t)(bf + 1); r <= 0; ++r) need_l(r);
    {
        size_t flat = 0; (void)flat;
        for (const QueryH& q : vk.instance_queries) {
            if (q.column.index >= NIC) { err = "instance query names a missing column"; return H2V_ERR_FORMAT; }
            if (!plan.wide_instances) for (size_t m = 0; m < M; ++m) for (size_t j = 0; j < col_lens[m * NIC + q.column.index]; ++j) need_l((int64_t)j - q.rotation);
        }
    }

    // ---------------- symbolic SHPLONK bookkeeping needs the query list; build evals first
    std::vector<Val> advice_evals_all(M * Qa), fixed_evals(Qf), sigma_evals(P);
    for (size_t i = 0; i < M * Qa; ++i) advice_evals_all[i] = b.load_scalar(s_adv[i]);
    for (size_t i = 0; i < Qf; ++i) fixed_evals[i] = b.load_scalar(s_fix[i]);
    for (size_t i = 0; i < P; ++i) sigma_evals[i] = b.load_scalar(s_sigma[i]);
    Val random_eval = b.load_scalar(s_random);

    // rotation sets (shplonk.rs:58-149), symbolic in the rotation; the evaluations are filled in below
    std::vector<SymQuery> queries;
    auto add_query = [&](CommitRef c, int64_t rot, Val e) { queries.push_back({c, norm_rot(rot), e}); };
    // (the eval Vals of h / instance-dependent values are patched after they exist; collect structure first)
    std::vector<Val> pz_all(M * nsets), pz_next_all(M * nsets), pz_last_all(M * nsets);
    struct LV { Val product, product_next, input, input_inv, table; };
    std::vector<LV> lk_all(M * L);
    struct SV2 { Val product, product_next; };
    std::vector<SV2> shv_all(M * Sh);
    for (size_t m = 0; m < M; ++m) {   // lib.rs:349-391: per instance advice, permutation, lookups, shuffles
        const int im = (int)m;
        for (size_t qi = 0; qi < Qa; ++qi) {
            if (vk.advice_queries[qi].column.index >= A) { err = "advice query names a missing column"; return H2V_ERR_FORMAT; }
            add_query({K_ADVICE, (int)vk.advice_queries[qi].column.index, im}, vk.advice_queries[qi].rotation, advice_evals_all[m * Qa + qi]);
        }
        Val* pz = &pz_all[m * nsets]; Val* pz_next = &pz_next_all[m * nsets]; Val* pz_last = &pz_last_all[m * nsets];
        for (size_t i = 0; i < nsets; ++i) {
            const PS& ps = s_perm[m * nsets + i];
            pz[i] = b.load_scalar(ps.eval); pz_next[i] = b.load_scalar(ps.next); if (ps.has_last) pz_last[i] = b.load_scalar(ps.last);
        }
        for (size_t i = 0; i < nsets; ++i) { add_query({K_PERM_PRODUCT, (int)i, im}, 0, pz[i]); add_query({K_PERM_PRODUCT, (int)i, im}, 1, pz_next[i]); }
        for (size_t i = nsets; i-- > 0;) { if (i + 1 == nsets) continue; add_query({K_PERM_PRODUCT, (int)i, im}, -(int64_t)(bf + 1), pz_last[i]); }
        for (size_t i = 0; i < L; ++i) {
            const LS& ls = s_lk[m * L + i];
            LV& e = lk_all[m * L + i];
            e = {b.load_scalar(ls.product), b.load_scalar(ls.product_next), b.load_scalar(ls.input), b.load_scalar(ls.input_inv), b.load_scalar(ls.table)};
            add_query({K_LOOKUP, (int)(3 * i + 0), im}, 0, e.product);
            add_query({K_LOOKUP, (int)(3 * i + 1), im}, 0, e.input);
            add_query({K_LOOKUP, (int)(3 * i + 2), im}, 0, e.table);
            add_query({K_LOOKUP, (int)(3 * i + 1), im}, -1, e.input_inv);
            add_query({K_LOOKUP, (int)(3 * i + 0), im}, 1, e.product_next);
        }
        for (size_t i = 0; i < Sh; ++i) {
            SV2& e = shv_all[m * Sh + i];
            e = {b.load_scalar(s_sh[m * Sh + i].product), b.load_scalar(s_sh[m * Sh + i].product_next)};
            add_query({K_SHUFFLE, (int)i, im}, 0, e.product);
            add_query({K_SHUFFLE, (int)i, im}, 1, e.product_next);
        }
    }
    for (size_t qi = 0; qi < Qf; ++qi) {
        if (vk.fixed_queries[qi].column.index >= vk.fixed_commitments.size()) { err = "fixed query names a missing column"; return H2V_ERR_FORMAT; }
        add_query({K_FIXED, (int)vk.fixed_queries[qi].column.index, 0}, vk.fixed_queries[qi].rotation, fixed_evals[qi]);
    }
    for (size_t i = 0; i < P; ++i) add_query({K_PERM_COMMON, (int)i, 0}, 0, sigma_evals[i]);
    const size_t q_hmsm = queries.size();
    add_query({K_H_MSM, 0, 0}, 0, 0 /* patched: expected_h_eval */);
    add_query({K_RANDOM, 0, 0}, 0, random_eval);

    struct RotSet { std::vector<int64_t> rots; std::vector<CommitRef> commits; };
    std::vector<RotSet> rsets; std::set<int64_t> super;
    {
        std::vector<std::pair<CommitRef, std::set<int64_t>>> cmap;
        for (const SymQuery& q : queries) {
            super.insert(q.rot);
            bool found = false;
            for (auto& e : cmap) if (e.first == q.c) { e.second.insert(q.rot); found = true; break; }
            if (!found) cmap.push_back({q.c, {q.rot}});
        }
        for (auto& e : cmap) {
            bool found = false;
            for (auto& r : rsets) if (std::set<int64_t>(r.rots.begin(), r.rots.end()) == e.second) { r.commits.push_back(e.first); found = true; break; }
            if (!found) rsets.push_back({std::vector<int64_t>(e.second.begin(), e.second.end()), {e.first}});
        }
    }

    // ---------------- batch inversion
    std::map<int64_t, Val> point_of;  // x * omega^rot for the opening points
    for (int64_t r : super) point_of[r] = b.mul(x, b.cst(omega_pow(r)));
    Val z_diff_0 = b.one();
    if (!gwc) for (int64_t r : super) if (std::find(rsets[0].rots.begin(), rsets[0].rots.end(), r) == rsets[0].rots.end()) z_diff_0 = b.mul(b.sub(su, point_of[r]), z_diff_0);
    // GWC needs neither 1/x nor 1/z_diff_0; keeping two harmless entries keeps the indices below fixed
    std::vector<Val> inv_list = {xn_m1, gwc ? b.one() : x, gwc ? b.one() : z_diff_0};
    for (int64_t r : l_rots) inv_list.push_back(b.sub(x, b.cst(omega_pow(r))));
    b.batch_invert(inv_list);
    Val xn_m1_inv = inv_list[0], x_inv = inv_list[1], z_0_diff_inverse = inv_list[2];
    Val common = b.mul(xn_m1, b.cst(n_inv));  // (xn - 1) * barycentric_weight   (poly/domain.rs:206)
    auto l_at = [&](int64_t r) { size_t i = l_index[norm_rot(r)]; return b.mul(b.mul(inv_list[3 + i], common), b.cst(omega_pow(l_rots[i]))); };

    // ---------------- instance evaluations (lib.rs:173-218): sum_j inst[col][j] * l_{j - rot}(x)
    std::vector<std::vector<Val>> instance_evals_all(M);
    {
        std::vector<uint32_t> col_base(col_lens.size(), 0);
        for (size_t c = 1; c < col_lens.size(); ++c) col_base[c] = col_base[c - 1] + (uint32_t)col_lens[c - 1];
        std::map<int64_t, Val> l_cache;
        for (size_t m = 0; m < M; ++m)
            for (const QueryH& q : vk.instance_queries) {
                const size_t col = m * NIC + q.column.index;
                if (plan.wide_instances) {   // evaluated by k_instance_eval before the program runs
                    plan.inst_queries.push_back({col_base[col], (uint32_t)col_lens[col], omega_pow(-(int64_t)q.rotation)});
                    instance_evals_all[m].push_back(b.load_insteval((uint32_t)plan.inst_queries.size() - 1));
                    continue;
                }
                Val acc = b.zero();
                for (size_t j = 0; j < col_lens[col]; ++j) {
                    int64_t r = norm_rot((int64_t)j - q.rotation);
                    if (!l_cache.count(r)) l_cache[r] = l_at(r);
                    acc = b.add(acc, b.mul(b.load_inst(col_base[col] + (uint32_t)j), l_cache[r]));
                }
                instance_evals_all[m].push_back(acc);
            }
    }
    // l_last, l_blind, l_0 (lib.rs:259-270)
    Val l_last = l_at(-(int64_t)(bf + 1));
    Val l_blind = b.zero();
    for (int64_t r = -(int64_t)bf; r <= -1; ++r) l_blind = b.add(l_blind, l_at(r));
    Val l_0 = l_at(0);

    // ---------------- expressions (lib.rs:273-346)
    const size_t Qi = vk.instance_queries.size();
    int expr_err = 0;
    std::vector<Val> exprs;
    Val active_rows = b.sub(b.one(), b.add(l_last, l_blind));
    for (size_t m = 0; m < M; ++m) {   // lib.rs:273-346: flat_map over the instances — gates, permutation, lookups, shuffles of each
        const Val* advice_evals = &advice_evals_all[m * Qa];
        const std::vector<Val>& instance_evals = instance_evals_all[m];
        std::map<std::pair<uint32_t, uint32_t>, Val> pow_cache;
        auto var_at = [&](uint32_t idx) -> Val {
            if (idx < Qa) return advice_evals[idx];
            if (idx < Qa + Qf) return fixed_evals[idx - Qa];
            if (idx < Qa + Qf + Qi) return instance_evals[idx - Qa - Qf];
            if (idx < Qa + Qf + Qi + Ch) return user_ch[idx - Qa - Qf - Qi];
            expr_err = 1; return b.zero();  // "index out of range" panic (vk.rs:501)
        };
        auto eval_expr = [&](const ExprH& e) -> Val {
            if (e.terms.empty()) { expr_err = 1; return b.zero(); }  // unwrap on empty terms (multilinear.rs:65)
            Val sum = 0; bool first = true;
            for (const TermH& t : e.terms) {
                if (t.coeff_idx >= vk.coeff_vals.size()) { expr_err = 1; return b.zero(); }
                Val prod = b.one();
                for (const auto& f : t.factors) {
                    auto key = std::make_pair(f.first, f.second);
                    auto it = pow_cache.find(key);
                    Val pv = it != pow_cache.end() ? it->second : (pow_cache[key] = b.pow(var_at(f.first), f.second));
                    prod = b.mul(prod, pv);
                }
                Val term = b.mul(b.cst(vk.coeff_vals[t.coeff_idx]), prod);
                sum = first ? term : b.add(sum, term);
                first = false;
            }
            return sum;
        };
        for (const ExprH& g : vk.gates) exprs.push_back(eval_expr(g));
        auto column_eval = [&](const ColumnH& c) -> Val {  // get_any_query_index(column, Rotation::cur()) (vk.rs:413-455)
            const std::vector<QueryH>& qs = c.type <= 2 ? vk.advice_queries : (c.type == COL_FIXED ? vk.fixed_queries : vk.instance_queries);
            for (size_t i = 0; i < qs.size(); ++i)
                if (qs[i].column.index == c.index && qs[i].column.type == c.type && qs[i].rotation == 0)
                    return c.type <= 2 ? advice_evals[i] : (c.type == COL_FIXED ? fixed_evals[i] : instance_evals[i]);
            expr_err = 1; return b.zero();
        };
        const Val* pz = &pz_all[m * nsets]; const Val* pz_next = &pz_next_all[m * nsets]; const Val* pz_last = &pz_last_all[m * nsets];
        if (nsets > 0) {  // permutation.rs:189-288
            exprs.push_back(b.mul(l_0, b.sub(b.one(), pz[0])));
            exprs.push_back(b.mul(b.sub(b.sqr(pz[nsets - 1]), pz[nsets - 1]), l_last));
            for (size_t i = 1; i < nsets; ++i) exprs.push_back(b.mul(b.sub(pz[i], pz_last[i - 1]), l_0));
            Val beta_x = b.mul(beta, x);
            for (size_t ci = 0; ci < nsets; ++ci) {
                size_t lo = ci * chunk, hi = std::min(P, lo + chunk);
                Val left = pz_next[ci], right = pz[ci];
                Fr dpow = delta.pow_u32((uint32_t)(ci * chunk));
                for (size_t j = lo; j < hi; ++j) {
                    Val v = column_eval(vk.permutation_columns[j]);
                    left = b.mul(left, b.add(b.add(v, b.mul(beta, sigma_evals[j])), gamma));
                    right = b.mul(right, b.add(b.add(v, b.mul(beta_x, b.cst(dpow))), gamma));
                    dpow = dpow * delta;
                }
                exprs.push_back(b.mul(b.sub(left, right), active_rows));
            }
        }
        auto compress = [&](const std::vector<ExprH>& es) { Val acc = b.zero(); for (const ExprH& e : es) acc = b.add(b.mul(acc, theta), eval_expr(e)); return acc; };
        for (size_t i = 0; i < L; ++i) {  // lookup.rs:159-230
            const LV& e = lk_all[m * L + i];
            exprs.push_back(b.mul(l_0, b.sub(b.one(), e.product)));
            exprs.push_back(b.mul(l_last, b.sub(b.sqr(e.product), e.product)));
            Val left = b.mul(b.mul(e.product_next, b.add(e.input, beta)), b.add(e.table, gamma));
            Val right = b.mul(b.mul(e.product, b.add(compress(vk.lookups[i].input), beta)), b.add(compress(vk.lookups[i].table), gamma));
            exprs.push_back(b.mul(b.sub(left, right), active_rows));
            exprs.push_back(b.mul(l_0, b.sub(e.input, e.table)));
            exprs.push_back(b.mul(b.mul(b.sub(e.input, e.table), b.sub(e.input, e.input_inv)), active_rows));
        }
        for (size_t i = 0; i < Sh; ++i) {  // shuffle.rs:148-203
            const SV2& e = shv_all[m * Sh + i];
            exprs.push_back(b.mul(l_0, b.sub(b.one(), e.product)));
            exprs.push_back(b.mul(l_last, b.sub(b.sqr(e.product), e.product)));
            Val left = b.mul(e.product_next, b.add(compress(vk.shuffles[i].shuffle), gamma));
            Val right = b.mul(e.product, b.add(compress(vk.shuffles[i].input), gamma));
            exprs.push_back(b.mul(b.sub(left, right), active_rows));
        }
    }
    if (expr_err) { err = "the VK makes the reference panic (empty expression polynomial or out-of-range index)"; return H2V_ERR_REFERENCE_PANIC; }
    // vanishing.rs:92-121
    Val h_eval = b.zero();
    for (Val v : exprs) h_eval = b.add(b.mul(h_eval, y), v);
    Val expected_h_eval = b.mul(h_eval, xn_m1_inv);
    queries[q_hmsm].eval = expected_h_eval;

    // ---------------- SHPLONK scalar preparation (shplonk.rs:202-264)
    auto eval_of = [&](const CommitRef& c, int64_t rot) -> Val {
        for (const SymQuery& q : queries) if (q.c == c && q.rot == rot) return q.eval;
        return b.zero();
    };
    Val mult = b.load_mult();
    std::vector<Val> msm_scalar(np, (Val)-1);                    // per point slot
    const size_t F = vk.fixed_commitments.size();
    plan.n_shared = (uint32_t)(F + P + 1);
    std::vector<Val> shared_scalar(plan.n_shared, (Val)-1);
    plan.shared_bases.clear();
    for (const G1A& c : vk.fixed_commitments) plan.shared_bases.push_back(c);
    for (const G1A& c : vk.permutation_commitments) plan.shared_bases.push_back(c);
    plan.shared_bases.push_back(params.g);
    auto slot_of = [&](const CommitRef& c) -> std::pair<uint8_t, uint32_t> {  // (is_shared, index)
        switch (c.kind) {
            case K_ADVICE: return {0, advice_slot[c.inst * A + c.idx]};
            case K_PERM_PRODUCT: return {0, perm_slot[c.inst * nsets + c.idx]};
            case K_LOOKUP: return {0, c.idx % 3 == 0 ? lk_product_slot[c.inst * L + c.idx / 3] : (c.idx % 3 == 1 ? lk_input_slot[c.inst * L + c.idx / 3] : lk_table_slot[c.inst * L + c.idx / 3])};
            case K_SHUFFLE: return {0, sh_slot[c.inst * Sh + c.idx]};
            case K_FIXED: return {1, (uint32_t)c.idx};
            case K_PERM_COMMON: return {1, (uint32_t)(F + c.idx)};
            case K_RANDOM: return {0, random_slot};
            default: return {0, 0};
        }
    };
    auto assign = [&](std::pair<uint8_t, uint32_t> where, Val v) {
        Val& dst = where.first ? shared_scalar[where.second] : msm_scalar[where.second];
        // SHPLONK: a commitment belongs to exactly one rotation set; GWC: a commitment opened at several points occurs once
        // per point — its scalars are summed and it is reported once (first appearance) in the Guard
        if (dst == (Val)-1) { plan.right_term_order.push_back(where); dst = v; }
        else dst = b.add(dst, v);
    };
    std::vector<Val> left_scalar(np, (Val)-1);
    if (gwc) {
        // gwc.rs:86-132: point group i has weight u^i, query j inside it weight v^j
        std::vector<std::vector<const SymQuery*>> groups(gwc_points.size());
        for (const SymQuery& q : queries) {
            size_t gi = std::find(gwc_points.begin(), gwc_points.end(), q.rot) - gwc_points.begin();
            if (gi >= groups.size()) { err = "internal: opening point missing from the GWC point list"; return H2V_ERR_BAD_ARGUMENT; }
            groups[gi].push_back(&q);
        }
        // reference term order of the right channel: witness_with_aux, commitment_multi, (eval_multi, -g)
        Val power_of_u = b.one();
        std::vector<Val> pu(groups.size());
        for (size_t i = 0; i < groups.size(); ++i) { pu[i] = power_of_u; power_of_u = b.mul(su, power_of_u); }
        auto guard_term = [&](std::pair<uint8_t, uint32_t> where, Val v) {
            if (!opts.guard_terms) return;
            b.store_guard(v, (uint32_t)plan.guard_term_order.size());
            plan.guard_term_order.push_back(where);
        };
        for (size_t i = 0; i < groups.size(); ++i) {
            const Val wz = b.mul(pu[i], point_of[gwc_points[i]]);
            assign({0, gwc_w_slot[i]}, wz);
            guard_term({0, gwc_w_slot[i]}, wz);            // witness_with_aux (gwc.rs:118-119, added to the right channel first :127)
            left_scalar[gwc_w_slot[i]] = pu[i];
            plan.left_term_order.push_back({0, gwc_w_slot[i]});
        }
        Val eval_multi = b.zero();
        for (size_t i = 0; i < groups.size(); ++i) {
            Val power_of_v = b.one(), eval_batch = b.zero();
            for (const SymQuery* q : groups[i]) {
                Val w = b.mul(power_of_v, pu[i]);
                if (q->c.kind == K_H_MSM) {
                    std::vector<Val> xnp(H); if (H) xnp[0] = b.one();
                    for (size_t t = 1; t < H; ++t) xnp[t] = b.mul(xnp[t - 1], xn);
                    for (size_t t = H; t-- > 0;) { const Val hw = b.mul(w, xnp[t]); assign({0, h_slot[t]}, hw); guard_term({0, h_slot[t]}, hw); }
                } else { assign(slot_of(q->c), w); guard_term(slot_of(q->c), w); }   // commitment_multi, query by query (gwc.rs:96-116)
                eval_batch = b.add(eval_batch, b.mul(power_of_v, q->eval));
                power_of_v = b.mul(sv, power_of_v);
            }
            eval_multi = b.add(eval_multi, b.mul(pu[i], eval_batch));
        }
        plan.shared_bases.back().y = plan.shared_bases.back().y.neg();  // the last VK-wide base is -g for GWC (gwc.rs:130-131)
        assign({1, (uint32_t)(F + P)}, eval_multi);
        guard_term({1, (uint32_t)(F + P)}, eval_multi);    // (eval_multi, -g) (gwc.rs:130-131)
    } else {
    Val z_0 = b.one();
    for (int64_t r : rsets[0].rots) z_0 = b.mul(b.sub(su, point_of[r]), z_0);
    Val r_outer = b.zero();
    Val power_of_v = b.one();
    // powers of x^-1 for the interpolation denominators
    std::vector<Val> xinv_pow = {b.one(), x_inv};
    for (size_t i = 0; i < rsets.size(); ++i) {
        const RotSet& rs = rsets[i];
        Val z_diff_i;
        if (i == 0) z_diff_i = b.one();
        else {
            Val zd = b.one();
            for (int64_t r : super) if (std::find(rs.rots.begin(), rs.rots.end(), r) == rs.rots.end()) zd = b.mul(b.sub(su, point_of[r]), zd);
            z_diff_i = b.mul(zd, z_0_diff_inverse);
        }
        // Lagrange weights at u over this set's points: W_k = prod_{m != k}(u - p_m) / prod_{m != k}(p_k - p_m),
        // p_k - p_m = x (omega^rk - omega^rm): the omega part is a per-VK constant, the x part one shared inverse.
        size_t s = rs.rots.size();
        std::vector<Val> W(s);
        if (s > 1) {
            while (xinv_pow.size() < s) xinv_pow.push_back(b.mul(xinv_pow.back(), x_inv));
            for (size_t k2 = 0; k2 < s; ++k2) {
                Fr cden = Fr::one();
                Val num = b.one();
                for (size_t m = 0; m < s; ++m) {
                    if (m == k2) continue;
                    cden = cden * (omega_pow(rs.rots[k2]) - omega_pow(rs.rots[m]));
                    num = b.mul(num, b.sub(su, point_of[rs.rots[m]]));
                }
                W[k2] = b.mul(b.mul(num, xinv_pow[s - 1]), b.cst(cden.inv()));
            }
        }
        Val set_weight = b.mul(power_of_v, z_diff_i);
        Val r_inner = b.zero();
        Val power_of_y = b.one();
        for (size_t j = 0; j < rs.commits.size(); ++j) {
            const CommitRef& c = rs.commits[j];
            Val r_u;  // r_ij(u)
            if (s == 1) r_u = eval_of(c, rs.rots[0]);
            else { r_u = b.zero(); for (size_t k2 = 0; k2 < s; ++k2) r_u = b.add(r_u, b.mul(eval_of(c, rs.rots[k2]), W[k2])); }
            r_inner = b.add(r_inner, b.mul(power_of_y, r_u));
            Val term_scalar = b.mul(power_of_y, set_weight);
            if (c.kind == K_H_MSM) {
                // nested MSM: bases h_{H-1} .. h_0 with scalars xn^{H-1} .. 1 (vanishing.rs:102-112)
                std::vector<Val> xnp(H); if (H) xnp[0] = b.one();
                for (size_t t = 1; t < H; ++t) xnp[t] = b.mul(xnp[t - 1], xn);
                for (size_t t = H; t-- > 0;) assign({0, h_slot[t]}, b.mul(term_scalar, xnp[t]));
            } else {
                assign(slot_of(c), term_scalar);
            }
            power_of_y = b.mul(sy, power_of_y);
        }
        r_outer = b.add(r_outer, b.mul(b.mul(power_of_v, r_inner), z_diff_i));
        power_of_v = b.mul(sv, power_of_v);
    }
    assign({1, (uint32_t)(F + P)}, b.neg(r_outer));
    assign({0, plan.slot_h1}, b.neg(z_0));
    assign({0, plan.slot_h2}, su);
    left_scalar[plan.slot_h2] = b.one();   // left channel: (1, h2) per proof (shplonk.rs:262)
    plan.left_term_order.push_back({0, plan.slot_h2});
    }
    // stores, scaled by the proof's batch multiplier (kzg/strategy.rs:129, msm.rs:173-176)
    for (uint32_t s2 = 0; s2 < np; ++s2) b.store_msm(msm_scalar[s2] == (Val)-1 ? b.zero() : b.mul(msm_scalar[s2], mult), s2);
    for (uint32_t j = 0; j < plan.n_shared; ++j) b.store_shared(shared_scalar[j] == (Val)-1 ? b.zero() : b.mul(shared_scalar[j], mult), j);
    for (uint32_t s2 = 0; s2 < np; ++s2) if (left_scalar[s2] != (Val)-1) b.store_left(b.mul(left_scalar[s2], mult), s2);

    b.emit(plan.code, plan.n_slots);
    for (int K = 2; K <= FRVM_MAX_STREAMS; ++K) { b.emit_streams(K, plan.code_k[K - 2], plan.n_slots_k[K - 2]); plan.makespan_k[K - 2] = b.makespan_k; }
    {   // diagnostics kept with the plan (tests/cpp/plan_host.hip prints them): the DAG's work and critical path in units of one Fr product
        std::vector<double> depth(b.nodes.size(), 0.0);
        double work = 0, cp = 0;
        for (size_t i = 0; i < b.nodes.size(); ++i) {
            const Node& nd = b.nodes[i];
            double d0 = 0;
            const bool two = nd.op == OP_MUL || nd.op == OP_ADD || nd.op == OP_SUB;
            const bool one = two || nd.op == OP_NEG || nd.op == OP_INV || nd.op == OP_POW || nd.op == OP_SQRN || nd.op == OP_STORE_MSM || nd.op == OP_STORE_SHARED || nd.op == OP_STORE_LEFT || nd.op == OP_STORE_GUARD;
            if (one) d0 = depth[nd.a];
            if (two) d0 = std::max(d0, depth[nd.b]);
            const double w = nd.op == OP_CONST ? 0.0 : Builder::op_weight(nd);
            depth[i] = d0 + w;
            work += w; cp = std::max(cp, depth[i]);
        }
        plan.dag_work = work; plan.dag_critical_path = cp;
    }
    plan.consts = b.consts;
    // LOAD_CHAL immediates already refer to squeeze order
    return 0;
}

// =============================================================================== device upload
template <class T> static int upload_vec(const std::vector<T>& v, T*& d) {
    size_t bytes = std::max<size_t>(v.size(), 1) * sizeof(T);
    H2V_HIP_CHECK(hipMalloc(&d, bytes));
    if (!v.empty()) H2V_HIP_CHECK(hipMemcpy(d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    return 0;
}
int PlanDevice::upload() {
    int rc;
    if ((rc = upload_vec(host.code, code))) return rc;
    for (int k = 0; k < 3; ++k) for (int q = 0; q < k + 2; ++q) if ((rc = upload_vec(host.code_k[k][q], code_k[k][q]))) return rc;
    if ((rc = upload_vec(host.consts, consts))) return rc;
    if ((rc = upload_vec(host.stream, stream))) return rc;
    if ((rc = upload_vec(host.squeeze_at, squeeze_at))) return rc;
    if ((rc = upload_vec(host.point_offsets, point_offsets))) return rc;
    if ((rc = upload_vec(host.scalar_offsets, scalar_offsets))) return rc;
    if ((rc = upload_vec(host.shared_bases, shared_bases))) return rc;
    std::vector<G1A> phi;
    for (const G1A& b : host.shared_bases) phi.push_back(g1_phi(b));
    if ((rc = upload_vec(phi, shared_phi))) return rc;
    return 0;
}
void PlanDevice::release() {
    hipFree(code); for (int k = 0; k < 3; ++k) for (int q = 0; q < FRVM_MAX_STREAMS; ++q) hipFree(code_k[k][q]); hipFree(consts); hipFree(stream); hipFree(squeeze_at); hipFree(point_offsets); hipFree(scalar_offsets); hipFree(shared_bases); hipFree(shared_phi);
    code = nullptr; consts = nullptr; stream = nullptr; squeeze_at = nullptr; point_offsets = nullptr; scalar_offsets = nullptr; shared_bases = nullptr; shared_phi = nullptr;
}

int ctx_load_vk(h2v_ctx* ctx, const uint8_t* vk, size_t vk_len, int vk_format) {
    VkDevice* v = new VkDevice();
    std::string err;
    if (!vk_from_bytes(vk, vk_len, vk_format, v->vk, err)) { set_last_error("VerifyingKey: " + err); delete v; return H2V_ERR_FORMAT; }
    if (v->vk.k != ctx->params.k) { set_last_error("VerifyingKey: k differs from ParamsKZG.k"); delete v; return H2V_ERR_BAD_ARGUMENT; }
    ctx->vk = v;
    return 0;
}
size_t ctx_total_instance_columns(const h2v_ctx* ctx) { return ctx->vk ? (size_t)ctx->circuit_instances * ctx->vk->vk.num_instance_columns : 0; }
void ctx_release_vk(h2v_ctx* ctx) {
    if (!ctx->vk) return;
    for (auto& kv : ctx->vk->plans) { kv.second->release(); delete kv.second; }
    delete ctx->vk;
    ctx->vk = nullptr;
}
int ctx_get_plan(h2v_ctx* ctx, const std::vector<size_t>& col_lens_in, PlanDevice** out, bool guard_terms) {
    if (!ctx->vk) { set_last_error("the context was created without a VerifyingKey"); return H2V_ERR_BAD_ARGUMENT; }
    std::lock_guard<std::mutex> lock(ctx->vk->mu);
    // the variant h2v_guard_msm runs (per-term Guard scalars; differs from the normal plan only for GWC) is cached under the
    // column lengths followed by a marker no real length can equal
    guard_terms = guard_terms && ctx->multiopen == H2V_MULTIOPEN_GWC;
    std::vector<size_t> key = col_lens_in;
    if (guard_terms) key.push_back((size_t)-1);
    const std::vector<size_t>& col_lens = col_lens_in;
    VkDevice& vd = *ctx->vk;
    auto it = vd.plans.find(key);
    if (it != vd.plans.end()) { it->second->last_use = ++vd.clock; ++it->second->pins; *out = it->second; return 0; }
    PlanDevice* pd = new PlanDevice();
    std::string err;
    PlanOptions po; po.multiopen = ctx->multiopen; po.transcript = ctx->transcript; po.circuit_instances = ctx->circuit_instances; po.guard_terms = guard_terms; po.instance_kernel_threshold = ctx->instance_kernel_threshold;
    int rc = compile_plan(ctx->vk->vk, ctx->params, col_lens, po, pd->host, err);
    if (rc) { set_last_error("plan: " + err); delete pd; return rc; }
    H2V_HIP_CHECK(hipSetDevice(ctx->device));
    rc = pd->upload();
    if (rc) { pd->release(); delete pd; return rc; }
    pd->last_use = ++vd.clock; pd->pins = 1;
    vd.plans[key] = pd;
    // bounded cache: release the least recently used plans nobody holds (never the one just made: it is pinned)
    while (vd.plans.size() > H2V_MAX_CACHED_PLANS) {
        auto victim = vd.plans.end();
        for (auto jt = vd.plans.begin(); jt != vd.plans.end(); ++jt)
            if (jt->second->pins == 0 && (victim == vd.plans.end() || jt->second->last_use < victim->second->last_use)) victim = jt;
        if (victim == vd.plans.end()) break;   // everything is in use: the cache may exceed its bound while that lasts
        victim->second->release(); delete victim->second;
        vd.plans.erase(victim);
    }
    *out = pd;
    return 0;
}
void ctx_put_plan(h2v_ctx* ctx, PlanDevice* pd) {
    if (!pd || !ctx->vk) return;
    std::lock_guard<std::mutex> lock(ctx->vk->mu);
    if (pd->pins > 0) --pd->pins;
}

}  // namespace h2v
