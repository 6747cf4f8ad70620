#pragma once
#include <mutex>
#include "internal.h"
#include "pairing.hip.h"

namespace h2v {

// Host view of ParamsKZG (poly/kzg/commitment.rs:22-29)
struct ParamsHost {
    uint32_t k = 0;
    G1A g;
    G2A g2, s_g2;
};
// Parse ParamsKZG::write_custom bytes (commitment.rs:142-152); false + message on a rejected encoding
bool params_from_bytes(const uint8_t* data, size_t len, int format, ParamsHost& out, std::string& err);

PairingConsts pairing_consts_host();
std::vector<uint32_t> pairing_program(bool merged_lines);
std::vector<uint32_t> pairing_program2();   // two operation streams per check: [step][2]
#define H2V_PAIR2_MAX_STEPS 352             // steps k_pairing2 holds in LDS (the table has 330)

#define H2V_PAIRING_LINES 102   // 64 doublings + popcount(ATE_LOW) = 36 additions + 2 Frobenius corrections
#define H2V_PAIRING_LINE_WS_BYTES ((size_t)66 * 6 * sizeof(Fq2))   // per check: k_pair_lines' output, one Fq12 per Miller iteration (+ 2 corrections)

struct PairingDevice {
    LineCoeff* l_sg2 = nullptr;  // line coefficients for s_g2
    LineCoeff* l_ng2 = nullptr;  // line coefficients for -g2
    // line coefficients of the multiples 2^(shift j) s_g2, -2^(shift j) g2, j < parts (for checks over split accumulators): made on
    // the host on first use of a (shift, parts) pair, kept for the life of the context.  Row 2 j + side, H2V_PAIRING_LINES entries each.
    struct SplitTable { uint32_t shift, parts; LineCoeff* lines; };
    std::vector<SplitTable> split;
    std::mutex split_mu;
    G2A h_sg2, h_ng2;            // host copies for those tables
    int split_lines(uint32_t shift, uint32_t parts, const LineCoeff** out);
    PairingConsts* consts = nullptr;
    uint32_t* prog = nullptr;    // the pairing's operation table (pairing.hip: pairing_program)
    uint32_t n_ops = 0;
    uint32_t* prog_merged = nullptr;   // the same with one line product per Miller iteration (checks over split accumulators)
    uint32_t n_ops_merged = 0;
    uint32_t* prog2 = nullptr;         // two streams per check (k_pairing2): uint2 per step
    uint32_t n_steps2 = 0;
    int upload(const ParamsHost& p);
    void release();
};

int pairing_check_enqueue(hipStream_t s, const PairingDevice& pd, const G1J* d_pairs, uint32_t n, uint32_t* d_ok);
// check g over split accumulators: left = sum_j 2^(shift j) piece[(2 g) parts + j], right likewise at 2 g + 1 (MsmSplit), the
// pieces given line-ready as (X Z, Y, Z^3); d_line_ws: n * H2V_PAIRING_LINE_WS_BYTES of device scratch owned by the caller
int pairing_check_split_enqueue(hipStream_t s, PairingDevice& pd, const G1JSlot* d_ready, uint32_t n, uint32_t parts, uint32_t shift, void* d_line_ws, uint32_t* d_ok, bool one_stream = false);

}  // namespace h2v
