#pragma once
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
std::vector<uint32_t> pairing_program();

struct PairingDevice {
    LineCoeff* l_sg2 = nullptr;  // line coefficients for s_g2
    LineCoeff* l_ng2 = nullptr;  // line coefficients for -g2
    PairingConsts* consts = nullptr;
    uint32_t* prog = nullptr;    // the pairing's operation table (pairing.hip: pairing_program)
    uint32_t n_ops = 0;
    int upload(const ParamsHost& p);
    void release();
};

int pairing_check_enqueue(hipStream_t s, const PairingDevice& pd, const G1J* d_pairs, uint32_t n, uint32_t* d_ok);

}  // namespace h2v
