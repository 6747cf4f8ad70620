// Final pairing check kernel:  e(left, s_g2) * e(right, -g2) == 1  (DualMSM::check,
// poly/kzg/msm.rs:185-203).  One lane per independent check: the batch path launches a single
// check per batch (AccumulatorStrategy), the per-proof path (SingleStrategy) one per proof.
// The G2 side is constant per context, so its Miller-loop line coefficients are precomputed
// once on the host (g2_prepare) and only evaluated at the two G1 points here.
#include "../../include/h2v.h"
#include "internal.h"
#include "pairing.cuh"
#include "pairing_api.h"

namespace h2v {

__global__ void __launch_bounds__(64) k_pairing_check(const G1J* __restrict__ pairs, uint32_t n, const LineCoeff* __restrict__ l_sg2,
                                                      const LineCoeff* __restrict__ l_ng2, const PairingConsts* __restrict__ consts,
                                                      uint32_t* __restrict__ ok) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    G1A left = g1_to_affine(pairs[2 * i]), right = g1_to_affine(pairs[2 * i + 1]);
    Fq12 f = miller_loop_2(left, l_sg2, right, l_ng2);
    ok[i] = final_exp_is_one(f, *consts) ? 1u : 0u;
}

int pairing_check_enqueue(hipStream_t s, const PairingDevice& pd, const G1J* d_pairs, uint32_t n, uint32_t* d_ok) {
    if (!n) return 0;
    hipLaunchKernelGGL(k_pairing_check, dim3((n + 63) / 64), dim3(64), 0, s, d_pairs, n, pd.l_sg2, pd.l_ng2, pd.consts, d_ok);
    H2V_HIP_CHECK(hipGetLastError());
    return 0;
}

}  // namespace h2v
