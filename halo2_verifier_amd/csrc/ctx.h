// The context object behind h2v_ctx (include/h2v.h).
#pragma once
#include "internal.h"
#include "pairing_api.h"
#include <mutex>

namespace h2v { struct VkDevice; }
struct h2v_batch;

namespace h2v {
// staging buffers of h2v_msm_g1, kept between calls (grow-only)
struct OneShotMsm {
    uint32_t cap = 0;
    DevBuf<uint8_t> sb, bb, out; DevBuf<uint32_t> s, flags; DevBuf<G1A> b; DevBuf<G1J> res;
};
}  // namespace h2v

struct h2v_ctx {
    int device = 0;
    h2v::ParamsHost params;
    h2v::PairingDevice pairing;
    h2v::MsmWorkspace msm_ws;      // used by h2v_msm_g1 only; batches own their workspaces
    h2v::OneShotMsm one_shot;
    hipStream_t stream = nullptr;  // used by the synchronous single-shot entry points
    std::mutex mu;                 // serialises the single-shot entry points
    h2v::VkDevice* vk = nullptr;   // per-VK compiled program and constants (vkplan.hip)
    int multiopen = 0, transcript = 0, circuit_instances = 1;  // h2v_options
    int instance_kernel_threshold = 0;                         // h2v_options (debug): 0 = default
    h2v::Tuning tuning;                                        // h2v_ctx_set_tuning (debug / test): forced kernel variants
    struct h2v_batch* scratch_batch = nullptr;  // kept between one-shot calls (h2v_verify_batch / _each): ~20 device allocations saved per call
};

namespace h2v {
// instance columns a proof of this context brings: circuit instances x the VK's instance columns (lib.rs:51-55)
size_t ctx_total_instance_columns(const h2v_ctx* ctx);
int ctx_load_vk(h2v_ctx* ctx, const uint8_t* vk, size_t vk_len, int vk_format);
void ctx_release_vk(h2v_ctx* ctx);
int affine_to_jacobian_enqueue(hipStream_t s, const G1A* d_in, G1J* d_out, uint32_t n);
}  // namespace h2v
