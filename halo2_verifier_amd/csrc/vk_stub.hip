#include "../../include/h2v.h"
#include "ctx.h"
namespace h2v {
int ctx_load_vk(h2v_ctx*, const uint8_t*, size_t, int) { set_last_error("VK support not built"); return H2V_ERR_UNSUPPORTED; }
void ctx_release_vk(h2v_ctx*) {}
}
// temporary stubs until batch.hip lands
extern "C" {
int h2v_ctx_proof_shape(const h2v_ctx*, size_t*, size_t*, size_t*, size_t*, size_t*) { return H2V_ERR_UNSUPPORTED; }
int h2v_verify_batch(h2v_ctx*, size_t, const uint8_t* const*, const size_t*, const uint8_t* const*, size_t, const size_t*, const uint8_t*, int*, int*, uint8_t*, uint8_t*) { return H2V_ERR_UNSUPPORTED; }
int h2v_verify_each(h2v_ctx*, size_t, const uint8_t* const*, const size_t*, const uint8_t* const*, size_t, const size_t*, int*) { return H2V_ERR_UNSUPPORTED; }
int h2v_guard_msm(h2v_ctx*, const uint8_t*, size_t, const uint8_t*, size_t, const size_t*, uint8_t*, uint8_t*, size_t*, uint8_t*, uint8_t*, size_t*, uint8_t*, size_t*) { return H2V_ERR_UNSUPPORTED; }
int h2v_batch_create(h2v_ctx*, size_t, size_t, h2v_batch**) { return H2V_ERR_UNSUPPORTED; }
void h2v_batch_destroy(h2v_batch*) {}
int h2v_batch_upload(h2v_batch*, size_t, const uint8_t*, size_t, const uint8_t*, size_t, const size_t*, const uint8_t*, size_t) { return H2V_ERR_UNSUPPORTED; }
int h2v_batch_launch(h2v_batch*, int) { return H2V_ERR_UNSUPPORTED; }
int h2v_batch_finish(h2v_batch*, int*, int*, uint8_t*, uint8_t*) { return H2V_ERR_UNSUPPORTED; }
int h2v_batch_accumulators(h2v_batch*, void**, size_t*) { return H2V_ERR_UNSUPPORTED; }
void* h2v_batch_stream(h2v_batch*) { return nullptr; }
int h2v_fold_check(h2v_ctx*, const void*, size_t, int*, uint8_t*, uint8_t*) { return H2V_ERR_UNSUPPORTED; }
int h2v_batch_timings(h2v_batch*, float*, int) { return H2V_ERR_UNSUPPORTED; }
int h2v_batch_set_profiling(h2v_batch*, int) { return H2V_ERR_UNSUPPORTED; }
}
