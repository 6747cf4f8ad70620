// C-ABI entry points (include/h2v.h): context, standalone MSM and pairing check.
// The batch-verification entry points live in batch.hip.
#include "../../include/h2v.h"
#include "ctx.h"
#include <mutex>
#include <string.h>
#include <stddef.h>

namespace h2v {
static thread_local std::string g_last_error;
void set_last_error(const std::string& s) { g_last_error = s; }
}  // namespace h2v

using namespace h2v;

extern "C" {

int h2v_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}
const char* h2v_last_error(void) { return g_last_error.c_str(); }

int h2v_ctx_create(const uint8_t* params, size_t params_len, int params_format, const uint8_t* vk, size_t vk_len, int vk_format,
                   int device, h2v_ctx** out) {
    return h2v_ctx_create_ex(params, params_len, params_format, vk, vk_len, vk_format, device, nullptr, out);
}

// h2v_options as the caller's header declares it -> the fields this library knows.  struct_size is the guard the review of round 2
// asked for: the struct grew a field without one, and a caller built against the older layout passed whatever followed it in memory.
static int read_options(const h2v_options* o, h2v_options& out) {
    out = h2v_options H2V_OPTIONS_INIT;
    if (!o) return 0;
    const size_t full = sizeof(h2v_options), v3 = offsetof(h2v_options, instance_kernel_threshold);   // layouts this library knows: up to circuit_instances, or all of it
    if (o->struct_size != full && o->struct_size != v3) {
        set_last_error("h2v_ctx_create_ex: h2v_options.struct_size is not a layout this library knows (built against another revision of h2v.h? H2V_ABI_VERSION " + std::to_string(H2V_ABI_VERSION) + ")");
        return H2V_ERR_BAD_ARGUMENT;
    }
    memcpy(&out, o, o->struct_size);
    if (o->struct_size < full) out.instance_kernel_threshold = 0;
    out.struct_size = full;
    return 0;
}

int h2v_abi_version(void) { return H2V_ABI_VERSION; }

int h2v_ctx_set_tuning(h2v_ctx* ctx, const h2v_tuning* t) {
    if (!ctx) { set_last_error("h2v_ctx_set_tuning: null context"); return H2V_ERR_BAD_ARGUMENT; }
    Tuning n;
    if (t) {
        if (t->struct_size != sizeof(h2v_tuning)) { set_last_error("h2v_ctx_set_tuning: h2v_tuning.struct_size does not match this library"); return H2V_ERR_BAD_ARGUMENT; }
        const int vals[] = {t->frvm_streams, t->frvm_lds_kb, t->msm_parts, t->msm_global_sort, t->msm_no_term_split, t->msm_window_threads, t->msm_window_wpw, t->msm_window_slots, t->msm_acc_waves, t->pairing_one_stream, t->upload_mode};
        for (int v : vals) if (v < 0) { set_last_error("h2v_ctx_set_tuning: negative field"); return H2V_ERR_BAD_ARGUMENT; }
        if (t->frvm_streams > 4 || t->msm_parts > MSM_MAX_PARTS || (t->msm_window_threads && t->msm_window_threads != 64 && t->msm_window_threads != 128 && t->msm_window_threads != 256) ||
            (t->msm_window_wpw && t->msm_window_wpw != 1 && t->msm_window_wpw != 2 && t->msm_window_wpw != 4) || (t->msm_window_slots && t->msm_window_slots != 3 && t->msm_window_slots != 5) || (t->msm_acc_waves && t->msm_acc_waves != 3 && t->msm_acc_waves != 4) || t->upload_mode > 3) {
            set_last_error("h2v_ctx_set_tuning: value out of range (see h2v.h)"); return H2V_ERR_BAD_ARGUMENT;
        }
        n.frvm_streams = t->frvm_streams; n.frvm_lds_kb = t->frvm_lds_kb; n.msm_parts = t->msm_parts; n.msm_global_sort = t->msm_global_sort; n.msm_no_term_split = t->msm_no_term_split;
        n.msm_window_threads = t->msm_window_threads; n.msm_window_wpw = t->msm_window_wpw; n.msm_window_slots = t->msm_window_slots; n.msm_acc_waves = t->msm_acc_waves; n.pairing_one_stream = t->pairing_one_stream; n.upload_mode = t->upload_mode;
    }
    std::lock_guard<std::mutex> lock(ctx->mu);
    ctx->tuning = n;
    return 0;
}

int h2v_ctx_create_ex(const uint8_t* params, size_t params_len, int params_format, const uint8_t* vk, size_t vk_len, int vk_format,
                      int device, const h2v_options* options_in, h2v_ctx** out) {
    if (!params || !out) { set_last_error("h2v_ctx_create: null argument"); return H2V_ERR_BAD_ARGUMENT; }
    h2v_options opt_v;
    { int rco = read_options(options_in, opt_v); if (rco) return rco; }
    const h2v_options* options = &opt_v;
    if (options->multiopen < 0 || options->multiopen > 1 || options->transcript < 0 || options->transcript > 1) {
        set_last_error("h2v_ctx_create_ex: unknown multiopen / transcript option"); return H2V_ERR_BAD_ARGUMENT;
    }
    if (options->circuit_instances < 0 || options->circuit_instances > 64) { set_last_error("h2v_ctx_create_ex: circuit_instances must be in 0..64 (0 = 1)"); return H2V_ERR_BAD_ARGUMENT; }
    if (options->instance_kernel_threshold < 0) { set_last_error("h2v_ctx_create_ex: instance_kernel_threshold must be >= 0"); return H2V_ERR_BAD_ARGUMENT; }
    int ndev = h2v_device_count();
    if (device < 0 || device >= ndev) { set_last_error("h2v_ctx_create: no such HIP device (the library has no CPU path)"); return H2V_ERR_DEVICE; }
    h2v_ctx* ctx = new h2v_ctx();
    std::string err;
    if (!params_from_bytes(params, params_len, params_format, ctx->params, err)) { set_last_error("ParamsKZG: " + err); delete ctx; return H2V_ERR_FORMAT; }
    ctx->device = device;
    ctx->multiopen = options->multiopen; ctx->transcript = options->transcript; ctx->circuit_instances = options->circuit_instances > 0 ? options->circuit_instances : 1;
    ctx->instance_kernel_threshold = options->instance_kernel_threshold;
    if (hipSetDevice(device) != hipSuccess) { set_last_error("hipSetDevice failed"); delete ctx; return H2V_ERR_DEVICE; }
    int rc = ctx->pairing.upload(ctx->params);
    if (rc) { delete ctx; return rc; }
    if (vk && vk_len) {
        rc = ctx_load_vk(ctx, vk, vk_len, vk_format);
        if (rc) { h2v_ctx_destroy(ctx); return rc; }
    }
    if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) { set_last_error("hipStreamCreate failed"); h2v_ctx_destroy(ctx); return H2V_ERR_DEVICE; }
    *out = ctx;
    return H2V_OK;
}

void h2v_ctx_destroy(h2v_ctx* ctx) {
    if (!ctx) return;
    hipSetDevice(ctx->device);
    if (ctx->scratch_batch) { h2v_batch_destroy(ctx->scratch_batch); ctx->scratch_batch = nullptr; }
    ctx->pairing.release();
    ctx->msm_ws.release();
    ctx_release_vk(ctx);
    if (ctx->stream) hipStreamDestroy(ctx->stream);
    delete ctx;
}

int h2v_msm_g1(h2v_ctx* ctx, const uint8_t* scalars32, const uint8_t* bases64, size_t n, uint8_t out_xy[64], int* out_is_identity) {
    if (!ctx || !out_xy || (n && (!scalars32 || !bases64))) { set_last_error("h2v_msm_g1: null argument"); return H2V_ERR_BAD_ARGUMENT; }
    if (n > (1u << 26)) { set_last_error("h2v_msm_g1: n too large"); return H2V_ERR_BAD_ARGUMENT; }
    std::lock_guard<std::mutex> lock(ctx->mu);
    H2V_HIP_CHECK(hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    const uint32_t nn = (uint32_t)n;
    int rc;
    // staging buffers and the Pippenger workspace are kept in the context and only grow (seven hipMalloc / hipFree pairs per
    // call before); they are owned by the context, so no return path below can leak them
    OneShotMsm& w = ctx->one_shot;
    if (nn > w.cap || !w.res.p) {
        const size_t cap = nn < 1024 ? 1024 : nn;
        w.cap = 0;
        if ((rc = w.sb.alloc(32 * cap)) || (rc = w.bb.alloc(64 * cap)) || (rc = w.s.alloc(8 * cap)) || (rc = w.b.alloc(cap)) || (rc = w.flags.alloc(2 * cap + 2)) ||
            (rc = w.res.alloc(1)) || (rc = w.out.alloc(64))) return rc;
        if ((rc = ctx->msm_ws.alloc((uint32_t)cap, 1))) return rc;
        w.cap = (uint32_t)cap;
    }
    std::vector<uint32_t> flags(2 * n + 1);
    if (n) {
        H2V_HIP_CHECK(hipMemcpyAsync(w.sb.p, scalars32, 32 * n, hipMemcpyHostToDevice, s));
        H2V_HIP_CHECK(hipMemcpyAsync(w.bb.p, bases64, 64 * n, hipMemcpyHostToDevice, s));
        if ((rc = scalars_from_bytes_enqueue(s, w.sb.p, w.s.p, w.flags.p, nn))) return rc;
        if ((rc = bases_from_bytes_enqueue(s, w.bb.p, w.b.p, w.flags.p + n, nn))) return rc;
    }
    ctx->msm_ws.tune = ctx->tuning;
    if ((rc = msm_enqueue(s, ctx->msm_ws, w.s.p, w.b.p, nn, w.res.p))) return rc;
    if ((rc = point_to_bytes_enqueue(s, w.res.p, w.out.p, w.flags.p + 2 * n, 1))) return rc;
    H2V_HIP_CHECK(hipMemcpyAsync(flags.data(), w.flags.p, 4 * (2 * n + 1), hipMemcpyDeviceToHost, s));
    H2V_HIP_CHECK(hipMemcpyAsync(out_xy, w.out.p, 64, hipMemcpyDeviceToHost, s));
    H2V_HIP_CHECK(hipStreamSynchronize(s));
    for (size_t i = 0; i < 2 * n; ++i) if (flags[i]) { set_last_error(i < n ? "h2v_msm_g1: scalar not canonical" : "h2v_msm_g1: base not on the curve"); return H2V_ERR_BAD_ARGUMENT; }
    if (out_is_identity) *out_is_identity = (int)flags[2 * n];
    return 0;
}

int h2v_pairing_check(h2v_ctx* ctx, const uint8_t left_xy[64], const uint8_t right_xy[64], int* ok) {
    if (!ctx || !left_xy || !right_xy || !ok) { set_last_error("h2v_pairing_check: null argument"); return H2V_ERR_BAD_ARGUMENT; }
    std::lock_guard<std::mutex> lock(ctx->mu);
    H2V_HIP_CHECK(hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    DevBuf<uint8_t> d_bytes; DevBuf<G1A> d_aff; DevBuf<uint32_t> d_flags; DevBuf<G1J> d_pairs;   // freed on every return path
    int rc;
    if ((rc = d_bytes.alloc(128)) || (rc = d_aff.alloc(2)) || (rc = d_flags.alloc(4)) || (rc = d_pairs.alloc(2))) return rc;
    uint32_t flags[3] = {0, 0, 0};
    uint8_t host[128]; memcpy(host, left_xy, 64); memcpy(host + 64, right_xy, 64);
    H2V_HIP_CHECK(hipMemcpyAsync(d_bytes.p, host, 128, hipMemcpyHostToDevice, s));
    if ((rc = bases_from_bytes_enqueue(s, d_bytes.p, d_aff.p, d_flags.p, 2))) return rc;
    if ((rc = affine_to_jacobian_enqueue(s, d_aff.p, d_pairs.p, 2))) return rc;
    if ((rc = pairing_check_enqueue(s, ctx->pairing, d_pairs.p, 1, d_flags.p + 2))) return rc;
    H2V_HIP_CHECK(hipMemcpyAsync(flags, d_flags.p, 12, hipMemcpyDeviceToHost, s));
    H2V_HIP_CHECK(hipStreamSynchronize(s));
    if (flags[0] || flags[1]) { set_last_error("h2v_pairing_check: point not on the curve"); return H2V_ERR_BAD_ARGUMENT; }
    *ok = (int)flags[2];
    return 0;
}

}  // extern "C"
