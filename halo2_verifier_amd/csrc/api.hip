// C-ABI entry points (include/h2v.h): context, standalone MSM and pairing check.
// The batch-verification entry points live in batch.hip.
#include "../../include/h2v.h"
#include "ctx.h"
#include <mutex>
#include <string.h>

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

int h2v_ctx_create_ex(const uint8_t* params, size_t params_len, int params_format, const uint8_t* vk, size_t vk_len, int vk_format,
                      int device, const h2v_options* options, h2v_ctx** out) {
    if (!params || !out) { set_last_error("h2v_ctx_create: null argument"); return H2V_ERR_BAD_ARGUMENT; }
    if (options && (options->multiopen < 0 || options->multiopen > 1 || options->transcript < 0 || options->transcript > 1)) {
        set_last_error("h2v_ctx_create_ex: unknown multiopen / transcript option"); return H2V_ERR_BAD_ARGUMENT;
    }
    int ndev = h2v_device_count();
    if (device < 0 || device >= ndev) { set_last_error("h2v_ctx_create: no such HIP device (the library has no CPU path)"); return H2V_ERR_DEVICE; }
    h2v_ctx* ctx = new h2v_ctx();
    std::string err;
    if (!params_from_bytes(params, params_len, params_format, ctx->params, err)) { set_last_error("ParamsKZG: " + err); delete ctx; return H2V_ERR_FORMAT; }
    ctx->device = device;
    if (options) { ctx->multiopen = options->multiopen; ctx->transcript = options->transcript; }
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
    uint32_t nn = (uint32_t)n;
    if (nn > ctx->msm_ws.cap_terms) { int rc = ctx->msm_ws.alloc(nn < 1024 ? 1024 : nn, 1); if (rc) return rc; }
    uint8_t *d_sb = nullptr, *d_bb = nullptr, *d_out = nullptr; uint32_t *d_s = nullptr, *d_flags = nullptr; G1A* d_b = nullptr; G1J* d_res = nullptr;
    size_t n1 = n ? n : 1;
    H2V_HIP_CHECK(hipMalloc(&d_sb, 32 * n1)); H2V_HIP_CHECK(hipMalloc(&d_bb, 64 * n1));
    H2V_HIP_CHECK(hipMalloc(&d_s, 32 * n1)); H2V_HIP_CHECK(hipMalloc(&d_b, sizeof(G1A) * n1));
    H2V_HIP_CHECK(hipMalloc(&d_flags, 8 * n1 + 8)); H2V_HIP_CHECK(hipMalloc(&d_res, sizeof(G1J))); H2V_HIP_CHECK(hipMalloc(&d_out, 64));
    int rc = 0;
    std::vector<uint32_t> flags(2 * n1 + 1);
    do {
        if (n) {
            if (hipMemcpyAsync(d_sb, scalars32, 32 * n, hipMemcpyHostToDevice, s) != hipSuccess || hipMemcpyAsync(d_bb, bases64, 64 * n, hipMemcpyHostToDevice, s) != hipSuccess) { rc = H2V_ERR_DEVICE; break; }
            if ((rc = scalars_from_bytes_enqueue(s, d_sb, d_s, d_flags, nn))) break;
            if ((rc = bases_from_bytes_enqueue(s, d_bb, d_b, d_flags + n, nn))) break;
        }
        if ((rc = msm_enqueue(s, ctx->msm_ws, d_s, d_b, nn, d_res))) break;
        if ((rc = point_to_bytes_enqueue(s, d_res, d_out, d_flags + 2 * n, 1))) break;
        if (hipMemcpyAsync(flags.data(), d_flags, 4 * (2 * n + 1), hipMemcpyDeviceToHost, s) != hipSuccess || hipMemcpyAsync(out_xy, d_out, 64, hipMemcpyDeviceToHost, s) != hipSuccess) { rc = H2V_ERR_DEVICE; break; }
        if (hipStreamSynchronize(s) != hipSuccess) { set_last_error(std::string("h2v_msm_g1: ") + hipGetErrorString(hipGetLastError())); rc = H2V_ERR_DEVICE; break; }
        for (size_t i = 0; i < 2 * n; ++i) if (flags[i]) { set_last_error(i < n ? "h2v_msm_g1: scalar not canonical" : "h2v_msm_g1: base not on the curve"); rc = H2V_ERR_BAD_ARGUMENT; break; }
        if (out_is_identity) *out_is_identity = (int)flags[2 * n];
    } while (0);
    hipFree(d_sb); hipFree(d_bb); hipFree(d_s); hipFree(d_b); hipFree(d_flags); hipFree(d_res); hipFree(d_out);
    return rc;
}

int h2v_pairing_check(h2v_ctx* ctx, const uint8_t left_xy[64], const uint8_t right_xy[64], int* ok) {
    if (!ctx || !left_xy || !right_xy || !ok) { set_last_error("h2v_pairing_check: null argument"); return H2V_ERR_BAD_ARGUMENT; }
    std::lock_guard<std::mutex> lock(ctx->mu);
    H2V_HIP_CHECK(hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    uint8_t* d_bytes = nullptr; G1A* d_aff = nullptr; uint32_t* d_flags = nullptr; G1J* d_pairs = nullptr;
    H2V_HIP_CHECK(hipMalloc(&d_bytes, 128)); H2V_HIP_CHECK(hipMalloc(&d_aff, 2 * sizeof(G1A)));
    H2V_HIP_CHECK(hipMalloc(&d_flags, 16)); H2V_HIP_CHECK(hipMalloc(&d_pairs, 2 * sizeof(G1J)));
    int rc = 0; uint32_t flags[3] = {0, 0, 0};
    do {
        uint8_t host[128]; memcpy(host, left_xy, 64); memcpy(host + 64, right_xy, 64);
        if (hipMemcpyAsync(d_bytes, host, 128, hipMemcpyHostToDevice, s) != hipSuccess) { rc = H2V_ERR_DEVICE; break; }
        if ((rc = bases_from_bytes_enqueue(s, d_bytes, d_aff, d_flags, 2))) break;
        if ((rc = affine_to_jacobian_enqueue(s, d_aff, d_pairs, 2))) break;
        if ((rc = pairing_check_enqueue(s, ctx->pairing, d_pairs, 1, d_flags + 2))) break;
        if (hipMemcpyAsync(flags, d_flags, 12, hipMemcpyDeviceToHost, s) != hipSuccess) { rc = H2V_ERR_DEVICE; break; }
        if (hipStreamSynchronize(s) != hipSuccess) { set_last_error(std::string("h2v_pairing_check: ") + hipGetErrorString(hipGetLastError())); rc = H2V_ERR_DEVICE; break; }
        if (flags[0] || flags[1]) { set_last_error("h2v_pairing_check: point not on the curve"); rc = H2V_ERR_BAD_ARGUMENT; break; }
        *ok = (int)flags[2];
    } while (0);
    hipFree(d_bytes); hipFree(d_aff); hipFree(d_flags); hipFree(d_pairs);
    return rc;
}

}  // extern "C"
