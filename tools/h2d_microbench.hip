// What does a host -> device copy of a batch's inputs cost, by how it is issued?  (h2v_batch_upload_launch, round 3.)
// 21 MB (20 x 1024 proofs of 1 KB), from malloc'ed (pageable) memory, as the C ABI receives it.
// Build: hipcc -O3 --offload-arch=gfx950 tools/h2d_microbench.hip -o tools/h2d_microbench
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
__global__ void spin(unsigned long long cycles, int* sink) { unsigned long long t0 = clock64(); while (clock64() - t0 < cycles) {} if (sink && threadIdx.x == 9999) *sink = 1; }
int main() {
    const size_t N = 20 * 1024 * 1024 + 5 * 1024 * 1024;
    char* h = (char*)malloc(N); memset(h, 3, N);
    char* d; (void)hipMalloc(&d, N);
    char* pinned; (void)hipHostMalloc(&pinned, N, hipHostMallocDefault);
    hipStream_t s1, s2; (void)hipStreamCreateWithFlags(&s1, hipStreamNonBlocking); (void)hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
    hipEvent_t ev[16]; for (auto& e : ev) (void)hipEventCreateWithFlags(&e, hipEventDisableTiming);
    auto rep = [&](const char* name, auto f) { f(); (void)hipDeviceSynchronize(); double best = 1e9, sum = 0; for (int i = 0; i < 7; ++i) { double t = now(); f(); (void)hipDeviceSynchronize(); t = now() - t; best = t < best ? t : best; sum += t; } printf("%-78s best %.3f ms  mean %.3f ms\n", name, best, sum / 7); };
    rep("one hipMemcpyAsync from pageable memory + stream sync", [&] { (void)hipMemcpyAsync(d, h, N, hipMemcpyHostToDevice, s1); (void)hipStreamSynchronize(s1); });
    rep("8 chunks hipMemcpyAsync from pageable memory, one stream", [&] { for (int c = 0; c < 8; ++c) (void)hipMemcpyAsync(d + c * (N / 8), h + c * (N / 8), N / 8, hipMemcpyHostToDevice, s1); (void)hipStreamSynchronize(s1); });
    rep("hipHostRegister + hipHostUnregister only", [&] { (void)hipHostRegister(h, N, hipHostRegisterDefault); (void)hipHostUnregister(h); });
    rep("hipHostRegister + one async copy + sync + unregister", [&] { (void)hipHostRegister(h, N, hipHostRegisterDefault); (void)hipMemcpyAsync(d, h, N, hipMemcpyHostToDevice, s1); (void)hipStreamSynchronize(s1); (void)hipHostUnregister(h); });
    rep("one async copy from hipHostMalloc'ed memory + sync", [&] { (void)hipMemcpyAsync(d, pinned, N, hipMemcpyHostToDevice, s1); (void)hipStreamSynchronize(s1); });
    rep("8 chunks from hipHostMalloc'ed memory, one stream + sync", [&] { for (int c = 0; c < 8; ++c) (void)hipMemcpyAsync(d + c * (N / 8), pinned + c * (N / 8), N / 8, hipMemcpyHostToDevice, s1); (void)hipStreamSynchronize(s1); });
    rep("memcpy pageable -> pinned staging (CPU, one thread)", [&] { memcpy(pinned, h, N); });
    rep("8 chunks pinned on copy stream, event each, other stream waits + 70 us kernel each", [&] {
        for (int c = 0; c < 8; ++c) { (void)hipMemcpyAsync(d + c * (N / 8), pinned + c * (N / 8), N / 8, hipMemcpyHostToDevice, s2); (void)hipEventRecord(ev[c], s2); (void)hipStreamWaitEvent(s1, ev[c], 0); hipLaunchKernelGGL(spin, dim3(256), dim3(64), 0, s1, 70 * 100ull, (int*)nullptr); }
        (void)hipStreamSynchronize(s1); });
    rep("8 x 70 us kernels alone", [&] { for (int c = 0; c < 8; ++c) hipLaunchKernelGGL(spin, dim3(256), dim3(64), 0, s1, 70 * 100ull, (int*)nullptr); (void)hipStreamSynchronize(s1); });
    rep("kernel reading 21 MB straight from pinned host memory (zero copy), 4096 x 256 lanes", [&] { (void)hipMemcpyAsync(d, pinned, 64, hipMemcpyHostToDevice, s1); (void)hipStreamSynchronize(s1); });
    // strided (2D) copies: the point bytes of a proof are 384 of its 1024 bytes (h2v_batch_upload_launch copies them first)
    const size_t rows = 20 * 1024, pitch = 1024, P21 = rows * pitch;
    rep("2D copy, 384 of every 1024 B x 20480 rows (7.9 MB), pageable + sync", [&] { (void)hipMemcpy2DAsync(d, pitch, h, pitch, 384, rows, hipMemcpyHostToDevice, s1); (void)hipStreamSynchronize(s1); });
    rep("2D copy, 640 of every 1024 B x 20480 rows (13.1 MB), pageable + sync", [&] { (void)hipMemcpy2DAsync(d + 384, pitch, h + 384, pitch, 640, rows, hipMemcpyHostToDevice, s1); (void)hipStreamSynchronize(s1); });
    rep("2D copy, 3 runs of 128 B of every 1024 B (7.9 MB), pageable + sync", [&] { for (int r = 0; r < 3; ++r) (void)hipMemcpy2DAsync(d + 256 * r, pitch, h + 256 * r, pitch, 128, rows, hipMemcpyHostToDevice, s1); (void)hipStreamSynchronize(s1); });
    rep("2D copy, 384 of every 1024 B from hipHostMalloc'ed memory + sync", [&] { (void)hipMemcpy2DAsync(d, pitch, pinned, pitch, 384, rows, hipMemcpyHostToDevice, s1); (void)hipStreamSynchronize(s1); });
    rep("21 MB contiguous + 5 MB contiguous, pageable + sync", [&] { (void)hipMemcpyAsync(d, h, P21, hipMemcpyHostToDevice, s1); (void)hipMemcpyAsync(d + P21, h + P21, N - P21, hipMemcpyHostToDevice, s1); (void)hipStreamSynchronize(s1); });
    // a stream held by hipStreamWaitValue32 on a flag the host sets (would let the later stages be enqueued before the copy they need has returned)
    {
        int can = 0; (void)hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0);
        printf("hipDeviceAttributeCanUseStreamWaitValue = %d\n", can);
        if (can) {
            uint32_t* flag = nullptr;
            if (hipExtMallocWithFlags((void**)&flag, 64, hipMallocSignalMemory) == hipSuccess && flag) {
                rep("hipStreamWaitValue32 (flag already set) + 70 us kernel + sync", [&] { *flag = 1; (void)hipStreamWaitValue32(s1, flag, 1, hipStreamWaitValueEq, 0xffffffffu); hipLaunchKernelGGL(spin, dim3(256), dim3(64), 0, s1, 70 * 100ull, (int*)nullptr); (void)hipStreamSynchronize(s1); });
                rep("hipStreamWaitValue32, flag set by the host after a 21 MB blocking copy on another stream, + 70 us kernel", [&] {
                    *flag = 0; (void)hipStreamWaitValue32(s1, flag, 1, hipStreamWaitValueEq, 0xffffffffu); hipLaunchKernelGGL(spin, dim3(256), dim3(64), 0, s1, 70 * 100ull, (int*)nullptr);
                    (void)hipMemcpyAsync(d, h, P21, hipMemcpyHostToDevice, s2); (void)hipStreamSynchronize(s2); *flag = 1; (void)hipStreamSynchronize(s1); });
                (void)hipFree(flag);
            } else printf("hipExtMallocWithFlags(hipMallocSignalMemory) failed\n");
        }
    }
    return 0;
}
