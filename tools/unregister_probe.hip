// does hipHostUnregister wait for unrelated kernels in flight?  and how fast is a zero-copy gather of 384-byte runs?
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
__global__ void spin(unsigned long long cycles) { unsigned long long t0 = wall_clock64(); while (wall_clock64() - t0 < cycles) {} }
__global__ void gather(const uint4* __restrict__ src, uint4* __restrict__ dst, unsigned n, unsigned row16, unsigned run16) {
    const unsigned t = blockIdx.x * blockDim.x + threadIdx.x, p = t / run16, j = t % run16;
    if (p < n) dst[(size_t)p * row16 + j] = src[(size_t)p * row16 + j];
}
int main() {
    const size_t N = 20 * 1024 * 1024;
    char* h = (char*)malloc(N); memset(h, 3, N);
    char* d; (void)hipMalloc(&d, N);
    hipStream_t s1, s2; (void)hipStreamCreateWithFlags(&s1, hipStreamNonBlocking); (void)hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
    hipEvent_t ev; (void)hipEventCreateWithFlags(&ev, hipEventDisableTiming);
    for (int it = 0; it < 3; ++it) {
        double t0 = now();
        hipError_t e = hipHostRegister(h, N, hipHostRegisterDefault);
        void* dv = nullptr; hipError_t e2 = hipHostGetDevicePointer(&dv, h, 0);
        double t1 = now();
        hipLaunchKernelGGL(gather, dim3(20480 * 24 / 256), dim3(256), 0, s1, (const uint4*)dv, (uint4*)d, 20480u, 64u, 24u);
        (void)hipEventRecord(ev, s1);
        hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s1, 300000ull);   // ~3 ms at 100 MHz
        double t2 = now();
        (void)hipEventSynchronize(ev);
        double t3 = now();
        hipError_t e3 = hipHostUnregister(h);
        double t4 = now();
        (void)hipStreamSynchronize(s1);
        double t5 = now();
        printf("register %d/%d %.3f ms | enqueue %.3f | wait gather %.3f | unregister %d %.3f ms | rest of stream %.3f ms\n", (int)e, (int)e2, t1 - t0, t2 - t1, t3 - t2, (int)e3, t4 - t3, t5 - t4);
    }
    return 0;
}
