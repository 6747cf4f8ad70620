// Issue cost of the integer instructions the field arithmetic is made of, per wave, on one wave per SIMD and on four:
// cycles per instruction = time * clock / instructions.  Eight independent chains per lane, so the figure is issue rate, not latency.
// Build: hipcc -O3 --offload-arch=gfx950 tools/instr_rate_microbench.hip -o tools/instr_rate_microbench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

template <int OP> __global__ void __launch_bounds__(64) k(uint32_t* io, int iters) {
    uint32_t a[8]; uint64_t w[8]; uint64_t cy[8];
    for (int i = 0; i < 8; ++i) { a[i] = io[threadIdx.x + 64 * i]; w[i] = a[i] * 0x9e3779b97f4a7c15ull; }
    const uint32_t c = io[512] | 1u;
    for (int it = 0; it < iters; ++it) {
#define MULLO(i) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
#define MULHI(i) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
#define MAD64(i) asm volatile("v_mad_u64_u32 %0, %1, %2, %3, %0" : "+v"(w[i]), "=s"(cy[i]) : "v"(a[i]), "v"(c));
#define MAD64D(i) asm volatile("v_mad_u64_u32 %0, %1, %2, %3, %0" : "+v"(w[0]), "=s"(cy[i]) : "v"(a[i]), "v"(c));   // ONE accumulator: the dependent chain of a product column
#define SHR64(i) asm volatile("v_lshrrev_b64 %0, 1, %0" : "+v"(w[i]));
#define ADD64(i) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(w[i]) : "v"(w[(i + 1) & 7]));
#define ADD32(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
#define AND32(i) asm volatile("v_and_b32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
#define MUL24(i) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(a[i]) : "v"(c));
#define MAD24(i) asm volatile("v_mad_u32_u24 %0, %0, %1, %0" : "+v"(a[i]) : "v"(c));
        if (OP == 0) { REP8(MULLO) REP8(MULLO) REP8(MULLO) REP8(MULLO) }
        if (OP == 1) { REP8(MULHI) REP8(MULHI) REP8(MULHI) REP8(MULHI) }
        if (OP == 2) { REP8(MAD64) REP8(MAD64) REP8(MAD64) REP8(MAD64) }
        if (OP == 3) { REP8(SHR64) REP8(SHR64) REP8(SHR64) REP8(SHR64) }
        if (OP == 4) { REP8(ADD64) REP8(ADD64) REP8(ADD64) REP8(ADD64) }
        if (OP == 5) { REP8(ADD32) REP8(ADD32) REP8(ADD32) REP8(ADD32) }
        if (OP == 6) { REP8(AND32) REP8(AND32) REP8(AND32) REP8(AND32) }
        if (OP == 7) { REP8(MUL24) REP8(MUL24) REP8(MUL24) REP8(MUL24) }
        if (OP == 8) { REP8(MAD24) REP8(MAD24) REP8(MAD24) REP8(MAD24) }
        if (OP == 9) { REP8(MAD64D) REP8(MAD64D) REP8(MAD64D) REP8(MAD64D) }
    }
    uint32_t r = 0;
    for (int i = 0; i < 8; ++i) r ^= a[i] ^ (uint32_t)w[i] ^ (uint32_t)(w[i] >> 32);
    io[threadIdx.x] = r;
}

template <int OP> void run(const char* name, uint32_t* d, double ghz) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    for (int waves_per_simd : {0, 1, 2, 4}) {   // 0: ONE wave on the whole chip
        const int blocks = waves_per_simd ? 1024 * waves_per_simd : 1;
        hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(64), 0, 0, d, 10); hipDeviceSynchronize();
        hipEventRecord(e0); hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(64), 0, 0, d, iters); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double instr = (double)iters * 32;
        printf("%-16s %d wave(s)/SIMD: %6.2f cycles per instruction per wave, %6.2f per SIMD\n", name, waves_per_simd, ms * 1e-3 * ghz * 1e9 / instr, ms * 1e-3 * ghz * 1e9 / instr / (waves_per_simd ? waves_per_simd : 1));
    }
}
int main() {
    uint32_t* d; hipMalloc(&d, 4096 * 4); hipMemset(d, 7, 4096 * 4);
    int clk = 0; hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0);
    const double ghz = clk / 1e6;
    printf("clock %.2f GHz (device attribute); 1024 SIMDs\n", ghz);
    run<0>("v_mul_lo_u32", d, ghz); run<1>("v_mul_hi_u32", d, ghz); run<2>("v_mad_u64_u32", d, ghz); run<3>("v_lshrrev_b64", d, ghz);
    run<4>("v_lshl_add_u64", d, ghz); run<5>("v_add_u32", d, ghz); run<6>("v_and_b32", d, ghz); run<7>("v_mul_u32_u24", d, ghz); run<8>("v_mad_u32_u24", d, ghz); run<9>("mad_u64 dependent", d, ghz);
    return 0;
}
