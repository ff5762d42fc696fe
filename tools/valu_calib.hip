// valu_calib.hip — pins the scale of the SQ VALU counters on gfx950.
//
// Three kernels whose VALU instruction count is known exactly (inline asm, nothing for the compiler to fold):
//   dep      one dependent chain of v_fma_f32 per lane                      (latency-bound unless enough waves hide it)
//   indep    8 independent chains of v_fma_f32 per lane                     (issue-bound)
//   packed   8 independent chains of v_pk_fma_f32 per lane                  (issue-bound, 2 flop-pairs per lane)
// each launched with W waves per SIMD on all 1024 SIMDs (256 CUs x 4).  Printed per kernel: wave-instructions issued,
// elapsed device time (hipEvents), wave-instructions per SIMD per microsecond; with the shader clock (measured below with
// s_memtime against wall_clock64) that is "cycles per wave64 VALU instruction", the number the roofline arithmetic needs.
// Run plain for the timing, and under
//   rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -- ./valu_calib
// for the counters: tools/valu_calib_report.py divides them by the known counts.
//
//   hipcc --offload-arch=gfx950 -O2 -std=c++17 tools/valu_calib.hip -o tools/valu_calib
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

constexpr int kUnroll = 64;          // VALU instructions per loop trip (per chain set)

__global__ void __launch_bounds__(256) k_dep(float* out, int trips)
{
    float a = threadIdx.x * 1e-9f, b = 0.999999f, c = 1e-7f;
    for (int t = 0; t < trips; ++t) {
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));
    }
    if (a == 123.456f) out[0] = a;
}
__global__ void __launch_bounds__(256) k_indep(float* out, int trips)
{
    float a[8];
    for (int k = 0; k < 8; ++k) a[k] = threadIdx.x * 1e-9f + k;
    const float b = 0.999999f, c = 1e-7f;
    for (int t = 0; t < trips; ++t) {
#pragma unroll
        for (int u = 0; u < kUnroll / 8; ++u)
#pragma unroll
            for (int k = 0; k < 8; ++k) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(c));
    }
    float s = 0; for (int k = 0; k < 8; ++k) s += a[k];
    if (s == 123.456f) out[0] = s;
}
typedef float f2 __attribute__((ext_vector_type(2)));
__global__ void __launch_bounds__(256) k_packed(float* out, int trips)
{
    f2 a[8];
    for (int k = 0; k < 8; ++k) a[k] = f2{threadIdx.x * 1e-9f + k, 1.0f + k};
    const f2 b = f2{0.999999f, 0.999998f}, c = f2{1e-7f, 2e-7f};
    for (int t = 0; t < trips; ++t) {
#pragma unroll
        for (int u = 0; u < kUnroll / 8; ++u)
#pragma unroll
            for (int k = 0; k < 8; ++k) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(c));
    }
    float s = 0; for (int k = 0; k < 8; ++k) s += a[k].x + a[k].y;
    if (s == 123.456f) out[0] = s;
}
// shader clock: s_memtime ticks (core clock domain) per wall_clock64 tick (100 MHz constant)
__global__ void k_clock(unsigned long long* out)
{
    const unsigned long long w0 = wall_clock64(), c0 = __builtin_readcyclecounter();
    float a = 1.0f;
    for (int t = 0; t < 200000; ++t) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(a));
    const unsigned long long w1 = wall_clock64(), c1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) { out[0] = w1 - w0; out[1] = c1 - c0; out[2] = (unsigned long long)a; }
}

int main(int argc, char** argv)
{
    const int trips = argc > 1 ? atoi(argv[1]) : 4096;
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount, simds = cus * 4;
    printf("device: %s, %d CUs (%d SIMDs), clockRate %.0f MHz (hipDeviceProp)\n", prop.gcnArchName, cus, simds, prop.clockRate / 1e3);
    float* dOut; CK(hipMalloc(&dOut, 64));
    unsigned long long* dClk; CK(hipMalloc(&dClk, 32));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    // warm the clocks up, then measure the shader clock while VALU-busy
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(k_indep, dim3(cus * 8), dim3(256), 0, 0, dOut, trips);
    hipLaunchKernelGGL(k_clock, dim3(1), dim3(64), 0, 0, dClk);
    unsigned long long hc[3]; CK(hipMemcpy(hc, dClk, 24, hipMemcpyDeviceToHost));
    const double mhz = 100.0 * (double)hc[1] / (double)hc[0];
    printf("shader clock while busy: %.0f MHz (s_memtime / wall_clock64; one wave: %llu core ticks per %llu x 10 ns)\n", mhz, hc[1], hc[0]);
    printf("%-8s %6s %16s %10s %14s %12s\n", "kernel", "w/SIMD", "wave-instr", "ms", "instr/SIMD/us", "clk/instr");
    for (int kind = 0; kind < 3; ++kind)
        for (int wps : {1, 2, 4, 8}) {
            const int blocks = simds * wps / 4;                  // 4 waves per 256-thread block
            auto launch = [&] {
                if (kind == 0) hipLaunchKernelGGL(k_dep, dim3(blocks), dim3(256), 0, 0, dOut, trips);
                else if (kind == 1) hipLaunchKernelGGL(k_indep, dim3(blocks), dim3(256), 0, 0, dOut, trips);
                else hipLaunchKernelGGL(k_packed, dim3(blocks), dim3(256), 0, 0, dOut, trips);
            };
            launch();
            CK(hipEventRecord(e0, 0)); launch(); CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
            float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
            const double instr = (double)blocks * 4.0 * (double)trips * kUnroll;          // loop VALU only (the asm statements)
            const double perSimdUs = instr / simds / (ms * 1e3);
            printf("%-8s %6d %16.0f %10.3f %14.1f %12.2f\n", kind == 0 ? "dep" : (kind == 1 ? "indep" : "packed"), wps, instr, ms, perSimdUs, mhz / perSimdUs);
        }
    return 0;
}
