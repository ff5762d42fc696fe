// ubench_ldsdma.hip — what one wave's LDS-DMA stream (global_load_lds_dwordx4) delivers on MI355X, by piece shape.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench_ldsdma.hip -o tools/ubench_ldsdma
// Each block = NW loader waves; a loader wave walks its share of a big array in "fills" of P pieces,
// waits vmcnt(0) after each fill.  Variants: piece = 64 lanes x 16 B (full) or `act` active lanes; per-lane addresses
// contiguous; M0 stepped per piece.  Also: plain global_load_dwordx4 + ds_write_b128 with the same shape.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;

template <int P, bool DMA>
__global__ void __launch_bounds__(1024) k_fill(const float4* __restrict__ src, float* __restrict__ sink, int fillsPerWave, int act, int nLoaders, int stride)
{
    extern __shared__ float4 lds[];
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    float acc = 0.f;
    if (wave < nLoaders) {
        float4* my = lds + (size_t)wave * P * 64;
        const float4* base = src + ((size_t)blockIdx.x * nLoaders + wave) * (size_t)fillsPerWave * P * stride;
#pragma unroll 1
        for (int f = 0; f < fillsPerWave; ++f) {
            if (DMA) {
#pragma unroll
                for (int p = 0; p < P; ++p)
                    if (lane < act) __builtin_amdgcn_global_load_lds((gbl_ptr_t)(base + ((size_t)f * P + p) * stride + lane), (lds_ptr_t)(my + p * 64), 16, 0, 0);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            } else {
                float4 r[P];
#pragma unroll
                for (int p = 0; p < P; ++p) if (lane < act) r[p] = base[((size_t)f * P + p) * stride + lane];
#pragma unroll
                for (int p = 0; p < P; ++p) if (lane < act) my[p * 64 + lane] = r[p];
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        acc = my[lane].x;
    }
    if (acc == 123.456f) sink[threadIdx.x] = acc;
}

int main(int argc, char** argv)
{
    const int fills = argc > 1 ? atoi(argv[1]) : 256;
    int numCUs = 256;
    { hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0)); numCUs = prop.multiProcessorCount; }
    const size_t maxBytes = (size_t)numCUs * 4 * fills * 16 * 64 * 16 + 4096;
    float4* src; float* sink; CK(hipMalloc(&src, maxBytes)); CK(hipMalloc(&sink, 4096)); CK(hipMemset(src, 0, maxBytes));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto run = [&](const char* name, int P, bool dma, int act, int nLoaders, int busy, int stride) {
        const int T = 64 * (nLoaders + busy);
        const size_t ldsBytes = (size_t)nLoaders * P * 64 * 16 + 16384;
        float ms = 0;
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipEventRecord(e0));
#define L(PP, DD) hipLaunchKernelGGL((k_fill<PP, DD>), dim3(numCUs), dim3(T), ldsBytes, 0, src, sink, fills, act, nLoaders, stride)
            if (dma) { if (P == 1) L(1, true); else if (P == 4) L(4, true); else L(16, true); }
            else { if (P == 1) L(1, false); else if (P == 4) L(4, false); else L(16, false); }
#undef L
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipGetLastError());
            CK(hipEventElapsedTime(&ms, e0, e1));
        }
        const double bytes = (double)numCUs * nLoaders * fills * P * act * 16.0;
        printf("%-58s %8.3f ms  %7.2f GB/s per CU  %7.2f GB/s per loader wave  %6.2f TB/s chip   (%.0f ns per piece per wave)\n", name, ms,
               bytes / numCUs / ms * 1e-6, bytes / numCUs / nLoaders / ms * 1e-6, bytes / ms * 1e-9, ms * 1e6 / ((double)fills * P));
    };
    char nm[128];
    for (int dma = 1; dma >= 0; --dma)
        for (int P : {1, 4, 16})
            for (int nl : {1, 2, 4})
                for (int act : {64, 45}) {
                    snprintf(nm, sizeof(nm), "%s P=%2d pieces/fill, %d loader wave(s), %2d lanes", dma ? "LDS-DMA" : "ld+ds_write", P, nl, act);
                    run(nm, P, dma, act, nl, 0, 64);
                }
    return 0;
}
