// global.h — constants and error macros shared by the drop-in C++ API.
//
// Mirrors the names a main.cpp-style driver of zhai-xiao/CPP-Fluid-Particles sees in its
// src/global.h:20-46 (block_size, EPSILON, PI, MAX_A, CUDA_CALL, CHECK_KERNEL, ThrustHelper),
// backed by the HIP runtime.  The three numeric constants are part of the arithmetic contract
// (SURVEY.md §2c): they appear verbatim in the smoothing kernels and clamps.
#pragma once

#include <cstdio>
#include <cmath>
#include <hip/hip_runtime.h>

constexpr int block_size = 256;          // launch width the reference uses for every kernel

#ifndef EPSILON
#define EPSILON (1e-6f)
#endif
#ifndef PI
#define PI (3.14159265358979323846f)
#endif
#ifndef MAX_A
#define MAX_A (1000.0f)
#endif

namespace sphx {
// Engine-wide HIP stream: every kernel, memset and copy of the engine is enqueued here, so
// ordering is implicit exactly as on the reference's default stream.  SPHSystem::step() ends with
// a stream synchronize, after which device pointers may be read from any stream.
hipStream_t stream();
// Records a failed HIP call (message retrievable through sphx_last_error()) and prints it,
// like the reference's print-and-continue CUDA_CALL.
void report_hip_error(hipError_t e, const char* file, int line);
}  // namespace sphx

#define HIP_CALL(x)                                                         \
    do {                                                                    \
        hipError_t sphx_e_ = (x);                                           \
        if (sphx_e_ != hipSuccess) ::sphx::report_hip_error(sphx_e_, __FILE__, __LINE__); \
    } while (0)
// API surface, not a compatibility layer: `CUDA_CALL` is the NAME the reference's header gives this macro (global.h:23) and a
// driver written against it uses.  It is the only CUDA-named symbol of the product and expands to the HIP macro above; there is
// no second code path behind it, and nothing in the engine uses it.
#define CUDA_CALL(x) HIP_CALL(x)
#define CHECK_KERNEL() HIP_CALL(hipGetLastError())

namespace ThrustHelper {
// unary "add a constant" and binary "|a|+|b|" functors (reference global.h:28-46); kept for
// drivers that use them — the engine itself fuses these passes into its kernels.
template <typename T>
struct plus {
    T addend;
    explicit plus(const T a) : addend(a) {}
    __host__ __device__ T operator()(const T& v) const { return v + addend; }
};
template <typename T>
struct abs_plus {
    __host__ __device__ T operator()(const T& a, const T& b) const { return std::fabs(a) + std::fabs(b); }
};
}  // namespace ThrustHelper
