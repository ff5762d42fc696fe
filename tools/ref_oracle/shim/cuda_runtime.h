// Stand-in for <cuda_runtime.h> of the DIAGNOSTIC CPU build of the reference sources (tools/ref_oracle/README.md): serial kernel
// launches, malloc-backed device memory.  Written for this repository; contains no reference text.
#pragma once
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <algorithm>
#include <chrono>
#include <iostream>
#include <limits>
#include <type_traits>
#include <memory>
#include <vector>
using std::abs;
#ifdef REFA_FLOAT_FABS
// nvcc's device headers (and MSVC's <cmath>) declare float fabs(float): CUDAFunctions.cuh:25 then stays in fp32
static inline float fabs(float x) { return fabsf(x); }
#endif
struct float3 { float x, y, z; };
struct int3 { int x, y, z; };
struct uint3s { unsigned x, y, z; };
extern thread_local uint3s threadIdx, blockIdx, blockDim;
#define __global__
#define __device__
#define __host__
static inline void __syncthreads() {}
static inline int __mul24(int a, int b) { return a * b; }
static inline int atomicAdd(int* p, int v) { int o = *p; *p += v; return o; }
typedef int cudaError_t;
enum { cudaSuccess = 0 };
enum cudaMemcpyKind { cudaMemcpyHostToHost, cudaMemcpyHostToDevice, cudaMemcpyDeviceToHost, cudaMemcpyDeviceToDevice };
static inline cudaError_t cudaMalloc(void** p, size_t n) { *p = malloc(n ? n : 1); return 0; }
static inline cudaError_t cudaFree(void* p) { free(p); return 0; }
static inline cudaError_t cudaMemcpy(void* d, const void* s, size_t n, cudaMemcpyKind) { memmove(d, s, n); return 0; }
static inline cudaError_t cudaMemset(void* d, int v, size_t n) { memset(d, v, n); return 0; }
static inline cudaError_t cudaGetLastError() { return 0; }
static inline const char* cudaGetErrorString(cudaError_t) { return ""; }
static inline cudaError_t cudaDeviceSynchronize() { return 0; }
typedef std::chrono::steady_clock::time_point cudaEvent_t;
static inline cudaError_t cudaEventCreate(cudaEvent_t*) { return 0; }
static inline cudaError_t cudaEventRecord(cudaEvent_t& e, int = 0) { e = std::chrono::steady_clock::now(); return 0; }
static inline cudaError_t cudaEventSynchronize(cudaEvent_t) { return 0; }
static inline cudaError_t cudaEventElapsedTime(float* ms, cudaEvent_t a, cudaEvent_t b) { *ms = std::chrono::duration<float, std::milli>(b - a).count(); return 0; }
static inline cudaError_t cudaEventDestroy(cudaEvent_t) { return 0; }
#define LAUNCH(G, B, K, ...) do { unsigned _g = (G), _b = (B); for (unsigned b_ = 0; b_ < _g; ++b_) for (unsigned t_ = 0; t_ < _b; ++t_) { blockDim.x = _b; blockIdx.x = b_; threadIdx.x = t_; K(__VA_ARGS__); } } while (0)
