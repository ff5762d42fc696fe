// sphx_math.h — the few float3/int3 helpers a main.cpp-style driver needs on top of HIP's own
// vector types.  HIP already provides component-wise + - * / for float3/int3; it does not
// provide the broadcast constructors, make_int3(float3), dot, length or normalize that the
// reference pulled from the CUDA-Samples header helper_math.h (not vendored there either).
// Semantics follow that header: C truncation for float->int, length = sqrtf(dot),
// normalize = v * (1/sqrtf(dot)).
#pragma once
#include <hip/hip_runtime.h>
#include <cmath>

__host__ __device__ inline float3 make_float3(float s) { return make_float3(s, s, s); }
__host__ __device__ inline float3 make_float3(int3 a) { return make_float3((float)a.x, (float)a.y, (float)a.z); }
__host__ __device__ inline int3 make_int3(int s) { return make_int3(s, s, s); }
__host__ __device__ inline int3 make_int3(float3 a) { return make_int3((int)a.x, (int)a.y, (int)a.z); }
__host__ __device__ inline float dot(float3 a, float3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__host__ __device__ inline float length(float3 v) { return sqrtf(dot(v, v)); }
__host__ __device__ inline float3 normalize(float3 v)
{
    const float inv = 1.0f / sqrtf(dot(v, v));
    return make_float3(v.x * inv, v.y * inv, v.z * inv);
}
