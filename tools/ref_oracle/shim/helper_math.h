// Stand-in of the diagnostic CPU build (tools/ref_oracle/README.md); written for this repository.
#pragma once
#include <cuda_runtime.h>
using std::max; using std::min;
static inline float3 make_float3(float x, float y, float z) { float3 r = {x, y, z}; return r; }
static inline float3 make_float3(float s) { return make_float3(s, s, s); }
static inline float3 make_float3(int3 a) { return make_float3((float)a.x, (float)a.y, (float)a.z); }
static inline int3 make_int3(int x, int y, int z) { int3 r = {x, y, z}; return r; }
static inline int3 make_int3(int s) { return make_int3(s, s, s); }
static inline int3 make_int3(float3 a) { return make_int3((int)a.x, (int)a.y, (int)a.z); }
static inline float3 operator-(float3 a) { return make_float3(-a.x, -a.y, -a.z); }
static inline float3 operator+(float3 a, float3 b) { return make_float3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline float3 operator-(float3 a, float3 b) { return make_float3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline float3 operator*(float3 a, float3 b) { return make_float3(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline float3 operator/(float3 a, float3 b) { return make_float3(a.x / b.x, a.y / b.y, a.z / b.z); }
static inline float3 operator+(float3 a, float b) { return make_float3(a.x + b, a.y + b, a.z + b); }
static inline float3 operator-(float3 a, float b) { return make_float3(a.x - b, a.y - b, a.z - b); }
static inline float3 operator*(float3 a, float b) { return make_float3(a.x * b, a.y * b, a.z * b); }
static inline float3 operator*(float b, float3 a) { return make_float3(b * a.x, b * a.y, b * a.z); }
static inline float3 operator/(float3 a, float b) { return make_float3(a.x / b, a.y / b, a.z / b); }
static inline void operator+=(float3& a, float3 b) { a.x += b.x; a.y += b.y; a.z += b.z; }
static inline void operator-=(float3& a, float3 b) { a.x -= b.x; a.y -= b.y; a.z -= b.z; }
static inline void operator*=(float3& a, float b) { a.x *= b; a.y *= b; a.z *= b; }
static inline void operator/=(float3& a, float b) { a.x /= b; a.y /= b; a.z /= b; }
static inline int3 operator+(int3 a, int3 b) { return make_int3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline int3 operator-(int3 a, int3 b) { return make_int3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline int3 operator*(int a, int3 b) { return make_int3(a * b.x, a * b.y, a * b.z); }
static inline float dot(float3 a, float3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline float length(float3 v) { return sqrtf(dot(v, v)); }
static inline float3 normalize(float3 v) { float inv = 1.0f / sqrtf(dot(v, v)); return v * inv; }
