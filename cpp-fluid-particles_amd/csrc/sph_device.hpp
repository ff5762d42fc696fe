// sph_device.hpp — device-side arithmetic of the SPH hot path (gfx950).
//
// Everything here is written so that the bits produced are a pure function of the IEEE-754
// binary32 operations add / mul / div / sqrt applied in the reference's association order
// (SURVEY.md §2c); the translation units including this header are compiled with
// -ffp-contract=off and correctly-rounded division / square root.  Hoisting a sub-expression that
// depends only on constants (wA, viscDen, stK, stC) does not change any bit.
//
// Smoothing kernels follow the reference's src/CUDAFunctions.cuh:23-98; `q`, `r` and `d` are
// shared between W and gradW of the same pair because they are the same expressions there.
#pragma once

#include <hip/hip_runtime.h>

namespace sphx {

constexpr float kEps = 1e-6f;                    // global.h:21
constexpr float kPi = 3.14159265358979323846f;   // global.h:22
constexpr float kMaxA = 1000.0f;                 // global.h:26

struct KernelConsts {
    float R;        // smoothing radius (support of every kernel)
    float wA;       // 0.25f / (PI*R*R*R)                      CUDAFunctions.cuh:32
    float viscDen;  // PI * powf(R, 6)                         CUDAFunctions.cuh:53
    float stK;      // PI*cube(R)*cube(R)*cube(R)              CUDAFunctions.cuh:93
    float stC;      // 0.0156f*cube(R)*cube(R)                 CUDAFunctions.cuh:95
    float tCut;     // largest squared distance at which ANY kernel can be non-zero (exact)
};

struct GridDesc {
    int gx, gy, gz, C;   // cells per axis, C = gx*gy*gz (also the out-of-grid sentinel id)
    float cellLength;
};

// ---- float3 helpers with helper_math.h semantics (explicit association) ----------------------
__device__ __forceinline__ float3 v3(float x, float y, float z) { return make_float3(x, y, z); }
__device__ __forceinline__ float3 add3(float3 a, float3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ float3 sub3(float3 a, float3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ float3 neg3(float3 a) { return v3(-a.x, -a.y, -a.z); }
__device__ __forceinline__ float3 mul3s(float3 a, float s) { return v3(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ float3 smul3(float s, float3 a) { return v3(s * a.x, s * a.y, s * a.z); }
__device__ __forceinline__ float3 div3s(float3 a, float s) { return v3(a.x / s, a.y / s, a.z / s); }
__device__ __forceinline__ float dot3(float3 a, float3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ float len3(float3 a) { return sqrtf(dot3(a, a)); }
__device__ __forceinline__ float cube(float x) { return x * x * x; }

// fmaxf/fminf with explicit compares (identical to the oracle's helpers; see oracle header)
__device__ __forceinline__ float max_eps(float x) { return (x > kEps) ? x : kEps; }
__device__ __forceinline__ float max0(float x) { return (x > 0.0f) ? x : 0.0f; }
__device__ __forceinline__ float min0(float x) { return (x < 0.0f) ? x : ((x == 0.0f) ? x : 0.0f); }

// ---- smoothing kernels ------------------------------------------------------------------------
// q = 2*|r|/R as computed by both cubic-spline functions
__device__ __forceinline__ float q_of(float r, const KernelConsts& k) { return 2.0f * r / k.R; }

// cubic_spline_kernel, CUDAFunctions.cuh:23-35
__device__ __forceinline__ float kW(float q, const KernelConsts& k)
{
    if (q > 2.0f || q < kEps) return 0.0f;
    return k.wA * ((q > 1.0f) ? (2.0f - q) * (2.0f - q) * (2.0f - q) : ((3.0f * q - 6.0f) * q * q + 4.0f));
}
// cubic_spline_kernel_gradient, CUDAFunctions.cuh:37-50
__device__ __forceinline__ float3 kGradW(float3 d, float q, const KernelConsts& k)
{
    if (q > 2.0f) return v3(0.0f, 0.0f, 0.0f);
    const float3 a = div3s(d, kPi * (q + kEps) * k.R * k.R * k.R * k.R * k.R);
    return mul3s(a, (q > 1.0f) ? ((12.0f - 3.0f * q) * q - 12.0f) : ((9.0f * q - 12.0f) * q));
}
// viscosity_kernel_laplacian, CUDAFunctions.cuh:52-54
__device__ __forceinline__ float kViscLap(float r, const KernelConsts& k)
{
    return (r <= k.R) ? (45.0f * (k.R - r) / k.viscDen) : 0.0f;
}
// surface_tension_kernel_gradient, CUDAFunctions.cuh:82-98
__device__ __forceinline__ float3 kSurfGrad(float3 d, float x, const KernelConsts& k)
{
    if (x > k.R || x < kEps) return v3(0.0f, 0.0f, 0.0f);
    const float3 a = div3s(smul3(136.0241f, neg3(d)), k.stK * x);
    return mul3s(a, (2.0f * x <= k.R) ? (2.0f * cube(k.R - x) * cube(x) - k.stC) : (cube(k.R - x) * cube(x)));
}

// x^7 of the Tait equation of state (BasicSPHSolver.cu:108): fp64 multiply chain, one rounding
__device__ __forceinline__ float pow7(float x)
{
    const double d = (double)x, d2 = d * d, d4 = d2 * d2, d6 = d4 * d2;
    return (float)(d6 * d);
}

// particlePos2cellIdx + make_int3(pos / cellLength), CUDAFunctions.cuh:64-78
__device__ __forceinline__ int3 cell_of(float3 p, float cellLength)
{
    return make_int3((int)(p.x / cellLength), (int)(p.y / cellLength), (int)(p.z / cellLength));
}
__device__ __forceinline__ int cell_id(int x, int y, int z, const GridDesc& g)
{
    return (x >= 0 && x < g.gx && y >= 0 && y < g.gy && z >= 0 && z < g.gz) ? ((x * g.gy + y) * g.gz + z) : g.C;
}

// box clamp of enforceBoundary_CUDA (BasicSPHSolver.cu:85-96 with velocity, PBDSolver.cu:212-223
// without): [0, 0.99*size] per axis, <= / >= tests, outward velocity component removed.
template <bool WITH_VEL>
__device__ __forceinline__ void clamp_box(float3& p, float3& v, const float3 space)
{
    const float lx = space.x * .00f, hx = space.x * .99f;
    const float ly = space.y * .00f, hy = space.y * .99f;
    const float lz = space.z * .00f, hz = space.z * .99f;
    if (p.x <= lx) { p.x = lx; if (WITH_VEL) v.x = max0(v.x); }
    if (p.x >= hx) { p.x = hx; if (WITH_VEL) v.x = min0(v.x); }
    if (p.y <= ly) { p.y = ly; if (WITH_VEL) v.y = max0(v.y); }
    if (p.y >= hy) { p.y = hy; if (WITH_VEL) v.y = min0(v.y); }
    if (p.z <= lz) { p.z = lz; if (WITH_VEL) v.z = max0(v.z); }
    if (p.z >= hz) { p.z = hz; if (WITH_VEL) v.z = min0(v.z); }
}

// ---- neighbour sweep skeleton -------------------------------------------------------------------
// Visits the 27 cells around the cell of `pi` in the reference order: dx outer, dy,
// dz inner; per cell the fluid range then the boundary range, j ascending.  Candidates farther
// than the common support are skipped: beyond tCut every kernel returns exactly +0 (or the zero
// vector), and adding that to a non-negative-zero-initialised accumulator changes no bit.
// Body::fluid / Body::boundary receive (j, d = pi - pj, r2 = |d|^2, w = pj.w).
template <bool FLUID, bool BOUNDARY, class Body>
__device__ __forceinline__ void sweep27(const GridDesc& g, const KernelConsts& k, const int* __restrict__ csF,
                                        const float4* __restrict__ posmF, const int* __restrict__ csB,
                                        const float4* __restrict__ posmB, const float3 pi, Body& body)
{
    const int3 c0 = cell_of(pi, g.cellLength);
    for (int dx = -1; dx <= 1; ++dx) {
        const int X = c0.x + dx;
        if (X < 0 || X >= g.gx) continue;
        for (int dy = -1; dy <= 1; ++dy) {
            const int Y = c0.y + dy;
            if (Y < 0 || Y >= g.gy) continue;
            for (int dz = -1; dz <= 1; ++dz) {
                const int Z = c0.z + dz;
                if (Z < 0 || Z >= g.gz) continue;
                const int c = (X * g.gy + Y) * g.gz + Z;
                if (FLUID) {
                    const int e = csF[c + 1];
                    for (int j = csF[c]; j < e; ++j) {
                        const float4 pj = posmF[j];
                        const float3 d = sub3(pi, v3(pj.x, pj.y, pj.z));
                        const float r2 = dot3(d, d);
                        if (r2 > k.tCut) continue;
                        body.fluid(j, d, r2, pj.w);
                    }
                }
                if (BOUNDARY) {
                    const int e = csB[c + 1];
                    for (int j = csB[c]; j < e; ++j) {
                        const float4 pj = posmB[j];
                        const float3 d = sub3(pi, v3(pj.x, pj.y, pj.z));
                        const float r2 = dot3(d, d);
                        if (r2 > k.tCut) continue;
                        body.boundary(j, d, r2, pj.w);
                    }
                }
            }
        }
    }
}

// ---- per-step compact neighbour list ---------------------------------------------------------------
// While positions are frozen (all sweeps of a WCSPH/DFSPH step; the two sweeps of one PBD
// iteration) every sweep of particle i meets the same candidates and rejects the same ones, and a
// rejected candidate contributes exactly +0.  The first pass therefore records, per particle, the
// candidates with r2 <= tCut IN VISIT ORDER (self excluded: its terms are exactly zero); later
// sweeps walk that list.  Order is preserved, so every accumulated bit is.
//
// Layout: wave-interleaved rows — entry k of particle i lives at ((i>>6)*cap + k)*64 + (i&63), so
// the 64 lanes of a wave read 256 contiguous bytes per k.  Bit 31 marks a boundary particle.
// count > cap means the list overflowed: that lane falls back to the direct 27-cell walk.
struct SweepCtx {
    GridDesc g; KernelConsts k;
    const int* csF; const float4* posm;     // fluid cell starts, packed (x,y,z,mass)
    const int* csB; const float4* bposm;    // boundary cell starts, packed (x,y,z,mass)
    const unsigned int* nbr; const int* nbrCount; int cap;   // nbr == nullptr: direct sweeps only
};
constexpr unsigned int kBoundaryBit = 0x80000000u;

// direct walk in reference order, Body::pair(idx, isBoundary, d, r2, mass_j)
template <bool WANT_BOUNDARY, class Body>
__device__ __forceinline__ void sweep_direct(const SweepCtx& c, const float3 pi, Body& body)
{
    const int3 c0 = cell_of(pi, c.g.cellLength);
    for (int dx = -1; dx <= 1; ++dx) {
        const int X = c0.x + dx;
        if (X < 0 || X >= c.g.gx) continue;
        for (int dy = -1; dy <= 1; ++dy) {
            const int Y = c0.y + dy;
            if (Y < 0 || Y >= c.g.gy) continue;
            for (int dz = -1; dz <= 1; ++dz) {
                const int Z = c0.z + dz;
                if (Z < 0 || Z >= c.g.gz) continue;
                const int cell = (X * c.g.gy + Y) * c.g.gz + Z;
                {
                    const int e = c.csF[cell + 1];
                    for (int j = c.csF[cell]; j < e; ++j) {
                        const float4 pj = c.posm[j];
                        const float3 d = sub3(pi, v3(pj.x, pj.y, pj.z));
                        const float r2 = dot3(d, d);
                        if (r2 > c.k.tCut) continue;
                        body.pair(j, false, d, r2, pj.w);
                    }
                }
                if (WANT_BOUNDARY) {
                    const int e = c.csB[cell + 1];
                    for (int j = c.csB[cell]; j < e; ++j) {
                        const float4 pj = c.bposm[j];
                        const float3 d = sub3(pi, v3(pj.x, pj.y, pj.z));
                        const float r2 = dot3(d, d);
                        if (r2 > c.k.tCut) continue;
                        body.pair(j, true, d, r2, pj.w);
                    }
                }
            }
        }
    }
}

template <bool WANT_BOUNDARY, class Body>
__device__ __forceinline__ void sweep(const SweepCtx& c, const int i, const float3 pi, Body& body)
{
    if (c.nbr) {
        const int cnt = c.nbrCount[i];
        if (cnt <= c.cap) {
            const unsigned int* row = c.nbr + ((size_t)(i >> 6) * (size_t)c.cap) * 64u + (unsigned)(i & 63);
            for (int t = 0; t < cnt; ++t) {
                const unsigned int e = row[(size_t)t * 64u];
                const bool isB = (e & kBoundaryBit) != 0u;
                if (!WANT_BOUNDARY && isB) continue;
                const int idx = (int)(e & ~kBoundaryBit);
                const float4 pj = isB ? c.bposm[idx] : c.posm[idx];
                const float3 d = sub3(pi, v3(pj.x, pj.y, pj.z));
                body.pair(idx, isB, d, dot3(d, d), pj.w);
            }
            return;
        }
    }
    sweep_direct<WANT_BOUNDARY>(c, pi, body);
}

// list construction: same walk; z-adjacent cells are contiguous in memory (z is the fastest cell
// axis), so when the three cells of a (dx,dy) column hold no boundary particles the fluid ranges
// are visited as one run (identical order).
__device__ __forceinline__ void build_neighbor_row(const SweepCtx& c, unsigned int* nbr, int* nbrCount, const int i)
{
    const float4 self = c.posm[i];
    const float3 pi = v3(self.x, self.y, self.z);
    unsigned int* row = nbr + ((size_t)(i >> 6) * (size_t)c.cap) * 64u + (unsigned)(i & 63);
    int cnt = 0;
    const int3 c0 = cell_of(pi, c.g.cellLength);
    const int zlo = max(c0.z - 1, 0), zhi = min(c0.z + 1, c.g.gz - 1);
    for (int dx = -1; dx <= 1; ++dx) {
        const int X = c0.x + dx;
        if (X < 0 || X >= c.g.gx) continue;
        for (int dy = -1; dy <= 1; ++dy) {
            const int Y = c0.y + dy;
            if (Y < 0 || Y >= c.g.gy || zlo > zhi) continue;
            const int base = (X * c.g.gy + Y) * c.g.gz;
            const bool noWall = c.csB[base + zlo] == c.csB[base + zhi + 1];
            const int step = noWall ? (zhi - zlo + 1) : 1;
            for (int z = zlo; z <= zhi; z += step) {
                const int cell = base + z;
                const int e = c.csF[cell + step];
                for (int j = c.csF[cell]; j < e; ++j) {
                    const float4 pj = c.posm[j];
                    const float3 d = sub3(pi, v3(pj.x, pj.y, pj.z));
                    if (dot3(d, d) > c.k.tCut || j == i) continue;
                    if (cnt < c.cap) row[(size_t)cnt * 64u] = (unsigned int)j;
                    ++cnt;
                }
                if (!noWall) {
                    const int eb = c.csB[cell + 1];
                    for (int j = c.csB[cell]; j < eb; ++j) {
                        const float4 pj = c.bposm[j];
                        const float3 d = sub3(pi, v3(pj.x, pj.y, pj.z));
                        if (dot3(d, d) > c.k.tCut) continue;
                        if (cnt < c.cap) row[(size_t)cnt * 64u] = (unsigned int)j | kBoundaryBit;
                        ++cnt;
                    }
                }
            }
        }
    }
    nbrCount[i] = cnt;
}

__device__ __forceinline__ float3 ld3(const float3* __restrict__ p, int i) { return p[i]; }
__device__ __forceinline__ float3 xyz(const float4 v) { return v3(v.x, v.y, v.z); }

// exact, order-independent |error| accumulation: 2^-32 fixed point
__device__ __forceinline__ long long error_fixed(float e)
{
    float s = fabsf(e) * 4294967296.0f;
    if (!(s < 4.0e18f)) s = 4.0e18f;
    return (long long)s;
}

}  // namespace sphx
