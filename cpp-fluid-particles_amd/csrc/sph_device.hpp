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
    int xOff;            // global x index of local cell column 0 (0 for a whole domain; slabs: x0-1)
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
__device__ __forceinline__ float3 xyz4(const float4 v) { return make_float3(v.x, v.y, v.z); }

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
// (local coordinates: the global x index minus g.xOff, so a slab's sub-grid sees the same cells)
__device__ __forceinline__ int3 cell_of(float3 p, const GridDesc& g)
{
    return make_int3((int)(p.x / g.cellLength) - g.xOff, (int)(p.y / g.cellLength), (int)(p.z / g.cellLength));
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
    const int3 c0 = cell_of(pi, g);
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

// ---- per-step compact neighbour list + LDS-staged tiles ---------------------------------------------
// While positions are frozen (all sweeps of a WCSPH/DFSPH step; the two sweeps of one PBD
// iteration) every sweep of particle i meets the same candidates and rejects the same ones, and a
// rejected candidate contributes exactly +0.  The first pass therefore records, per particle, the
// candidates with r2 <= tCut IN VISIT ORDER (self excluded: its terms are exactly zero); later
// sweeps walk that list.  Order is preserved, so every accumulated bit is.
//
// Row layout: wave-interleaved — entry k of particle i lives at ((i>>6)*cap + k)*64 + (i&63), so
// the 64 lanes of a wave read 256 contiguous bytes per k.  Bit 31 marks a boundary particle.
// count > cap means the row overflowed: that lane falls back to the direct 27-cell walk.
//
// Tiles: a tile is 64 consecutive (cell-sorted) particles = one wave = one workgroup.  Because the
// linear cell id runs z fastest, the 27-cell neighbourhoods of a tile are covered by 9 contiguous
// cell-id ranges [first+off-1, last+off+1], off = (dx*gy+dy)*gz, i.e. 9 contiguous particle ranges
// of the fluid array and 9 of the boundary array.  A tiled sweep copies those ranges (position+mass
// and the one per-particle field the sweep reads from neighbours) into LDS with coalesced loads and
// then gathers from LDS; row entries of a tiled tile are LDS slots instead of global indices.
// Divergent global gathers cost ~64 cycles per wave-instruction on a CU's single texture-address
// pipe and bound the un-tiled sweeps; LDS gathers cost ~10.  Tiles whose ranges exceed kTileSlots,
// that touch the out-of-grid sentinel, or whose positions are not the binned ones (PBD) keep
// global indices (tileFmt = 0).
constexpr int kTile = 64;
constexpr int kWideBlock = 256;     // threads per block of the un-tiled kernels (4 adjacent tiles share a CU's L1)
constexpr int kTileSlots = 1280;
constexpr unsigned int kBoundaryBit = 0x80000000u;

struct SweepCtx {
    GridDesc g; KernelConsts k;
    const int* csF; const float4* posm;     // fluid cell starts, packed (x,y,z,mass)
    const int* csB; const float4* bposm;    // boundary cell starts, packed (x,y,z,mass)
    const unsigned int* nbr; const int* nbrCount; int cap;   // nbr == nullptr: direct sweeps only
    const int* tileFmt;                     // per tile: 1 = row entries are LDS slots (nullptr: never)
    float4* vel4;                           // 16-byte aligned mirror of the fluid velocities (one gather)
    float4* cg4;                            // 16-byte aligned mirror of the colour gradient
    int n;
};

// XCD-aware block order: the dispatcher places block b on XCD b % 8 (MI355X_MICROARCH.md), each XCD
// has its own L2.  Logical block = (b % 8) * chunk + b / 8 gives every XCD one contiguous run of
// logical blocks = one spatial slab of the cell-sorted particles, so the neighbour data an XCD
// re-reads stays in ITS L2 instead of being fetched into all eight.  Grids are launched with
// 8 * chunk blocks; logical blocks past the end exit.  (Speed only: any placement is correct.)
__device__ __forceinline__ int logical_block() { return (int)(blockIdx.x & 7u) * (int)(gridDim.x >> 3) + (int)(blockIdx.x >> 3); }
inline unsigned int xcd_grid(int n, int block) { const int nb = n > 0 ? (n - 1) / block + 1 : 1; return (unsigned int)(((nb + 7) / 8) * 8); }

struct TileTable { int start[18]; int off[19]; };   // ranges 0..8 fluid (dx,dy), 9..17 boundary

// All 64 lanes call this.  Returns true when the tile can be staged.
__device__ __forceinline__ bool tile_table(const SweepCtx& c, const int i0, TileTable& tab)
{
    const int lane = threadIdx.x;
    const int i1 = min(i0 + kTile, c.n);
    const int3 cf = cell_of(xyz4(c.posm[i0]), c.g);
    const int3 cl = cell_of(xyz4(c.posm[i1 - 1]), c.g);
    const int idF = cell_id(cf.x, cf.y, cf.z, c.g), idL = cell_id(cl.x, cl.y, cl.z, c.g);
    const bool ok = idF < c.g.C && idL < c.g.C && idF <= idL;
    int len = 0;
    if (lane < 18) {
        const int r = lane % 9;
        const int off = ((r / 3 - 1) * c.g.gy + (r % 3 - 1)) * c.g.gz;
        const int lo = max(idF + off - 1, 0), hi = min(idL + off + 1, c.g.C - 1);
        const int* cs = lane < 9 ? c.csF : c.csB;
        int s0 = 0;
        if (ok && lo <= hi) { s0 = cs[lo]; len = cs[hi + 1] - s0; }
        tab.start[lane] = s0;
    }
    int incl = len;
#pragma unroll
    for (int o = 1; o < 32; o <<= 1) {
        const int t = __shfl_up(incl, o, 64);
        if (lane >= o) incl += t;
    }
    if (lane < 18) tab.off[lane] = incl - len;
    if (lane == 17) tab.off[18] = incl;
    __syncthreads();
    return ok && tab.off[18] <= kTileSlots;
}

// direct walk in reference order; `visit(j, isBoundary, d, r2, mass_j)`
template <bool WANT_BOUNDARY, class Visit>
__device__ __forceinline__ void walk_cells(const SweepCtx& c, const float3 pi, Visit&& visit)
{
    const int3 c0 = cell_of(pi, c.g);
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
                        visit(j, false, d, r2, pj.w);
                    }
                }
                if (WANT_BOUNDARY) {
                    const int e = c.csB[cell + 1];
                    for (int j = c.csB[cell]; j < e; ++j) {
                        const float4 pj = c.bposm[j];
                        const float3 d = sub3(pi, v3(pj.x, pj.y, pj.z));
                        const float r2 = dot3(d, d);
                        if (r2 > c.k.tCut) continue;
                        visit(j, true, d, r2, pj.w);
                    }
                }
            }
        }
    }
}

// The sweep of one particle.  Op supplies `Field` (the per-neighbour value its pair term reads) and
// `stage(isBoundary, j)` (its global load; boundaries yield zeros); Body::pair(field, isBoundary,
// d, r2, mass_j, j_or_-1) accumulates.  ldsPos/ldsField are the staged tile (nullptr: global).
template <bool WANT_BOUNDARY, class Op, class Body>
__device__ __forceinline__ void sweep(const Op& op, const SweepCtx& c, const float4* ldsPos,
                                      const typename Op::Field* ldsField, const int i, const float3 pi, Body& body)
{
    if (c.nbr) {
        const int cnt = c.nbrCount[i];
        if (cnt <= c.cap) {
            const unsigned int* row = c.nbr + ((size_t)(i >> 6) * (size_t)c.cap) * 64u + (unsigned)(i & 63);
            if (ldsPos) {
                for (int t = 0; t < cnt; ++t) {
                    const unsigned int e = row[(size_t)t * 64u];
                    const bool isB = (e & kBoundaryBit) != 0u;
                    if (!WANT_BOUNDARY && isB) continue;
                    const int slot = (int)(e & ~kBoundaryBit);
                    const float4 pj = ldsPos[slot];
                    const float3 d = sub3(pi, v3(pj.x, pj.y, pj.z));
                    body.pair(ldsField[slot], isB, d, dot3(d, d), pj.w, -1);
                }
            } else {
                for (int t = 0; t < cnt; ++t) {
                    const unsigned int e = row[(size_t)t * 64u];
                    const bool isB = (e & kBoundaryBit) != 0u;
                    if (!WANT_BOUNDARY && isB) continue;
                    const int idx = (int)(e & ~kBoundaryBit);
                    const float4 pj = isB ? c.bposm[idx] : c.posm[idx];
                    const float3 d = sub3(pi, v3(pj.x, pj.y, pj.z));
                    body.pair(op.stage(isB, idx), isB, d, dot3(d, d), pj.w, idx);
                }
            }
            return;
        }
    }
    walk_cells<WANT_BOUNDARY>(c, pi, [&](int j, bool isB, float3 d, float r2, float mj) {
        body.pair(op.stage(isB, j), isB, d, r2, mj, j);
    });
}

// Row construction for particle i: the same walk; the three z-adjacent cells of a (dx,dy) column
// are contiguous in memory, so when they hold no boundary particles the fluid ranges are visited
// as one run (identical order).  With a staged tile candidates are read from LDS and entries are
// LDS slots: slot = tab.off[r] + (j - tab.start[r]), r = (dx+1)*3 + (dy+1) (+9 for boundaries).
__device__ __forceinline__ void build_neighbor_row(const SweepCtx& c, const float4* ldsPos, const TileTable* tab,
                                                   unsigned int* nbr, int* nbrCount, const int i)
{
    const float4 self = c.posm[i];
    const float3 pi = v3(self.x, self.y, self.z);
    unsigned int* row = nbr + ((size_t)(i >> 6) * (size_t)c.cap) * 64u + (unsigned)(i & 63);
    int cnt = 0;
    const int3 c0 = cell_of(pi, c.g);
    const int zlo = max(c0.z - 1, 0), zhi = min(c0.z + 1, c.g.gz - 1);
    for (int dx = -1; dx <= 1; ++dx) {
        const int X = c0.x + dx;
        if (X < 0 || X >= c.g.gx) continue;
        for (int dy = -1; dy <= 1; ++dy) {
            const int Y = c0.y + dy;
            if (Y < 0 || Y >= c.g.gy || zlo > zhi) continue;
            const int base = (X * c.g.gy + Y) * c.g.gz;
            const int r = (dx + 1) * 3 + (dy + 1);
            const int fShift = ldsPos ? tab->off[r] - tab->start[r] : 0;       // slot = j + shift
            const int bShift = ldsPos ? tab->off[r + 9] - tab->start[r + 9] : 0;
            const bool noWall = c.csB[base + zlo] == c.csB[base + zhi + 1];
            const int step = noWall ? (zhi - zlo + 1) : 1;
            for (int z = zlo; z <= zhi; z += step) {
                const int cell = base + z;
                const int e = c.csF[cell + step];
                for (int j = c.csF[cell]; j < e; ++j) {
                    const float4 pj = ldsPos ? ldsPos[j + fShift] : c.posm[j];
                    const float3 d = sub3(pi, v3(pj.x, pj.y, pj.z));
                    if (dot3(d, d) > c.k.tCut || j == i) continue;
                    if (cnt < c.cap) row[(size_t)cnt * 64u] = (unsigned int)(j + fShift);
                    ++cnt;
                }
                if (!noWall) {
                    const int eb = c.csB[cell + 1];
                    for (int j = c.csB[cell]; j < eb; ++j) {
                        const float4 pj = ldsPos ? ldsPos[j + bShift] : c.bposm[j];
                        const float3 d = sub3(pi, v3(pj.x, pj.y, pj.z));
                        if (dot3(d, d) > c.k.tCut) continue;
                        if (cnt < c.cap) row[(size_t)cnt * 64u] = (unsigned int)(j + bShift) | kBoundaryBit;
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
