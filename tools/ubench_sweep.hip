// ubench_sweep.hip — stand-alone micro-benchmark behind DESIGN.md's choice of sweep structure.
//
// Question it answers on a real MI355X: with the neighbour rows given, what bounds a DFSPH-style
// sweep — the divergent 16-byte global gathers (vector-memory address/tag pipe), LDS reads of the same
// records staged per block, the row stream, or the VALU work per pair?  It runs the SAME pair
// arithmetic (the engine's own kGradW from csrc/sph_device.hpp in "exact" mode, an rsq/rcp + FMA form
// in "tol" mode) through three data paths:
//   G32   rows of 32-bit global indices, wave-interleaved dword stream, 4 entries in flight (the r01 engine)
//   G32c  same indices, but 4 entries per lane stored contiguously (one dwordx4 row load per 4 pairs)
//   L16   256-particle (or 128) blocks stage their 9 neighbour ranges in LDS with coalesced loads; rows are
//         16-bit LDS slots, 8 per lane per 16-byte chunk; pairs read ds_read_b128
// each with one gathered record per pair ("1f": position+scalar, the correction sweeps) or two ("2f":
// position+mass and velocity, the rate sweeps).  Results of all paths are compared (exact: bitwise).
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt \
//         -fno-gpu-flush-denormals-to-zero -I cpp-fluid-particles_amd/csrc -I include tools/ubench_sweep.hip -o ubench_sweep
//   ./ubench_sweep [nx=88] [reps=20] [p = address-pattern probes | b = compact-brick LDS variants] [x: tile schedules] [x: L16]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "sph_device.hpp"

using namespace sphx;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

constexpr int kCap = 64;          // row capacity (entries) of the benchmark scene
constexpr int kChunk = 8;         // 16-bit entries per 16-byte row chunk

struct Consts { KernelConsts k; float twoOverR, gradScale; };   // gradScale = 1 / (PI R^5)

// ---- pair arithmetic ---------------------------------------------------------------------------------
// exact: the engine's formulas on its validated fast paths (bit-exact IEEE results)
__device__ __forceinline__ float pair_exact(const Consts& c, float3 pi, float3 vi, float4 pj, float4 vj)
{
    const float3 d = sub3(pi, v3(pj.x, pj.y, pj.z));
    const float r2 = dot3(d, d);
    const float3 g = kGradW<true>(d, q_of<true>(sqrt_sel<true>(r2), c.k), c.k);
    return pj.w * dot3(sub3(vi, v3(vj.x, vj.y, vj.z)), g);
}
// tolerance: v_rsq / v_rcp and fused multiply-adds
__device__ __forceinline__ float pair_tol(const Consts& c, float3 pi, float3 vi, float4 pj, float4 vj)
{
    const float dx = pi.x - pj.x, dy = pi.y - pj.y, dz = pi.z - pj.z;
    const float r2 = __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
    const float r = r2 * __builtin_amdgcn_rsqf(fmaxf(r2, 1e-30f));
    const float q = r * c.twoOverR;
    const float poly = (q > 1.0f) ? __builtin_fmaf(__builtin_fmaf(-3.0f, q, 12.0f), q, -12.0f) : __builtin_fmaf(9.0f, q, -12.0f) * q;
    const float s = poly * c.gradScale * __builtin_amdgcn_rcpf(q + kEps);
    const float dv = __builtin_fmaf(vi.z - vj.z, dz, __builtin_fmaf(vi.y - vj.y, dy, (vi.x - vj.x) * dx));
    return pj.w * s * dv;
}
template <bool EXACT>
__device__ __forceinline__ float pair_term(const Consts& c, float3 pi, float3 vi, float4 pj, float4 vj)
{
    return EXACT ? pair_exact(c, pi, vi, pj, vj) : pair_tol(c, pi, vi, pj, vj);
}

// ---- G32: the r01 engine's data path -----------------------------------------------------------------
template <bool EXACT, bool TWO>
__global__ void __launch_bounds__(256) k_g32(Consts c, const float4* __restrict__ posm, const float4* __restrict__ vel4,
                                             const unsigned int* __restrict__ rows, const int* __restrict__ cnt,
                                             float* __restrict__ out, int n, int numTiles)
{
    const int tile = logical_block() * 4 + (int)(threadIdx.x >> 6);
    if (tile >= numTiles) return;
    const int i = tile * 64 + (int)(threadIdx.x & 63);
    if (i >= n) return;
    const float4 self = posm[i];
    const float3 pi = v3(self.x, self.y, self.z);
    const float4 sv = vel4[i];
    const float3 vi = v3(sv.x, sv.y, sv.z);
    const unsigned int* row = rows + ((size_t)tile * kCap) * 64u + (unsigned)(i & 63);
    const int m = cnt[i];
    float e = 0.0f;
    int t = 0;
    for (; t + 4 <= m; t += 4) {
        unsigned int idx[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) idx[u] = row[(size_t)(t + u) * 64u];
        float4 pj[4], vj[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            pj[u] = gather16(posm, idx[u] << 4);
            vj[u] = TWO ? gather16(vel4, idx[u] << 4) : make_float4(pj[u].w, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) e += pair_term<EXACT>(c, pi, vi, pj[u], vj[u]);
    }
    for (; t < m; ++t) {
        const unsigned int idx = row[(size_t)t * 64u];
        const float4 pj = gather16(posm, idx << 4);
        const float4 vj = TWO ? gather16(vel4, idx << 4) : make_float4(pj.w, 0.f, 0.f, 0.f);
        e += pair_term<EXACT>(c, pi, vi, pj, vj);
    }
    out[i] = e;
}

// ---- G32c: chunked 32-bit rows (one dwordx4 per 4 pairs), rows padded with the dummy particle n ---------
template <bool EXACT, bool TWO>
__global__ void __launch_bounds__(256) k_g32c(Consts c, const float4* __restrict__ posm, const float4* __restrict__ vel4,
                                              const uint4* __restrict__ rows, const int* __restrict__ tileChunks4,
                                              float* __restrict__ out, int n, int numTiles, int cap4)
{
    const int tile = logical_block() * 4 + (int)(threadIdx.x >> 6);
    if (tile >= numTiles) return;
    const int lane = threadIdx.x & 63;
    const int i = min(tile * 64 + lane, n - 1);
    const float4 self = posm[i];
    const float3 pi = v3(self.x, self.y, self.z);
    const float4 sv = vel4[i];
    const float3 vi = v3(sv.x, sv.y, sv.z);
    const uint4* row = rows + ((size_t)tile * cap4) * 64u + (unsigned)lane;
    const int chunks = tileChunks4[tile];
    float e = 0.0f;
    uint4 nxt = row[0];
    for (int ch = 0; ch < chunks; ++ch) {
        const uint4 cur = nxt;
        if (ch + 1 < chunks) nxt = row[(size_t)(ch + 1) * 64u];
        const unsigned int idx[4] = {cur.x, cur.y, cur.z, cur.w};
        float4 pj[4], vj[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            pj[u] = gather16(posm, idx[u] << 4);
            vj[u] = TWO ? gather16(vel4, idx[u] << 4) : make_float4(pj[u].w, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) e += pair_term<EXACT>(c, pi, vi, pj[u], vj[u]);
    }
    if (tile * 64 + lane < n) out[i] = e;
}

// ---- G32i: position and velocity interleaved in ONE array of 32-byte records: the two gathers of a pair hit
// the same 128-byte line ------------------------------------------------------------------------------------------
template <bool EXACT, bool TWO>
__global__ void __launch_bounds__(256) k_g32i(Consts c, const float4* __restrict__ pv, const uint4* __restrict__ rows,
                                              const int* __restrict__ tileChunks4, float* __restrict__ out, int n, int numTiles, int cap4)
{
    const int tile = logical_block() * 4 + (int)(threadIdx.x >> 6);
    if (tile >= numTiles) return;
    const int lane = threadIdx.x & 63;
    const int i = min(tile * 64 + lane, n - 1);
    const float4 self = pv[2 * i];
    const float3 pi = v3(self.x, self.y, self.z);
    const float4 sv = pv[2 * i + 1];
    const float3 vi = v3(sv.x, sv.y, sv.z);
    const uint4* row = rows + ((size_t)tile * cap4) * 64u + (unsigned)lane;
    const int chunks = tileChunks4[tile];
    float e = 0.0f;
    uint4 nxt = row[0];
    for (int ch = 0; ch < chunks; ++ch) {
        const uint4 cur = nxt;
        if (ch + 1 < chunks) nxt = row[(size_t)(ch + 1) * 64u];
        const unsigned int idx[4] = {cur.x, cur.y, cur.z, cur.w};
        float4 pj[4], vj[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            pj[u] = gather16(pv, idx[u] << 5);
            vj[u] = TWO ? gather16(pv, (idx[u] << 5) + 16u) : make_float4(pj[u].w, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) e += pair_term<EXACT>(c, pi, vi, pj[u], vj[u]);
    }
    if (tile * 64 + lane < n) out[i] = e;
}

// ---- G32o: G32c with a tile schedule (which tile each launched wave works on) and a block size -------------
template <int T, bool EXACT, bool TWO>
__global__ void __launch_bounds__(T) k_g32o(Consts c, const float4* __restrict__ posm, const float4* __restrict__ vel4,
                                            const uint4* __restrict__ rows, const int* __restrict__ tileChunks4,
                                            const int* __restrict__ tileOrder, float* __restrict__ out, int n, int numTiles, int cap4)
{
    const int lt = logical_block() * (T / 64) + (int)(threadIdx.x >> 6);
    if (lt >= numTiles) return;
    const int tile = tileOrder[lt];
    const int lane = threadIdx.x & 63;
    const int i = min(tile * 64 + lane, n - 1);
    const float4 self = posm[i];
    const float3 pi = v3(self.x, self.y, self.z);
    const float4 sv = vel4[i];
    const float3 vi = v3(sv.x, sv.y, sv.z);
    const uint4* row = rows + ((size_t)tile * cap4) * 64u + (unsigned)lane;
    const int chunks = tileChunks4[tile];
    float e = 0.0f;
    uint4 nxt = row[0];
    for (int ch = 0; ch < chunks; ++ch) {
        const uint4 cur = nxt;
        if (ch + 1 < chunks) nxt = row[(size_t)(ch + 1) * 64u];
        const unsigned int idx[4] = {cur.x, cur.y, cur.z, cur.w};
        float4 pj[4], vj[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            pj[u] = gather16(posm, idx[u] << 4);
            vj[u] = TWO ? gather16(vel4, idx[u] << 4) : make_float4(pj[u].w, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) e += pair_term<EXACT>(c, pi, vi, pj[u], vj[u]);
    }
    if (tile * 64 + lane < n) out[i] = e;
}

// ---- L16: block-staged neighbour ranges in LDS, 16-bit slot rows ------------------------------------------

// ---- QG: G lanes per particle.  A wave holds 64/G particles; in one step the G lanes of a particle evaluate G CONSECUTIVE
// row entries, i.e. neighbours that sit next to each other in memory, so the lanes of a quad (G = 4) share cache lines.
// The per-particle sum stays sequential in row order: the G terms are added one after the other (quad broadcasts).
template <int G>
__device__ __forceinline__ float group_term(float t, int e, int lane)
{
    if (G == 4) {
        switch (e) {
        case 0: return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, t), 0x00, 0xf, 0xf, true));
        case 1: return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, t), 0x55, 0xf, 0xf, true));
        case 2: return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, t), 0xAA, 0xf, 0xf, true));
        default: return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, t), 0xFF, 0xf, 0xf, true));
        }
    }
    return __shfl(t, (lane & ~(G - 1)) + e, 64);
}

template <int G, int E, int U, bool EXACT, bool TWO>
__global__ void __launch_bounds__(256) k_qg(Consts c, const float4* __restrict__ posm, const float4* __restrict__ vel4,
                                            const unsigned int* __restrict__ rows, const int* __restrict__ tileSteps,
                                            float* __restrict__ out, int n, int numTilesQ, int capSteps)
{
    // E entries per lane per step (consecutive), U steps in flight; a step of a particle covers G * E consecutive entries
    constexpr int PPW = 64 / G;                     // particles per wave
    const int tile = logical_block() * 4 + (int)(threadIdx.x >> 6);
    if (tile >= numTilesQ) return;
    const int lane = threadIdx.x & 63;
    const int ip = tile * PPW + lane / G;
    const int i = min(ip, n - 1);
    const float4 self = posm[i];
    const float3 pi = v3(self.x, self.y, self.z);
    const float4 sv = vel4[i];
    const float3 vi = v3(sv.x, sv.y, sv.z);
    const unsigned int* row = rows + ((size_t)tile * capSteps) * (64u * E) + (unsigned)lane * E;
    const int steps = tileSteps[tile];
    float e = 0.0f;
    for (int s = 0; s < steps; s += U) {
        unsigned int idx[U][E];
        float4 pj[U][E], vj[U][E];
        float t[U][E];
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int k = 0; k < E; ++k) idx[u][k] = (s + u < steps) ? row[(size_t)(s + u) * (64u * E) + k] : (unsigned)n;
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int k = 0; k < E; ++k) {
                pj[u][k] = gather16(posm, idx[u][k] << 4);
                vj[u][k] = TWO ? gather16(vel4, idx[u][k] << 4) : make_float4(pj[u][k].w, 0.f, 0.f, 0.f);
            }
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int k = 0; k < E; ++k) t[u][k] = pair_term<EXACT>(c, pi, vi, pj[u][k], vj[u][k]);
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int g = 0; g < G; ++g)
#pragma unroll
                for (int k = 0; k < E; ++k) e += group_term<G>(t[u][k], g, lane);
    }
    if (ip < n && (lane % G) == 0) out[i] = e;
}


// ---- QGI: the quad walk (G = 4, E = 1) on 32-byte interleaved (position | velocity) records: the two gathers of a pair fall
// into the same 128-byte line, and the 4 consecutive records of a quad are ONE line instead of two half lines
template <int U, bool EXACT, bool TWO>
__global__ void __launch_bounds__(256) k_qgi(Consts c, const float4* __restrict__ pv, const unsigned int* __restrict__ rows,
                                             const int* __restrict__ tileSteps, float* __restrict__ out, int n, int numTilesQ, int capSteps)
{
    constexpr int G = 4, PPW = 16;
    const int tile = logical_block() * 4 + (int)(threadIdx.x >> 6);
    if (tile >= numTilesQ) return;
    const int lane = threadIdx.x & 63;
    const int ip = tile * PPW + lane / G;
    const int i = min(ip, n - 1);
    const float4 self = pv[2 * i];
    const float3 pi = v3(self.x, self.y, self.z);
    const float4 sv = pv[2 * i + 1];
    const float3 vi = v3(sv.x, sv.y, sv.z);
    const unsigned int* row = rows + ((size_t)tile * capSteps) * 64u + (unsigned)lane;
    const int steps = tileSteps[tile];
    float e = 0.0f;
    for (int s = 0; s < steps; s += U) {
        unsigned int idx[U];
        float4 pj[U], vj[U];
        float t[U];
#pragma unroll
        for (int u = 0; u < U; ++u) idx[u] = (s + u < steps) ? row[(size_t)(s + u) * 64u] : (unsigned)n;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            pj[u] = gather16(pv, idx[u] << 5);
            vj[u] = TWO ? gather16(pv, (idx[u] << 5) + 16u) : make_float4(pj[u].w, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) t[u] = pair_term<EXACT>(c, pi, vi, pj[u], vj[u]);
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int g = 0; g < G; ++g) e += group_term<G>(t[u], g, lane);
    }
    if (ip < n && (lane % G) == 0) out[i] = e;
}


// ---- QGX: the quad walk on 32-byte interleaved records with PAIRED half gathers: in one gather instruction the two lanes of a
// lane pair fetch the two 16-byte halves of ONE record (first the even lane's entry, then the odd lane's), so a wave instruction
// touches the lines of 32 records instead of 64 — every line is visited by one instruction, not by two.  The halves change lanes
// through a select and a quad-permute (DPP) afterwards.  Results are bit-identical to QGI.
__device__ __forceinline__ float swap_pair(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, true));
}
template <int U, bool EXACT>
__global__ void __launch_bounds__(256) k_qgx(Consts c, const float4* __restrict__ pv, const unsigned int* __restrict__ rows,
                                             const int* __restrict__ tileSteps, float* __restrict__ out, int n, int numTilesQ, int capSteps)
{
    constexpr int G = 4, PPW = 16;
    const int tile = logical_block() * 4 + (int)(threadIdx.x >> 6);
    if (tile >= numTilesQ) return;
    const int lane = threadIdx.x & 63;
    const bool odd = lane & 1;
    const int ip = tile * PPW + lane / G;
    const int i = min(ip, n - 1);
    const float4 self = pv[2 * i];
    const float3 pi = v3(self.x, self.y, self.z);
    const float4 sv = pv[2 * i + 1];
    const float3 vi = v3(sv.x, sv.y, sv.z);
    const unsigned int* row = rows + ((size_t)tile * capSteps) * 64u + (unsigned)lane;
    const int steps = tileSteps[tile];
    float e = 0.0f;
    for (int s = 0; s < steps; s += U) {
        unsigned int idx[U];
        float4 a[U], b[U];
        float t[U];
#pragma unroll
        for (int u = 0; u < U; ++u) idx[u] = (s + u < steps) ? row[(size_t)(s + u) * 64u] : (unsigned)n;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const unsigned own = idx[u] << 5;
            const unsigned partner = ((unsigned)__builtin_amdgcn_mov_dpp((int)idx[u], 0xB1, 0xf, 0xf, true) << 5) + 16u;
            a[u] = gather16(pv, odd ? partner : own);       // the even lane's record: position half | velocity half
            b[u] = gather16(pv, odd ? own : partner);       // the odd lane's record:  velocity half | position half
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const float4 pj = odd ? b[u] : a[u];
            const float zx = odd ? a[u].x : b[u].x, zy = odd ? a[u].y : b[u].y, zz = odd ? a[u].z : b[u].z;
            const float4 vj = make_float4(swap_pair(zx), swap_pair(zy), swap_pair(zz), 0.f);
            t[u] = pair_term<EXACT>(c, pi, vi, pj, vj);
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int g = 0; g < G; ++g) e += group_term<G>(t[u], g, lane);
    }
    if (ip < n && (lane % G) == 0) out[i] = e;
}


// ---- QG16: the quad walk (G = 4, E = 1, 4 steps in flight) on 16-bit tile-relative row entries: entry = column (4 bits) | offset
// (12 bits) inside the tile's candidate window of that (dx,dy) column; 9 window bases per 16-particle tile.  Halves the row stream.
template <bool EXACT, bool TWO>
__global__ void __launch_bounds__(256) k_qg16(Consts c, const float4* __restrict__ posm, const float4* __restrict__ vel4,
                                              const unsigned short* __restrict__ rows, const int* __restrict__ tileSteps,
                                              const int* __restrict__ tileBases, float* __restrict__ out, int n, int numTilesQ, int capSteps)
{
    constexpr int G = 4, PPW = 16, U = 4;
    __shared__ int bases[4][16];
    const int wave = threadIdx.x >> 6;
    const int tile = logical_block() * 4 + wave;
    if (tile >= numTilesQ) return;
    const int lane = threadIdx.x & 63;
    if (lane < 16) bases[wave][lane] = lane < 10 ? tileBases[tile * 10 + lane] : n;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const int ip = tile * PPW + lane / G;
    const int i = min(ip, n - 1);
    const float4 self = posm[i];
    const float3 pi = v3(self.x, self.y, self.z);
    const float4 sv = vel4[i];
    const float3 vi = v3(sv.x, sv.y, sv.z);
    const unsigned short* row = rows + ((size_t)tile * capSteps) * 64u + (unsigned)lane;
    const int steps = tileSteps[tile];
    float e = 0.0f;
    for (int s = 0; s < steps; s += U) {
        unsigned int idx[U];
        float4 pj[U], vj[U];
        float t[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const unsigned int raw = (s + u < steps) ? row[(size_t)(s + u) * 64u] : 0x9000u;        // column 9 = the dummy record
            idx[u] = (unsigned int)bases[wave][raw >> 12] + (raw & 0xfffu);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            pj[u] = gather16(posm, idx[u] << 4);
            vj[u] = TWO ? gather16(vel4, idx[u] << 4) : make_float4(pj[u].w, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) t[u] = pair_term<EXACT>(c, pi, vi, pj[u], vj[u]);
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int g = 0; g < G; ++g) e += group_term<G>(t[u], g, lane);
    }
    if (ip < n && (lane % G) == 0) out[i] = e;
}


// ---- QG3: the quad walk gathering 12-byte records (float3 arrays, global_load_dwordx3) instead of 16-byte ones: if the gather
// path is bound by bytes per lane, 24 instead of 32 bytes per pair should show
struct __attribute__((packed, aligned(4))) F3 { float x, y, z; };
template <bool EXACT, bool TWO>
__global__ void __launch_bounds__(256) k_qg3(Consts c, const F3* __restrict__ pos3, const F3* __restrict__ vel3, float m0,
                                             const unsigned int* __restrict__ rows, const int* __restrict__ tileSteps,
                                             float* __restrict__ out, int n, int numTilesQ, int capSteps)
{
    constexpr int G = 4, PPW = 16, U = 4;
    const int tile = logical_block() * 4 + (int)(threadIdx.x >> 6);
    if (tile >= numTilesQ) return;
    const int lane = threadIdx.x & 63;
    const int ip = tile * PPW + lane / G;
    const int i = min(ip, n - 1);
    const F3 sp = pos3[i], sv = vel3[i];
    const float3 pi = v3(sp.x, sp.y, sp.z), vi = v3(sv.x, sv.y, sv.z);
    const unsigned int* row = rows + ((size_t)tile * capSteps) * 64u + (unsigned)lane;
    const int steps = tileSteps[tile];
    float e = 0.0f;
    for (int s = 0; s < steps; s += U) {
        unsigned int idx[U];
        F3 pj[U], vj[U];
        float t[U];
#pragma unroll
        for (int u = 0; u < U; ++u) idx[u] = (s + u < steps) ? row[(size_t)(s + u) * 64u] : (unsigned)n;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            pj[u] = *reinterpret_cast<const F3*>(reinterpret_cast<const char*>(pos3) + idx[u] * 12u);
            if (TWO) vj[u] = *reinterpret_cast<const F3*>(reinterpret_cast<const char*>(vel3) + idx[u] * 12u);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const float mj = idx[u] == (unsigned)n ? 0.0f : m0;
            const float4 p4 = make_float4(pj[u].x, pj[u].y, pj[u].z, mj);
            const float4 v4 = TWO ? make_float4(vj[u].x, vj[u].y, vj[u].z, 0.f) : make_float4(mj, 0.f, 0.f, 0.f);
            t[u] = pair_term<EXACT>(c, pi, vi, p4, v4);
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int g = 0; g < G; ++g) e += group_term<G>(t[u], g, lane);
    }
    if (ip < n && (lane % G) == 0) out[i] = e;
}

// ---- CQ: lane-per-particle arithmetic and accumulation (as G32c), but the gathers are issued quad-cooperatively: in
// gather k the 4 lanes of a quad fetch the 4 entries of particle k of the quad (adjacent records), park them in LDS and
// every lane then reads its own 4 records back.  Rows in the engine's chunk layout: [tile][chunk][lane][4].
template <bool EXACT, bool TWO>
__global__ void __launch_bounds__(256) k_cq(Consts c, const float4* __restrict__ posm, const float4* __restrict__ vel4,
                                            const unsigned int* __restrict__ rows, const int* __restrict__ cnt,
                                            float* __restrict__ out, int n, int numTiles, int capChunks)
{
    constexpr int kStride = 64 * 4 + 4;                 // float4 slots per source index k, padded
    __shared__ float4 lpos[4][4 * kStride];             // [wave][k][quad * 4 + entry]   (linear in the lane when written)
    __shared__ float4 lvel[TWO ? 4 : 1][TWO ? 4 * kStride : 1];
    const int wave = threadIdx.x >> 6;
    const int tile = logical_block() * 4 + wave;
    if (tile >= numTiles) return;
    const int lane = threadIdx.x & 63, l = lane & 3;
    const int ip = tile * 64 + lane;
    const int i = min(ip, n - 1);
    const float4 self = posm[i];
    const float3 pi = v3(self.x, self.y, self.z);
    const float4 sv = vel4[i];
    const float3 vi = v3(sv.x, sv.y, sv.z);
    const int m = ip < n ? cnt[i] : 0;
    int chunks = (m + 3) >> 2;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) chunks = max(chunks, __shfl_xor(chunks, off, 64));
    const unsigned int* tileRows = rows + (size_t)tile * capChunks * 256u;
    // counts of the 4 particles of my quad
    int mk[4];
    mk[0] = __builtin_amdgcn_mov_dpp(m, 0x00, 0xf, 0xf, true); mk[1] = __builtin_amdgcn_mov_dpp(m, 0x55, 0xf, 0xf, true);
    mk[2] = __builtin_amdgcn_mov_dpp(m, 0xAA, 0xf, 0xf, true); mk[3] = __builtin_amdgcn_mov_dpp(m, 0xFF, 0xf, 0xf, true);
    float e = 0.0f;
    for (int s = 0; s < chunks; ++s) {
        const unsigned int* ch = tileRows + (size_t)s * 256u;
        unsigned int idx[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {                    // entry 4s + l of particle k of my quad
            const unsigned int raw = ch[((lane & ~3) + k) * 4 + l];
            idx[k] = (4 * s + l < mk[k]) ? raw : (unsigned)n;
        }
        float4 pj[4], vj[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            pj[k] = gather16(posm, idx[k] << 4);
            if (TWO) vj[k] = gather16(vel4, idx[k] << 4);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            lpos[wave][k * kStride + lane] = pj[k];
            if (TWO) lvel[wave][k * kStride + lane] = vj[k];
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        float4 qj[4], wj[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {                    // my particle = index l of the quad; its entry t
            qj[t] = lpos[wave][l * kStride + (lane & ~3) + t];
            wj[t] = TWO ? lvel[wave][l * kStride + (lane & ~3) + t] : make_float4(qj[t].w, 0.f, 0.f, 0.f);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int t = 0; t < 4; ++t) e += pair_term<EXACT>(c, pi, vi, qj[t], wj[t]);   // padded entries: the dummy record, term +-0
    }
    if (ip < n) out[i] = e;
}

struct BlockRanges { int start[9]; int len[9]; int base[9]; int total; };

template <int T, bool EXACT, bool TWO>
__global__ void __launch_bounds__(T) k_l16(Consts c, const float4* __restrict__ posm, const float4* __restrict__ vel4,
                                           const uint4* __restrict__ rows, const int* __restrict__ tileChunks,
                                           const BlockRanges* __restrict__ ranges, float* __restrict__ out, int n, int numTiles,
                                           int slots)
{
    extern __shared__ float4 lds[];
    float4* lpos = lds;
    float4* lvel = lds + slots;
    constexpr int kWaves = T / 64;
    const int blk = logical_block();
    const int tile0 = blk * kWaves;
    if (tile0 >= numTiles) return;
    const BlockRanges& R = ranges[blk];
#pragma unroll 1
    for (int r = 0; r < 9; ++r) {
        const int s0 = R.start[r], ln = R.len[r], b0 = R.base[r];
        for (int t = threadIdx.x; t < ln; t += T) {
            lpos[b0 + t] = posm[s0 + t];
            if (TWO) lvel[b0 + t] = vel4[s0 + t];
        }
    }
    if (threadIdx.x == 0) {                     // the padding slot: zero mass, far away, zero field
        lpos[R.total] = make_float4(1.0e3f, 1.0e3f, 1.0e3f, 0.0f);
        if (TWO) lvel[R.total] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    __syncthreads();
    const int tile = tile0 + (int)(threadIdx.x >> 6);
    if (tile >= numTiles) return;
    const int lane = threadIdx.x & 63;
    const int i = min(tile * 64 + lane, n - 1);
    const float4 self = posm[i];
    const float3 pi = v3(self.x, self.y, self.z);
    const float4 sv = vel4[i];
    const float3 vi = v3(sv.x, sv.y, sv.z);
    const uint4* row = rows + ((size_t)tile * (kCap / kChunk)) * 64u + (unsigned)lane;
    const int chunks = tileChunks[tile];
    float e = 0.0f;
    uint4 nxt = row[0];
    for (int ch = 0; ch < chunks; ++ch) {
        const uint4 cur = nxt;
        if (ch + 1 < chunks) nxt = row[(size_t)(ch + 1) * 64u];
        const unsigned int w[4] = {cur.x, cur.y, cur.z, cur.w};
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            float4 pj[4], vj[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const unsigned int word = w[h * 2 + (u >> 1)];
                const unsigned int slot = (u & 1) ? (word >> 16) : (word & 0xffffu);
                pj[u] = lpos[slot];
                vj[u] = TWO ? lvel[slot] : make_float4(pj[u].w, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) e += pair_term<EXACT>(c, pi, vi, pj[u], vj[u]);
        }
    }
    if (tile * 64 + lane < n) out[i] = e;
}


// ---- BRICK: the compact-brick LDS stage SURVEY 8(a) sized ---------------------------------------------------
// A block owns the particles of a brick of BX x BY cell columns x BZ cells along z (z is the fastest cell axis, so the brick's part
// of every column is ONE contiguous particle run) and stages the (BX+2) x (BY+2) surrounding runs, z range [z0-1, z0+BZ], with
// coalesced loads: position+mass (and, for the two-field sweeps, velocity) as 16-byte LDS records.  Rows hold 16-bit LDS slots,
// 8 per 16-byte chunk; a pair costs one (two) ds_read_b128 instead of one (two) divergent global gathers.  Own particles are
// handed to the T threads in rounds of T (a brick of the reference lattice owns ~500-700 particles).
struct BrickDesc { int runFirst, numRuns, staged, own, ownFirst, rowBase, rounds; };   // runs / own map / rows: offsets into flat arrays
struct BrickRun { int start, len, base; };

template <int T, bool EXACT, bool TWO>
__global__ void __launch_bounds__(T) k_brick(Consts c, const float4* __restrict__ posm, const float4* __restrict__ vel4,
                                             const BrickDesc* __restrict__ bricks, const BrickRun* __restrict__ runs,
                                             const int* __restrict__ ownIndex, const unsigned short* __restrict__ ownSlot,
                                             const uint4* __restrict__ rows, const unsigned char* __restrict__ waveChunks,
                                             float* __restrict__ out, int numBricks, int slots)
{
    extern __shared__ float4 lds[];
    float4* lpos = lds;
    float4* lvel = lds + slots;
    constexpr int kWaves = T / 64;
    const int blk = logical_block();
    if (blk >= numBricks) return;
    const BrickDesc B = bricks[blk];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int r = wave; r < B.numRuns; r += kWaves) {             // one wave per run: coalesced 16-byte loads
        const BrickRun R = runs[B.runFirst + r];
        for (int t = lane; t < R.len; t += 64) {
            lpos[R.base + t] = posm[R.start + t];
            if (TWO) lvel[R.base + t] = vel4[R.start + t];
        }
    }
    if (threadIdx.x == 0) {                                      // the padding slot
        lpos[B.staged] = make_float4(1.0e3f, 1.0e3f, 1.0e3f, 0.0f);
        if (TWO) lvel[B.staged] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    __syncthreads();
    int rowAt = B.rowBase;                                       // in chunk-rows of T lanes
    for (int rd = 0; rd < B.rounds; ++rd) {
        const int p = rd * T + (int)threadIdx.x;
        const bool has = p < B.own;
        const int i = has ? ownIndex[B.ownFirst + p] : -1;
        const int self = has ? (int)ownSlot[B.ownFirst + p] : B.staged;
        const float4 sp = lpos[self];
        const float3 pi = v3(sp.x, sp.y, sp.z);
        const float4 sv = TWO ? lvel[self] : make_float4(sp.w, 0.f, 0.f, 0.f);
        const float3 vi = v3(sv.x, sv.y, sv.z);
        const int chunksRound = waveChunks[(size_t)(blk * 8 + rd) * 17 + 16];      // [brick][round][wave 0..15 | 16 = max of the round]
        const int chunks = waveChunks[(size_t)(blk * 8 + rd) * 17 + wave];
        const uint4* row = rows + (size_t)rowAt * T + threadIdx.x;
        float e = 0.0f;
        uint4 nxt = chunks > 0 ? row[0] : make_uint4(0, 0, 0, 0);
        for (int ch = 0; ch < chunks; ++ch) {
            const uint4 cur = nxt;
            if (ch + 1 < chunks) nxt = row[(size_t)(ch + 1) * T];
            const unsigned int w[4] = {cur.x, cur.y, cur.z, cur.w};
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                float4 pj[4], vj[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const unsigned int word = w[h * 2 + (u >> 1)];
                    const unsigned int slot = (u & 1) ? (word >> 16) : (word & 0xffffu);
                    pj[u] = lpos[slot];
                    vj[u] = TWO ? lvel[slot] : make_float4(pj[u].w, 0.f, 0.f, 0.f);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) e += pair_term<EXACT>(c, pi, vi, pj[u], vj[u]);
            }
        }
        if (has) out[i] = e;
        rowAt += chunksRound;
    }
}


// ---- row BUILDERS: the lane-per-particle walk over global memory (the engine's k_build_list, fluid only) against the same walk
// reading its candidates from a brick's LDS stage.  Both write the engine's row format (chunks of 4 x 32-bit global indices).
__device__ __forceinline__ void ub_put(unsigned int* row, int cnt, unsigned int e, uint4& pend, int cap)
{
    const int w = cnt & 3;
    pend.x = w == 0 ? e : pend.x; pend.y = w == 1 ? e : pend.y; pend.z = w == 2 ? e : pend.z; pend.w = w == 3 ? e : pend.w;
    if (w == 3 && cnt < cap) *reinterpret_cast<uint4*>(row + (size_t)(cnt >> 2) * 256u) = pend;
}
__global__ void __launch_bounds__(256) k_build_global(const float4* __restrict__ posm, const int* __restrict__ cs, int gx, int gy, int gz,
                                                      float cellLength, float cut, unsigned int* __restrict__ rows, int* __restrict__ counts,
                                                      int n, int numTiles, int cap)
{
    const int tile = logical_block() * 4 + (int)(threadIdx.x >> 6);
    if (tile >= numTiles) return;
    const int i = tile * 64 + (int)(threadIdx.x & 63);
    if (i >= n) return;
    const float4 self = posm[i];
    const float3 pi = v3(self.x, self.y, self.z);
    const int cx = (int)(pi.x / cellLength), cy = (int)(pi.y / cellLength), cz = (int)(pi.z / cellLength);
    unsigned int* row = rows + ((size_t)(i >> 6) * cap) * 64u + (size_t)(i & 63) * 4u;
    const int zlo = max(cz - 1, 0), zhi = min(cz + 1, gz - 1);
    int cnt = 0; uint4 pend = make_uint4(0, 0, 0, 0);
    for (int dx = -1; dx <= 1; ++dx) {
        const int X = cx + dx; if (X < 0 || X >= gx) continue;
        for (int dy = -1; dy <= 1; ++dy) {
            const int Y = cy + dy; if (Y < 0 || Y >= gy) continue;
            const int base = (X * gy + Y) * gz;
            const int e = cs[base + zhi + 1];
            int j = cs[base + zlo];
            for (; j + 4 <= e; j += 4) {
                float4 pj[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) pj[u] = posm[j + u];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const float3 d = sub3(pi, v3(pj[u].x, pj[u].y, pj[u].z));
                    const float r2 = dot3(d, d);
                    if (r2 > cut || j + u == i) continue;
                    ub_put(row, cnt, (unsigned)(j + u), pend, cap); ++cnt;
                }
            }
            for (; j < e; ++j) {
                const float4 pj = posm[j];
                const float3 d = sub3(pi, v3(pj.x, pj.y, pj.z));
                const float r2 = dot3(d, d);
                if (r2 > cut || j == i) continue;
                ub_put(row, cnt, (unsigned)j, pend, cap); ++cnt;
            }
        }
    }
    counts[i] = cnt;
    if ((cnt & 3) != 0 && cnt < cap) *reinterpret_cast<uint4*>(row + (size_t)(cnt >> 2) * 256u) = pend;
}
// the same builder reading its candidates from a float3 position array: 4 consecutive candidates = 48 bytes = THREE 16-byte loads
// (dword-aligned), 12 bytes per candidate instead of 16 on the vector memory path
struct __attribute__((packed, aligned(4))) U4 { float a, b, c, d; };
__global__ void __launch_bounds__(256) k_build_global3(const F3* __restrict__ pos3, const int* __restrict__ cs, int gx, int gy, int gz,
                                                       float cellLength, float cut, unsigned int* __restrict__ rows, int* __restrict__ counts,
                                                       int n, int numTiles, int cap)
{
    const int tile = logical_block() * 4 + (int)(threadIdx.x >> 6);
    if (tile >= numTiles) return;
    const int i = tile * 64 + (int)(threadIdx.x & 63);
    if (i >= n) return;
    const F3 self = pos3[i];
    const float3 pi = v3(self.x, self.y, self.z);
    const int cx = (int)(pi.x / cellLength), cy = (int)(pi.y / cellLength), cz = (int)(pi.z / cellLength);
    unsigned int* row = rows + ((size_t)(i >> 6) * cap) * 64u + (size_t)(i & 63) * 4u;
    const int zlo = max(cz - 1, 0), zhi = min(cz + 1, gz - 1);
    int cnt = 0; uint4 pend = make_uint4(0, 0, 0, 0);
    for (int dx = -1; dx <= 1; ++dx) {
        const int X = cx + dx; if (X < 0 || X >= gx) continue;
        for (int dy = -1; dy <= 1; ++dy) {
            const int Y = cy + dy; if (Y < 0 || Y >= gy) continue;
            const int base = (X * gy + Y) * gz;
            const int e = cs[base + zhi + 1];
            int j = cs[base + zlo];
            for (; j + 4 <= e; j += 4) {
                const U4* q = reinterpret_cast<const U4*>(pos3 + j);
                const U4 a = q[0], b = q[1], c = q[2];
                const float3 pj[4] = {v3(a.a, a.b, a.c), v3(a.d, b.a, b.b), v3(b.c, b.d, c.a), v3(c.b, c.c, c.d)};
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const float3 d = sub3(pi, pj[u]);
                    const float r2 = dot3(d, d);
                    if (r2 > cut || j + u == i) continue;
                    ub_put(row, cnt, (unsigned)(j + u), pend, cap); ++cnt;
                }
            }
            for (; j < e; ++j) {
                const F3 pj = pos3[j];
                const float3 d = sub3(pi, v3(pj.x, pj.y, pj.z));
                const float r2 = dot3(d, d);
                if (r2 > cut || j == i) continue;
                ub_put(row, cnt, (unsigned)j, pend, cap); ++cnt;
            }
        }
    }
    counts[i] = cnt;
    if ((cnt & 3) != 0 && cnt < cap) *reinterpret_cast<uint4*>(row + (size_t)(cnt >> 2) * 256u) = pend;
}

template <int T>
__global__ void __launch_bounds__(T) k_build_brick(const float4* __restrict__ posm, const int* __restrict__ cs, int gx, int gy, int gz,
                                                   float cellLength, float cut, const BrickDesc* __restrict__ bricks,
                                                   const BrickRun* __restrict__ runs, const int* __restrict__ ownIndex,
                                                   const int* __restrict__ brickOrigin,     // x0, y0, z0 of the brick (cells)
                                                   unsigned int* __restrict__ rows, int* __restrict__ counts, int numBricks, int cap,
                                                   int bx, int by, int bz)
{
    extern __shared__ float4 lds[];
    __shared__ int runStart[64], runBase[64];
    constexpr int kWaves = T / 64;
    const int blk = logical_block();
    if (blk >= numBricks) return;
    const BrickDesc B = bricks[blk];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int r = wave; r < B.numRuns; r += kWaves) {
        const BrickRun R = runs[B.runFirst + r];
        if (lane == 0) { runStart[r] = R.start; runBase[r] = R.base; }
        for (int t = lane; t < R.len; t += 64) lds[R.base + t] = posm[R.start + t];
    }
    __syncthreads();
    const int x0 = brickOrigin[3 * blk], y0 = brickOrigin[3 * blk + 1], z0 = brickOrigin[3 * blk + 2];
    const int hx0 = max(x0 - 1, 0), hy0 = max(y0 - 1, 0), hz0 = max(z0 - 1, 0);           // origin of the halo
    const int hny = min(y0 + by, gy - 1) - hy0 + 1;                                        // halo columns along y
    for (int rd = 0; rd < B.rounds; ++rd) {
        const int p = rd * T + (int)threadIdx.x;
        if (p >= B.own) continue;
        const int i = ownIndex[B.ownFirst + p];
        const float4 self = posm[i];
        const float3 pi = v3(self.x, self.y, self.z);
        const int cx = (int)(pi.x / cellLength), cy = (int)(pi.y / cellLength), cz = (int)(pi.z / cellLength);
        unsigned int* row = rows + ((size_t)(i >> 6) * cap) * 64u + (size_t)(i & 63) * 4u;
        const int zlo = max(cz - 1, 0), zhi = min(cz + 1, gz - 1);
        int cnt = 0; uint4 pend = make_uint4(0, 0, 0, 0);
        for (int dx = -1; dx <= 1; ++dx) {
            const int X = cx + dx; if (X < 0 || X >= gx) continue;
            for (int dy = -1; dy <= 1; ++dy) {
                const int Y = cy + dy; if (Y < 0 || Y >= gy) continue;
                const int base = (X * gy + Y) * gz;
                const int e = cs[base + zhi + 1];
                int j = cs[base + zlo];
                // the staged run of this halo column (runs are emitted x-major, y-minor; empty columns have no run: host guarantees
                // the table is dense for this benchmark scene by construction -- see the host code)
                const int r = (X - hx0) * hny + (Y - hy0);
                const int shift = runBase[r] - runStart[r];
                for (; j + 4 <= e; j += 4) {
                    float4 pj[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) pj[u] = lds[j + u + shift];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const float3 d = sub3(pi, v3(pj[u].x, pj[u].y, pj[u].z));
                        const float r2 = dot3(d, d);
                        if (r2 > cut || j + u == i) continue;
                        ub_put(row, cnt, (unsigned)(j + u), pend, cap); ++cnt;
                    }
                }
                for (; j < e; ++j) {
                    const float4 pj = lds[j + shift];
                    const float3 d = sub3(pi, v3(pj.x, pj.y, pj.z));
                    const float r2 = dot3(d, d);
                    if (r2 > cut || j == i) continue;
                    ub_put(row, cnt, (unsigned)j, pend, cap); ++cnt;
                }
            }
        }
        counts[i] = cnt;
        if ((cnt & 3) != 0 && cnt < cap) *reinterpret_cast<uint4*>(row + (size_t)(cnt >> 2) * 256u) = pend;
    }
    (void)hz0; (void)bx; (void)bz;
}

// ---- host: scene, grid, rows ---------------------------------------------------------------------------
static float bits_to_float(unsigned int b) { float f; memcpy(&f, &b, 4); return f; }

int main(int argc, char** argv)
{
    const int nx = argc > 1 ? atoi(argv[1]) : 88;
    const int reps = argc > 2 ? atoi(argv[2]) : 20;
    const float spacing = 0.02f, R = 0.04f, cellLength = 1.01f * R, scale = nx / 24.0f;
    const int ny = 3 * nx / 2, nz = nx;
    const int n = nx * ny * nz;
    const int gx = (int)ceilf(scale / cellLength), gy = gx, gz = gx, C = gx * gy * gz;
    printf("scene: %d x %d x %d = %d particles, grid %d^3, R = %g\n", nx, ny, nz, n, gx, R);

    // jittered lattice (deterministic): positions as in the dam-break block, +-4 %% of the spacing
    std::vector<float4> P0(n);
    std::vector<int> cell(n);
    unsigned int rng = 12345u;
    auto jitter = [&]() { rng = rng * 1664525u + 1013904223u; return ((rng >> 8) * (1.0f / 16777216.0f) - 0.5f) * 0.08f * spacing; };
    {
        int q = 0;
        for (int iy = 0; iy < ny; ++iy) for (int ix = 0; ix < nx; ++ix) for (int iz = 0; iz < nz; ++iz, ++q) {
            P0[q] = make_float4(0.27f * scale + spacing * ix + jitter(), 0.10f * scale + spacing * iy + jitter(),
                                0.27f * scale + spacing * iz + jitter(), 76.596750762082e-6f);
            const int cx = (int)(P0[q].x / cellLength), cy = (int)(P0[q].y / cellLength), cz = (int)(P0[q].z / cellLength);
            cell[q] = (cx * gy + cy) * gz + cz;
        }
    }
    std::vector<int> cs(C + 2, 0), order(n);
    for (int q = 0; q < n; ++q) cs[cell[q] + 1]++;
    for (int k = 0; k < C + 1; ++k) cs[k + 1] += cs[k];
    {
        std::vector<int> cur(cs.begin(), cs.begin() + C + 1);
        for (int q = 0; q < n; ++q) order[cur[cell[q]]++] = q;
    }
    std::vector<float4> posm(n + 1), vel4(n + 1);
    std::vector<int> scell(n);
    for (int q = 0; q < n; ++q) {
        posm[q] = P0[order[q]]; scell[q] = cell[order[q]];
        rng = rng * 1664525u + 1013904223u;
        vel4[q] = make_float4(jitter() * 50.f, -0.04f + jitter() * 50.f, jitter() * 50.f, 0.0f);
    }
    posm[n] = make_float4(1.0e3f, 1.0e3f, 1.0e3f, 0.0f);   // dummy record for padded rows
    vel4[n] = make_float4(0.f, 0.f, 0.f, 0.f);

    const int numTiles = (n + 63) / 64;
    const float tCut = R * R;
    // global rows (visit order dx, dy, dz, ascending index), capacity kCap
    std::vector<unsigned int> rowsG((size_t)numTiles * kCap * 64, 0u);
    std::vector<uint4> rowsGc((size_t)numTiles * (kCap / 4) * 64, make_uint4(n, n, n, n));
    std::vector<int> cnt(n, 0), tileChunks(numTiles, 0);
    std::vector<std::vector<int>> nbr(n);
    long long pairs = 0; int maxCnt = 0;
#pragma omp parallel for reduction(+ : pairs) reduction(max : maxCnt) schedule(dynamic, 4096)
    for (int i = 0; i < n; ++i) {
        const int c0 = scell[i], cz = c0 % gz, cy = (c0 / gz) % gy, cx = c0 / (gz * gy);
        std::vector<int>& my = nbr[i];
        for (int dx = -1; dx <= 1; ++dx) for (int dy = -1; dy <= 1; ++dy) for (int dz = -1; dz <= 1; ++dz) {
            const int X = cx + dx, Y = cy + dy, Z = cz + dz;
            if (X < 0 || X >= gx || Y < 0 || Y >= gy || Z < 0 || Z >= gz) continue;
            const int cc = (X * gy + Y) * gz + Z;
            for (int j = cs[cc]; j < cs[cc + 1]; ++j) {
                if (j == i) continue;
                const float ddx = posm[i].x - posm[j].x, ddy = posm[i].y - posm[j].y, ddz = posm[i].z - posm[j].z;
                if (ddx * ddx + ddy * ddy + ddz * ddz <= tCut) my.push_back(j);
            }
        }
        pairs += (long long)my.size();
        maxCnt = std::max(maxCnt, (int)my.size());
    }
    if (maxCnt > kCap) { printf("row capacity %d exceeded (%d)\n", kCap, maxCnt); return 1; }
    for (int i = 0; i < n; ++i) {
        const int tile = i >> 6, lane = i & 63, m = (int)nbr[i].size();
        cnt[i] = m;
        tileChunks[tile] = std::max(tileChunks[tile], (m + kChunk - 1) / kChunk);
        for (int t = 0; t < m; ++t) {
            rowsG[((size_t)tile * kCap + t) * 64 + lane] = (unsigned)nbr[i][t];
            unsigned int* w = reinterpret_cast<unsigned int*>(&rowsGc[((size_t)tile * (kCap / 4) + t / 4) * 64 + lane]);
            w[t & 3] = (unsigned)nbr[i][t];
        }
    }
    double waveIters = 0;
    for (int t = 0; t < numTiles; ++t) waveIters += tileChunks[t] * kChunk;
    printf("pairs: %lld (%.1f per particle), max %d; wave-iterations per tile %.1f (padding x%.2f)\n", pairs, (double)pairs / n,
           maxCnt, waveIters / numTiles, waveIters * 64.0 / pairs);

    // block ranges + 16-bit slot rows for T = 256 and T = 128
    struct L16Set { int T; std::vector<BlockRanges> ranges; std::vector<uint4> rows; int slots; };
    L16Set sets[2];
    for (int v = 0; v < 2; ++v) {
        L16Set& S = sets[v];
        S.T = v == 0 ? 256 : 128;
        const int blocks = (n + S.T - 1) / S.T;
        S.ranges.resize(blocks);
        S.rows.assign((size_t)numTiles * (kCap / kChunk) * 64, make_uint4(0, 0, 0, 0));
        S.slots = 0;
        for (int b = 0; b < blocks; ++b) {
            const int i0 = b * S.T, i1 = std::min(n, i0 + S.T);
            const int idF = scell[i0], idL = scell[i1 - 1];
            BlockRanges& Rg = S.ranges[b];
            int base = 0;
            for (int r = 0; r < 9; ++r) {
                const int off = ((r / 3 - 1) * gy + (r % 3 - 1)) * gz;
                const int lo = std::max(idF + off - 1, 0), hi = std::min(idL + off + 1, C - 1);
                Rg.start[r] = 0; Rg.len[r] = 0; Rg.base[r] = base;
                if (lo <= hi) { Rg.start[r] = cs[lo]; Rg.len[r] = cs[hi + 1] - cs[lo]; }
                base += Rg.len[r];
            }
            Rg.total = base;
            S.slots = std::max(S.slots, base + 1);
            for (int i = i0; i < i1; ++i) {
                const int c0 = scell[i], cy = (c0 / gz) % gy, cx = c0 / (gz * gy);
                const int tile = i >> 6, lane = i & 63, m = cnt[i];
                unsigned short* w = nullptr;
                for (int t = 0; t < tileChunks[tile] * kChunk; ++t) {
                    if ((t % kChunk) == 0) w = reinterpret_cast<unsigned short*>(&S.rows[((size_t)tile * (kCap / kChunk) + t / kChunk) * 64 + lane]);
                    int slot = Rg.total;           // padding
                    if (t < m) {
                        const int j = nbr[i][t];
                        const int cj = scell[j], jy = (cj / gz) % gy, jx = cj / (gz * gy);
                        const int r = (jx - cx + 1) * 3 + (jy - cy + 1);
                        slot = Rg.base[r] + (j - Rg.start[r]);
                        if (j < Rg.start[r] || j >= Rg.start[r] + Rg.len[r]) { printf("range bug\n"); return 1; }
                    }
                    w[t % kChunk] = (unsigned short)slot;
                }
            }
        }
        if (S.slots > 65535) { printf("slots overflow\n"); return 1; }
        printf("L16 T=%d: max staged slots per block %d (%.1f KB per field)\n", S.T, S.slots, S.slots * 16.0 / 1024);
    }

    // constants
    Consts c;
    memset(&c, 0, sizeof(c));
    c.k.R = R; c.k.wA = 0.25f / (kPi * R * R * R); c.k.rcpR = 1.0f / R; c.k.fastQ = 1; c.k.fastDiv = 1; c.k.q2Free = 1; c.k.tCut = tCut;
    c.twoOverR = 2.0f / R; c.gradScale = 1.0f / (kPi * R * R * R * R * R);
    (void)bits_to_float;

    // device buffers
    float4 *dPos, *dVel; unsigned int* dRowsG; uint4 *dRowsGc, *dRowsL[2]; int *dCnt, *dTileChunks; BlockRanges* dRanges[2]; float* dOut;
    CK(hipMalloc(&dPos, sizeof(float4) * (n + 1))); CK(hipMalloc(&dVel, sizeof(float4) * (n + 1)));
    CK(hipMalloc(&dRowsG, sizeof(unsigned int) * rowsG.size())); CK(hipMalloc(&dRowsGc, sizeof(uint4) * rowsGc.size()));
    CK(hipMalloc(&dCnt, sizeof(int) * n)); CK(hipMalloc(&dTileChunks, sizeof(int) * numTiles)); CK(hipMalloc(&dOut, sizeof(float) * n));
    CK(hipMemcpy(dPos, posm.data(), sizeof(float4) * (n + 1), hipMemcpyHostToDevice));
    CK(hipMemcpy(dVel, vel4.data(), sizeof(float4) * (n + 1), hipMemcpyHostToDevice));
    CK(hipMemcpy(dRowsG, rowsG.data(), sizeof(unsigned int) * rowsG.size(), hipMemcpyHostToDevice));
    CK(hipMemcpy(dRowsGc, rowsGc.data(), sizeof(uint4) * rowsGc.size(), hipMemcpyHostToDevice));
    CK(hipMemcpy(dCnt, cnt.data(), sizeof(int) * n, hipMemcpyHostToDevice));
    CK(hipMemcpy(dTileChunks, tileChunks.data(), sizeof(int) * numTiles, hipMemcpyHostToDevice));
    int* dTileChunks4;
    {
        std::vector<int> t4(numTiles);
        for (int t = 0; t < numTiles; ++t) t4[t] = tileChunks[t] * 2;
        CK(hipMalloc(&dTileChunks4, sizeof(int) * numTiles));
        CK(hipMemcpy(dTileChunks4, t4.data(), sizeof(int) * numTiles, hipMemcpyHostToDevice));
    }
    for (int v = 0; v < 2; ++v) {
        CK(hipMalloc(&dRowsL[v], sizeof(uint4) * sets[v].rows.size()));
        CK(hipMalloc(&dRanges[v], sizeof(BlockRanges) * sets[v].ranges.size()));
        CK(hipMemcpy(dRowsL[v], sets[v].rows.data(), sizeof(uint4) * sets[v].rows.size(), hipMemcpyHostToDevice));
        CK(hipMemcpy(dRanges[v], sets[v].ranges.data(), sizeof(BlockRanges) * sets[v].ranges.size(), hipMemcpyHostToDevice));
    }

    // QG rows: [tile of 64/G particles][step][lane = particle * G + g][E entries]; padded with the dummy particle n
    struct QSet { int G, E, U; int numTiles; int capSteps; unsigned int* dRows; int* dSteps; };
    static const int qcfg[6][3] = {{4, 1, 2}, {4, 1, 4}, {4, 2, 2}, {4, 2, 1}, {8, 1, 4}, {2, 2, 2}};
    QSet qsets[6];
    for (int v = 0; v < 6; ++v) {
        QSet& Q = qsets[v];
        Q.G = qcfg[v][0]; Q.E = qcfg[v][1]; Q.U = qcfg[v][2];
        const int ppw = 64 / Q.G, per = Q.G * Q.E;
        Q.numTiles = (n + ppw - 1) / ppw;
        Q.capSteps = kCap / per;
        std::vector<unsigned int> rq((size_t)Q.numTiles * Q.capSteps * 64 * Q.E, (unsigned)n);
        std::vector<int> steps(Q.numTiles, 0);
        double slots = 0;
        for (int i = 0; i < n; ++i) {
            const int tile = i / ppw, p = i % ppw, m = cnt[i];
            steps[tile] = std::max(steps[tile], (m + per - 1) / per);
            for (int t = 0; t < m; ++t) {
                const int st = t / per, g = (t % per) / Q.E, k = t % Q.E;
                rq[(((size_t)tile * Q.capSteps + st) * 64 + p * Q.G + g) * Q.E + k] = (unsigned)nbr[i][t];
            }
        }
        for (int t = 0; t < Q.numTiles; ++t) slots += steps[t] * 64.0 * Q.E;
        printf("Q%dx%d rows: %d tiles, padding x%.2f\n", Q.G, Q.E, Q.numTiles, slots / pairs);
        CK(hipMalloc(&Q.dRows, sizeof(unsigned int) * rq.size())); CK(hipMalloc(&Q.dSteps, sizeof(int) * Q.numTiles));
        CK(hipMemcpy(Q.dRows, rq.data(), sizeof(unsigned int) * rq.size(), hipMemcpyHostToDevice));
        CK(hipMemcpy(Q.dSteps, steps.data(), sizeof(int) * Q.numTiles, hipMemcpyHostToDevice));
    }
    float4* dPV;
    {
        std::vector<float4> pv(2 * (size_t)(n + 1));
        for (int q = 0; q <= n; ++q) { pv[2 * (size_t)q] = posm[q]; pv[2 * (size_t)q + 1] = vel4[q]; }
        CK(hipMalloc(&dPV, sizeof(float4) * pv.size()));
        CK(hipMemcpy(dPV, pv.data(), sizeof(float4) * pv.size(), hipMemcpyHostToDevice));
    }
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));

    std::vector<float> ref[2][2], got(n);
    auto run = [&](const char* name, int exact, int two, auto&& launch) {
        CK(hipMemsetAsync(dOut, 0, sizeof(float) * n, st));
        for (int w = 0; w < 3; ++w) launch();
        CK(hipEventRecord(e0, st));
        for (int r = 0; r < reps; ++r) launch();
        CK(hipEventRecord(e1, st));
        CK(hipEventSynchronize(e1));
        CK(hipGetLastError());
        float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
        ms /= reps;
        CK(hipMemcpy(got.data(), dOut, sizeof(float) * n, hipMemcpyDeviceToHost));
        std::vector<float>& rf = ref[exact][two];
        const char* verdict = "reference";
        if (rf.empty()) rf = got;
        else {
            long long diffBits = 0; double maxRel = 0;
            for (int i = 0; i < n; ++i) {
                if (memcmp(&got[i], &rf[i], 4) != 0) ++diffBits;
                const double den = std::max(1e-12, (double)fabsf(rf[i]));
                maxRel = std::max(maxRel, fabs((double)got[i] - rf[i]) / den);
            }
            static char buf[96];
            snprintf(buf, sizeof(buf), "%lld values differ bitwise, max rel %.2e", diffBits, maxRel);
            verdict = buf;
        }
        printf("%-40s %8.3f ms   %7.1f Gpair/s   alg %6.1f GB/s (44 B/particle)   [%s]\n", name, ms, pairs / ms * 1e-6,
               44.0 * n / ms * 1e-6, verdict);
    };

    const unsigned gridG = xcd_grid(n, 256);

    // ---- row builders (argv[3] = 'B'): lane-per-particle over global memory vs the same walk over a brick's LDS stage ----------
    if (argc > 3 && argv[3][0] == 'B') {
        const int capB = 48;
        int* dCs; CK(hipMalloc(&dCs, sizeof(int) * (C + 2))); CK(hipMemcpy(dCs, cs.data(), sizeof(int) * (C + 2), hipMemcpyHostToDevice));
        unsigned int *dRowsA, *dRowsB2; int *dCntA, *dCntB;
        const size_t rowWords = (size_t)numTiles * capB * 64;
        CK(hipMalloc(&dRowsA, 4 * rowWords)); CK(hipMalloc(&dRowsB2, 4 * rowWords)); CK(hipMalloc(&dCntA, 4 * n)); CK(hipMalloc(&dCntB, 4 * n));
        CK(hipMemset(dRowsA, 0, 4 * rowWords)); CK(hipMemset(dRowsB2, 0, 4 * rowWords));
        auto timeit = [&](const char* name, auto&& launch) {
            for (int w = 0; w < 2; ++w) launch();
            CK(hipEventRecord(e0, st));
            for (int r = 0; r < reps; ++r) launch();
            CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1)); CK(hipGetLastError());
            float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
            printf("%-44s %8.3f ms   (%.2f G candidate tests/s at ~%d per particle)\n", name, ms / reps, 1e-6 * 6.8 * 27 * n / (ms / reps), (int)(6.8 * 27));
        };
        timeit("BUILD global lane-per-particle", [&] {
            hipLaunchKernelGGL(k_build_global, dim3(gridG), dim3(256), 0, st, dPos, dCs, gx, gy, gz, cellLength, tCut, dRowsA, dCntA, n, numTiles, capB); });
        {
            std::vector<F3> p3(n + 4);
            for (int q = 0; q <= n; ++q) p3[q] = F3{posm[q].x, posm[q].y, posm[q].z};
            F3* dP3; CK(hipMalloc(&dP3, 12 * p3.size())); CK(hipMemcpy(dP3, p3.data(), 12 * p3.size(), hipMemcpyHostToDevice));
            timeit("BUILD global, float3 candidates (3 x 16 B / 4)", [&] {
                hipLaunchKernelGGL(k_build_global3, dim3(gridG), dim3(256), 0, st, dP3, dCs, gx, gy, gz, cellLength, tCut, dRowsB2, dCntB, n, numTiles, capB); });
            std::vector<int> ca(n), cb(n); std::vector<unsigned int> ra(rowWords), rb(rowWords);
            CK(hipMemcpy(ca.data(), dCntA, 4 * n, hipMemcpyDeviceToHost)); CK(hipMemcpy(cb.data(), dCntB, 4 * n, hipMemcpyDeviceToHost));
            CK(hipMemcpy(ra.data(), dRowsA, 4 * rowWords, hipMemcpyDeviceToHost)); CK(hipMemcpy(rb.data(), dRowsB2, 4 * rowWords, hipMemcpyDeviceToHost));
            long long badCnt = 0, badEnt = 0;
            for (int i = 0; i < n; ++i) {
                if (ca[i] != cb[i]) { ++badCnt; continue; }
                for (int k = 0; k < std::min(ca[i], capB); ++k) {
                    const size_t at = ((size_t)(i >> 6) * capB) * 64u + (size_t)(i & 63) * 4u + (size_t)(k >> 2) * 256u + (k & 3);
                    if (ra[at] != rb[at]) ++badEnt;
                }
            }
            printf("   float3 rows vs the float4 builder: %lld counts differ, %lld entries differ\n", badCnt, badEnt);
            CK(hipFree(dP3));
        }
        struct Shape { int bx, by, bz, T; };
        const Shape shapes[] = {{4, 4, 4, 256}, {4, 4, 4, 512}, {4, 4, 8, 512}, {2, 2, 8, 256}};
        for (const Shape& S : shapes) {
            const int nbx = (gx + S.bx - 1) / S.bx, nby = (gy + S.by - 1) / S.by, nbz = (gz + S.bz - 1) / S.bz;
            std::vector<BrickDesc> descs; std::vector<BrickRun> bruns; std::vector<int> ownIdx, origin;
            int maxStaged = 0;
            for (int bxi = 0; bxi < nbx; ++bxi) for (int byi = 0; byi < nby; ++byi) for (int bzi = 0; bzi < nbz; ++bzi) {
                const int x0 = bxi * S.bx, y0 = byi * S.by, z0 = bzi * S.bz;
                const int x1 = std::min(x0 + S.bx, gx), y1 = std::min(y0 + S.by, gy), z1 = std::min(z0 + S.bz, gz);
                BrickDesc D; D.runFirst = (int)bruns.size(); D.numRuns = 0; D.staged = 0; D.own = 0; D.ownFirst = (int)ownIdx.size(); D.rowBase = 0;
                std::vector<int> own;
                for (int X = x0; X < x1; ++X) for (int Y = y0; Y < y1; ++Y)
                    for (int j = cs[(X * gy + Y) * gz + z0]; j < cs[(X * gy + Y) * gz + z1]; ++j) own.push_back(j);
                if (own.empty()) continue;
                const int zlo = std::max(z0 - 1, 0), zhi = std::min(z1, gz - 1);
                for (int X = std::max(x0 - 1, 0); X <= std::min(x1, gx - 1); ++X) for (int Y = std::max(y0 - 1, 0); Y <= std::min(y1, gy - 1); ++Y) {
                    const int a = cs[(X * gy + Y) * gz + zlo], b = cs[(X * gy + Y) * gz + zhi + 1];
                    bruns.push_back({a, b - a, D.staged});          // dense table: empty columns keep a zero-length run
                    D.staged += b - a; D.numRuns++;
                }
                D.own = (int)own.size(); D.rounds = (D.own + S.T - 1) / S.T;
                for (int j : own) ownIdx.push_back(j);
                origin.push_back(x0); origin.push_back(y0); origin.push_back(z0);
                maxStaged = std::max(maxStaged, D.staged);
                descs.push_back(D);
            }
            const int numBricks = (int)descs.size();
            BrickDesc* dDesc; BrickRun* dRuns; int *dOwnIdx, *dOrigin;
            CK(hipMalloc(&dDesc, sizeof(BrickDesc) * descs.size())); CK(hipMalloc(&dRuns, sizeof(BrickRun) * bruns.size()));
            CK(hipMalloc(&dOwnIdx, 4 * ownIdx.size())); CK(hipMalloc(&dOrigin, 4 * origin.size()));
            CK(hipMemcpy(dDesc, descs.data(), sizeof(BrickDesc) * descs.size(), hipMemcpyHostToDevice));
            CK(hipMemcpy(dRuns, bruns.data(), sizeof(BrickRun) * bruns.size(), hipMemcpyHostToDevice));
            CK(hipMemcpy(dOwnIdx, ownIdx.data(), 4 * ownIdx.size(), hipMemcpyHostToDevice));
            CK(hipMemcpy(dOrigin, origin.data(), 4 * origin.size(), hipMemcpyHostToDevice));
            const unsigned grid = (unsigned)(((numBricks + 7) / 8) * 8);
            const size_t ldsBytes = (size_t)(maxStaged + 1) * 16;
            char nm[96]; snprintf(nm, sizeof(nm), "BUILD brick %dx%dx%d T=%d (%zu KB LDS)", S.bx, S.by, S.bz, S.T, ldsBytes / 1024);
            if (ldsBytes > 150 * 1024) { printf("%s: too much LDS\n", nm); continue; }
#define LBB(TT) do { if (ldsBytes > 48 * 1024) CK(hipFuncSetAttribute((const void*)k_build_brick<TT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsBytes)); \
                     hipLaunchKernelGGL((k_build_brick<TT>), dim3(grid), dim3(TT), ldsBytes, st, dPos, dCs, gx, gy, gz, cellLength, tCut, dDesc, dRuns, dOwnIdx, dOrigin, dRowsB2, dCntB, numBricks, capB, S.bx, S.by, S.bz); } while (0)
            timeit(nm, [&] { if (S.T == 256) LBB(256); else LBB(512); });
#undef LBB
            std::vector<int> ca(n), cb(n);
            CK(hipMemcpy(ca.data(), dCntA, 4 * n, hipMemcpyDeviceToHost)); CK(hipMemcpy(cb.data(), dCntB, 4 * n, hipMemcpyDeviceToHost));
            std::vector<unsigned int> ra(rowWords), rb(rowWords);
            CK(hipMemcpy(ra.data(), dRowsA, 4 * rowWords, hipMemcpyDeviceToHost)); CK(hipMemcpy(rb.data(), dRowsB2, 4 * rowWords, hipMemcpyDeviceToHost));
            long long badCnt = 0, badEnt = 0;
            for (int i = 0; i < n; ++i) {
                if (ca[i] != cb[i]) { ++badCnt; continue; }
                for (int k = 0; k < std::min(ca[i], capB); ++k) {
                    const size_t at = ((size_t)(i >> 6) * capB) * 64u + (size_t)(i & 63) * 4u + (size_t)(k >> 2) * 256u + (k & 3);
                    if (ra[at] != rb[at]) ++badEnt;
                }
            }
            printf("   rows vs the global builder: %lld counts differ, %lld entries differ\n", badCnt, badEnt);
            CK(hipFree(dDesc)); CK(hipFree(dRuns)); CK(hipFree(dOwnIdx)); CK(hipFree(dOrigin));
        }
        return 0;
    }
    for (int exact = 1; exact >= 0; --exact) {
        for (int two = 1; two >= 0; --two) {
            char nm[64];
            auto tag = [&](const char* path) { snprintf(nm, sizeof(nm), "%s %s %s", path, exact ? "exact" : "tol", two ? "2f" : "1f"); return nm; };
#define PICK(KERNEL, ...)                                                                                   \
    do {                                                                                                    \
        if (exact) { if (two) hipLaunchKernelGGL((KERNEL<true, true>), __VA_ARGS__); else hipLaunchKernelGGL((KERNEL<true, false>), __VA_ARGS__); } \
        else { if (two) hipLaunchKernelGGL((KERNEL<false, true>), __VA_ARGS__); else hipLaunchKernelGGL((KERNEL<false, false>), __VA_ARGS__); }      \
    } while (0)
            run(tag("G32 "), exact, two, [&] { PICK(k_g32, dim3(gridG), dim3(256), 0, st, c, dPos, dVel, dRowsG, dCnt, dOut, n, numTiles); });
            run(tag("G32c"), exact, two, [&] { PICK(k_g32c, dim3(gridG), dim3(256), 0, st, c, dPos, dVel, dRowsGc, dTileChunks4, dOut, n, numTiles, kCap / 4); });
            run(tag("CQ  "), exact, two, [&] { PICK(k_cq, dim3(gridG), dim3(256), 0, st, c, dPos, dVel, reinterpret_cast<const unsigned int*>(dRowsGc), dCnt, dOut, n, numTiles, kCap / 4); });
            run(tag("G32i"), exact, two, [&] { PICK(k_g32i, dim3(gridG), dim3(256), 0, st, c, dPV, dRowsGc, dTileChunks4, dOut, n, numTiles, kCap / 4); });
            for (int v = 0; v < 6; ++v) {
                const QSet& Q = qsets[v];
                const unsigned gridQ = xcd_grid(Q.numTiles * 64, 256);
                snprintf(nm, sizeof(nm), "Q%dx%d u%d %s %s", Q.G, Q.E, Q.U, exact ? "exact" : "tol", two ? "2f" : "1f");
#define LQ(GG, EE, UU, EX, TW) hipLaunchKernelGGL((k_qg<GG, EE, UU, EX, TW>), dim3(gridQ), dim3(256), 0, st, c, dPos, dVel, Q.dRows, Q.dSteps, dOut, n, Q.numTiles, Q.capSteps)
#define LQ2(GG, EE, UU) do { if (exact) { if (two) LQ(GG, EE, UU, true, true); else LQ(GG, EE, UU, true, false); } else { if (two) LQ(GG, EE, UU, false, true); else LQ(GG, EE, UU, false, false); } } while (0)
                run(nm, exact, two, [&] {
                    switch (v) {
                    case 0: LQ2(4, 1, 2); break;
                    case 1: LQ2(4, 1, 4); break;
                    case 2: LQ2(4, 2, 2); break;
                    case 3: LQ2(4, 2, 1); break;
                    case 4: LQ2(8, 1, 4); break;
                    default: LQ2(2, 2, 2); break;
                    }
                });
#undef LQ2
#undef LQ
            }
            {   // the quad walk gathering 12-byte records
                static F3 *dPos3 = nullptr, *dVel3 = nullptr;
                const QSet& Q = qsets[1];
                if (!dPos3) {
                    std::vector<F3> p3(n + 1), v3h(n + 1);
                    for (int q = 0; q <= n; ++q) { p3[q] = F3{posm[q].x, posm[q].y, posm[q].z}; v3h[q] = F3{vel4[q].x, vel4[q].y, vel4[q].z}; }
                    CK(hipMalloc(&dPos3, 12 * (size_t)(n + 1))); CK(hipMalloc(&dVel3, 12 * (size_t)(n + 1)));
                    CK(hipMemcpy(dPos3, p3.data(), 12 * (size_t)(n + 1), hipMemcpyHostToDevice));
                    CK(hipMemcpy(dVel3, v3h.data(), 12 * (size_t)(n + 1), hipMemcpyHostToDevice));
                }
                const unsigned gridQ = xcd_grid(Q.numTiles * 64, 256);
                const float m0 = posm[0].w;
                snprintf(nm, sizeof(nm), "Q4x1 u4 float3 gathers %s %s", exact ? "exact" : "tol", two ? "2f" : "1f");
                run(nm, exact, two, [&] {
                    if (exact) { if (two) hipLaunchKernelGGL((k_qg3<true, true>), dim3(gridQ), dim3(256), 0, st, c, dPos3, dVel3, m0, Q.dRows, Q.dSteps, dOut, n, Q.numTiles, Q.capSteps);
                                 else hipLaunchKernelGGL((k_qg3<true, false>), dim3(gridQ), dim3(256), 0, st, c, dPos3, dVel3, m0, Q.dRows, Q.dSteps, dOut, n, Q.numTiles, Q.capSteps); }
                    else { if (two) hipLaunchKernelGGL((k_qg3<false, true>), dim3(gridQ), dim3(256), 0, st, c, dPos3, dVel3, m0, Q.dRows, Q.dSteps, dOut, n, Q.numTiles, Q.capSteps);
                           else hipLaunchKernelGGL((k_qg3<false, false>), dim3(gridQ), dim3(256), 0, st, c, dPos3, dVel3, m0, Q.dRows, Q.dSteps, dOut, n, Q.numTiles, Q.capSteps); }
                });
            }
            {   // the quad walk on 16-bit tile-relative rows
                static unsigned short* dRows16 = nullptr; static int* dBases = nullptr;
                const QSet& Q = qsets[1];
                if (!dRows16) {
                    std::vector<unsigned short> r16((size_t)Q.numTiles * Q.capSteps * 64, (unsigned short)0x9000u);
                    std::vector<int> bases((size_t)Q.numTiles * 10, n);
                    bool fits = true;
                    for (int tile = 0; tile < Q.numTiles; ++tile) {
                        const int i0 = tile * 16, i1 = std::min(n, i0 + 16) - 1;
                        if (i0 >= n) break;
                        const int cmin = scell[i0], cmax = scell[i1];
                        int lo9[9], hi9[9];
                        for (int m = 0; m < 9; ++m) {
                            const int off = ((m / 3 - 1) * gy + (m % 3 - 1)) * gz;
                            const int lo = std::min(std::max(cmin + off - 1, 0), C), hi = std::min(std::max(cmax + off + 1, 0), C - 1);
                            lo9[m] = cs[lo]; hi9[m] = hi >= lo ? cs[hi + 1] : cs[lo];
                            bases[(size_t)tile * 10 + m] = lo9[m];
                        }
                        bases[(size_t)tile * 10 + 9] = n;
                        for (int p = 0; p < 16 && i0 + p < n; ++p) {
                            const int i = i0 + p, m_ = cnt[i];
                            for (int t = 0; t < m_; ++t) {
                                const int j = nbr[i][t];
                                int col = -1;
                                for (int m = 0; m < 9; ++m) if (j >= lo9[m] && j < hi9[m]) { col = m; break; }
                                if (col < 0 || j - lo9[col] > 4095) { fits = false; continue; }
                                r16[((size_t)tile * Q.capSteps + t / 4) * 64 + p * 4 + (t % 4)] = (unsigned short)((col << 12) | (j - lo9[col]));
                            }
                        }
                    }
                    printf("Q4x1 r16 rows: %s\n", fits ? "every entry fits 4+12 bits" : "SOME ENTRIES DO NOT FIT");
                    CK(hipMalloc(&dRows16, 2 * r16.size())); CK(hipMalloc(&dBases, 4 * bases.size()));
                    CK(hipMemcpy(dRows16, r16.data(), 2 * r16.size(), hipMemcpyHostToDevice));
                    CK(hipMemcpy(dBases, bases.data(), 4 * bases.size(), hipMemcpyHostToDevice));
                }
                const unsigned gridQ = xcd_grid(Q.numTiles * 64, 256);
                snprintf(nm, sizeof(nm), "Q4x1 u4 rows16 %s %s", exact ? "exact" : "tol", two ? "2f" : "1f");
                run(nm, exact, two, [&] {
                    if (exact) { if (two) hipLaunchKernelGGL((k_qg16<true, true>), dim3(gridQ), dim3(256), 0, st, c, dPos, dVel, dRows16, Q.dSteps, dBases, dOut, n, Q.numTiles, Q.capSteps);
                                 else hipLaunchKernelGGL((k_qg16<true, false>), dim3(gridQ), dim3(256), 0, st, c, dPos, dVel, dRows16, Q.dSteps, dBases, dOut, n, Q.numTiles, Q.capSteps); }
                    else { if (two) hipLaunchKernelGGL((k_qg16<false, true>), dim3(gridQ), dim3(256), 0, st, c, dPos, dVel, dRows16, Q.dSteps, dBases, dOut, n, Q.numTiles, Q.capSteps);
                           else hipLaunchKernelGGL((k_qg16<false, false>), dim3(gridQ), dim3(256), 0, st, c, dPos, dVel, dRows16, Q.dSteps, dBases, dOut, n, Q.numTiles, Q.capSteps); }
                });
            }
            {   // the quad walk on interleaved records (rows of the Q4x1 u4 set)
                const QSet& Q = qsets[1];
                const unsigned gridQ = xcd_grid(Q.numTiles * 64, 256);
                snprintf(nm, sizeof(nm), "Q4x1 u4 interleaved %s %s", exact ? "exact" : "tol", two ? "2f" : "1f");
                run(nm, exact, two, [&] {
                    if (exact) { if (two) hipLaunchKernelGGL((k_qgi<4, true, true>), dim3(gridQ), dim3(256), 0, st, c, dPV, Q.dRows, Q.dSteps, dOut, n, Q.numTiles, Q.capSteps);
                                 else hipLaunchKernelGGL((k_qgi<4, true, false>), dim3(gridQ), dim3(256), 0, st, c, dPV, Q.dRows, Q.dSteps, dOut, n, Q.numTiles, Q.capSteps); }
                    else { if (two) hipLaunchKernelGGL((k_qgi<4, false, true>), dim3(gridQ), dim3(256), 0, st, c, dPV, Q.dRows, Q.dSteps, dOut, n, Q.numTiles, Q.capSteps);
                           else hipLaunchKernelGGL((k_qgi<4, false, false>), dim3(gridQ), dim3(256), 0, st, c, dPV, Q.dRows, Q.dSteps, dOut, n, Q.numTiles, Q.capSteps); }
                });
            }
            if (two) {   // ... and with paired half gathers (one instruction per record instead of two)
                const QSet& Q = qsets[1];
                const unsigned gridQ = xcd_grid(Q.numTiles * 64, 256);
                snprintf(nm, sizeof(nm), "Q4x1 u4 paired-half %s 2f", exact ? "exact" : "tol");
                run(nm, exact, two, [&] {
                    if (exact) hipLaunchKernelGGL((k_qgx<4, true>), dim3(gridQ), dim3(256), 0, st, c, dPV, Q.dRows, Q.dSteps, dOut, n, Q.numTiles, Q.capSteps);
                    else hipLaunchKernelGGL((k_qgx<4, false>), dim3(gridQ), dim3(256), 0, st, c, dPV, Q.dRows, Q.dSteps, dOut, n, Q.numTiles, Q.capSteps);
                });
            }
            for (int v = 0; v < ((argc > 5 && argv[3][0] != 'b') ? 2 : 0); ++v) {
                const int T = sets[v].T, slots = sets[v].slots + 1;
                const size_t ldsBytes = (size_t)slots * 16 * (two ? 2 : 1);
                const unsigned grid = xcd_grid(n, T);
                snprintf(nm, sizeof(nm), "L16 T=%d %s %s (%zu KB)", T, exact ? "exact" : "tol", two ? "2f" : "1f", ldsBytes / 1024);
#define LAUNCH_L(TT, EX, TW) hipLaunchKernelGGL((k_l16<TT, EX, TW>), dim3(grid), dim3(TT), ldsBytes, st, c, dPos, dVel, dRowsL[v], dTileChunks, dRanges[v], dOut, n, numTiles, slots)
                auto launch = [&] {
                    if (T == 256) { if (exact) { if (two) LAUNCH_L(256, true, true); else LAUNCH_L(256, true, false); } else { if (two) LAUNCH_L(256, false, true); else LAUNCH_L(256, false, false); } }
                    else { if (exact) { if (two) LAUNCH_L(128, true, true); else LAUNCH_L(128, true, false); } else { if (two) LAUNCH_L(128, false, true); else LAUNCH_L(128, false, false); } }
                };
                if (ldsBytes > 64 * 1024) {
                    const void* fn = T == 256 ? (exact ? (two ? (const void*)k_l16<256, true, true> : (const void*)k_l16<256, true, false>)
                                                        : (two ? (const void*)k_l16<256, false, true> : (const void*)k_l16<256, false, false>))
                                              : (exact ? (two ? (const void*)k_l16<128, true, true> : (const void*)k_l16<128, true, false>)
                                                        : (two ? (const void*)k_l16<128, false, true> : (const void*)k_l16<128, false, false>));
                    CK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsBytes));
                }
                run(nm, exact, two, launch);
#undef LAUNCH_L
            }
#undef PICK
        }
    }


    // ---- BRICK variants (argv[6] present): brick shapes BXxBYxBZ, T threads ---------------------------------------------------
    if (argc > 3 && argv[3][0] == 'b') {
        struct Shape { int bx, by, bz, T; };
        const Shape shapes[] = {{4, 4, 4, 256}, {4, 4, 4, 512}, {4, 4, 8, 512}, {4, 4, 8, 1024}, {2, 2, 8, 256}, {3, 3, 6, 256}, {6, 6, 4, 512}};
        for (const Shape& S : shapes) {
            const int nbx = (gx + S.bx - 1) / S.bx, nby = (gy + S.by - 1) / S.by, nbz = (gz + S.bz - 1) / S.bz;
            std::vector<BrickDesc> descs; std::vector<BrickRun> bruns; std::vector<int> ownIdx; std::vector<unsigned short> ownSlot;
            std::vector<unsigned char> wch;
            std::vector<int> slotOf(n, -1);
            int maxStaged = 0, maxOwn = 0; long long stagedSum = 0, ownSum = 0; long long rowChunkRows = 0;
            struct Tmp { int first, count; };
            std::vector<std::vector<unsigned short>> rowsPerBrick;   // flattened later
            std::vector<uint4> rowsFlat;
            bool ok = true;
            for (int bxi = 0; bxi < nbx && ok; ++bxi) for (int byi = 0; byi < nby && ok; ++byi) for (int bzi = 0; bzi < nbz && ok; ++bzi) {
                const int x0 = bxi * S.bx, y0 = byi * S.by, z0 = bzi * S.bz;
                const int x1 = std::min(x0 + S.bx, gx), y1 = std::min(y0 + S.by, gy), z1 = std::min(z0 + S.bz, gz);   // exclusive
                BrickDesc D; D.runFirst = (int)bruns.size(); D.numRuns = 0; D.staged = 0; D.own = 0; D.ownFirst = (int)ownIdx.size();
                // own particles first (to skip empty bricks)
                std::vector<int> own;
                for (int X = x0; X < x1; ++X) for (int Y = y0; Y < y1; ++Y) {
                    const int a = cs[(X * gy + Y) * gz + z0], b = cs[(X * gy + Y) * gz + z1 - 1 + 1];
                    for (int j = a; j < b; ++j) own.push_back(j);
                }
                if (own.empty()) continue;
                const int zlo = std::max(z0 - 1, 0), zhi = std::min(z1, gz - 1);
                std::vector<std::pair<int, int>> touched;
                for (int X = std::max(x0 - 1, 0); X <= std::min(x1, gx - 1); ++X) for (int Y = std::max(y0 - 1, 0); Y <= std::min(y1, gy - 1); ++Y) {
                    const int a = cs[(X * gy + Y) * gz + zlo], b = cs[(X * gy + Y) * gz + zhi + 1];
                    if (b == a) continue;
                    bruns.push_back({a, b - a, D.staged});
                    for (int j = a; j < b; ++j) slotOf[j] = D.staged + (j - a);
                    touched.push_back({a, b});
                    D.staged += b - a; D.numRuns++;
                }
                if (D.staged + 1 > 65535) { ok = false; break; }
                D.own = (int)own.size();
                D.rounds = (D.own + S.T - 1) / S.T;
                if (D.rounds > 8) { ok = false; break; }
                D.rowBase = (int)rowChunkRows;
                for (int j : own) { ownIdx.push_back(j); ownSlot.push_back((unsigned short)slotOf[j]); }
                const int blk = (int)descs.size();
                wch.resize((size_t)(blk + 1) * 8 * 17, 0);
                for (int rd = 0; rd < D.rounds; ++rd) {
                    int roundMax = 0;
                    for (int t = 0; t < S.T; ++t) {
                        const int p = rd * S.T + t;
                        const int m = p < D.own ? cnt[own[p]] : 0;
                        const int ch = (m + kChunk - 1) / kChunk;
                        unsigned char& wv = wch[(size_t)(blk * 8 + rd) * 17 + t / 64];
                        wv = (unsigned char)std::max<int>(wv, ch);
                        roundMax = std::max(roundMax, ch);
                    }
                    wch[(size_t)(blk * 8 + rd) * 17 + 16] = (unsigned char)roundMax;
                    const size_t at = rowsFlat.size();
                    rowsFlat.resize(at + (size_t)roundMax * S.T, make_uint4(0, 0, 0, 0));
                    for (int t = 0; t < S.T; ++t) {
                        const int p = rd * S.T + t;
                        const int i = p < D.own ? own[p] : -1;
                        const int m = i >= 0 ? cnt[i] : 0;
                        for (int k = 0; k < roundMax * kChunk; ++k) {
                            int slot = D.staged;
                            if (k < m) { slot = slotOf[nbr[i][k]]; if (slot < 0) { printf("brick: neighbour outside the halo\n"); return 1; } }
                            reinterpret_cast<unsigned short*>(&rowsFlat[at + (size_t)(k / kChunk) * S.T + t])[k % kChunk] = (unsigned short)slot;
                        }
                    }
                    rowChunkRows += roundMax;
                }
                for (auto& ab : touched) for (int j = ab.first; j < ab.second; ++j) slotOf[j] = -1;
                maxStaged = std::max(maxStaged, D.staged); maxOwn = std::max(maxOwn, D.own);
                stagedSum += D.staged; ownSum += D.own;
                descs.push_back(D);
            }
            if (!ok) { printf("BRICK %dx%dx%d T=%d: shape does not fit (slots or rounds)\n", S.bx, S.by, S.bz, S.T); continue; }
            const int numBricks = (int)descs.size(), slots = maxStaged + 1;
            double waveIt = 0; for (size_t k = 0; k < wch.size(); k += 17) for (int w = 0; w < 16; ++w) waveIt += wch[k + w] * 8.0 * 64.0;
            printf("BRICK %dx%dx%d T=%d: %d bricks, own avg %.0f max %d, staged avg %.0f max %d (x%.2f of own, %.1f KB per field), row slots x%.2f of pairs\n",
                   S.bx, S.by, S.bz, S.T, numBricks, (double)ownSum / numBricks, maxOwn, (double)stagedSum / numBricks, maxStaged,
                   (double)stagedSum / ownSum, slots * 16.0 / 1024, waveIt / pairs);
            BrickDesc* dDesc; BrickRun* dRuns; int* dOwnIdx; unsigned short* dOwnSlot; uint4* dRowsB; unsigned char* dWch;
            CK(hipMalloc(&dDesc, sizeof(BrickDesc) * descs.size())); CK(hipMalloc(&dRuns, sizeof(BrickRun) * bruns.size()));
            CK(hipMalloc(&dOwnIdx, sizeof(int) * ownIdx.size())); CK(hipMalloc(&dOwnSlot, sizeof(unsigned short) * ownSlot.size()));
            CK(hipMalloc(&dRowsB, sizeof(uint4) * std::max<size_t>(rowsFlat.size(), 1))); CK(hipMalloc(&dWch, wch.size()));
            CK(hipMemcpy(dDesc, descs.data(), sizeof(BrickDesc) * descs.size(), hipMemcpyHostToDevice));
            CK(hipMemcpy(dRuns, bruns.data(), sizeof(BrickRun) * bruns.size(), hipMemcpyHostToDevice));
            CK(hipMemcpy(dOwnIdx, ownIdx.data(), sizeof(int) * ownIdx.size(), hipMemcpyHostToDevice));
            CK(hipMemcpy(dOwnSlot, ownSlot.data(), sizeof(unsigned short) * ownSlot.size(), hipMemcpyHostToDevice));
            CK(hipMemcpy(dRowsB, rowsFlat.data(), sizeof(uint4) * rowsFlat.size(), hipMemcpyHostToDevice));
            CK(hipMemcpy(dWch, wch.data(), wch.size(), hipMemcpyHostToDevice));
            const unsigned grid = (unsigned)(((numBricks + 7) / 8) * 8);
            for (int exact = 1; exact >= 0; --exact)
                for (int two = 1; two >= 0; --two) {
                    const size_t ldsBytes = (size_t)slots * 16 * (two ? 2 : 1);
                    if (ldsBytes > 160 * 1024) { printf("   (%s %s: %zu KB of LDS, skipped)\n", exact ? "exact" : "tol", two ? "2f" : "1f", ldsBytes / 1024); continue; }
                    char nm[96];
                    snprintf(nm, sizeof(nm), "BRICK %dx%dx%d T=%d %s %s (%zu KB)", S.bx, S.by, S.bz, S.T, exact ? "exact" : "tol", two ? "2f" : "1f", ldsBytes / 1024);
#define LB(TT, EX, TW) do { if (ldsBytes > 64 * 1024) CK(hipFuncSetAttribute((const void*)k_brick<TT, EX, TW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsBytes)); \
                            hipLaunchKernelGGL((k_brick<TT, EX, TW>), dim3(grid), dim3(TT), ldsBytes, st, c, dPos, dVel, dDesc, dRuns, dOwnIdx, dOwnSlot, dRowsB, dWch, dOut, numBricks, slots); } while (0)
#define LB2(TT) do { if (exact) { if (two) LB(TT, true, true); else LB(TT, true, false); } else { if (two) LB(TT, false, true); else LB(TT, false, false); } } while (0)
                    run(nm, exact, two, [&] { if (S.T == 256) LB2(256); else if (S.T == 512) LB2(512); else LB2(1024); });
#undef LB2
#undef LB
                }
            CK(hipFree(dDesc)); CK(hipFree(dRuns)); CK(hipFree(dOwnIdx)); CK(hipFree(dOwnSlot)); CK(hipFree(dRowsB)); CK(hipFree(dWch));
        }
    }

    // ---- tile schedules: which tiles share a CU's L1 / an XCD's L2 at the same time ---------------------------
    if (argc > 4 && argv[3][0] != 'b') {
        std::vector<int> order(numTiles);
        int* dOrder; CK(hipMalloc(&dOrder, sizeof(int) * numTiles));
        auto spread3 = [](unsigned long long v) { unsigned long long r = 0; for (int b = 0; b < 20; ++b) r |= ((v >> b) & 1ull) << (3 * b); return r; };
        for (int sched = 0; sched < 5; ++sched) {
            static const char* names[5] = {"linear", "ychunk8", "morton(x,y,z/8)", "morton(x,y,z/16)", "brick 2x2 cols"};
            std::vector<std::pair<unsigned long long, int>> keyed(numTiles);
            for (int t = 0; t < numTiles; ++t) {
                const int c0 = scell[std::min(n - 1, t * 64)], cz = c0 % gz, cy = (c0 / gz) % gy, cx = c0 / (gz * gy);
                unsigned long long key = (unsigned long long)t;
                if (sched == 1) key = ((unsigned long long)(cy / ((gy + 7) / 8)) * gx + cx) * (unsigned long long)numTiles + t;
                if (sched == 2) key = (spread3(cx) << 2) | (spread3(cy) << 1) | spread3(cz / 8);
                if (sched == 3) key = (spread3(cx) << 2) | (spread3(cy) << 1) | spread3(cz / 16);
                if (sched == 4) key = (((unsigned long long)(cx / 2) * gy + (cy / 2)) * gz + cz / 8) * 4ull + (cx & 1) * 2 + (cy & 1);
                keyed[t] = {key, t};
            }
            std::stable_sort(keyed.begin(), keyed.end());
            for (int t = 0; t < numTiles; ++t) order[t] = keyed[t].second;
            CK(hipMemcpy(dOrder, order.data(), sizeof(int) * numTiles, hipMemcpyHostToDevice));
            for (int bs = 0; bs < 3; ++bs) {
                const int T = bs == 0 ? 256 : (bs == 1 ? 512 : 1024);
                const unsigned grid = xcd_grid(n, T);
                for (int exact = 1; exact >= 0; --exact)
                    for (int two = 1; two >= 0; --two) {
                        ref[exact][two].clear();
                        char nm[80];
                        snprintf(nm, sizeof(nm), "%s T=%d %s %s", names[sched], T, exact ? "exact" : "tol", two ? "2f" : "1f");
#define LO(TT, EX, TW) hipLaunchKernelGGL((k_g32o<TT, EX, TW>), dim3(grid), dim3(TT), 0, st, c, dPos, dVel, dRowsGc, dTileChunks4, dOrder, dOut, n, numTiles, kCap / 4)
#define LO2(TT) do { if (exact) { if (two) LO(TT, true, true); else LO(TT, true, false); } else { if (two) LO(TT, false, true); else LO(TT, false, false); } } while (0)
                        run(nm, exact, two, [&] { if (T == 256) LO2(256); else if (T == 512) LO2(512); else LO2(1024); });
#undef LO2
#undef LO
                    }
            }
        }
    }

    // ---- address-pattern probes: what does one divergent 16-byte gather instruction cost? ------------------
    // Same kernel (G32c, tolerance arithmetic), synthetic rows of exactly 32 entries per lane:
    //   same      every lane of the wave reads ONE record                      (pure issue cost)
    //   contig    lane l reads record base + l                                  (8 cache lines of 128 B)
    //   win80     every lane reads a pseudo-random record of an 80-record window around the tile (11 lines)
    //   cell24    lanes of the same cell (groups of 8) share a 24-record window (what phase-aligned rows give)
    //   real      the scene's rows as they are
    //   aligned   the scene's rows with every (dx,dy) run padded to the wave's longest run, so all lanes are in
    //             the same run at the same time
    if (argc > 3 && argv[3][0] == 'p') {
        const int capP = 96, cap4 = capP / 4;
        std::vector<uint4> rows((size_t)numTiles * cap4 * 64);
        std::vector<int> chunks4(numTiles);
        uint4* dRows; int* dChunks;
        CK(hipMalloc(&dRows, sizeof(uint4) * rows.size())); CK(hipMalloc(&dChunks, sizeof(int) * numTiles));
        auto hash = [](unsigned int a, unsigned int b, unsigned int c3) { unsigned int v = a * 0x9E3779B1u ^ b * 0x85EBCA77u ^ c3 * 0xC2B2AE3Du; v ^= v >> 15; v *= 0x2C1B3C6Du; v ^= v >> 12; return v; };
        for (int probe = 0; probe < 6; ++probe) {
            static const char* names[6] = {"same", "contig", "win80", "cell24", "real", "aligned"};
            std::fill(rows.begin(), rows.end(), make_uint4(n, n, n, n));
            double iters = 0;
            for (int tile = 0; tile < numTiles; ++tile) {
                const int i0 = tile * 64;
                int len = 32;
                if (probe == 4) len = tileChunks[tile] * kChunk;
                std::vector<int> off(10, 0);
                if (probe == 5) {
                    for (int r = 0; r < 9; ++r) {
                        int mx = 0;
                        for (int l = 0; l < 64 && i0 + l < n; ++l) {
                            const int i = i0 + l, c0 = scell[i], cy = (c0 / gz) % gy, cx = c0 / (gz * gy);
                            int k = 0;
                            for (int j : nbr[i]) { const int cj = scell[j]; if ((cj / (gz * gy) - cx + 1) * 3 + ((cj / gz) % gy - cy + 1) == r) ++k; }
                            mx = std::max(mx, k);
                        }
                        off[r + 1] = off[r] + mx;
                    }
                    len = (off[9] + 3) / 4 * 4;
                    if (len > capP) { printf("aligned rows exceed capacity\n"); return 1; }
                }
                chunks4[tile] = len / 4;
                iters += len;
                for (int l = 0; l < 64; ++l) {
                    const int i = std::min(i0 + l, n - 1);
                    std::vector<unsigned int> e(len, (unsigned)n);
                    if (probe == 0) for (int t = 0; t < len; ++t) e[t] = (unsigned)std::min(n - 1, i0 + (int)(hash(tile, t, 0) % 64));
                    if (probe == 1) for (int t = 0; t < len; ++t) e[t] = (unsigned)std::min(n - 1, std::max(0, i0 + l + (int)(hash(tile, t, 0) % 17) - 8));
                    if (probe == 2) for (int t = 0; t < len; ++t) e[t] = (unsigned)std::min(n - 1, std::max(0, i0 - 8 + (int)(hash(tile, t, l) % 80)));
                    if (probe == 3) for (int t = 0; t < len; ++t) e[t] = (unsigned)std::min(n - 1, std::max(0, i0 + (l / 8) * 8 - 8 + (int)(hash(tile, t, l) % 24)));
                    if (probe == 4) for (int t = 0; t < cnt[i] && i0 + l < n; ++t) e[t] = (unsigned)nbr[i][t];
                    if (probe == 5 && i0 + l < n) {
                        const int c0 = scell[i], cy = (c0 / gz) % gy, cx = c0 / (gz * gy);
                        std::vector<int> fill(9, 0);
                        for (int j : nbr[i]) {
                            const int cj = scell[j], r = (cj / (gz * gy) - cx + 1) * 3 + ((cj / gz) % gy - cy + 1);
                            e[off[r] + fill[r]++] = (unsigned)j;
                        }
                    }
                    for (int t = 0; t < len; ++t) reinterpret_cast<unsigned int*>(&rows[((size_t)tile * cap4 + t / 4) * 64 + l])[t & 3] = e[t];
                }
            }
            CK(hipMemcpy(dRows, rows.data(), sizeof(uint4) * rows.size(), hipMemcpyHostToDevice));
            CK(hipMemcpy(dChunks, chunks4.data(), sizeof(int) * numTiles, hipMemcpyHostToDevice));
            for (int exact = 1; exact >= 0; --exact)
                for (int two = 1; two >= 0; --two) {
                    ref[exact][two].clear();
                    char nm[64];
                    snprintf(nm, sizeof(nm), "probe %-7s %s %s", names[probe], exact ? "exact" : "tol", two ? "2f" : "1f");
                    float msBefore = 0; (void)msBefore;
                    run(nm, exact, two, [&] {
                        if (exact) { if (two) hipLaunchKernelGGL((k_g32c<true, true>), dim3(gridG), dim3(256), 0, st, c, dPos, dVel, dRows, dChunks, dOut, n, numTiles, cap4);
                                     else hipLaunchKernelGGL((k_g32c<true, false>), dim3(gridG), dim3(256), 0, st, c, dPos, dVel, dRows, dChunks, dOut, n, numTiles, cap4); }
                        else { if (two) hipLaunchKernelGGL((k_g32c<false, true>), dim3(gridG), dim3(256), 0, st, c, dPos, dVel, dRows, dChunks, dOut, n, numTiles, cap4);
                               else hipLaunchKernelGGL((k_g32c<false, false>), dim3(gridG), dim3(256), 0, st, c, dPos, dVel, dRows, dChunks, dOut, n, numTiles, cap4); }
                    });
                }
            printf("   (%s: %.1f wave-iterations per tile)\n", names[probe], iters / numTiles);
        }
    }
    return 0;
}
