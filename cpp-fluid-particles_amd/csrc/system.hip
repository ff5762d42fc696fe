// system.hip — Particles / SPHParticles / SPHSystem of the drop-in API: uploads, the uniform-grid
// neighbour search (stable counting sort by cell id), boundary-mass precompute and the timed,
// optionally hipGraph-replayed, step loop.
//
// Reference behaviour restated (never copied): src/Particles.h:22-25, src/Particles.cu:28-36,
// src/SPHParticles.h:22-29, src/SPHSystem.cu:33-158, src/CUDAFunctions.cuh:56-78.
#include <algorithm>
#include <cstdlib>
#include <iostream>

#include "DFSPHSolver.h"
#include "PBDSolver.h"
#include "SPHSystem.h"
#include "sphx_c.h"
#include "engine.hpp"
#include "scan_chain.hpp"

using namespace sphx;

// ================================================================================ particle sets
Particles::Particles(const std::vector<float3>& p) : pos((unsigned)p.size()), vel((unsigned)p.size()), _active((unsigned)p.size())
{
    if (!p.empty()) {
        HIP_CALL(hipMemcpyAsync(pos.addr(), p.data(), sizeof(float3) * p.size(), hipMemcpyHostToDevice, sphx::stream()));
        HIP_CALL(hipStreamSynchronize(sphx::stream()));
    }
}

__global__ void k_advect_only(float3* __restrict__ pos, const float3* __restrict__ vel, float dt, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) pos[i] = add3(pos[i], smul3(dt, vel[i]));
}

// Particles::advect, Particles.cu:28-36: pos += dt * vel
void Particles::advect(float dt)
{
    const int n = (int)size();
    if (n <= 0) return;
    ScopedKernel t("advect");
    k_advect_only<<<blocks_for(n), 256, 0, sphx::stream()>>>(pos.addr(), vel.addr(), dt, n);
}

Particles::Particles(Uninitialised u) : pos(u.count), vel(u.count), _active(u.count) {}

SPHParticles::SPHParticles(Uninitialised u)
    : Particles(u), pressure(u.count), density(u.count), mass(u.count), particle2Cell(u.count), sortPerm(u.count), ids(u.count)
{
    ew_iota(ids.addr(), (int)u.count);
    ew_iota(sortPerm.addr(), (int)u.count);
    HIP_CALL(hipStreamSynchronize(sphx::stream()));
}

SPHParticles::SPHParticles(const std::vector<float3>& p)
    : Particles(p), pressure((unsigned)p.size()), density((unsigned)p.size()), mass((unsigned)p.size()),
      particle2Cell((unsigned)p.size()), sortPerm((unsigned)p.size()), ids((unsigned)p.size())
{
    ew_iota(ids.addr(), (int)p.size());
    ew_iota(sortPerm.addr(), (int)p.size());
    HIP_CALL(hipStreamSynchronize(sphx::stream()));
}

// ================================================================================ grid kernels
namespace sphx {

struct GridScratch {
    explicit GridScratch(int maxParticles, int cells)
        : slot((unsigned)maxParticles), order((unsigned)maxParticles), tmp3((unsigned)maxParticles),
          tmpi((unsigned)maxParticles), blockSums((unsigned)(std::max(cells, maxParticles) / 2048 + 2)),
          posm(4u * (unsigned)maxParticles), outRank((unsigned)maxParticles + 1u),
          chain(std::max(cells, maxParticles) / 2048 + 2, 1)
    {
    }
    DArray<int> slot;       // arrival slot of particle i inside its cell (atomic order, arbitrary)
    DArray<int> order;      // particle index stored at each bucket position (unordered inside a cell)
    DArray<float3> tmp3;    // gather target
    DArray<int> tmpi;
    DArray<int> blockSums;  // scan scratch
    DArray<float> posm;     // packed boundary positions for the boundary-mass sweep
    DArray<int> outRank;    // stable rank of each out-of-grid particle inside the sentinel bucket
    ChainScratch chain;     // tile states of the single-launch scans (scan_chain.hpp)
};

// SPHSystem's persistent mode (SPHSystem.h): the map from API slots (the reference's order) to the working arrays
struct PersistState {
    explicit PersistState(int capacity) : slotToWork((unsigned)capacity), tmpMap((unsigned)capacity), tmpMass((unsigned)capacity) {}
    std::unique_ptr<DArray<int>> nearWall;   // per cell: 1 when one of the 27 cells around it holds boundary particles (static)
    DArray<int> slotToWork;   // P: API slot s holds working particle P[s]
    DArray<int> tmpMap;
    DArray<float> tmpMass;
    bool wanted = false;
    bool primed = false;      // the working copy mirrors the API arrays through P
    // controller (SPHSystem::persistentController): rebuild and step counters at the last look, steps since, suspension
    int seenBuilds = 0, seenSteps = 0, sinceLook = 0;
    bool suspended = false; int suspendedSteps = 0;
};

struct StepGraph {
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    bool tried = false;
    long stepsRun = 0;       // steps so far
    long warmSteps = 0;      // eager steps that ran the solver's full schedule (its lazy allocations are done after one)
    unsigned int capturedCount = 0;   // active fluid particles the captured launches were sized for
    unsigned int capturedGeneration = 0;   // BaseSolver::graphGeneration() at capture time
    void drop()
    {
        if (exec) (void)hipGraphExecDestroy(exec);
        if (graph) (void)hipGraphDestroy(graph);
        exec = nullptr; graph = nullptr; tried = false;
    }
    ~StepGraph() { drop(); }
};

}  // namespace sphx

// mapParticles2Cells_CUDA + countingInCell_CUDA (CUDAFunctions.cuh:56-78) fused: cell id by true
// fp32 division and truncation, histogram by atomics; the returned arrival slot is remembered.
// The input is nearly cell-sorted (it was sorted one step ago), so neighbouring lanes mostly hit the same cell: one
// atomic per RUN of equal ids inside a wave, the run's lanes take consecutive slots (any slot assignment inside a
// cell is acceptable here: the stable rank fix-up below restores the reference's order).
__global__ void __launch_bounds__(256) k_cell_and_count(int* __restrict__ p2c, int* __restrict__ slot, int* __restrict__ counts,
                                                        const float3* __restrict__ pos, GridDesc g, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const bool valid = i < n;
    int id = -1;
    if (valid) {
        const int3 c = cell_of(pos[i], g);
        id = cell_id(c.x, c.y, c.z, g);
        p2c[i] = id;
    }
    const int prev = __shfl_up(id, 1, 64);
    const bool head = valid && (lane == 0 || prev != id);
    const unsigned long long heads = __ballot(head);
    const unsigned long long live = __ballot(valid);
    if (!valid) return;
    // my run: from the closest head at or below my lane up to the lane before the next head (or the last live lane)
    const unsigned long long below = heads & (~0ull >> (63 - lane));
    const int first = 63 - __builtin_clzll(below);
    const unsigned long long above = heads & ~(~0ull >> (63 - lane));            // heads strictly above me
    const int end = above ? __builtin_ctzll(above) : (64 - __builtin_clzll(live));   // one past the run's last lane
    int base = 0;
    if (head) base = atomicAdd(&counts[id], end - first);
    base = __shfl(base, first, 64);
    slot[i] = base + (lane - first);
}

// ---- exclusive scan over C+1 ints (replaces thrust::exclusive_scan, SPHSystem.cu:125) ----------
constexpr int kScanItems = 8;                      // items per thread
constexpr int kScanTile = 256 * kScanItems;        // items per block

__device__ __forceinline__ int block_exclusive_scan_256(int v, int* total)
{
    __shared__ int waveSums[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int incl = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int t = __shfl_up(incl, off, 64);
        if (lane >= off) incl += t;
    }
    if (lane == 63) waveSums[wave] = incl;
    __syncthreads();
    int base = 0;
    for (int w = 0; w < wave; ++w) base += waveSums[w];
    const int tot = waveSums[0] + waveSums[1] + waveSums[2] + waveSums[3];
    __syncthreads();
    *total = tot;
    return base + incl - v;
}

__global__ void __launch_bounds__(256) k_scan_tiles(int* __restrict__ data, int* __restrict__ blockSums, int n)
{
    const int base = blockIdx.x * kScanTile + threadIdx.x * kScanItems;
    int v[kScanItems];
    int sum = 0;
#pragma unroll
    for (int t = 0; t < kScanItems; ++t) { v[t] = (base + t < n) ? data[base + t] : 0; sum += v[t]; }
    int total;
    int run = block_exclusive_scan_256(sum, &total);
#pragma unroll
    for (int t = 0; t < kScanItems; ++t) { if (base + t < n) data[base + t] = run; run += v[t]; }
    if (threadIdx.x == 0) blockSums[blockIdx.x] = total;
}
// one block walks all tile totals with a running carry (any length)
__global__ void __launch_bounds__(256) k_scan_block_sums(int* __restrict__ blockSums, int m)
{
    __shared__ int carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int base = 0; base < m; base += 256) {
        const int idx = base + threadIdx.x;
        const int v = idx < m ? blockSums[idx] : 0;
        int total;
        const int ex = block_exclusive_scan_256(v, &total);
        const int c = carry;
        if (idx < m) blockSums[idx] = c + ex;
        __syncthreads();
        if (threadIdx.x == 0) carry = c + total;
        __syncthreads();
    }
}
__global__ void k_scan_add_offsets(int* __restrict__ data, const int* __restrict__ blockSums, int n, int* __restrict__ last = nullptr)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int v = data[i] + blockSums[i / kScanTile];
    data[i] = v;
    if (last && i == n - 1) *last = v;
}

// The same exclusive scan as ONE launch (scan_chain.hpp): a tile's block scans its 2048 items, looks back for the sum of the tiles
// before it and writes.  FLAGS: the items are not read from `data` but formed on the fly -- item i = 1 when particle i sits in the
// out-of-grid bucket (the flag pass and the three scan launches of the sentinel ranks in one).  `last`, when given, also receives
// the final item (the total of a windowed cell table goes where the out-of-grid bucket starts).
template <bool FLAGS>
__global__ void __launch_bounds__(256) k_scan_chain(int* __restrict__ data, int n, ScanChain c, const int* __restrict__ guard, int guardEq,
                                                    const int* __restrict__ p2c, int sentinel, int live, int* __restrict__ last)
{
    if (guard && *guard == guardEq) return;
    unsigned int gen;
    const int tile = chain_enter(c, gen);
    const int base = tile * kScanTile + threadIdx.x * kScanItems;
    int v[kScanItems];
    int sum = 0;
#pragma unroll
    for (int t = 0; t < kScanItems; ++t) {
        const int i = base + t;
        if (FLAGS) v[t] = (i < live && p2c[i] == sentinel) ? 1 : 0;
        else v[t] = i < n ? data[i] : 0;
        sum += v[t];
    }
    int total;
    int run = block_exclusive_scan_256(sum, &total);
    __shared__ int before;
    if (threadIdx.x < 64) {
        const int b = chain_exclusive(c, 0, gen, tile, total);
        if (threadIdx.x == 0) before = b;
    }
    __syncthreads();
    run += before;
#pragma unroll
    for (int t = 0; t < kScanItems; ++t) {
        if (base + t < n) {
            data[base + t] = run;
            if (last && base + t == n - 1) *last = run;
        }
        run += v[t];
    }
    chain_leave(c, gen, (int)gridDim.x);
}

// three independent scans of equal length in the same three launches (blockIdx.y = channel; the slab layer's three stable compactions)
struct Scan3 { int* data[3]; };
__global__ void __launch_bounds__(256) k_scan3_tiles(Scan3 s, int* __restrict__ blockSums, int n, int sumsStride)
{
    int* __restrict__ data = s.data[blockIdx.y];
    const int base = blockIdx.x * kScanTile + threadIdx.x * kScanItems;
    int v[kScanItems];
    int sum = 0;
#pragma unroll
    for (int t = 0; t < kScanItems; ++t) { v[t] = (base + t < n) ? data[base + t] : 0; sum += v[t]; }
    int total;
    int run = block_exclusive_scan_256(sum, &total);
#pragma unroll
    for (int t = 0; t < kScanItems; ++t) { if (base + t < n) data[base + t] = run; run += v[t]; }
    if (threadIdx.x == 0) blockSums[blockIdx.y * sumsStride + blockIdx.x] = total;
}
__global__ void __launch_bounds__(256) k_scan3_block_sums(int* __restrict__ blockSums, int m, int sumsStride)
{
    int* __restrict__ sums = blockSums + blockIdx.x * sumsStride;
    __shared__ int carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int base = 0; base < m; base += 256) {
        const int idx = base + threadIdx.x;
        const int v = idx < m ? sums[idx] : 0;
        int total;
        const int ex = block_exclusive_scan_256(v, &total);
        const int c = carry;
        if (idx < m) sums[idx] = c + ex;
        __syncthreads();
        if (threadIdx.x == 0) carry = c + total;
        __syncthreads();
    }
}
__global__ void k_scan3_add_offsets(Scan3 s, const int* __restrict__ blockSums, int n, int sumsStride)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) s.data[blockIdx.y][i] += blockSums[blockIdx.y * sumsStride + i / kScanTile];
}

// bucket placement in arrival order, then the stable fix-up: inside a cell the reference order is
// ascending original index (stable sort), so the rank of particle i is the number of bucket
// mates with a smaller index.  Cells hold a handful of particles, so the count is a short loop.
__global__ void k_place(int* __restrict__ order, const int* __restrict__ p2c, const int* __restrict__ slot,
                        const int* __restrict__ cellStart, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) order[cellStart[p2c[i]] + slot[i]] = i;
}
// The out-of-grid sentinel bucket can hold any number of particles (a blown-up run, parked slots),
// so its stable rank comes from an exclusive scan of the "is out of grid" flags (k_scan_chain<true>) instead of the
// quadratic bucket loop.
__global__ void k_stable_rank(int* __restrict__ perm, const int* __restrict__ order, const int* __restrict__ p2c,
                              const int* __restrict__ cellStart, const int* __restrict__ outRank, int n, int cellsPlusOne)
{
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= n) return;
    const int i = order[q];
    const int c = p2c[i];
    const int s = cellStart[c];
    if (c + 1 >= cellsPlusOne) {           // sentinel bucket
        perm[s + outRank[i]] = i;
        return;
    }
    const int e = cellStart[c + 1];
    int rank = 0;
    for (int t = s; t < e; ++t) rank += (order[t] < i) ? 1 : 0;
    perm[s + rank] = i;
}

// the sort's payload in one pass each way: positions, velocities and ids follow the permutation (into scratch), then go back
__global__ void k_gather_sorted(float3* __restrict__ tp, float3* __restrict__ tv, int* __restrict__ ti, const float3* __restrict__ pos,
                                const float3* __restrict__ vel, const int* __restrict__ id, const int* __restrict__ perm, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int p = perm[i];
    tp[i] = pos[p]; tv[i] = vel[p]; ti[i] = id[p];
}
__global__ void k_copy_back_sorted(float3* __restrict__ pos, float3* __restrict__ vel, int* __restrict__ id, const float3* __restrict__ tp,
                                   const float3* __restrict__ tv, const int* __restrict__ ti, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    pos[i] = tp[i]; vel[i] = tv[i]; id[i] = ti[i];
}

// ---- the same sort fed from payload rows (slab drivers, SPHSystem::setStagedInput) ----------------------------------------
// row t of the pre-sort order [from left | kept | from right]; W = 7 + E floats: pos(3) vel(3) id(1) extras(E)
struct RowSource { const float* rows[3]; int count[3]; int W; };
__device__ __forceinline__ const float* staged_row(const RowSource& r, int t)
{
    if (t < r.count[0]) return r.rows[0] + (size_t)t * r.W;
    t -= r.count[0];
    if (t < r.count[1]) return r.rows[1] + (size_t)t * r.W;
    return r.rows[2] + (size_t)(t - r.count[1]) * r.W;
}
// k_cell_and_count with the position read from the staged rows (same keys, same run-compressed histogram atomics)
// (cells outside the window [winLo, winHi) count as out of grid: SPHSystem::setCellWindow)
__global__ void __launch_bounds__(256) k_cell_and_count_rows(int* __restrict__ p2c, int* __restrict__ slot, int* __restrict__ counts, RowSource src,
                                                             GridDesc g, int n, int winLo, int winHi)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const bool valid = i < n;
    int id = -1;
    if (valid) {
        const float* row = staged_row(src, i);
        const int3 c = cell_of(v3(row[0], row[1], row[2]), g);
        id = cell_id(c.x, c.y, c.z, g);
        if (id < winLo || id >= winHi) id = g.C;
        p2c[i] = id;
    }
    const int prev = __shfl_up(id, 1, 64);
    const bool head = valid && (lane == 0 || prev != id);
    const unsigned long long heads = __ballot(head);
    const unsigned long long live = __ballot(valid);
    if (!valid) return;
    const unsigned long long below = heads & (~0ull >> (63 - lane));
    const int first = 63 - __builtin_clzll(below);
    const unsigned long long above = heads & ~(~0ull >> (63 - lane));
    const int end = above ? __builtin_ctzll(above) : (64 - __builtin_clzll(live));
    int base = 0;
    if (head) base = atomicAdd(&counts[id], end - first);
    base = __shfl(base, first, 64);
    slot[i] = base + (lane - first);
}
// the sort's payload in ONE pass: sorted slot q takes row perm[q]
__global__ void k_gather_rows_sorted(float3* __restrict__ pos, float3* __restrict__ vel, int* __restrict__ id, float* __restrict__ extra, int E,
                                     RowSource src, const int* __restrict__ perm, int n)
{
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= n) return;
    const float* row = staged_row(src, perm[q]);
    pos[q] = make_float3(row[0], row[1], row[2]);
    vel[q] = make_float3(row[3], row[4], row[5]);
    id[q] = __float_as_int(row[6]);
    for (int e = 0; e < E; ++e) extra[(size_t)q * E + e] = row[7 + e];
}

// computeBoundaryMass_CUDA, SPHSystem.cu:79-105: mass_b = rhoB / max(EPS, sum_j W_ij) over the
// boundary grid (the self term is excluded by W(0) = 0).
struct BoundaryMassBody {
    const KernelConsts& k;
    float sum;
    __device__ __forceinline__ void fluid(int, float3, float, float) {}
    __device__ __forceinline__ void boundary(int, float3, float r2, float) { sum += kW<false>(q_of<false>(sqrtf(r2), k), k); }
};
__global__ void __launch_bounds__(256) k_boundary_mass(float* __restrict__ mass, const float4* __restrict__ posm,
                                                       const int* __restrict__ csB, GridDesc g, KernelConsts k,
                                                       float rhoB, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    BoundaryMassBody body{k, 0.0f};
    sweep27<false, true>(g, k, nullptr, nullptr, csB, posm, xyz(posm[i]), body);
    mass[i] = rhoB / max_eps(body.sum);
}
__global__ void k_pack_pos_only(float4* __restrict__ dst, const float3* __restrict__ pos, int n)
{
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q < n) { const float3 p = pos[q]; dst[q] = make_float4(p.x, p.y, p.z, 0.0f); }
}

// The scan of a cell table.  Up to kChainTiles tiles (262,144 cells: the reference scene, BASELINE config 2) as one launch: every
// block is resident at once and a look-back ends within two rounds.  Beyond that the look-back is what a tile waits for (measured:
// ~3 us per round of 64 tiles, 0.19 ms instead of 0.06 at 7.5 M cells), so large tables keep the three passes.
constexpr int kChainTiles = 128;
static void chain_scan(GridScratch& gs, int* data, int count, int* last)
{
    if (count <= 0) return;
    hipStream_t st = sphx::stream();
    const int tiles = (count - 1) / kScanTile + 1;
    if (tiles <= kChainTiles) {
        k_scan_chain<false><<<tiles, 256, 0, st>>>(data, count, gs.chain.chain(), nullptr, 0, nullptr, 0, 0, last);
        return;
    }
    k_scan_tiles<<<tiles, 256, 0, st>>>(data, gs.blockSums.addr(), count);
    k_scan_block_sums<<<1, 256, 0, st>>>(gs.blockSums.addr(), tiles);
    k_scan_add_offsets<<<blocks_for(count), 256, 0, st>>>(data, gs.blockSums.addr(), count, last);
}
// a look-back that gave up (scan_chain.hpp) left a wrong cell table behind: reported where the solver's own faults are, behind the step
static void grid_fault_check(GridScratch& gs)
{
    if (!gs.chain.faulted()) return;
    *gs.chain.fault = 0;
    throw "SPHSystem: a scan of the grid pass gave up waiting for the tiles in front of it: the cell table of that step and everything computed from it are NOT valid";
}
// outRank[i] = number of out-of-grid particles in front of particle i (nothing to do while *guard == num: no such particle)
static void chain_rank_out_of_grid(GridScratch& gs, const int* p2c, int sentinel, int num, const int* guard)
{
    const int count = num + 1, tiles = (count - 1) / kScanTile + 1;
    k_scan_chain<true><<<tiles, 256, 0, sphx::stream()>>>(gs.outRank.addr(), count, gs.chain.chain(), guard, num, p2c, sentinel, num, nullptr);
}

// ================================================================================ SPHSystem
#define SPHX_SYSTEM_INIT_LIST                                                                          \
    _fluids(std::move(fluidParticles)), _boundaries(std::move(boundaryParticles)), _solver(std::move(solver)), \
        _sc{spaceSize, cellSize, sphCellLength, sphSmoothingRadius, dt, sphRho0, sphRhoBoundary, sphStiff, sphVisc, \
            sphSurfaceTensionIntensity, sphAirPressure, sphG},                                          \
        _fluidCellStart((unsigned)(cellSize.x * cellSize.y * cellSize.z + 1)),                         \
        _wallCellStart((unsigned)(cellSize.x * cellSize.y * cellSize.z + 1)),                          \
        _intScratch((unsigned)std::max(totalSize(), cellSize.x * cellSize.y * cellSize.z + 1))

SPHSystem::SPHSystem(std::shared_ptr<SPHParticles>& fluidParticles, std::shared_ptr<SPHParticles>& boundaryParticles,
                     std::shared_ptr<BaseSolver>& solver, const float3 spaceSize, const float sphCellLength,
                     const float sphSmoothingRadius, const float dt, const float sphM0, const float sphRho0,
                     const float sphRhoBoundary, const float sphStiff, const float sphVisc,
                     const float sphSurfaceTensionIntensity, const float sphAirPressure, const float3 sphG,
                     const int3 cellSize)
    : SPHX_SYSTEM_INIT_LIST
{
    initialise(sphM0, true);
}

SPHSystem::SPHSystem(NoInitialStep, std::shared_ptr<SPHParticles>& fluidParticles,
                     std::shared_ptr<SPHParticles>& boundaryParticles, std::shared_ptr<BaseSolver>& solver,
                     const float3 spaceSize, const float sphCellLength, const float sphSmoothingRadius, const float dt,
                     const float sphM0, const float sphRho0, const float sphRhoBoundary, const float sphStiff,
                     const float sphVisc, const float sphSurfaceTensionIntensity, const float sphAirPressure,
                     const float3 sphG, const int3 cellSize)
    : SPHX_SYSTEM_INIT_LIST
{
    initialise(sphM0, false);
}

SPHSystem::SPHSystem(Restored, std::shared_ptr<SPHParticles>& fluidParticles,
                     std::shared_ptr<SPHParticles>& boundaryParticles, std::shared_ptr<BaseSolver>& solver,
                     const float3 spaceSize, const float sphCellLength, const float sphSmoothingRadius, const float dt,
                     const float sphM0, const float sphRho0, const float sphRhoBoundary, const float sphStiff,
                     const float sphVisc, const float sphSurfaceTensionIntensity, const float sphAirPressure,
                     const float3 sphG, const int3 cellSize)
    : SPHX_SYSTEM_INIT_LIST
{
    initialise(sphM0, false, false);
}

SPHSystem::SPHSystem(Slab slab, std::shared_ptr<SPHParticles>& fluidParticles,
                     std::shared_ptr<SPHParticles>& boundaryParticles, std::shared_ptr<BaseSolver>& solver,
                     const float3 spaceSize, const float sphCellLength, const float sphSmoothingRadius, const float dt,
                     const float sphM0, const float sphRho0, const float sphRhoBoundary, const float sphStiff,
                     const float sphVisc, const float sphSurfaceTensionIntensity, const float sphAirPressure,
                     const float3 sphG, const int3 cellSize)
    : SPHX_SYSTEM_INIT_LIST
{
    _cellOffsetX = slab.cellOffsetX;
    _slab = true;
    if (auto* s = dynamic_cast<BasicSPHSolver*>(_solver.get())) s->setCellOffsetX(slab.cellOffsetX);
    initialise(sphM0, false);
}

SPHSystem::~SPHSystem() noexcept {}

namespace sphx {
// exclusive prefix sum of `count` ints in place; blockSums: scratch of at least count / 2048 + 2 ints
void device_exclusive_scan(int* data, int count, int* blockSums)
{
    if (count <= 0) return;
    hipStream_t st = sphx::stream();
    const int tiles = (count - 1) / kScanTile + 1;
    k_scan_tiles<<<tiles, 256, 0, st>>>(data, blockSums, count);
    if (tiles > 1) {
        k_scan_block_sums<<<1, 256, 0, st>>>(blockSums, tiles);
        k_scan_add_offsets<<<blocks_for(count), 256, 0, st>>>(data, blockSums, count);
    }
}
// the same for three arrays of `count` ints at once; blockSums: 3 x (count / 2048 + 2) ints
void device_exclusive_scan3(int* a, int* b, int* c, int count, int* blockSums)
{
    if (count <= 0) return;
    hipStream_t st = sphx::stream();
    const int tiles = (count - 1) / kScanTile + 1, stride = tiles + 1;
    const Scan3 s{{a, b, c}};
    k_scan3_tiles<<<dim3(tiles, 3), 256, 0, st>>>(s, blockSums, count, stride);
    if (tiles > 1) {
        k_scan3_block_sums<<<3, 256, 0, st>>>(blockSums, tiles, stride);
        k_scan3_add_offsets<<<dim3(blocks_for(count), 3), 256, 0, st>>>(s, blockSums, count, stride);
    }
}
}  // namespace sphx

// a stage restricted to particles [lo, hi) (lo < 0: all), optionally accumulating the |error| total of
// particles [sumLo, sumHi) (DFSPH error stages); keepAccum: add to the running total (a later part of a split stage)
// Behind the last stage of a step taken stage by stage: a whole-domain system tunes its solver; a slab's engine only
// lets its neighbour rows grow (BasicSPHSolver::tune -> SweepCache::tuneRowCapacity) -- the loop windows and PBD skins of the derived
// solvers belong to whole-domain stepping, a slab's loops are its driver's.
static void tune_after_stages(bool slab, BaseSolver* solver)
{
    if (!slab) { solver->tune(1); return; }
    if (auto* basic = dynamic_cast<BasicSPHSolver*>(solver)) basic->BasicSPHSolver::tune(1);
}

void SPHSystem::phaseEx(int p, int lo, int hi, bool reduce, int sumLo, int sumHi, bool keepAccum, int lo2, int hi2)
{
    auto* basic = dynamic_cast<BasicSPHSolver*>(_solver.get());
    if (!basic) throw "SPHSystem::phaseEx: needs an engine solver";
    if (lo2 >= 0 && (lo < 0 || lo2 < hi || dynamic_cast<PBDSolver*>(_solver.get())))
        throw "SPHSystem::phaseEx: a second range must lie behind the first (WCSPH / DFSPH sweep stages only)";
    basic->setSweepRange(lo, hi, keepAccum, lo2, hi2);
    try {
        if (reduce) phaseReduce(p, sumLo, sumHi);
        else phase(p);
    } catch (...) {
        basic->setSweepRange(-1, -1, false);
        throw;
    }
    basic->setSweepRange(-1, -1, false);
}

// one stage of the DFSPH step (distributed drivers; see sphx_phase)
void SPHSystem::phaseReduce(int p, int sumLo, int sumHi)
{
    auto* dfsph = dynamic_cast<DFSPHSolver*>(_solver.get());
    if (!dfsph) throw "SPHSystem::phaseReduce: needs a DFSPHSolver";
    dfsph->setErrorSumRange(sumLo, sumHi);
    dfsph->runPhase(p, _fluids, _boundaries, _fluidCellStart, _wallCellStart, _sc.space, _sc.cells, _sc.cellLength,
                    _sc.radius, _sc.dt, _sc.rho0, _sc.rhoBoundary, _sc.visc, _sc.gravity, _sc.surfaceTension,
                    _sc.airPressure, true);
}

void SPHSystem::resetErrorTotal()
{
    auto* dfsph = dynamic_cast<DFSPHSolver*>(_solver.get());
    if (!dfsph) throw "SPHSystem::resetErrorTotal: needs a DFSPHSolver";
    dfsph->resetErrorTotal();
}

long long SPHSystem::errorTotalFixed()
{
    auto* dfsph = dynamic_cast<DFSPHSolver*>(_solver.get());
    if (!dfsph) throw "SPHSystem::errorTotalFixed: needs a DFSPHSolver";
    return dfsph->readErrorTotalFixed();
}

void SPHSystem::phase(int p)
{
    if (_persist && _persist->wanted) throw "SPHSystem::phase: stage-wise stepping is not available with persistent rows";
    if (p == SPHX_PH_W_SURFACE_PRESSURE) {
        auto* w = dynamic_cast<BasicSPHSolver*>(_solver.get());
        if (!w || dynamic_cast<DFSPHSolver*>(_solver.get())) throw "SPHSystem::phase: WCSPH stages need a BasicSPHSolver";
        w->runWcsphPhase(p, _fluids, _boundaries, _fluidCellStart, _wallCellStart, _sc.space, _sc.cells, _sc.cellLength,
                         _sc.radius, _sc.dt, _sc.rho0, _sc.rhoBoundary, _sc.stiff, _sc.visc, _sc.gravity,
                         _sc.surfaceTension, _sc.airPressure);
        return;
    }
    if ((p >= SPHX_PH_P_SEARCH && p <= SPHX_PH_P_TAIL) || p == SPHX_PH_P_DELTA_SWEEP || p == SPHX_PH_P_APPLY) {
        auto* pbd = dynamic_cast<PBDSolver*>(_solver.get());
        if (!pbd) throw "SPHSystem::phase: PBD stages need a PBDSolver";
        if (p == SPHX_PH_P_SEARCH) { neighborSearch(_fluids, _fluidCellStart); if (_afterSort) _afterSort(); }
        pbd->runPhase(p, _fluids, _boundaries, _fluidCellStart, _wallCellStart, _sc.space, _sc.cells, _sc.cellLength,
                      _sc.radius, _sc.dt, _sc.rho0, _sc.rhoBoundary, _sc.gravity, _sc.surfaceTension, _sc.airPressure);
        if (p == SPHX_PH_P_TAIL) { _graph->stepsRun++; tune_after_stages(_slab, _solver.get()); }
        return;
    }
    if ((p >= SPHX_PH_W_SEARCH && p <= SPHX_PH_W_PRESSURE) || (p == SPHX_PH_ADVECT && !dynamic_cast<DFSPHSolver*>(_solver.get()))) {
        auto* w = dynamic_cast<BasicSPHSolver*>(_solver.get());
        if (!w || dynamic_cast<DFSPHSolver*>(_solver.get())) throw "SPHSystem::phase: WCSPH stages need a BasicSPHSolver";
        if (p == SPHX_PH_W_SEARCH) {
            if (_hasStaged) { _hasStaged = false; neighborSearchStaged(_staged); } else neighborSearch(_fluids, _fluidCellStart);
            if (_afterSort) _afterSort();
        }
        w->runWcsphPhase(p, _fluids, _boundaries, _fluidCellStart, _wallCellStart, _sc.space, _sc.cells, _sc.cellLength,
                         _sc.radius, _sc.dt, _sc.rho0, _sc.rhoBoundary, _sc.stiff, _sc.visc, _sc.gravity,
                         _sc.surfaceTension, _sc.airPressure);
        if (p == SPHX_PH_ADVECT) { _graph->stepsRun++; tune_after_stages(_slab, _solver.get()); }
        return;
    }
    auto* dfsph = dynamic_cast<DFSPHSolver*>(_solver.get());
    if (!dfsph) throw "SPHSystem::phase: stage-wise stepping needs a DFSPHSolver";
    if (p == SPHX_PH_SEARCH) {
        if (_hasStaged) {      // (the rows carry the warm-start stiffness: it arrives sorted, DFSPHSolver.cu:170-171 has nothing left to do)
            _hasStaged = false;
            neighborSearchStaged(_staged);
            if (_staged.extraOut && _staged.extraFloats == 1) dfsph->noteWarmStiffnessSorted();
        } else neighborSearch(_fluids, _fluidCellStart);
        if (_afterSort) _afterSort();
    }
    dfsph->runPhase(p, _fluids, _boundaries, _fluidCellStart, _wallCellStart, _sc.space, _sc.cells, _sc.cellLength,
                    _sc.radius, _sc.dt, _sc.rho0, _sc.rhoBoundary, _sc.visc, _sc.gravity, _sc.surfaceTension,
                    _sc.airPressure, false);
    if (p == SPHX_PH_ADVECT) { _graph->stepsRun++; tune_after_stages(_slab, _solver.get()); }      // (whole-domain systems stepped stage by stage: ADVICE r03)
}

// the constructor sequence of SPHSystem.cu:68-76
void SPHSystem::initialise(float sphM0, bool runStep, bool sortFluid)
{
    const int cells = _sc.cells.x * _sc.cells.y * _sc.cells.z;
    _grid.reset(new GridScratch(std::max(std::max((int)_fluids->capacity(), (int)_boundaries->capacity()), 1), cells));
    _graph.reset(new StepGraph());
    if (auto* w = dynamic_cast<BasicSPHSolver*>(_solver.get())) w->reserveBoundary((int)_boundaries->capacity());
    neighborSearch(_boundaries, _wallCellStart);
    computeBoundaryMass();
    ew_fill_float(_fluids->getMassPtr(), sphM0, (int)_fluids->capacity());
    if (!_slab && sortFluid) neighborSearch(_fluids, _fluidCellStart);   // a slab's particles arrive with the first exchange
    HIP_CALL(hipStreamSynchronize(sphx::stream()));
    if (runStep) step();
}

void SPHSystem::computeBoundaryMass()
{
    const int nb = boundarySize();
    if (nb <= 0) return;
    const KernelConsts k = make_kernel_consts(_sc.radius);
    const GridDesc g = make_grid_desc(_sc.cells, _sc.cellLength, _cellOffsetX);
    float4* posm = reinterpret_cast<float4*>(_grid->posm.addr());
    ScopedKernel t("boundary_mass");
    k_pack_pos_only<<<blocks_for(nb), 256, 0, sphx::stream()>>>(posm, _boundaries->getPosPtr(), nb);
    k_boundary_mass<<<blocks_for(nb), 256, 0, sphx::stream()>>>(_boundaries->getMassPtr(), posm, _wallCellStart.addr(), g, k,
                                                                 _sc.rhoBoundary, nb);
}

// SPHSystem::neighborSearch, SPHSystem.cu:114-127, as one stable counting sort:
//   cell id + histogram -> exclusive scan -> bucket placement -> stable rank -> gather pos, vel, id.
// particle2Cell keeps the PRE-sort keys; the permutation is published through
// SPHParticles::getSortPerm() for solver-owned persistent arrays.
void SPHSystem::neighborSearch(const std::shared_ptr<SPHParticles>& particles, DArray<int>& cellStart)
{
    const int num = (int)particles->size();
    const int cellsPlusOne = _sc.cells.x * _sc.cells.y * _sc.cells.z + 1;
    const GridDesc g = make_grid_desc(_sc.cells, _sc.cellLength, _cellOffsetX);
    hipStream_t st = sphx::stream();
    int* p2c = particles->getParticle2Cell();
    int* perm = particles->getSortPerm();

    HIP_CALL(hipMemsetAsync(cellStart.addr(), 0, sizeof(int) * cellsPlusOne, st));
    if (num > 0) {
        ScopedKernel t("grid_cell_count");
        k_cell_and_count<<<blocks_for(num), 256, 0, st>>>(p2c, _grid->slot.addr(), cellStart.addr(), particles->getPosPtr(), g, num);
    }
    {
        ScopedKernel t("grid_scan");
        chain_scan(*_grid, cellStart.addr(), cellsPlusOne, nullptr);
    }
    if (num <= 0) return;
    {
        ScopedKernel t("grid_stable_rank");
        // the sentinel bucket's ranks: after the scan cellStart[C] = num - (out-of-grid particles), so "== num" means there are
        // none and the launch returns at once
        chain_rank_out_of_grid(*_grid, p2c, cellsPlusOne - 1, num, cellStart.addr() + (cellsPlusOne - 1));
        k_place<<<blocks_for(num), 256, 0, st>>>(_grid->order.addr(), p2c, _grid->slot.addr(), cellStart.addr(), num);
        k_stable_rank<<<blocks_for(num), 256, 0, st>>>(perm, _grid->order.addr(), p2c, cellStart.addr(), _grid->outRank.addr(),
                                                       num, cellsPlusOne);
    }
    {
        ScopedKernel t("grid_gather");
        // (the boundary-mass scratch, 16 bytes per particle and idle after construction, is the second float3 target)
        float3* tv = reinterpret_cast<float3*>(_grid->posm.addr());
        k_gather_sorted<<<blocks_for(num), 256, 0, st>>>(_grid->tmp3.addr(), tv, _grid->tmpi.addr(), particles->getPosPtr(), particles->getVelPtr(),
                                                         particles->getIdPtr(), perm, num);
        k_copy_back_sorted<<<blocks_for(num), 256, 0, st>>>(particles->getPosPtr(), particles->getVelPtr(), particles->getIdPtr(), _grid->tmp3.addr(), tv,
                                                            _grid->tmpi.addr(), num);
    }
}

// The fluid sort of a slab step fed from payload rows (SPHSystem.h, StagedRows): neighborSearch with the cell keys read from the
// staged positions and the gather writing every array the rows carry in sorted order.
void SPHSystem::neighborSearchStaged(const StagedRows& staged)
{
    const int num = (int)_fluids->size();
    if (staged.count[0] + staged.count[1] + staged.count[2] != num) throw "SPHSystem: the staged rows do not add up to the active particle count";
    const int cellsPlusOne = _sc.cells.x * _sc.cells.y * _sc.cells.z + 1;
    const GridDesc g = make_grid_desc(_sc.cells, _sc.cellLength, _cellOffsetX);
    hipStream_t st = sphx::stream();
    int* p2c = _fluids->getParticle2Cell();
    int* perm = _fluids->getSortPerm();
    DArray<int>& cellStart = _fluidCellStart;
    const RowSource src{{staged.rows[0], staged.rows[1], staged.rows[2]}, {staged.count[0], staged.count[1], staged.count[2]}, 7 + staged.extraFloats};
    // the window of the cell table this step touches: cells [a, b), element b receiving the in-window total (b = C: the sentinel itself)
    const int cells = cellsPlusOne - 1, perColumn = _sc.cells.y * _sc.cells.z;
    const bool windowed = _winLo >= 0 && _winHi > _winLo;
    const int a = windowed ? std::min(std::max(_winLo, 0), _sc.cells.x) * perColumn : 0;
    const int b = windowed ? std::min(std::max(_winHi, 0), _sc.cells.x) * perColumn : cells;
    HIP_CALL(hipMemsetAsync(cellStart.addr(a), 0, sizeof(int) * (size_t)(b - a + 1), st));
    if (b < cells) HIP_CALL(hipMemsetAsync(cellStart.addr(cells), 0, sizeof(int), st));
    if (num > 0) {
        ScopedKernel t("grid_cell_count");
        k_cell_and_count_rows<<<blocks_for(num), 256, 0, st>>>(p2c, _grid->slot.addr(), cellStart.addr(), src, g, num, a, b);
    }
    {
        ScopedKernel t("grid_scan");
        // (the in-window total also goes where the out-of-grid bucket starts)
        chain_scan(*_grid, cellStart.addr(a), b - a + 1, b < cells ? cellStart.addr(cells) : nullptr);
    }
    if (num <= 0) return;
    {
        ScopedKernel t("grid_stable_rank");
        chain_rank_out_of_grid(*_grid, p2c, cellsPlusOne - 1, num, cellStart.addr() + (cellsPlusOne - 1));
        k_place<<<blocks_for(num), 256, 0, st>>>(_grid->order.addr(), p2c, _grid->slot.addr(), cellStart.addr(), num);
        k_stable_rank<<<blocks_for(num), 256, 0, st>>>(perm, _grid->order.addr(), p2c, cellStart.addr(), _grid->outRank.addr(), num, cellsPlusOne);
    }
    {
        ScopedKernel t("grid_gather");
        k_gather_rows_sorted<<<blocks_for(num), 256, 0, st>>>(_fluids->getPosPtr(), _fluids->getVelPtr(), _fluids->getIdPtr(), staged.extraOut,
                                                               staged.extraOut ? staged.extraFloats : 0, src, perm, num);
    }
}

// ================================================================================ persistent mode
// flags: [0] rebuild in this step, [1] forced by the host (consumed here), [2] rebuilds so far, [3] steps so far
__global__ void k_persist_begin(int* __restrict__ flags)
{
    flags[0] = flags[1] != 0 ? 1 : 0;
    flags[1] = 0;
    flags[3] += 1;
}
__global__ void k_persist_count(int* __restrict__ flags) { if (flags[0] != 0) flags[2] += 1; }

// k_cell_and_count for the API slots of the previous step: slot s holds working particle P[s].  Also the displacement check of
// the rows: d = (position now - position at the build) - the same difference of ONE reference particle (any common vector
// works: a pair's separation changes by d_i - d_j, which is bounded by twice the largest |d - c|; with the reference particle's
// own displacement as c a block of fluid in free fall has d - c = 0 to rounding).
// Boundary particles do not move, so against THEM a fluid particle's displacement counts in full (no common drift to take out):
// a particle that is within reach of the boundary now -- one of the 27 cells around its cell holds boundary particles, nearWall --
// asks for a rebuild once it has moved 0.98 skin in absolute terms.
__global__ void k_near_wall(int* __restrict__ nearWall, const int* __restrict__ csB, GridDesc g)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c > g.C) return;
    int any = 0;
    if (c < g.C) {
        const int z = c % g.gz, y = (c / g.gz) % g.gy, x = c / (g.gz * g.gy);
        for (int dx = -1; dx <= 1; ++dx) for (int dy = -1; dy <= 1; ++dy) {
            const int X = x + dx, Y = y + dy;
            if (X < 0 || X >= g.gx || Y < 0 || Y >= g.gy) continue;
            const int base = (X * g.gy + Y) * g.gz, zlo = max(z - 1, 0), zhi = min(z + 1, g.gz - 1);
            any |= csB[base + zhi + 1] > csB[base + zlo] ? 1 : 0;
        }
    }
    nearWall[c] = any;      // (slot C, the out-of-grid bucket: 0)
}
__global__ void __launch_bounds__(256) k_persist_cells(int* __restrict__ p2c, int* __restrict__ slot, int* __restrict__ counts,
                                                       const int* __restrict__ P, const float3* __restrict__ wpos,
                                                       const float4* __restrict__ posBuild, const int* __restrict__ nearWall,
                                                       GridDesc g, int n, int ref, float limit2, float wallLimit2,
                                                       int* __restrict__ flags)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const bool valid = i < n;
    int id = -1;
    if (valid) {
        const int p = P[i];
        const float3 x = wpos[p];
        const int3 c = cell_of(x, g);
        id = cell_id(c.x, c.y, c.z, g);
        p2c[i] = id;
        const float4 b = posBuild[p], br = posBuild[ref];
        const float3 xr = wpos[ref];
        const float3 da = sub3(x, v3(b.x, b.y, b.z));
        const float3 d = sub3(da, sub3(xr, v3(br.x, br.y, br.z)));
        if (!(dot3(d, d) <= limit2) || (nearWall[id] != 0 && !(dot3(da, da) <= wallLimit2))) flags[0] = 1;
    }
    const int prev = __shfl_up(id, 1, 64);
    const bool head = valid && (lane == 0 || prev != id);
    const unsigned long long heads = __ballot(head);
    const unsigned long long live = __ballot(valid);
    if (!valid) return;
    const unsigned long long below = heads & (~0ull >> (63 - lane));
    const int first = 63 - __builtin_clzll(below);
    const unsigned long long above = heads & ~(~0ull >> (63 - lane));
    const int end = above ? __builtin_ctzll(above) : (64 - __builtin_clzll(live));
    int base = 0;
    if (head) base = atomicAdd(&counts[id], end - first);
    base = __shfl(base, first, 64);
    slot[i] = base + (lane - first);
}
__global__ void k_persist_compose(int* __restrict__ dst, const int* __restrict__ P, const int* __restrict__ perm, int n)
{
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s < n) dst[s] = P[perm[s]];
}
__global__ void k_copy_int(int* __restrict__ dst, const int* __restrict__ src, int n)
{
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s < n) dst[s] = src[s];
}
// the re-sort of the working arrays into the API order of this step, only when the rows are being rebuilt
__global__ void k_persist_resort_gather(float3* __restrict__ tp, float3* __restrict__ tv, int* __restrict__ ti, float* __restrict__ tm,
                                        const float3* __restrict__ pos, const float3* __restrict__ vel, const int* __restrict__ id,
                                        const float* __restrict__ mass, const int* __restrict__ P, int n, const int* __restrict__ flags)
{
    if (flags[0] == 0) return;
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n) return;
    const int p = P[s];
    tp[s] = pos[p]; tv[s] = vel[p]; ti[s] = id[p]; tm[s] = mass[p];
}
__global__ void k_persist_resort_back(float3* __restrict__ pos, float3* __restrict__ vel, int* __restrict__ id, float* __restrict__ mass,
                                      int* __restrict__ workPerm, int* __restrict__ P, const float3* __restrict__ tp,
                                      const float3* __restrict__ tv, const int* __restrict__ ti, const float* __restrict__ tm, int n,
                                      const int* __restrict__ flags)
{
    if (flags[0] == 0) return;
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n) return;
    pos[s] = tp[s]; vel[s] = tv[s]; id[s] = ti[s]; mass[s] = tm[s];
    workPerm[s] = P[s];        // the permutation the solver's own persistent arrays follow (getSortPerm() of the working set)
    P[s] = s;
}
__global__ void k_persist_export(float3* __restrict__ apos, float3* __restrict__ avel, float* __restrict__ aden, float* __restrict__ apre,
                                 float* __restrict__ amass, int* __restrict__ aid, const float3* __restrict__ wpos,
                                 const float3* __restrict__ wvel, const float* __restrict__ wden, const float* __restrict__ wpre,
                                 const float* __restrict__ wmass, const int* __restrict__ wid, const int* __restrict__ P, int n)
{
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n) return;
    const int p = P[s];
    apos[s] = wpos[p]; avel[s] = wvel[p]; aden[s] = wden[p]; apre[s] = wpre[p]; amass[s] = wmass[p]; aid[s] = wid[p];
}

bool SPHSystem::setPersistentRows(bool on)
{
    auto* basic = dynamic_cast<BasicSPHSolver*>(_solver.get());
    if (!basic || dynamic_cast<PBDSolver*>(_solver.get()) || _slab) return !on;
    if (!_persist) {
        if (!on) return true;
        _persist.reset(new PersistState((int)_fluids->capacity()));
        _work = std::make_shared<SPHParticles>(Particles::Uninitialised{_fluids->capacity()});
    }
    if (_persist->wanted == on) return true;
    invalidatePersistentOrder();
    _persist->wanted = on;
    _persist->suspended = false; _persist->sinceLook = 0;
    basic->requestPersistentRows(on);
    _graph->drop();
    return on ? persistentActive() : true;
}

bool SPHSystem::persistentRows() const { return _persist && _persist->wanted; }
const int* SPHSystem::persistentSlotMap() const { return (_persist && _persist->primed) ? _persist->slotToWork.addr() : nullptr; }

bool SPHSystem::persistentActive()
{
    if (!_persist || !_persist->wanted || _persist->suspended) return false;
    auto* basic = static_cast<BasicSPHSolver*>(_solver.get());
    return basic->preparePersistent(_sc.cells, _sc.cellLength, _sc.radius).active;
}

// Rows that are rebuilt in (nearly) every step cost more than plain per-step rows: the skin re-test of every pair, ~3 % longer
// rows, the slot-map grid pass and the export.  Every 64+ steps the host looks at the device-side counters {row builds, steps}
// (one 16-byte read between steps, never inside a captured graph): at >= 90 % rebuilding steps the mode is left for 256 steps --
// the solver's arrays go back to the API order, the ordinary tolerance step runs -- and tried again afterwards.
void SPHSystem::persistentController(int stepsSinceLastCall)
{
    if (!_persist || !_persist->wanted || tuning().persist_controller == 0) return;
    PersistState& ps = *_persist;
    auto* basic = static_cast<BasicSPHSolver*>(_solver.get());
    if (ps.suspended) {
        ps.suspendedSteps += stepsSinceLastCall;
        if (ps.suspendedSteps >= 256) { ps.suspended = false; ps.sinceLook = 0; basic->requestPersistentRows(true); _graph->drop(); }
        return;
    }
    ps.sinceLook += stepsSinceLastCall;
    if (ps.sinceLook < 64 || !ps.primed) return;
    const int* flags = basic->enginePersistFlags();
    if (!flags) return;
    int words[4] = {0, 0, 0, 0};
    HIP_CALL(hipMemcpyAsync(words, flags, sizeof(words), hipMemcpyDeviceToHost, sphx::stream()));
    HIP_CALL(hipStreamSynchronize(sphx::stream()));
    const int builds = words[2] - ps.seenBuilds, steps = words[3] - ps.seenSteps;
    ps.seenBuilds = words[2]; ps.seenSteps = words[3]; ps.sinceLook = 0;
    if (steps >= 32 && 10 * builds >= 9 * steps) {
        invalidatePersistentOrder();
        basic->requestPersistentRows(false);
        ps.suspended = true; ps.suspendedSteps = 0;
        _graph->drop();
    }
}

void SPHSystem::invalidatePersistentOrder()
{
    if (!_persist || !_persist->primed) return;
    // the API arrays are current (exported by every step); the solver's own arrays are in the working order: bring them along
    static_cast<BasicSPHSolver*>(_solver.get())->permuteState(_persist->slotToWork.addr(), (int)_fluids->size());
    _persist->primed = false;
    _graph->drop();
}

void SPHSystem::persistentPrime()
{
    const int n = (int)_fluids->size();
    _work->setActiveCount((unsigned)n);
    ew_copy(_work->getPosPtr(), _fluids->getPosPtr(), sizeof(float3) * (size_t)n);
    ew_copy(_work->getVelPtr(), _fluids->getVelPtr(), sizeof(float3) * (size_t)n);
    ew_copy(_work->getMassPtr(), _fluids->getMassPtr(), sizeof(float) * (size_t)_fluids->capacity());
    ew_copy(_work->getDensityPtr(), _fluids->getDensityPtr(), sizeof(float) * (size_t)n);
    ew_copy(_work->getPressurePtr(), _fluids->getPressurePtr(), sizeof(float) * (size_t)n);
    ew_copy(_work->getIdPtr(), _fluids->getIdPtr(), sizeof(int) * (size_t)n);
    ew_iota(_persist->slotToWork.addr(), n);
    static_cast<BasicSPHSolver*>(_solver.get())->requestRowRebuild();
    _persist->primed = true;
}

// The grid pass of a persistent step.  What SPHSystem::neighborSearch does for the API order -- keys, histogram, scan, stable
// ranks -- runs on the API slots of the previous step with the positions read through the slot map, so that particle2Cell,
// getSortPerm(), the cell table and (after the export) every API array are what the reference's stable sort gives; the payload
// of the sort is the slot map alone.  The working arrays move only when the rows are rebuilt.
void SPHSystem::persistentSearch()
{
    auto* basic = static_cast<BasicSPHSolver*>(_solver.get());
    const BasicSPHSolver::PersistentView pv = basic->preparePersistent(_sc.cells, _sc.cellLength, _sc.radius);
    const int num = (int)_fluids->size();
    const int cellsPlusOne = _sc.cells.x * _sc.cells.y * _sc.cells.z + 1;
    const GridDesc g = make_grid_desc(_sc.cells, _sc.cellLength, _cellOffsetX);
    hipStream_t st = sphx::stream();
    int* p2c = _fluids->getParticle2Cell();
    int* perm = _fluids->getSortPerm();
    int* P = _persist->slotToWork.addr();
    DArray<int>& cellStart = _fluidCellStart;
    _work->setActiveCount((unsigned)num);
    if (!_persist->nearWall) {          // (boundary particles are static: once)
        _persist->nearWall.reset(new DArray<int>((unsigned)cellsPlusOne));
        k_near_wall<<<blocks_for(cellsPlusOne), 256, 0, st>>>(_persist->nearWall->addr(), _wallCellStart.addr(), g);
    }
    k_persist_begin<<<1, 1, 0, st>>>(pv.flags);
    HIP_CALL(hipMemsetAsync(cellStart.addr(), 0, sizeof(int) * cellsPlusOne, st));
    if (num > 0) {
        ScopedKernel t("grid_cell_count");
        k_persist_cells<<<blocks_for(num), 256, 0, st>>>(p2c, _grid->slot.addr(), cellStart.addr(), P, _work->getPosPtr(),
                                                         static_cast<const float4*>(pv.posBuild), _persist->nearWall->addr(), g, num, num / 2,
                                                         pv.limit2, 4.0f * pv.limit2, pv.flags);      // 4 x (0.49 skin)^2 = (0.98 skin)^2
    }
    {
        ScopedKernel t("grid_scan");
        chain_scan(*_grid, cellStart.addr(), cellsPlusOne, nullptr);
    }
    if (num <= 0) return;
    {
        ScopedKernel t("grid_stable_rank");
        chain_rank_out_of_grid(*_grid, p2c, cellsPlusOne - 1, num, cellStart.addr() + (cellsPlusOne - 1));
        k_place<<<blocks_for(num), 256, 0, st>>>(_grid->order.addr(), p2c, _grid->slot.addr(), cellStart.addr(), num);
        k_stable_rank<<<blocks_for(num), 256, 0, st>>>(perm, _grid->order.addr(), p2c, cellStart.addr(), _grid->outRank.addr(), num, cellsPlusOne);
    }
    {
        ScopedKernel t("grid_gather");
        k_persist_compose<<<blocks_for(num), 256, 0, st>>>(_persist->tmpMap.addr(), P, perm, num);
        k_copy_int<<<blocks_for(num), 256, 0, st>>>(P, _persist->tmpMap.addr(), num);
        // rows to be rebuilt: the working arrays take the API order of this step (the order the builder's cell walk needs)
        float3* tv = reinterpret_cast<float3*>(_grid->posm.addr());
        k_persist_count<<<1, 1, 0, st>>>(pv.flags);
        k_persist_resort_gather<<<blocks_for(num), 256, 0, st>>>(_grid->tmp3.addr(), tv, _grid->tmpi.addr(), _persist->tmpMass.addr(), _work->getPosPtr(),
                                                                 _work->getVelPtr(), _work->getIdPtr(), _work->getMassPtr(), P, num, pv.flags);
        k_persist_resort_back<<<blocks_for(num), 256, 0, st>>>(_work->getPosPtr(), _work->getVelPtr(), _work->getIdPtr(), _work->getMassPtr(),
                                                               _work->getSortPerm(), P, _grid->tmp3.addr(), tv, _grid->tmpi.addr(),
                                                               _persist->tmpMass.addr(), num, pv.flags);
    }
}

void SPHSystem::persistentExport()
{
    const int num = (int)_fluids->size();
    if (num <= 0) return;
    ScopedKernel t("export_api_order");
    k_persist_export<<<blocks_for(num), 256, 0, sphx::stream()>>>(_fluids->getPosPtr(), _fluids->getVelPtr(), _fluids->getDensityPtr(),
                                                                  _fluids->getPressurePtr(), _fluids->getMassPtr(), _fluids->getIdPtr(),
                                                                  _work->getPosPtr(), _work->getVelPtr(), _work->getDensityPtr(),
                                                                  _work->getPressurePtr(), _work->getMassPtr(), _work->getIdPtr(),
                                                                  _persist->slotToWork.addr(), num);
}

void SPHSystem::enqueueStep()
{
    if (persistentActive()) {
        if (!_persist->primed) persistentPrime();
        persistentSearch();
        _solver->step(_work, _boundaries, _fluidCellStart, _wallCellStart, _sc.space, _sc.cells, _sc.cellLength,
                      _sc.radius, _sc.dt, _sc.rho0, _sc.rhoBoundary, _sc.stiff, _sc.visc, _sc.gravity,
                      _sc.surfaceTension, _sc.airPressure);
        persistentExport();
        return;
    }
    if (_persist && _persist->primed) invalidatePersistentOrder();      // (the mode became unusable: back to the ordinary step)
    neighborSearch(_fluids, _fluidCellStart);
    _solver->step(_fluids, _boundaries, _fluidCellStart, _wallCellStart, _sc.space, _sc.cells, _sc.cellLength,
                  _sc.radius, _sc.dt, _sc.rho0, _sc.rhoBoundary, _sc.stiff, _sc.visc, _sc.gravity,
                  _sc.surfaceTension, _sc.airPressure);
}

// SPHSystem::step, SPHSystem.cu:129-158
float SPHSystem::step()
{
    hipStream_t st = sphx::stream();
    hipEvent_t start, stop;
    HIP_CALL(hipEventCreate(&start));
    HIP_CALL(hipEventCreate(&stop));
    HIP_CALL(hipEventRecord(start, st));
    const bool fullSchedule = _solver->graphSafe();   // e.g. false for PBD's first call, which only records positions
    try {
        enqueueStep();
        HIP_CALL(hipStreamSynchronize(st));
        CHECK_KERNEL();
        if (fullSchedule) _graph->warmSteps++;
    } catch (const char* s) {
        std::cout << s << "\n";
    } catch (...) {
        std::cout << "Unknown exception in SPHSystem::step\n";
    }
    float milliseconds = 0.0f;
    HIP_CALL(hipEventRecord(stop, st));
    HIP_CALL(hipEventSynchronize(stop));
    HIP_CALL(hipEventElapsedTime(&milliseconds, start, stop));
    HIP_CALL(hipEventDestroy(start));
    HIP_CALL(hipEventDestroy(stop));
    _graph->stepsRun++;
    grid_fault_check(*_grid);
    _solver->tune(1);
    persistentController(1);
    return milliseconds;
}

// n steps, one sync.  When the solver is graph-safe the step is captured once and replayed.
float SPHSystem::stepN(int n)
{
    if (n <= 0) return 0.0f;
    hipStream_t st = sphx::stream();
    float extra = 0.0f;
    // a capture must not contain allocations: the engine's lazily created buffers (neighbour rows, tile
    // buckets) appear during the first step that runs the solver's full schedule, so that one is eager
    // ... and so is the step that primes the working copy of the persistent mode (host-side copies outside the captured schedule)
    while (n > 0 && (_graph->stepsRun == 0 || (_solver->graphSafe() && _graph->warmSteps == 0) ||
                     (_persist && !_persist->primed && persistentActive()))) { extra += step(); --n; }
    if (n == 0) return extra;
    const bool wantGraph = _solver->graphSafe() && !KernelTimer::enabled && !tuning().no_graph;
    auto ensureGraph = [&] {
        // launch sizes are baked into a capture: a changed active count (sphx_set_count) needs a new one
        // ... and so do host-side invalidations (boundary masses rewritten, arrays regrown, engine switches)
        if (_graph->tried && (_graph->capturedCount != _fluids->size() || _graph->capturedGeneration != _solver->graphGeneration()))
            _graph->drop();
        if (!(wantGraph && !_graph->exec && !_graph->tried)) return;
        _graph->tried = true;
        _graph->capturedCount = _fluids->size();
        _graph->capturedGeneration = _solver->graphGeneration();
        _solver->prepareForCapture();
        const bool say = tuning().graph_debug != 0;       // which stage of a capture failed (the step then runs eagerly)
        bool ok = hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal) == hipSuccess;
        if (ok) {
            try { enqueueStep(); } catch (const char* msg) { ok = false; if (say) std::cout << "sphx: capture: enqueue threw: " << msg << "\n"; }
            catch (...) { ok = false; if (say) std::cout << "sphx: capture: enqueue threw\n"; }
            hipGraph_t gph = nullptr;
            const hipError_t ended = hipStreamEndCapture(st, &gph);
            if (ended != hipSuccess) { ok = false; if (say) std::cout << "sphx: capture: end of capture: " << hipGetErrorString(ended) << "\n"; }
            const hipError_t made = (ok && gph) ? hipGraphInstantiate(&_graph->exec, gph, nullptr, nullptr, 0) : hipErrorUnknown;
            if (made == hipSuccess) {
                _graph->graph = gph;
                if (say) std::cout << "sphx: capture: step graph instantiated\n";
            } else {
                if (say && ok) std::cout << "sphx: capture: instantiate: " << hipGetErrorString(made) << "\n";
                if (gph) (void)hipGraphDestroy(gph);
                _graph->exec = nullptr;
                (void)hipGetLastError();
                _solver->captureFailed();         // the recorded launches never ran: their "valid" marks must not survive
            }
        } else if (say) std::cout << "sphx: capture: could not begin\n";
    };
    hipEvent_t start, stop;
    HIP_CALL(hipEventCreate(&start));
    HIP_CALL(hipEventCreate(&stop));
    HIP_CALL(hipEventRecord(start, st));
    // The batch runs in chunks of (normally) 16 steps with the solver's host-side tuning hook between them (one 4-byte read-back: row
    // capacity, PBD skin controller): a long batch that runs into denser states must not wait for its end to get longer rows.
    // The hook may invalidate the capture (regrown rows); the next chunk then re-captures.
    const int kChunk = std::max(1, _solver->tuneInterval());
    for (int done = 0; done < n;) {
        // the persistent mode may have been resumed by the controller between two chunks (256 steps after a suspension): the step
        // that primes the working copy -- host-side copies, a stream sync in the rebuild request -- must run eagerly, outside
        // any capture, exactly as in front of the loop
        if (_persist && !_persist->primed && persistentActive()) {
            try { enqueueStep(); } catch (const char* msg) { std::cout << msg << "\n"; }
            ++done; _graph->stepsRun++;
            continue;
        }
        ensureGraph();
        const int chunk = std::min(n - done, kChunk);
        for (int s = 0; s < chunk; ++s) {
            if (wantGraph && _graph->exec) {
                HIP_CALL(hipGraphLaunch(_graph->exec, st));
            } else {
                try { enqueueStep(); } catch (const char* msg) { std::cout << msg << "\n"; }
            }
        }
        done += chunk;
        _graph->stepsRun += chunk;
        if (done < n) { grid_fault_check(*_grid); _solver->tune(chunk); persistentController(chunk); }
    }
    HIP_CALL(hipEventRecord(stop, st));
    HIP_CALL(hipEventSynchronize(stop));
    HIP_CALL(hipStreamSynchronize(st));
    float milliseconds = 0.0f;
    HIP_CALL(hipEventElapsedTime(&milliseconds, start, stop));
    HIP_CALL(hipEventDestroy(start));
    HIP_CALL(hipEventDestroy(stop));
    grid_fault_check(*_grid);
    _solver->tune(n % kChunk == 0 ? kChunk : n % kChunk);
    persistentController(n % kChunk == 0 ? kChunk : n % kChunk);
    return milliseconds + extra;
}
