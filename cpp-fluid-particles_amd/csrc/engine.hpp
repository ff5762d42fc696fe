// engine.hpp — host-side internals shared by the solver / system translation units.
#pragma once

#include <memory>
#include <string>
#include <vector>

#include "DArray.h"
#include "SPHParticles.h"
#include "sph_device.hpp"
#include "sphx_c.h"

namespace sphx {

// the process-wide tuning block (include/sphx_c.h, sphx_set_tuning): the engine's behaviour switches.  The library reads no
// environment variable for them; tests and tools install what they need through the C ABI.
const sphx_tuning& tuning();
sphx_tuning default_tuning();
void install_tuning(const sphx_tuning& t);
// which kernel the last launch_rate_kernel() chose (bench.py labels its roofline block with what actually ran)
enum RateVariant { kRateNone = 0, kRateQuadStrict, kRateQuadTol, kRateQuadStrictInTol, kRateDuo, kRateLane, kRateLdsTiles, kRateBrick };
extern int g_lastRateVariant;

// Builds the constant block of the smoothing kernels on the host with the same fp32 expressions
// the device would use, and finds tCut by bisection over float bit patterns: the three support
// tests (q > 2, r <= R, x > R) are monotone in r2 because correctly-rounded sqrt and division are
// monotone, so "largest r2 that passes" is an exact threshold.
KernelConsts make_kernel_consts(float radius);
void validate_fast_math(KernelConsts& k);
void fastmath_selftest(float R, unsigned long long samples, unsigned int out[3], int flags[2]);
GridDesc make_grid_desc(int3 cellSize, float cellLength, int cellOffsetX = 0);

// Per-solver packed views of the particle sets and the per-step neighbour list, refreshed when
// positions move.
//   posm         : float4 (x, y, z, mass) in cell-sorted order, fluid then boundary, one 16-byte load per neighbour
//   pterm        : p_j / max(EPS, rho_j^2), the per-particle half of the pressure-force weight
//   nbr/nbrCount : wave-interleaved compact neighbour rows (sph_device.hpp), `cap` entries/particle
//   aux3         : second float3 scratch (viscosity delta-v while bufferFloat3 holds the colour
//                  gradient in the fused sweeps)
// kFlagTiles: LDS-streamed tiles (sph_device.hpp, entry format 2)
// kFlagLinearTiles: launch tiles in array order instead of the (y-chunk, x) schedule
// kFlagNoQuad: lane-per-particle walks everywhere (no quad-per-particle sweep variants)
constexpr int kQuadSurfaceBit = 256;      // = kQuadSurface (sweep_ops.hpp)
enum EngineFlags { kFlagUnfused = 1, kFlagNoList = 2, kFlagTiles = 4, kFlagLinearTiles = 8, kFlagNoQuad = 16 };

// The neighbour rows are the one array that outgrows DArray's 32-bit element count (cap = 96 entries per particle:
// 2^32 entries at 44.7 M particles): a plain device allocation with a 64-bit length.  Not zero-filled: the builder
// writes every slot a sweep reads (rows are read up to nbrCount only).
struct RowStore {
    explicit RowStore(unsigned long long entries_) : entries(entries_)
    {
        void* raw = nullptr;
        const hipError_t e = hipMalloc(&raw, sizeof(unsigned int) * (size_t)(entries ? entries : 1ull));
        if (e != hipSuccess || !raw) {
            report_hip_error(e, __FILE__, __LINE__);
            throw DeviceAllocError("neighbour rows: hipMalloc of " + std::to_string(sizeof(unsigned int) * entries) + " bytes failed");
        }
        rows = static_cast<unsigned int*>(raw);
    }
    RowStore(const RowStore&) = delete;
    RowStore& operator=(const RowStore&) = delete;
    ~RowStore() { HIP_CALL(hipFree(rows)); }
    unsigned long long entries;
    unsigned int* rows = nullptr;
};

struct SweepCache {
    explicit SweepCache(int num);
    int n;
    DArray<float> posm;                      // 4 floats per fluid particle
    DArray<float> pterm;
    DArray<float3> aux3;
    DArray<float> vel4;                      // float4 mirror of vel, kept in step by every velocity writer
    DArray<float> cg4;                       // float4 mirror of the colour gradient
    DArray<float> posf;                      // float4 (x, y, z, scalar neighbour field): one-gather sweeps
    // PBD (r05): the other halves of posm / posf for Jacobi iterations that store the moved positions from inside the delta-p sweep;
    // swapped with the live ones behind such a launch.  Copies of the live arrays when made (boundary tail included), dropped whenever
    // the boundary part is repacked or regrown.
    std::unique_ptr<DArray<float>> posmAlt, posfAlt;
    void ensureAltPositions();
    void swapAltPositions() { posm.swap(*posmAlt); posf.swap(*posfAlt); }
    DArray<int> massUniform;                 // device flag set by the pack pass: all fluid masses equal
    bool allowPacked = true;                 // one-gather sweeps allowed (slab drivers refresh posf next to the scalar's own array)
    DArray<int> nbrCount;
    DArray<int> tileFmt;                     // per 64-particle tile: entry format of its rows (0 or 2)
    DArray<int> tileOrder;                   // launch schedule of the tiles (wave_tile)
    DArray<int> tileKey;                     // bucket of each tile while the schedule is built
    std::unique_ptr<DArray<int>> tileBuckets; // histogram / cursors of the (y-chunk, x) buckets
    bool orderValid = false;
    int orderBuiltTiles = -1;                // tile count of the schedule the array holds (a complete permutation of that many tiles)
    std::unique_ptr<RowStore> nbr;           // allocated on first use
    // Unified neighbour space: posm, posf, vel4 and cg4 hold [capN fluid slots | nbCap boundary slots].
    // A row entry carries ONE index into it, so a sweep gathers with a uniform base pointer and a
    // 32-bit offset, with no per-entry pointer select; the boundary tails of vel4 / cg4 stay +0.
    int capN = 0;                            // fluid slots (the capacity given at construction)
    int nbCap = 0;                           // boundary slots currently reserved
    int nb = 0;
    int cap = 96;
    // Row capacity.  Fixed (SPHX_NBR_CAP, slabs: 96) or adaptive: rows start at 48 entries per particle -- the lattice needs 32,
    // the settled dam-break 42-46 -- the builder records the longest row that did not fit (such a particle walks the
    // cells directly meanwhile: same bits, slower), and between steps the host enlarges the rows to that length + 8.  Halves
    // the row slab (192 instead of 384 bytes per particle).
    bool capAuto = false;
    // compact-brick LDS stage (tolerance arithmetic): rows hold 16-bit LDS slots, every sweep runs one block per brick
    bool brickWanted = false, brickFailed = false, listIsBrick = false;      // opt-in (SPHX_BRICK=1): measured 18 % slower than the quad walks
    int brickMin = 2000000;
    int brickBlocks = 0;                     // blocks per brick launch (about the number of non-empty bricks)
    std::unique_ptr<DArray<int>> brickTab;   // BrickTables of the non-empty bricks of this step
    bool brickMode() const;
    DArray<int> rowOverflow;                 // [0]: longest row beyond `cap` since the last check; [1]: brick stage fault flag; [2]: non-empty bricks of this step
    int capCheckSteps = 0;
    void tuneRowCapacity(int stepsSinceLastCall);
    int cellOffsetX = 0;                     // sub-grid offset of slab decompositions (GridDesc::xOff)
    int flags = 0;
    int quadMask = 1;                        // QuadBits (sweep_ops.hpp): sweeps that run quad-per-particle when rows exist
    int duoMask = 0;                         // QuadBits: sweeps that run with two lanes per particle
    int duoMaskLarge = 6;                    // ... additionally from 4 M particles on: head (2) and viscosity+colour (4)
    int quadMaskTol = 15;                    // tolerance arithmetic, quad walks with per-lane partial sums + one DPP reduction (r03, 10.3 M particles):
                                             // head -11 %, viscosity+colour -26 %, corrections -2..4 %; the surface sweeps (3 gathers, ~100 VGPRs)
                                             // lose 5x and stay lane-per-particle.  Below 4 M particles the corrections stay lane-per-particle too (mask & 7)
    // Small scenes (r05): below `smallBelow` particles a lane-per-particle launch has fewer waves than the device has SIMDs (20,736
    // particles: 324 waves for 1024 SIMDs) and a sweep takes as long as ONE wave's row walk; four lanes per particle cut that walk to a
    // quarter.  Measured (tools/quad_size_probe.py, free fall, strict): PBD(20) 0.787 -> 0.481 ms per step at 20,736 particles, 0.958 ->
    // 0.804 at 96,000, break-even near 130,000; adaptive DFSPH 0.228 -> 0.203; WCSPH 0.116 -> 0.109.  Every sweep that has the variant
    // then walks quad-per-particle (strict: same bits -- the schedule tests run all-quads; the surface sweeps stay lane-per-particle).
    int smallBelow = 131072;
    int quadMaskSmall = 255, quadMaskTolSmall = 255;
    // bumped whenever a host-side change invalidates launches recorded in a captured hipGraph (boundary
    // repack pending, arrays reallocated, engine switches changed); SPHSystem::stepN compares it
    unsigned int generation = 0;
    // range-restricted sweeps (slab drivers split a stage into edge and interior particles so that halo
    // traffic overlaps the interior): particles [rangeLo, rangeHi) of the next launches; -1 = all
    int rangeLo = -1, rangeHi = -1;
    bool rangeOrder = true;                 // SPHX_RANGE_ORDER=0: ranged launches always walk their tiles linearly (experiments)
    int rangeOrderMin = 3000000;            // particles a range must hold to keep the tile schedule (SPHX_RANGE_ORDER_MIN; tests lower it)
    int rangeLo2 = -1, rangeHi2 = -1;       // a second range behind the first, swept by the SAME launches (the two edge layers of a slab)
    bool keepErrorAccum = false;             // a later part of a split error stage adds to the running |error| total
    bool strictRateInTol = false;            // tolerance mode, >= 4 M particles: rate sweeps on the strict quad kernel (SweepCache::ctx); r05: off,
                                             // the tolerance walk with written-out FMAs fits 8 waves per SIMD too and wins (12.13 vs 12.32 ms per step)
    const int* gate = nullptr;               // device word that switches the following sweeps off (SweepCtx::gate)
    const int* advectSkipIf = nullptr;       // device word that, when raised, keeps the advect pass from moving anything (DFSPH loop-tail fault)
    // Skin rows (PBD, whole-domain systems): ONE row build per step with the cutoff enlarged by `skin`; sweeps
    // re-test every pair against the true support, `staleFlag` (device) is raised by the position update when a
    // particle has moved more than 0.45 * skin since the build (then every sweep walks the cells directly).
    bool tolerance = false;                  // row walks use the tolerance arithmetic (sph_device.hpp)
    bool skinRows = false;
    bool isSlab = false;
    float skin = 0.0f;                       // absolute length
    std::unique_ptr<DArray<float>> posBuild; // (x, y, z, -) of every fluid particle when the rows were built
    std::unique_ptr<DArray<int>> rowCell;    // ... and the cell its row was built around
    DArray<int> staleFlag;                   // two flags: [activeFlag] is raised by the coming position updates; [2] counts rebuilds;
                                             // [3 + activeFlag]: particles on the changed-cell list of the coming updates; [5] counts partial rebuilds
    int activeFlag = 0;
    std::unique_ptr<DArray<int>> changedList; // two lists of changedCap particle indices (SkinWatch)
    int changedCap = 0;
    SkinWatch skinWatch() const;
    float staleLimit2() const { return (0.45f * skin) * (0.45f * skin); }
    // persistent rows: a pair's separation changes by at most the sum of the two displacements (relative to the common drift), so
    // 0.49 skin each keeps every pair within R now inside the R + skin of the build; against the static boundary: 0.98 skin in full
    float persistLimit2() const { return (0.49f * skin) * (0.49f * skin); }
    // Persistent rows (SPHSystem's persistent mode; tolerance arithmetic, WCSPH / DFSPH, whole-domain systems): the solver steps
    // arrays that keep the order of the last row build; rows carry a skin no larger than the slack of the cell length
    // (cellLength - R, so that the 27-cell candidate walk still sees every pair within R + skin) and are rebuilt only when the
    // device-side check of the step's grid pass raises persistFlags[0].  csBuild = the fluid cell table of the build (the order
    // the arrays are in), for the cell walks of particles without a row.
    bool persistWanted = false, persistRows = false;
    DArray<int> persistFlags;                // [0] rebuild in this step (device), [1] forced by the host, [2] rebuilds so far, [3] steps so far
    std::unique_ptr<DArray<int>> csBuild;
    void requestRebuild();                   // host side: the next step rebuilds (rows reallocated, state rewritten)
    bool fluidValid = false;
    bool boundaryValid = false;
    bool listValid = false;
    bool allowTiles = true;                  // false when sweeps run on positions that were not binned (PBD)
    const int* listCsF = nullptr;            // cell tables the current rows were built from
    const int* listCsB = nullptr;
    const void* boundaryKey = nullptr;       // boundary pos pointer the packed copy was made from
    KernelConsts k{};
    GridDesc g{};
    float radiusKey = -1.0f, cellKey = -1.0f;
    int3 cellsKey = make_int3(0, 0, 0);

    void setup(int3 cellSize, float cellLength, float radius);
    void packFluid(const SPHParticles& fluids);
    // pack and apply the gravity kick vel += dv in one pass (only valid right after a re-sort)
    void packFluidKick(const SPHParticles& fluids, float3 dv);
    void packBoundary(const SPHParticles& boundaries);
    // make room for `count` boundary slots (contents of the fluid part are kept; pointers change)
    void reserveBoundary(int count);
    // (the tile schedule is only a launch order: it is kept for a few steps, particles drift slowly through the grid)
    void invalidatePositions() { fluidValid = false; listValid = false; if (++orderAge >= 8) orderValid = false; }
    int orderAge = 0, orderTiles = 0;
    void ensureTileOrder();
    // build the neighbour rows for the current positions (no-op when valid or disabled)
    void ensureList(const DArray<int>& csF, const DArray<int>& csB);
    void rebuildIfStale(const DArray<int>& csF, const DArray<int>& csB);
    void buildListForRange(const DArray<int>& csF, const DArray<int>& csB);     // rows of [rangeLo, rangeHi) only, unconditionally
    void launchBuild(const SweepCtx& c, float4* posBuildOut, const int* flagNow, int* flagNext);
    SweepCtx ctx(const DArray<int>& csF, const DArray<int>& csB) const;
    bool fused() const { return (flags & kFlagUnfused) == 0; }
    const float4* fluid4() const { return reinterpret_cast<const float4*>(posm.addr()); }
    float4* fluid4w() { return reinterpret_cast<float4*>(posm.addr()); }
    float4* vel4w() const { return reinterpret_cast<float4*>(vel4.addr()); }
    float4* cg4w() const { return reinterpret_cast<float4*>(cg4.addr()); }
    float4* posfw() const { return reinterpret_cast<float4*>(posf.addr()); }
    const float4* boundary4() const { return fluid4() + capN; }
};

inline unsigned int sweep_grid_for(int numTiles) { return xcd_grid(numTiles * kTile, kWideBlock); }
inline unsigned int blocks_for(int n, int block = 256) { return n > 0 ? (unsigned int)((n - 1) / block + 1) : 1u; }

// optional per-kernel timing (sphx_profile_step): when enabled every launch helper brackets the
// kernel with hipEvents on sphx::stream().
struct KernelTimer {
    static bool enabled;
    static std::string filter;   // when non-empty only spans with exactly this name are recorded
    static void begin(const char* name);
    static void end();
    static void collect(std::vector<std::string>& names, std::vector<float>& ms);
    static void reset();
};
struct ScopedKernel {
    explicit ScopedKernel(const char* name)
        : on(KernelTimer::enabled && (KernelTimer::filter.empty() || KernelTimer::filter == name))
    {
        if (on) KernelTimer::begin(name);
    }
    ~ScopedKernel() { if (on) KernelTimer::end(); }
    bool on;
};

// generic element-wise device helpers implemented in elementwise.hip
void ew_gather_float3(float3* dst, const float3* src, const int* perm, int n);
void ew_gather_float(float* dst, const float* src, const int* perm, int n);
void ew_gather_int(int* dst, const int* src, const int* perm, int n);
void ew_copy(void* dst, const void* src, size_t bytes);
// the same, executed only when the device word *flag is non-zero (persistent rows: launched every step, replayable in a graph)
void ew_gather_float_if(float* dst, const float* src, const int* perm, int n, const int* flag);
void ew_copy_float_if(float* dst, const float* src, int n, const int* flag);
void ew_fill_float(float* dst, float value, int n);
void ew_iota(int* dst, int n);

void device_exclusive_scan(int* data, int count, int* blockSums);
void device_exclusive_scan3(int* a, int* b, int* c, int count, int* blockSums);   // blockSums: 3 x (count / 2048 + 2) ints
void use_external_stream(hipStream_t s);
// every launch of the enclosed scope goes to `s` instead of the engine stream (single host thread; the slab layer sweeps the edge
// layers of a stage on a stream of their own beside the interior)
struct ScopedStream {
    explicit ScopedStream(hipStream_t s);
    ~ScopedStream();
    ScopedStream(const ScopedStream&) = delete;
    ScopedStream& operator=(const ScopedStream&) = delete;
    hipStream_t previous;
};
const std::string& last_error_text();
void set_error_text(const std::string& s);

}  // namespace sphx
