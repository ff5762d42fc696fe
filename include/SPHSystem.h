// SPHSystem.h — owner of the particle sets, the uniform grid and the time-step loop of the
// drop-in API (reference: src/SPHSystem.h:20-84, src/SPHSystem.cu:33-158).
//
// Construction takes ownership by MOVING from the caller's shared_ptrs (they become null), then
// runs: boundary neighbour search -> boundary masses -> uniform fluid mass -> fluid neighbour
// search -> one full step().  step() = neighbour search + solver->step() + device
// sync, returns the hipEvent-timed milliseconds.  The neighbour search is a stable counting sort
// by linear cell id (x slowest) written as HIP kernels; it reproduces thrust::sort_by_key's
// permutation exactly (DESIGN.md "Neighbour grid").
#pragma once

#include <functional>
#include <memory>
#include "BaseSolver.h"

namespace sphx { struct GridScratch; struct StepGraph; struct PersistState; }

class SPHSystem {
public:
    SPHSystem(std::shared_ptr<SPHParticles>& fluidParticles,
              std::shared_ptr<SPHParticles>& boundaryParticles, std::shared_ptr<BaseSolver>& solver,
              float3 spaceSize, float sphCellLength, float sphSmoothingRadius, float dt, float sphM0,
              float sphRho0, float sphRhoBoundary, float sphStiff, float sphVisc,
              float sphSurfaceTensionIntensity, float sphAirPressure, float3 sphG, int3 cellSize);
    SPHSystem(const SPHSystem&) = delete;
    SPHSystem& operator=(const SPHSystem&) = delete;
    ~SPHSystem() noexcept;

    float step();

    int size() const { return fluidSize(); }
    int fluidSize() const { return (*_fluids).size(); }
    int boundarySize() const { return (*_boundaries).size(); }
    int totalSize() const { return (*_fluids).size() + (*_boundaries).size(); }
    auto getFluids() const { return static_cast<const std::shared_ptr<SPHParticles>>(_fluids); }
    auto getBoundaries() const { return static_cast<const std::shared_ptr<SPHParticles>>(_boundaries); }

    // --- engine extensions -------------------------------------------------------------------
    // n steps back to back with a single sync at the end; replays a captured hipGraph when the
    // solver reports graphSafe().  Returns the hipEvent time of the whole batch in ms.
    float stepN(int n);
    // engine construction without the trailing step() of the reference constructor (used by the
    // C ABI when run_ctor_step = 0, for per-kernel tests)
    struct NoInitialStep {};
    SPHSystem(NoInitialStep, std::shared_ptr<SPHParticles>& fluidParticles,
              std::shared_ptr<SPHParticles>& boundaryParticles, std::shared_ptr<BaseSolver>& solver,
              float3 spaceSize, float sphCellLength, float sphSmoothingRadius, float dt, float sphM0,
              float sphRho0, float sphRhoBoundary, float sphStiff, float sphVisc,
              float sphSurfaceTensionIntensity, float sphAirPressure, float3 sphG, int3 cellSize);
    // continuation of a saved run (sphx_snapshot_load): the fluid arrays are taken in the order given —
    // no initial fluid sort, no step — so that the next step() sorts exactly what an uninterrupted run
    // would have sorted (a stable sort's result depends on the incoming order)
    struct Restored {};
    SPHSystem(Restored, std::shared_ptr<SPHParticles>& fluidParticles,
              std::shared_ptr<SPHParticles>& boundaryParticles, std::shared_ptr<BaseSolver>& solver,
              float3 spaceSize, float sphCellLength, float sphSmoothingRadius, float dt, float sphM0,
              float sphRho0, float sphRhoBoundary, float sphStiff, float sphVisc,
              float sphSurfaceTensionIntensity, float sphAirPressure, float3 sphG, int3 cellSize);
    // slab decompositions: this system covers a sub-grid whose local cell column 0 is global column
    // `cellOffsetX` (cellSize.x = local columns incl. one ghost layer per side); no initial step.
    // phase(p) runs one stage of the DFSPH step (see sphx_phase in sphx_c.h) so that a distributed
    // driver can refresh halo fields between stages.
    struct Slab { int cellOffsetX; };
    SPHSystem(Slab, std::shared_ptr<SPHParticles>& fluidParticles,
              std::shared_ptr<SPHParticles>& boundaryParticles, std::shared_ptr<BaseSolver>& solver,
              float3 spaceSize, float sphCellLength, float sphSmoothingRadius, float dt, float sphM0,
              float sphRho0, float sphRhoBoundary, float sphStiff, float sphVisc,
              float sphSurfaceTensionIntensity, float sphAirPressure, float3 sphG, int3 cellSize);
    void phase(int p);
    void phaseReduce(int p, int sumLo, int sumHi);   // error stages with the |error| total over [lo, hi)
    long long errorTotalFixed();
    void resetErrorTotal();
    // a stage on particles [lo, hi) only (lo < 0: all): slab drivers sweep the edge layers first, start the halo
    // exchange of the stage's output, then sweep the interior
    void phaseEx(int p, int lo, int hi, bool reduce, int sumLo, int sumHi, bool keepAccum, int lo2 = -1, int hi2 = -1);
    // called inside a SEARCH stage right after the sort, before packing and row construction are enqueued: a slab driver reads
    // its layer boundaries from the fresh cell table there, so that the read-back overlaps the row build
    void setAfterSortHook(std::function<void()> hook) { _afterSort = std::move(hook); }
    // Slab drivers hand the particles of a step over as PAYLOAD ROWS in three device buffers -- [from the left neighbour | kept |
    // from the right neighbour], each row = pos(3) vel(3) id(1, bit pattern) + `extraFloats` solver floats -- which is the pre-sort
    // order of the step.  The next SEARCH stage (DFSPH / WCSPH) then sorts the rows STRAIGHT into the particle arrays: cell keys from
    // the staged positions, the same stable permutation, one gather that writes positions, velocities, ids and the solver's extra
    // array (DFSPH: the warm-start stiffness) in sorted order -- no unpack pass, no copy back, no separate permutation of the extras
    // (r05; same bits as unpacking and sorting in place).  One-shot: consumed by that stage.
    struct StagedRows { const float* rows[3]; int count[3]; int extraFloats; float* extraOut; };
    void setStagedInput(const StagedRows& staged) { _staged = staged; _hasStaged = true; }
    // ... and the cell columns [colLo, colHi) its held particles can lie in, plus an empty column on either side for the 27-cell
    // walks: the staged sort then clears, counts and scans only that window of the whole-grid cell table (every slab's engine works
    // on the global grid: 7.5 M cells at 10 M particles, of which a slab of an 8-GPU run touches a sixth).  Cells outside the window
    // keep stale values and are never read; a particle found outside it is filed in the out-of-grid bucket (the driver's layer check
    // then reports it).  colLo < 0: the whole grid.
    void setCellWindow(int colLo, int colHi) { _winLo = colLo; _winHi = colHi; }
    // Persistent neighbour rows (opt-in; tolerance arithmetic, WCSPH / DFSPH, whole-domain systems; C ABI: reserved[3] = 2).
    // The solver steps a working copy of the fluid arrays that stays in the order of the last row build, the rows carry a skin
    // and are rebuilt only when a device-side check finds that some particle has moved more than 0.49 skin relative to the
    // others; getFluids()' arrays, particle2Cell, getSortPerm() and the cell table are brought up to date in the reference's
    // order (stable sort by the cells of the step's starting positions) at the end of every step() as always.  Results meet the
    // tolerance contract, not the strict one (the order of a particle's sums follows the rows).  A caller that WRITES the fluid
    // arrays through raw pointers between steps must call invalidatePersistentOrder() afterwards; the C ABI's setters do.
    // Returns false when the mode cannot be used as the system stands (solver without the engine's rows, PBD, slab systems,
    // strict arithmetic, cellLength <= radius, engine switches that exclude the rows): the request is remembered all the same
    // and takes effect once the obstacle is gone (e.g. setToleranceArithmetic(true) afterwards).
    // While (nearly) every step rebuilds its rows -- violent phases -- a host-side controller (sphx_tuning.persist_controller)
    // leaves the mode for 256 steps and runs the plain tolerance step, as PBD's skin rows do.
    bool setPersistentRows(bool on);
    bool persistentRows() const;
    // API slot -> working index while the solver's own arrays are in the working order (nullptr otherwise): readers of
    // solver-internal fields gather through it instead of flushing the mode
    const int* persistentSlotMap() const;
    // make every per-particle array of the solver follow the API order again (host-side readers of solver-internal fields,
    // snapshots, setters); the next step() re-primes the working copy from the API arrays and rebuilds the rows
    void invalidatePersistentOrder();
    const DArray<int>& getCellStartFluid() const { return _fluidCellStart; }
    const DArray<int>& getCellStartBoundary() const { return _wallCellStart; }
    BaseSolver* getSolver() const { return _solver.get(); }

private:
    void initialise(float sphM0, bool runStep, bool sortFluid = true);
    void computeBoundaryMass();
    void neighborSearch(const std::shared_ptr<SPHParticles>& particles, DArray<int>& cellStart);
    void neighborSearchStaged(const StagedRows& staged);      // the fluid sort of a slab step, fed from payload rows
    void enqueueStep();   // neighbour search + solver step, no sync
    bool persistentActive();          // the mode is on AND usable for this solver and grid AND not suspended by the controller
    void persistentController(int stepsSinceLastCall);   // between steps: suspend / resume the mode by its rebuild rate
    void persistentPrime();           // working copy := API arrays, identity map, rows to be rebuilt
    void persistentSearch();          // cells of the API slots, displacement check, stable sort of the slot map, conditional re-sort
    void persistentExport();          // API arrays := working copy through the slot map
    std::shared_ptr<SPHParticles> _work;                 // persistent mode: the arrays the solver steps (order of the last row build)
    std::unique_ptr<sphx::PersistState> _persist;

    // who: the two particle sets and the solver plugin (owned; moved in by the constructor)
    std::shared_ptr<SPHParticles> _fluids;
    const std::shared_ptr<SPHParticles> _boundaries;
    std::shared_ptr<BaseSolver> _solver;
    // what: the scalars handed to BaseSolver::step every frame
    struct Scalars {
        float3 space; int3 cells; float cellLength, radius, dt, rho0, rhoBoundary, stiff, visc, surfaceTension, airPressure;
        float3 gravity;
    };
    const Scalars _sc;
    // where: uniform grid tables (C+1 exclusive prefix sums; slot C = out-of-grid bucket) and scratch
    DArray<int> _fluidCellStart;
    DArray<int> _wallCellStart;
    DArray<int> _intScratch;
    std::unique_ptr<sphx::GridScratch> _grid;
    std::unique_ptr<sphx::StepGraph> _graph;
    std::function<void()> _afterSort;
    StagedRows _staged{};
    bool _hasStaged = false;
    int _winLo = -1, _winHi = -1;
    int _cellOffsetX = 0;     // slab decompositions: global x index of local cell column 0
    bool _slab = false;
};
