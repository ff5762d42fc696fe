// BasicSPHSolver.h — weakly-compressible SPH solver (reference: src/BasicSPHSolver.h:20-52,
// src/BasicSPHSolver.cu:32-381).
//
// step() = gravity -> viscosity -> [colour-gradient surface detection + surface tension / air
// pressure] -> density -> Tait pressure -> pressure force -> advect + box clamp.
// The protected virtuals keep the reference's names and argument lists so derived solvers
// (DFSPHSolver, PBDSolver, user subclasses) compose the same building blocks.
//
// Engine side: every neighbour sweep is a hand-written HIP kernel over cell-sorted, float4-packed
// views (position+mass) that the solver refreshes whenever positions change; see DESIGN.md.
#pragma once

#include <memory>
#include "BaseSolver.h"

namespace sphx { struct SweepCache; }

class BasicSPHSolver : public BaseSolver {
public:
    explicit BasicSPHSolver(int num);
    virtual ~BasicSPHSolver() noexcept;

    virtual void step(std::shared_ptr<SPHParticles>& fluids,
                      const std::shared_ptr<SPHParticles>& boundaries,
                      const DArray<int>& cellStartFluid, const DArray<int>& cellStartBoundary,
                      float3 spaceSize, int3 cellSize, float cellLength, float radius, float dt,
                      float rho0, float rhoB, float stiff, float visc, float3 G,
                      float surfaceTensionIntensity, float airPressure) override;

    bool graphSafe() const override { return true; }
    unsigned int graphGeneration() const override;
    void prepareForCapture() override;
    void captureFailed() override;
    void tune(int stepsSinceLastCall) override;      // engine: adaptive neighbour-row capacity
    // engine extensions: the colour gradient left by handleSurface(), and engine switches used by
    // the tests (bit 0: run the reference-structure, unfused sequence of building blocks;
    // bit 1: walk the 27 cells directly instead of the per-step neighbour list;
    // bit 2: stage neighbour ranges in LDS per 64-particle tile)
    const DArray<float3>& getColorGradient() const { return bufferFloat3; }
    void setEngineFlags(int flags);
    // arithmetic of the neighbour sweeps: strict (default; every bit equals the IEEE evaluation of the reference's
    // expressions) or tolerance (hardware rsq / rcp and fused multiply-adds: deviations of a few 1e-7 per pair term)
    void setToleranceArithmetic(bool on);
    // Persistent neighbour rows (tolerance arithmetic only; driven by SPHSystem::setPersistentRows): the rows carry a skin and
    // survive from step to step until a device-side displacement check asks for a rebuild; the solver then steps particle
    // arrays that stay in the order of the last build.  preparePersistent() decides whether the mode can be used for this
    // grid (it needs cellLength > radius: the skin lives in that slack) and returns what SPHSystem's grid pass needs.
    struct PersistentView { bool active; int* flags; const void* posBuild; float limit2; };
    void requestPersistentRows(bool on);
    PersistentView preparePersistent(int3 cellSize, float cellLength, float radius);
    void requestRowRebuild();
    const int* enginePersistFlags() const;  // device words {rebuild now, forced, row builds so far, steps so far}; nullptr when not in use
    // bring the solver's own per-particle arrays from the persistent order into the order slot -> perm[slot]
    virtual void permuteState(const int* perm, int n);
    // slab decompositions: global x index of this solver's local cell column 0
    void setCellOffsetX(int cellOffsetX);
    // raw device pointers of the float4 mirrors the sweeps gather from (halo exchange targets)
    void* engineVel4() const;
    void* engineCg4() const;
    void* enginePterm() const;
    void* enginePos4() const;
    void* enginePosf() const;
    // per-particle neighbour-row lengths of the most recent row build (bench statistics)
    const int* engineRowCounts() const;
    int engineRowCapacity() const;
    const int* engineStaleFlag() const;     // device counter of conditional skin-row rebuilds (nullptr when not in use)
    // reserve the boundary part of the engine's unified neighbour arrays (called once by SPHSystem
    // before any engine pointer is handed out; otherwise done lazily by the first step)
    void reserveBoundary(int count);
    // one stage of the fused WCSPH schedule (SPHX_PH_W_* and SPHX_PH_ADVECT of sphx_c.h), for
    // distributed drivers that refresh halo fields between stages
    void runWcsphPhase(int phase, std::shared_ptr<SPHParticles>& fluids,
                       const std::shared_ptr<SPHParticles>& boundaries, const DArray<int>& cellStartFluid,
                       const DArray<int>& cellStartBoundary, float3 spaceSize, int3 cellSize, float cellLength,
                       float radius, float dt, float rho0, float rhoB, float stiff, float visc, float3 G,
                       float surfaceTensionIntensity, float airPressure);
    // restrict the sweeps of the following stages to particles [lo, hi) (lo < 0: all).  Slab drivers run a
    // stage on the edge particles first, start the halo exchange, then run it on the interior.
    void setSweepRange(int lo, int hi, bool keepErrorAccum = false, int lo2 = -1, int hi2 = -1);
    // call after writing boundary positions/masses through raw pointers
    void invalidateBoundary();

protected:
    virtual void force(std::shared_ptr<SPHParticles>& fluids, float dt, float3 G) override final;
    virtual void advect(std::shared_ptr<SPHParticles>& fluids, float dt, float3 spaceSize) override final;
    virtual void project(std::shared_ptr<SPHParticles>& fluids,
                         const std::shared_ptr<SPHParticles>& boundaries,
                         const DArray<int>& cellStartFluid, const DArray<int>& cellStartBoundary,
                         float rho0, float stiff, int3 cellSize, float cellLength, float radius,
                         float dt);
    virtual void diffuse(std::shared_ptr<SPHParticles>& fluids, const DArray<int>& cellStartFluid,
                         int3 cellSize, float cellLength, float rho0, float radius, float visc,
                         float dt);
    virtual void handleSurface(std::shared_ptr<SPHParticles>& fluids,
                               const std::shared_ptr<SPHParticles>& boundaries,
                               const DArray<int>& cellStartFluid,
                               const DArray<int>& cellStartBoundary, float rho0, float rhoB,
                               int3 cellSize, float cellLength, float radius, float dt,
                               float surfaceTensionIntensity, float airPressure);

    // engine: packed per-step views shared with the derived solvers
    sphx::SweepCache& cache() { return *_cache; }
    // marks the packed position view stale (call after anything that moves particles)
    void invalidatePositions();
    DArray<float3>& colorGradientBuffer() { return bufferFloat3; }

private:
    DArray<float3> bufferFloat3;   // viscosity delta-v, then the colour gradient
    std::unique_ptr<sphx::SweepCache> _cache;
};
