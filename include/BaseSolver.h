// BaseSolver.h — the solver plug-in boundary (reference: src/BaseSolver.h:20-31).
//
// SPHSystem::step() calls step() once per time step on the host thread, after the neighbour
// search, with the fluid/boundary particle sets (cell-sorted), both cell-start tables (C+1
// entries, exclusive prefix sums) and the 14 simulation scalars.  Implementations enqueue work
// on sphx::stream(); they may throw `const char*` (SPHSystem catches and prints, as the
// reference does).  Third-party solvers written against the reference interface plug in here.
#pragma once

#include <memory>
#include "SPHParticles.h"

class BaseSolver {
public:
    virtual void step(std::shared_ptr<SPHParticles>& fluids,
                      const std::shared_ptr<SPHParticles>& boundaries,
                      const DArray<int>& cellStartFluid, const DArray<int>& cellStartBoundary,
                      float3 spaceSize, int3 cellSize, float cellLength, float radius, float dt,
                      float rho0, float rhoB, float stiff, float visc, float3 G,
                      float surfaceTensionIntensity, float airPressure) = 0;
    virtual ~BaseSolver() {}

    // engine extension: true when step() performs no host synchronisation or host-side branching
    // on device results, i.e. when a whole SPHSystem::step() may be captured into a hipGraph.
    virtual bool graphSafe() const { return false; }
    // changes whenever host-side state a captured graph depends on changed (buffers reallocated,
    // boundary data rewritten, engine switches); a replaying caller must re-capture
    virtual unsigned int graphGeneration() const { return 0; }
    // called right before a step is captured into a hipGraph: anything the solver refreshes only every few steps
    // must be part of the captured step (a replay repeats exactly what was recorded)
    virtual void prepareForCapture() {}
    // called when that capture was thrown away (the step then runs eagerly): whatever the recorded-but-never-run launches were
    // meant to refresh is stale and must be redone by the next eager step
    virtual void captureFailed() {}
    // called between steps (never inside a captured graph) with the number of steps run since the last call: a place
    // for host-side adaptation from device-side statistics; may change graphGeneration()
    virtual void tune(int stepsSinceLastCall) { (void)stepsSinceLastCall; }
    // how many steps of a batch (SPHSystem::stepN) may be enqueued between two calls of tune()
    virtual int tuneInterval() const { return 16; }

protected:
    virtual void advect(std::shared_ptr<SPHParticles>& fluids, float dt, float3 spaceSize) = 0;
    virtual void force(std::shared_ptr<SPHParticles>& fluids, float dt, float3 G) = 0;
};
