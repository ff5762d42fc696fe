// DFSPHSolver.h — divergence-free SPH solver (reference: src/DFSPHSolver.h:20-65,
// src/DFSPHSolver.cu:33-363).
//
// step() = density+alpha -> divergence-free solve -> gravity, viscosity, surface effects ->
// constant-density solve with warm start -> advect.  Loop semantics follow
// SURVEY.md Q9; the termination sum is an exact fixed-point reduction (deviation D2).
// Engine extension: setFixedIterations(v, d) runs exactly v divergence and d density iterations
// with no device->host read-back, which makes the whole step hipGraph-capturable.
#pragma once

#include "BasicSPHSolver.h"

class DFSPHSolver final : public BasicSPHSolver {
public:
    explicit DFSPHSolver(int num, float defaultDensityErrorThreshold = 1e-3f,
                         float defaultDivergenceErrorThreshold = 1e-3f, int defaultMaxIter = 20);
    virtual ~DFSPHSolver() noexcept;

    virtual void step(std::shared_ptr<SPHParticles>& fluids,
                      const std::shared_ptr<SPHParticles>& boundaries,
                      const DArray<int>& cellStartFluid, const DArray<int>& cellStartBoundary,
                      float3 spaceSize, int3 cellSize, float cellLength, float radius, float dt,
                      float rho0, float rhoB, float stiff, float visc, float3 G,
                      float surfaceTensionIntensity, float airPressure) override;

    // --- engine extensions ---------------------------------------------------------------
    void setFixedIterations(int divergenceIters, int densityIters)
    {
        fixedDiv = divergenceIters;
        fixedDen = densityIters;
    }
    // Fixed iteration counts: no host decision inside the step.  Adaptive control (the reference's default) is graph-safe too
    // when the termination test runs on the device (whole-domain systems; SPHX_DFSPH_HOST_LOOP=1 restores the host loop)
    bool graphSafe() const override;
    // one stage of the fused schedule (values of sphx_phase in sphx_c.h, except SEARCH which here
    // means "prepare": pack, neighbour rows, carry the warm stiffness through the sort).
    // `reduce` accumulates the |error| total of the error stages for readErrorTotal().
    void runPhase(int phase, std::shared_ptr<SPHParticles>& fluids,
                  const std::shared_ptr<SPHParticles>& boundaries, const DArray<int>& cellStartFluid,
                  const DArray<int>& cellStartBoundary, float3 spaceSize, int3 cellSize, float cellLength,
                  float radius, float dt, float rho0, float rhoB, float visc, float3 G,
                  float surfaceTensionIntensity, float airPressure, bool reduce = false);
    int lastDivergenceIterations() { fetchIterations(); return lastDiv; }
    int lastDensityIterations() { fetchIterations(); return lastDen; }
    // distributed adaptive mode: restrict the |error| total to [lo, hi) and read it as the raw integer
    void setErrorSumRange(int lo, int hi) { sumLo = lo; sumHi = hi; }
    long long readErrorTotalFixed();
    void resetErrorTotal();          // zero the |error| accumulators on the current stream (split error stages then all ADD)
    void noteIterations(int div, int den) { lastDiv = div; lastDen = den; itersPending = false; }
    // Slab drivers sweep owned particles only: after the VISC_COLOR stage (which leaves the warm-start stiffness of the particles it
    // swept in posf.w) this writes it for the GHOST particles [lo, hi) as well, so that the fused surface + warm-start sweep can
    // take the stiffness from the record it gathers anyway (two gathers per pair instead of three), as whole-domain steps do
    void packWarmIntoPosf(int lo, int hi);
    // the warm-start stiffness of this step is in sorted order already (a slab driver's staged sort wrote it there): the coming
    // SEARCH stage leaves its permutation out (one-shot)
    void noteWarmStiffnessSorted() { warmSorted = true; }
    // stage-wise drivers with fixed counts: the DIV_CORRECT stages run while this is on also apply the gravity kick
    // vel += dt G (BasicSPHSolver::force, BasicSPHSolver.cu:227-235) in their store, as DFSPHSolver::step does for the last
    // divergence correction of a whole-domain step; the driver then leaves the FORCE stage out
    void setKickInCorrect(bool on, float dt = 0.0f, float3 G = {0.0f, 0.0f, 0.0f})
    {
        kickInCorrect = on;
        if (on) kickDv = make_float3(dt * G.x, dt * G.y, dt * G.z);
    }
    const DArray<float>& getAlpha() const { return alpha; }
    const DArray<float>& getStiffness() const { return bufferFloat; }
    const DArray<float>& getError() const { return error; }
    DArray<float>& getWarmStiffness() { return denWarmStiff; }
    void permuteState(const int* perm, int n) override;

    void tune(int stepsSinceLastCall) override;      // engine: row capacity (BasicSPHSolver) + the windows of the device-decided loops
    int tuneInterval() const override;               // device-decided loops on large scenes: the windows follow the counts every 4 steps
protected:
    // hides BasicSPHSolver::project (different signature), as in the reference
    virtual int project(std::shared_ptr<SPHParticles>& fluids,
                        const std::shared_ptr<SPHParticles>& boundaries,
                        const DArray<int>& cellStartFluid, const DArray<int>& cellStartBoundary,
                        float rho0, int3 cellSize, float cellLength, float radius, float dt,
                        float errorThreshold, int maxIter);

private:
    void computeDensityAlpha(std::shared_ptr<SPHParticles>& fluids,
                             const std::shared_ptr<SPHParticles>& boundaries,
                             const DArray<int>& cellStartFluid,
                             const DArray<int>& cellStartBoundary, int3 cellSize, float cellLength,
                             float radius);
    int correctDivergenceError(std::shared_ptr<SPHParticles>& fluids,
                               const std::shared_ptr<SPHParticles>& boundaries,
                               const DArray<int>& cellStartFluid,
                               const DArray<int>& cellStartBoundary, float rho0, int3 cellSize,
                               float cellLength, float radius, float dt, float errorThreshold,
                               int maxIter);
    float readErrorTotal();
    int sumLo = 0, sumHi = 0x7fffffff;   // particles entering the |error| total (slab drivers: owned range)

    DArray<float> alpha;
    DArray<float> bufferFloat;     // stiffness kappa
    DArray<float> error;
    DArray<float> denWarmStiff;
    DArray<float> scratch;         // permutation target for denWarmStiff
    DArray<int> errorAccum;        // 64-bit fixed-point accumulators: kErrorSlots partial sums, one cache line apart
    const float densityErrorThreshold;
    const float divergenceErrorThreshold;
    const int maxIter;
    int fixedDiv = -1, fixedDen = -1;
    bool warmSorted = false;
    bool warmInPosfAll = false, warmInPosfGhosts = false;   // this step's posf.w carries the warm stiffness: for every held particle / for the ghosts too
    bool headDidFirstError = false;   // the fused head sweep already produced the first divergence error
    int lastDiv = 0, lastDen = 0;
    // device-side adaptive loops: {done, iteration, divergence iterations, density iterations, grid-barrier word} of the current step on the device,
    // copied to pinned host memory at the end of every step and read when somebody asks
    bool deviceLoops() const;
    void fetchIterations();
    // iterations beyond the first one(s) of a device-decided loop in ONE persistent launch (sweep_ops.hpp, k_dfsph_loop_tail);
    // false: not available in this configuration, the caller enqueues gated launches
    // how many iterations of each loop are enqueued as ordinary (gated) launches before the tail: the count of the last steps + 2.
    // A prediction only -- the tail runs whatever is left, so the results do not depend on it; a change re-captures the step graph.
    int windowDiv = 3, windowDen = 4;
    // fixed counts: the coming divergence correction is the last one of the step and also applies the gravity kick (OpCorrect::addKick)
    bool tailFailed = false;         // a grid barrier of the tail timed out once: gated launches from then on
    bool kickInCorrect = false;
    float3 kickDv = {0.0f, 0.0f, 0.0f};
    void adaptWindows();
    bool runLoopTail(bool densityLoop, std::shared_ptr<SPHParticles>& fluids, const DArray<int>& cellStartFluid,
                     const DArray<int>& cellStartBoundary, float dt, float rho0, float threshold, int minIter, int which);
    DArray<int> loopState;
    int* hostIters = nullptr;
    bool itersPending = false;
};
