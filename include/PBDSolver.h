// PBDSolver.h — position-based fluids solver (reference: src/PBDSolver.h:20-86,
// src/PBDSolver.cu:34-258).
//
// step() = [first call: remember positions and throw] -> carry last positions through this
// step's sort -> maxIter x {lambda sweep, delta-p sweep, apply + clamp} on a FIXED cell table ->
// velocity from displacement -> XSPH viscosity -> surface effects -> gravity -> remember
// positions, advect + clamp.  XSPH is evaluated Jacobi-style (read old, write
// new); the reference kernel updates in place and is racy.
#pragma once

#include "BasicSPHSolver.h"

class PBDSolver final : public BasicSPHSolver {
public:
    explicit PBDSolver(int num, int defaultMaxIter = 20, float defaultXSPH_c = 0.05f,
                       float defaultRelaxation = 0.75f);
    explicit PBDSolver(const std::shared_ptr<SPHParticles>& particles, int defaultMaxIter = 20,
                       float defaultXSPH_c = 0.1f, float defaultRelaxation = 1.0f);
    virtual ~PBDSolver() noexcept;

    virtual void step(std::shared_ptr<SPHParticles>& fluids,
                      const std::shared_ptr<SPHParticles>& boundaries,
                      const DArray<int>& cellStartFluid, const DArray<int>& cellStartBoundary,
                      float3 spaceSize, int3 cellSize, float cellLength, float radius, float dt,
                      float rho0, float rhoB, float stiff, float visc, float3 G,
                      float surfaceTensionIntensity, float airPressure) override;

    void initializePosLast(const DArray<float3>& posFluid);

    bool graphSafe() const override { return posLastInitialized; }
    void tune(int stepsSinceLastCall) override;
    const DArray<float3>& getPosLast() const { return fluidPosLast; }
    // engine extension (snapshot restore): the last positions were written through the raw pointer
    void markPosLastInitialized() { posLastInitialized = true; }
    const DArray<float>& getLambda() const { return bufferFloat; }
    // one stage of the PBD schedule (SPHX_PH_P_* of sphx_c.h), for distributed drivers that refresh
    // halo fields between stages
    void runPhase(int phase, std::shared_ptr<SPHParticles>& fluids, const std::shared_ptr<SPHParticles>& boundaries,
                  const DArray<int>& cellStartFluid, const DArray<int>& cellStartBoundary, float3 spaceSize,
                  int3 cellSize, float cellLength, float radius, float dt, float rho0, float rhoB, float3 G,
                  float surfaceTensionIntensity, float airPressure);

protected:
    void predict(std::shared_ptr<SPHParticles>& fluids, float dt, float3 spaceSize);
    virtual int project(std::shared_ptr<SPHParticles>& fluids,
                        const std::shared_ptr<SPHParticles>& boundaries,
                        const DArray<int>& cellStartFluid, const DArray<int>& cellStartBoundary,
                        float rho0, int3 cellSize, float3 spaceSize, float cellLength, float radius,
                        int maxIter);
    virtual void diffuse(std::shared_ptr<SPHParticles>& fluids, const DArray<int>& cellStartFluid,
                         int3 cellSize, float cellLength, float rho0, float radius, float visc);

private:
    void updateNeighborhood(const std::shared_ptr<SPHParticles>& particles);
    void applyDelta(std::shared_ptr<SPHParticles>& fluids, float3 spaceSize, int num);
    void configureSkin(float radius);
    // skin-row controller: rows with a skin pay off while particles stay in their cells between Jacobi iterations;
    // when the device-side rebuild counter shows that most iterations rebuild anyway, plain per-iteration rebuilds
    // (shorter rows) are cheaper.  Checked every 32 steps, retried after 256.
    bool skinWanted = true;
    int tuneSteps = 0, skinOffSteps = 0, lastRebuilds = 0;

    bool posLastInitialized = false;
    const int maxIter;
    const float xSPH_c;
    const float relaxation;
    DArray<float3> fluidPosLast;
    DArray<float3> bufferFloat3;   // delta-p, XSPH output, sort scratch
    DArray<float> bufferFloat;     // lambda
};
