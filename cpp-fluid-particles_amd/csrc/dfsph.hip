// dfsph.hip — DFSPHSolver (divergence-free SPH, Bender & Koschier 2015) as HIP kernels for gfx950.
//
// Reference behaviour restated from src/DFSPHSolver.cu:33-363: step order, the two
// solver loops and their warm-start bookkeeping (Q9).  Differences in mechanism, not in result:
// the warm-stiffness array follows the particles through the published sort permutation instead
// of a second key sort; the warm reset / accumulate passes and the |error| reduction are fused
// into the error kernel; the termination sum is an exact fixed-point integer, read
// back only in adaptive mode.
#include <limits>

#include "DFSPHSolver.h"
#include "engine.hpp"
#include "sweep_ops.hpp"

using namespace sphx;

DFSPHSolver::DFSPHSolver(int num, float defaultDensityErrorThreshold, float defaultDivergenceErrorThreshold,
                         int defaultMaxIter)
    : BasicSPHSolver(num), alpha((unsigned)num), bufferFloat((unsigned)num), error((unsigned)num),
      denWarmStiff((unsigned)num), scratch((unsigned)num), errorAccum(2u),
      densityErrorThreshold(defaultDensityErrorThreshold), divergenceErrorThreshold(defaultDivergenceErrorThreshold),
      maxIter(defaultMaxIter)
{
}
DFSPHSolver::~DFSPHSolver() noexcept {}

float DFSPHSolver::readErrorTotal()
{
    unsigned long long acc = 0;
    HIP_CALL(hipMemcpyAsync(&acc, errorAccum.addr(), sizeof(acc), hipMemcpyDeviceToHost, sphx::stream()));
    HIP_CALL(hipStreamSynchronize(sphx::stream()));
    return (float)((double)(long long)acc * (1.0 / 4294967296.0));
}

namespace {
template <bool DENSITY_MODE, int WARM>
void launch_rate(const OpRate& op, int n, bool reduce)
{
    if (n <= 0) return;
    OpRate o = op;
    if (!reduce) o.out.accum = nullptr;
    else HIP_CALL(hipMemsetAsync(o.out.accum, 0, sizeof(unsigned long long), sphx::stream()));
    launch_rate_kernel<DENSITY_MODE, WARM>(o, n);
}
}  // namespace

// computeDensityAlpha, DFSPHSolver.cu:251-259
void DFSPHSolver::computeDensityAlpha(std::shared_ptr<SPHParticles>& fluids, const std::shared_ptr<SPHParticles>& boundaries,
                                      const DArray<int>& cellStartFluid, const DArray<int>& cellStartBoundary,
                                      int3 cellSize, float cellLength, float radius)
{
    SweepCache& c = cache();
    c.setup(cellSize, cellLength, radius);
    c.packFluid(*fluids);
    c.packBoundary(*boundaries);
    c.ensureList(cellStartFluid, cellStartBoundary);
    const int num = (int)fluids->size();
    if (num <= 0) return;
    ScopedKernel t("density_alpha");
    OpDfsphHead op{c.ctx(cellStartFluid, cellStartBoundary), nullptr, fluids->getDensityPtr(), alpha.addr(), RateOut{}};
    launch_dfsph_head<false>(op, num);
}

// correctDivergenceError, DFSPHSolver.cu:331-363.  `firstErrorDone`: the fused head sweep has
// already produced error/stiffness for the current velocities (DFSPHSolver.cu:341).
int DFSPHSolver::correctDivergenceError(std::shared_ptr<SPHParticles>& fluids, const std::shared_ptr<SPHParticles>& boundaries,
                                        const DArray<int>& cellStartFluid, const DArray<int>& cellStartBoundary, float rho0,
                                        int3 cellSize, float cellLength, float radius, float dt, float errorThreshold,
                                        int maxIterations)
{
    SweepCache& c = cache();
    c.setup(cellSize, cellLength, radius);
    c.packFluid(*fluids);
    c.packBoundary(*boundaries);
    c.ensureList(cellStartFluid, cellStartBoundary);
    const SweepCtx ctx = c.ctx(cellStartFluid, cellStartBoundary);
    const int num = (int)fluids->size();
    const bool adaptive = fixedDiv < 0;
    auto totalError = std::numeric_limits<float>::max();
    auto iter = 0;
    const OpRate rate{ctx, fluids->getVelPtr(), fluids->getDensityPtr(), alpha.addr(),
                      RateOut{error.addr(), bufferFloat.addr(), nullptr,
                              reinterpret_cast<unsigned long long*>(errorAccum.addr()), dt, rho0}};
    const OpCorrect<false> correct{ctx, bufferFloat.addr(), fluids->getVelPtr(), dt};
    if (!headDidFirstError) {
        ScopedKernel t("divergence_error");
        launch_rate<false, 0>(rate, num, false);
    }
    headDidFirstError = false;
    while (adaptive ? ((iter < 1 || totalError > errorThreshold * num * rho0) && iter < maxIterations) : (iter < fixedDiv)) {
        {
            ScopedKernel t("divergence_correct");
            launch_op(correct, num);
        }
        {
            ScopedKernel t("divergence_error");
            launch_rate<false, 0>(rate, num, adaptive);
        }
        ++iter;
        if (adaptive) totalError = readErrorTotal();
    }
    return iter;
}

// DFSPHSolver::project, DFSPHSolver.cu:160-210
int DFSPHSolver::project(std::shared_ptr<SPHParticles>& fluids, const std::shared_ptr<SPHParticles>& boundaries,
                         const DArray<int>& cellStartFluid, const DArray<int>& cellStartBoundary, float rho0, int3 cellSize,
                         float cellLength, float radius, float dt, float errorThreshold, int maxIterations)
{
    SweepCache& c = cache();
    c.setup(cellSize, cellLength, radius);
    c.packFluid(*fluids);
    c.packBoundary(*boundaries);
    c.ensureList(cellStartFluid, cellStartBoundary);
    const SweepCtx ctx = c.ctx(cellStartFluid, cellStartBoundary);
    const int num = (int)fluids->size();
    const bool adaptive = fixedDen < 0;
    auto totalError = std::numeric_limits<float>::max();
    auto iter = 0;
    // carry last step's warm stiffness through this step's sort (DFSPHSolver.cu:170-171)
    {
        ScopedKernel t("warm_permute");
        ew_gather_float(scratch.addr(), denWarmStiff.addr(), fluids->getSortPerm(), num);
        ew_copy(denWarmStiff.addr(), scratch.addr(), sizeof(float) * num);
    }
    const OpRate rate{ctx, fluids->getVelPtr(), fluids->getDensityPtr(), alpha.addr(),
                      RateOut{error.addr(), bufferFloat.addr(), denWarmStiff.addr(),
                              reinterpret_cast<unsigned long long*>(errorAccum.addr()), dt, rho0}};
    OpCorrect<true> correct{ctx, denWarmStiff.addr(), fluids->getVelPtr(), dt};
    {
        ScopedKernel t("density_correct");   // warm start
        launch_op(correct, num);
    }
    {
        ScopedKernel t("density_error");     // also resets the warm stiffness to this stiffness
        launch_rate<true, 1>(rate, num, false);
    }
    correct.kappa = bufferFloat.addr();
    while (adaptive ? ((iter < 2 || totalError > errorThreshold * num * rho0) && iter < maxIterations) : (iter < fixedDen)) {
        {
            ScopedKernel t("density_correct");
            launch_op(correct, num);
        }
        ++iter;
        const bool needTotal = adaptive && iter >= 2;
        {
            ScopedKernel t("density_error");   // accumulates the warm stiffness
            launch_rate<true, 2>(rate, num, needTotal);
        }
        if (needTotal) totalError = readErrorTotal();
    }
    return iter;
}

// DFSPHSolver::step, DFSPHSolver.cu:33-72
void DFSPHSolver::step(std::shared_ptr<SPHParticles>& fluids, const std::shared_ptr<SPHParticles>& boundaries,
                       const DArray<int>& cellStartFluid, const DArray<int>& cellStartBoundary, float3 spaceSize,
                       int3 cellSize, float cellLength, float radius, float dt, float rho0, float rhoB, float stiff,
                       float visc, float3 G, float surfaceTensionIntensity, float airPressure)
{
    (void)stiff;
    invalidatePositions();
    SweepCache& c = cache();
    const bool surface = surfaceTensionIntensity > EPSILON || airPressure > EPSILON;
    const int num = (int)fluids->size();
    if (!c.fused()) {
        computeDensityAlpha(fluids, boundaries, cellStartFluid, cellStartBoundary, cellSize, cellLength, radius);
        lastDiv = correctDivergenceError(fluids, boundaries, cellStartFluid, cellStartBoundary, rho0, cellSize, cellLength,
                                         radius, dt, divergenceErrorThreshold, maxIter);
        force(fluids, dt, G);
        BasicSPHSolver::diffuse(fluids, cellStartFluid, cellSize, cellLength, rho0, radius, visc, dt);
        if (surface)
            handleSurface(fluids, boundaries, cellStartFluid, cellStartBoundary, rho0, rhoB, cellSize, cellLength, radius,
                          dt, surfaceTensionIntensity, airPressure);
        lastDen = project(fluids, boundaries, cellStartFluid, cellStartBoundary, rho0, cellSize, cellLength, radius, dt,
                          densityErrorThreshold, maxIter);
        advect(fluids, dt, spaceSize);
        return;
    }
    // fused schedule: [density+alpha+first divergence error] -> divergence loop -> gravity ->
    // [viscosity+colour gradient] -> [surface (+ vel += deltaV)] -> density solve -> advect
    c.setup(cellSize, cellLength, radius);
    c.packFluid(*fluids);
    c.packBoundary(*boundaries);
    c.ensureList(cellStartFluid, cellStartBoundary);
    const SweepCtx ctx = c.ctx(cellStartFluid, cellStartBoundary);
    if (num > 0) {
        ScopedKernel t("density_alpha_diverr");
        OpDfsphHead op{ctx, fluids->getVelPtr(), fluids->getDensityPtr(), alpha.addr(),
                       RateOut{error.addr(), bufferFloat.addr(), nullptr, nullptr, dt, rho0}};
        launch_dfsph_head<true>(op, num);
    }
    headDidFirstError = true;
    lastDiv = correctDivergenceError(fluids, boundaries, cellStartFluid, cellStartBoundary, rho0, cellSize, cellLength,
                                     radius, dt, divergenceErrorThreshold, maxIter);
    force(fluids, dt, G);
    DArray<float3>& cg = colorGradientBuffer();
    if (surface) {
        {
            ScopedKernel t("visc_color");
            OpFluidProps<true, true, false> op{ctx, fluids->getVelPtr(), c.aux3.addr(), cg.addr(), nullptr, nullptr, nullptr,
                                               rho0, rhoB, visc, dt, 0.0f};
            launch_op(op, num);
        }
        ScopedKernel t("surface_tension");
        OpSurface op{ctx, cg.addr(), fluids->getVelPtr(), c.aux3.addr(), fluids->getVelPtr(), rho0, surfaceTensionIntensity,
                     airPressure, dt};
        launch_op(op, num);
    } else {
        BasicSPHSolver::diffuse(fluids, cellStartFluid, cellSize, cellLength, rho0, radius, visc, dt);
    }
    lastDen = project(fluids, boundaries, cellStartFluid, cellStartBoundary, rho0, cellSize, cellLength, radius, dt,
                      densityErrorThreshold, maxIter);
    advect(fluids, dt, spaceSize);
}
