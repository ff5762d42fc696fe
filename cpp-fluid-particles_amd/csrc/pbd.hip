// pbd.hip — PBDSolver (position-based fluids, Macklin & Mueller 2013) as HIP kernels for gfx950.
//
// Reference behaviour restated from src/PBDSolver.cu:34-258: fixed number of
// Jacobi iterations on a fixed cell table while positions move, velocity from displacement, XSPH,
// surface effects, gravity, then predict.  XSPH writes to a separate buffer (Jacobi) instead of
// the reference's racy in-place update.
#include <algorithm>
#include <cstdlib>

#include "PBDSolver.h"
#include "sphx_c.h"
#include "engine.hpp"
#include "sweep_ops.hpp"

using namespace sphx;

PBDSolver::PBDSolver(int num, int defaultMaxIter, float defaultXSPH_c, float defaultRelaxation)
    : BasicSPHSolver(num), maxIter(defaultMaxIter), xSPH_c(defaultXSPH_c), relaxation(defaultRelaxation),
      fluidPosLast((unsigned)num), bufferFloat3((unsigned)num), bufferFloat((unsigned)num)
{
}

PBDSolver::PBDSolver(const std::shared_ptr<SPHParticles>& particles, int defaultMaxIter, float defaultXSPH_c,
                     float defaultRelaxation)
    : BasicSPHSolver((int)particles->size()), maxIter(defaultMaxIter), xSPH_c(defaultXSPH_c),
      relaxation(defaultRelaxation), fluidPosLast(particles->size()), bufferFloat3(particles->size()),
      bufferFloat(particles->size())
{
    initializePosLast(particles->getPos());
}

PBDSolver::~PBDSolver() noexcept {}

// PBDSolver.h:56-60
void PBDSolver::initializePosLast(const DArray<float3>& posFluid)
{
    ew_copy(fluidPosLast.addr(), posFluid.addr(), sizeof(float3) * fluidPosLast.length());
    posLastInitialized = true;
}

// One row build per step instead of one per Jacobi iteration: rows with a skin (engine.hpp).  Whole-domain systems
// only: a slab's ghost particles move by halo messages, which the displacement watch does not see.
void PBDSolver::configureSkin(float radius)
{
    SweepCache& c = cache();
    const float factor = tuning().pbd_skin >= 0.0f ? tuning().pbd_skin : 0.05f;
    c.skinRows = !c.isSlab && factor > 0.0f && skinWanted;
    c.skin = c.skinRows ? factor * radius : 0.0f;
}

void PBDSolver::tune(int stepsSinceLastCall)
{
    BasicSPHSolver::tune(stepsSinceLastCall);
    SweepCache& c = cache();
    if (c.isSlab || tuning().pbd_skin_fixed) return;
    tuneSteps += stepsSinceLastCall;
    if (tuneSteps < 32) return;
    if (skinWanted && c.skinRows) {
        int rebuilds = 0;
        HIP_CALL(hipMemcpyAsync(&rebuilds, c.staleFlag.addr(2), sizeof(int), hipMemcpyDeviceToHost, sphx::stream()));
        HIP_CALL(hipStreamSynchronize(sphx::stream()));
        const int fired = rebuilds - lastRebuilds;
        lastRebuilds = rebuilds;
        if (fired > tuneSteps * maxIter / 2) { skinWanted = false; skinOffSteps = 0; c.listValid = false; ++c.generation; }
    } else if (!skinWanted) {
        skinOffSteps += tuneSteps;
        if (skinOffSteps >= 256) { skinWanted = true; c.listValid = false; ++c.generation; }
    }
    tuneSteps = 0;
}

// PBDSolver::updateNeighborhood, PBDSolver.cu:81-87: carry last positions through this step's sort
void PBDSolver::updateNeighborhood(const std::shared_ptr<SPHParticles>& particles)
{
    const int num = (int)particles->size();
    ScopedKernel t("poslast_permute");
    ew_gather_float3(bufferFloat3.addr(), fluidPosLast.addr(), particles->getSortPerm(), num);
    ew_copy(fluidPosLast.addr(), bufferFloat3.addr(), sizeof(float3) * num);
}

// PBDSolver::diffuse, PBDSolver.cu:117-125 (XSPH viscosity)
void PBDSolver::diffuse(std::shared_ptr<SPHParticles>& fluids, const DArray<int>& cellStartFluid, int3 cellSize,
                        float cellLength, float rho0, float radius, float visc)
{
    SweepCache& c = cache();
    c.setup(cellSize, cellLength, radius);
    c.packFluid(*fluids);
    const int num = (int)fluids->size();
    ScopedKernel t("xsph");
    OpXsph<false> op{c.ctx(cellStartFluid, cellStartFluid), fluids->getVelPtr(), bufferFloat3.addr(), nullptr, visc, rho0, 0.0f};
    launch_op(op, num);
    launch_copy3_mirror(fluids->getVelPtr(), c.vel4w(), bufferFloat3.addr(), num);
}

// PBDSolver::project, PBDSolver.cu:225-258.  Positions move every iteration while the cell table
// stays fixed, so the neighbour rows are rebuilt per iteration (shared by the
// lambda and delta-p sweeps of that iteration).
int PBDSolver::project(std::shared_ptr<SPHParticles>& fluids, const std::shared_ptr<SPHParticles>& boundaries,
                       const DArray<int>& cellStartFluid, const DArray<int>& cellStartBoundary, float rho0, int3 cellSize,
                       float3 spaceSize, float cellLength, float radius, int maxIterations)
{
    SweepCache& c = cache();
    c.setup(cellSize, cellLength, radius);
    c.packFluid(*fluids);
    c.packBoundary(*boundaries);
    const int num = (int)fluids->size();
    auto iter = 0;
    // r05: the delta-p sweep of an iteration also applies it (OpDeltaPos::apply: the moved positions go into the other half of the
    // double-buffered posm / posf, which then becomes the live one) -- one launch and one pass per iteration less.  An EVEN number of
    // iterations is run that way, so that a step ends on the buffers it began with (a captured step graph holds their addresses); a last
    // odd iteration applies in place with the separate pass.  Whole-domain steps only (slabs apply range by range behind their halos).
    const int fusedIters = (c.isSlab || c.rangeLo >= 0 || num <= 0) ? 0 : (maxIterations / 2) * 2;
    if (fusedIters > 0) c.ensureAltPositions();
    while (iter < maxIterations) {
        c.ensureList(cellStartFluid, cellStartBoundary);
        if (iter > 0) c.rebuildIfStale(cellStartFluid, cellStartBoundary);
        const SweepCtx ctx = c.ctx(cellStartFluid, cellStartBoundary);
        {
            ScopedKernel t("pbd_lambda");
            OpLambda lam{ctx, fluids->getDensityPtr(), bufferFloat.addr(), rho0, (rho0 != 0.0f) ? 1.0f : 0.0f, relaxation};
            launch_op(lam, num);
        }
        if (iter < fusedIters) {
            ScopedKernel t("pbd_delta_pos");
            const bool skin = c.skinRows && c.skin > 0.0f && c.listValid && c.posBuild;
            OpDeltaPos dp{ctx, bufferFloat.addr(), bufferFloat3.addr(), rho0, true};
            dp.apply.pos = fluids->getPosPtr();
            dp.apply.posmNext = reinterpret_cast<float4*>(c.posmAlt->addr());
            dp.apply.posfNext = reinterpret_cast<float4*>(c.posfAlt->addr());
            dp.apply.space = spaceSize;
            if (skin) dp.apply.watch = c.skinWatch();
            launch_op(dp, num);
            c.swapAltPositions();
            if (!skin) c.listValid = false;
        } else {
            {
                ScopedKernel t("pbd_delta_pos");
                OpDeltaPos dp{ctx, bufferFloat.addr(), bufferFloat3.addr(), rho0, true};
                launch_op(dp, num);
            }
            ScopedKernel t("pbd_apply_clamp");   // keeps the packed position view in step with pos
            applyDelta(fluids, spaceSize, num);
        }
        ++iter;
    }
    return iter;
}

// pos += delta-p with the box clamp (PBDSolver.cu:212-223, :247-253).  Ordinary rows are invalid afterwards; skin
// rows stay, the update itself watches how far particles have moved since the build.
void PBDSolver::applyDelta(std::shared_ptr<SPHParticles>& fluids, float3 spaceSize, int num)
{
    SweepCache& c = cache();
    const bool skin = c.skinRows && c.skin > 0.0f && c.listValid && c.posBuild;
    launch_apply_delta_clamp(fluids->getPosPtr(), c.fluid4w(), c.posfw(), bufferFloat3.addr(), spaceSize, num,
                             skin ? c.skinWatch() : SkinWatch{}, c.g);
    if (!skin) c.listValid = false;
}

// PBDSolver::predict, PBDSolver.cu:75-79
void PBDSolver::predict(std::shared_ptr<SPHParticles>& fluids, float dt, float3 spaceSize)
{
    ew_copy(fluidPosLast.addr(), fluids->getPosPtr(), sizeof(float3) * fluids->size());
    advect(fluids, dt, spaceSize);
}

// PBDSolver::step, PBDSolver.cu:34-73
void PBDSolver::step(std::shared_ptr<SPHParticles>& fluids, const std::shared_ptr<SPHParticles>& boundaries,
                     const DArray<int>& cellStartFluid, const DArray<int>& cellStartBoundary, float3 spaceSize,
                     int3 cellSize, float cellLength, float radius, float dt, float rho0, float rhoB, float stiff,
                     float visc, float3 G, float surfaceTensionIntensity, float airPressure)
{
    (void)stiff; (void)visc;
    if (!posLastInitialized) {
        initializePosLast(fluids->getPos());
        throw "PBD: The last position of fluids is initialized.";
    }
    invalidatePositions();
    SweepCache& c = cache();
    c.allowTiles = false;   // PBD sweeps run on positions that moved after binning
    configureSkin(radius);
    const bool surface = surfaceTensionIntensity > EPSILON || airPressure > EPSILON;
    const int num = (int)fluids->size();
    updateNeighborhood(fluids);
    project(fluids, boundaries, cellStartFluid, cellStartBoundary, rho0, cellSize, spaceSize, cellLength, radius, maxIter);
    {
        ScopedKernel t("pbd_velocity");
        launch_velocity_from_displacement(fluids->getVelPtr(), cache().vel4w(), fluids->getPosPtr(), fluidPosLast.addr(), dt, num);
    }
    if (!c.fused()) {
        diffuse(fluids, cellStartFluid, cellSize, cellLength, rho0, radius, xSPH_c);
        if (surface)
            handleSurface(fluids, boundaries, cellStartFluid, cellStartBoundary, rho0, rhoB, cellSize, cellLength, radius,
                          dt, surfaceTensionIntensity, airPressure);
        force(fluids, dt, G);
        predict(fluids, dt, spaceSize);
        return;
    }
    // fused tail: [XSPH + colour gradient] -> [surface, reading the XSPH result] -> one pass for
    // gravity + remember positions + advect + clamp
    c.setup(cellSize, cellLength, radius);
    c.packFluid(*fluids);
    c.packBoundary(*boundaries);
    c.ensureList(cellStartFluid, cellStartBoundary);
    c.rebuildIfStale(cellStartFluid, cellStartBoundary);      // the last position update may have outrun the skin
    const SweepCtx ctx = c.ctx(cellStartFluid, cellStartBoundary);
    DArray<float3>& cg = colorGradientBuffer();
    if (surface) {
        {
            ScopedKernel t("xsph_color");
            OpXsph<true> op{ctx, fluids->getVelPtr(), bufferFloat3.addr(), cg.addr(), xSPH_c, rho0, rhoB};
            launch_op(op, num);
        }
        ScopedKernel t("surface_tension");
        OpSurface op{ctx, cg.addr(), bufferFloat3.addr(), nullptr, fluids->getVelPtr(), rho0, surfaceTensionIntensity,
                     airPressure, dt};
        launch_op(op, num);
    } else {
        diffuse(fluids, cellStartFluid, cellSize, cellLength, rho0, radius, xSPH_c);
    }
    {
        ScopedKernel t("kick_remember_advect");
        launch_kick_remember_advect(fluids->getPosPtr(), fluids->getVelPtr(), fluidPosLast.addr(),
                                    make_float3(dt * G.x, dt * G.y, dt * G.z), dt, spaceSize, num);
        invalidatePositions();
    }
}

// One stage of the schedule above, for distributed drivers (multi_gpu.py): the cuts sit where a
// sweep reads what an earlier one wrote for the neighbours (lambda, moved positions, velocities,
// colour gradient).  Always the fused tail; same kernels, same order as step().
void PBDSolver::runPhase(int phase, std::shared_ptr<SPHParticles>& fluids, const std::shared_ptr<SPHParticles>& boundaries,
                         const DArray<int>& cellStartFluid, const DArray<int>& cellStartBoundary, float3 spaceSize,
                         int3 cellSize, float cellLength, float radius, float dt, float rho0, float rhoB, float3 G,
                         float surfaceTensionIntensity, float airPressure)
{
    SweepCache& c = cache();
    c.allowTiles = false;
    const bool surface = surfaceTensionIntensity > EPSILON || airPressure > EPSILON;
    const int num = (int)fluids->size();
    if (phase == SPHX_PH_P_SEARCH) {
        invalidatePositions();
        updateNeighborhood(fluids);
        posLastInitialized = true;
        c.setup(cellSize, cellLength, radius);
        c.packFluid(*fluids);
        c.packBoundary(*boundaries);
        return;
    }
    // range-restricted launches (slab layer): element-wise passes cover [lo, hi) only, and the two stages that meet moved
    // positions first (P_LAMBDA after a position update, P_XSPH after the last one) build the rows of their own range
    const bool ranged = c.rangeLo >= 0;
    const int lo = ranged ? std::min(c.rangeLo, num) : 0, hi = ranged ? std::min(std::max(c.rangeHi, lo), num) : num;
    if (phase == SPHX_PH_P_VELOCITY) {
        ScopedKernel t("pbd_velocity");
        launch_velocity_from_displacement(fluids->getVelPtr() + lo, c.vel4w() + lo, fluids->getPosPtr() + lo, fluidPosLast.addr() + lo, dt, hi - lo);
        return;
    }
    if (phase == SPHX_PH_P_APPLY) {
        ScopedKernel t("pbd_apply_clamp");
        launch_apply_delta_clamp(fluids->getPosPtr() + lo, c.fluid4w() + lo, c.posfw() + lo, bufferFloat3.addr() + lo, spaceSize, hi - lo);
        c.listValid = false;
        return;
    }
    if (phase == SPHX_PH_P_TAIL) {
        ScopedKernel t("kick_remember_advect");
        launch_kick_remember_advect(fluids->getPosPtr(), fluids->getVelPtr(), fluidPosLast.addr(),
                                    make_float3(dt * G.x, dt * G.y, dt * G.z), dt, spaceSize, num);
        invalidatePositions();
        return;
    }
    c.setup(cellSize, cellLength, radius);
    c.packFluid(*fluids);
    c.packBoundary(*boundaries);
    if (ranged && (phase == SPHX_PH_P_LAMBDA || phase == SPHX_PH_P_XSPH)) c.buildListForRange(cellStartFluid, cellStartBoundary);
    else if (ranged) c.listValid = c.nbr != nullptr && !(c.flags & kFlagNoList);   // the rows this range got from its P_LAMBDA / P_XSPH
    else c.ensureList(cellStartFluid, cellStartBoundary);   // rebuilt whenever positions moved since the last build
    const SweepCtx ctx = c.ctx(cellStartFluid, cellStartBoundary);
    DArray<float3>& cg = colorGradientBuffer();
    switch (phase) {
    case SPHX_PH_P_DELTA_SWEEP: {
        ScopedKernel t("pbd_delta_pos");
        launch_op(OpDeltaPos{ctx, bufferFloat.addr(), bufferFloat3.addr(), rho0, true}, num);
        break;
    }
    case SPHX_PH_P_LAMBDA: {
        ScopedKernel t("pbd_lambda");
        launch_op(OpLambda{ctx, fluids->getDensityPtr(), bufferFloat.addr(), rho0, (rho0 != 0.0f) ? 1.0f : 0.0f, relaxation}, num);
        break;
    }
    case SPHX_PH_P_DELTA: {
        {
            ScopedKernel t("pbd_delta_pos");
            launch_op(OpDeltaPos{ctx, bufferFloat.addr(), bufferFloat3.addr(), rho0, true}, num);
        }
        ScopedKernel t("pbd_apply_clamp");
        applyDelta(fluids, spaceSize, num);
        break;
    }
    case SPHX_PH_P_XSPH: {
        if (surface) {
            ScopedKernel t("xsph_color");
            launch_op(OpXsph<true>{ctx, fluids->getVelPtr(), bufferFloat3.addr(), cg.addr(), xSPH_c, rho0, rhoB}, num);
        } else {
            ScopedKernel t("xsph");
            launch_op(OpXsph<false>{ctx, fluids->getVelPtr(), bufferFloat3.addr(), nullptr, xSPH_c, rho0, 0.0f}, num);
            // (the live velocities are read by the neighbours' XSPH sums: a ranged schedule copies them in once EVERY range
            // has been swept -- SPHX_PH_P_SURFACE without surface effects does it)
            if (!ranged) launch_copy3_mirror(fluids->getVelPtr(), c.vel4w(), bufferFloat3.addr(), num);
        }
        break;
    }
    case SPHX_PH_P_SURFACE: {
        if (!surface && ranged) {
            ScopedKernel t("xsph_commit");
            launch_copy3_mirror(fluids->getVelPtr() + lo, c.vel4w() + lo, bufferFloat3.addr() + lo, hi - lo);
        }
        if (surface) {
            ScopedKernel t("surface_tension");
            launch_op(OpSurface{ctx, cg.addr(), bufferFloat3.addr(), nullptr, fluids->getVelPtr(), rho0, surfaceTensionIntensity,
                                airPressure, dt}, num);
        }
        break;
    }
    default: throw "PBDSolver::runPhase: unknown stage";
    }
}
