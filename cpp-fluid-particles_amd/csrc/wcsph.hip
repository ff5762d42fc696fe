// wcsph.hip — BasicSPHSolver (weakly compressible SPH) as hand-written HIP kernels for gfx950.
//
// One lane per fluid particle walks its compact neighbour row (built once per step, see
// sph_device.hpp) and accumulates into registers in the reference's order.  Reference behaviour
// restated from src/BasicSPHSolver.cu:32-381 (kernel-by-kernel citations in sweep_ops.hpp).
//
// step() has two equivalent schedules:
//   fused    gravity folded into the pack pass; ONE sweep for viscosity + colour gradient +
//            density/pressure; the surface sweep also applies vel += deltaV; pressure force;
//            advect+clamp.  3 sweeps instead of 5.
//   unfused  the reference's sequence of protected building blocks (force, diffuse, handleSurface,
//            project, advect), used when a subclass overrides any of them or when asked for.
#include <algorithm>
#include <typeinfo>

#include "BasicSPHSolver.h"
#include "engine.hpp"
#include "sweep_ops.hpp"
#include "sphx_c.h"

using namespace sphx;

BasicSPHSolver::BasicSPHSolver(int num) : bufferFloat3((unsigned)num), _cache(new SweepCache(num)) {}
BasicSPHSolver::~BasicSPHSolver() noexcept {}

void BasicSPHSolver::invalidatePositions() { _cache->invalidatePositions(); }
void BasicSPHSolver::setEngineFlags(int flags) { _cache->flags = flags; _cache->listValid = false; ++_cache->generation; }
unsigned int BasicSPHSolver::graphGeneration() const { return _cache->generation; }
void BasicSPHSolver::prepareForCapture() { _cache->orderAge = 1 << 20; _cache->orderValid = false; }
// the packs / row build of a discarded capture never ran: the flags they set are taken back
void BasicSPHSolver::captureFailed() { _cache->boundaryValid = false; _cache->fluidValid = false; _cache->listValid = false; _cache->orderValid = false; }
void BasicSPHSolver::tune(int stepsSinceLastCall) { _cache->tuneRowCapacity(stepsSinceLastCall); }
void BasicSPHSolver::setToleranceArithmetic(bool on) { _cache->tolerance = on; ++_cache->generation; }
void* BasicSPHSolver::engineVel4() const { return _cache->vel4.addr(); }
void* BasicSPHSolver::engineCg4() const { return _cache->cg4.addr(); }
void* BasicSPHSolver::enginePterm() const { return _cache->pterm.addr(); }
void* BasicSPHSolver::enginePos4() const { return _cache->posm.addr(); }
void* BasicSPHSolver::enginePosf() const { return _cache->posf.addr(); }
const int* BasicSPHSolver::engineRowCounts() const { return _cache->nbrCount.addr(); }
int BasicSPHSolver::engineRowCapacity() const { return _cache->cap; }
const int* BasicSPHSolver::engineStaleFlag() const { return (_cache->skinRows && _cache->skin > 0.0f) ? _cache->staleFlag.addr(2) : nullptr; }
void BasicSPHSolver::reserveBoundary(int count) { _cache->reserveBoundary(count); }
void BasicSPHSolver::invalidateBoundary() { _cache->boundaryValid = false; _cache->listValid = false; ++_cache->generation; }
void BasicSPHSolver::setSweepRange(int lo, int hi, bool keepErrorAccum, int lo2, int hi2)
{
    _cache->rangeLo = lo; _cache->rangeHi = hi; _cache->keepErrorAccum = keepErrorAccum;
    _cache->rangeLo2 = lo >= 0 ? lo2 : -1; _cache->rangeHi2 = lo >= 0 ? hi2 : -1;
}
void BasicSPHSolver::requestPersistentRows(bool on)
{
    if (_cache->persistWanted == on) return;
    _cache->persistWanted = on;
    if (!on && _cache->persistRows) { _cache->persistRows = false; _cache->skin = 0.0f; _cache->listValid = false; }
    ++_cache->generation;
}
BasicSPHSolver::PersistentView BasicSPHSolver::preparePersistent(int3 cellSize, float cellLength, float radius)
{
    SweepCache& c = *_cache;
    c.setup(cellSize, cellLength, radius);
    if (c.persistRows && !c.posBuild) {      // the grid pass measures against the build positions from its first launch on
        c.posBuild.reset(new DArray<float>(4u * (unsigned)c.capN));
        c.rowCell.reset(new DArray<int>((unsigned)c.capN));
        ++c.generation;
        c.requestRebuild();
    }
    return PersistentView{c.persistRows, c.persistFlags.addr(), c.persistRows ? (const void*)c.posBuild->addr() : nullptr, c.persistLimit2()};
}
const int* BasicSPHSolver::enginePersistFlags() const { return _cache->persistRows ? _cache->persistFlags.addr() : nullptr; }
void BasicSPHSolver::requestRowRebuild() { if (_cache->persistRows) _cache->requestRebuild(); }
void BasicSPHSolver::permuteState(const int* perm, int n)
{
    if (n <= 0) return;
    SweepCache& c = *_cache;
    ew_gather_float3(c.aux3.addr(), bufferFloat3.addr(), perm, n);          // (aux3 is per-step scratch)
    ew_copy(bufferFloat3.addr(), c.aux3.addr(), sizeof(float3) * (size_t)n);
}
void BasicSPHSolver::setCellOffsetX(int cellOffsetX)
{
    _cache->cellOffsetX = cellOffsetX; _cache->cellKey = -1.0f; _cache->isSlab = true;
    // (until r05 a slab's rows were fixed at 96 entries; now they start at 48 and grow like a whole-domain system's -- SPHSystem::phase
    // calls BasicSPHSolver::tune behind a slab's last stage: half the row store, 0.2-0.6 % per step at 10.3 M)
}

// BasicSPHSolver::force, BasicSPHSolver.cu:227-235: vel += dt * G
void BasicSPHSolver::force(std::shared_ptr<SPHParticles>& fluids, float dt, float3 G)
{
    const int n = (int)fluids->size();
    const float3 dv = make_float3(dt * G.x, dt * G.y, dt * G.z);
    ScopedKernel t("force");
    launch_add_const3(fluids->getVelPtr(), cache().vel4w(), dv, n);
}

// BasicSPHSolver::advect, BasicSPHSolver.cu:98-101 (+ Particles::advect): pos += dt*vel, then the
// box clamp with velocity correction, fused into one pass.
void BasicSPHSolver::advect(std::shared_ptr<SPHParticles>& fluids, float dt, float3 spaceSize)
{
    const int n = (int)fluids->size();
    ScopedKernel t("advect_clamp");
    launch_advect_clamp(fluids->getPosPtr(), fluids->getVelPtr(), dt, spaceSize, n, cache().advectSkipIf);
    invalidatePositions();
}

// BasicSPHSolver::diffuse, BasicSPHSolver.cu:211-225: viscosity sweep into deltaV, then vel += deltaV
void BasicSPHSolver::diffuse(std::shared_ptr<SPHParticles>& fluids, const DArray<int>& cellStartFluid, int3 cellSize,
                             float cellLength, float rho0, float radius, float visc, float dt)
{
    SweepCache& c = cache();
    c.setup(cellSize, cellLength, radius);
    c.packFluid(*fluids);
    const int n = (int)fluids->size();
    // fluid-only sweep: walk the cells directly unless a list for these positions already exists
    SweepCtx ctx = c.ctx(cellStartFluid, cellStartFluid);
    {
        ScopedKernel t("viscosity");
        OpFluidProps<true, false, false> op{ctx, fluids->getVelPtr(), bufferFloat3.addr(), nullptr, nullptr, nullptr, nullptr,
                                            rho0, 0.0f, visc, dt, 0.0f};
        launch_op(op, n);
    }
    {
        ScopedKernel t("add_delta_v");
        launch_add3(fluids->getVelPtr(), c.vel4w(), bufferFloat3.addr(), n);
    }
}

// BasicSPHSolver::handleSurface, BasicSPHSolver.cu:262-275 (colour gradient, then surface tension
// and air pressure; method of He et al. 2014)
void BasicSPHSolver::handleSurface(std::shared_ptr<SPHParticles>& fluids, const std::shared_ptr<SPHParticles>& boundaries,
                                   const DArray<int>& cellStartFluid, const DArray<int>& cellStartBoundary, float rho0,
                                   float rhoB, int3 cellSize, float cellLength, float radius, float dt,
                                   float surfaceTensionIntensity, float airPressure)
{
    SweepCache& c = cache();
    c.setup(cellSize, cellLength, radius);
    c.packFluid(*fluids);
    c.packBoundary(*boundaries);
    c.ensureList(cellStartFluid, cellStartBoundary);
    const SweepCtx ctx = c.ctx(cellStartFluid, cellStartBoundary);
    const int n = (int)fluids->size();
    {
        ScopedKernel t("color_grad");
        OpFluidProps<false, true, false> op{ctx, nullptr, nullptr, bufferFloat3.addr(), nullptr, nullptr, nullptr,
                                            rho0, rhoB, 0.0f, dt, 0.0f};
        launch_op(op, n);
    }
    {
        ScopedKernel t("surface_tension");
        OpSurface op{ctx, bufferFloat3.addr(), fluids->getVelPtr(), nullptr, fluids->getVelPtr(), rho0,
                     surfaceTensionIntensity, airPressure, dt};
        launch_op(op, n);
    }
}

// BasicSPHSolver::project, BasicSPHSolver.cu:167-181: density, Tait pressure, pressure force
void BasicSPHSolver::project(std::shared_ptr<SPHParticles>& fluids, const std::shared_ptr<SPHParticles>& boundaries,
                             const DArray<int>& cellStartFluid, const DArray<int>& cellStartBoundary, float rho0,
                             float stiff, int3 cellSize, float cellLength, float radius, float dt)
{
    SweepCache& c = cache();
    c.setup(cellSize, cellLength, radius);
    c.packFluid(*fluids);
    c.packBoundary(*boundaries);
    c.ensureList(cellStartFluid, cellStartBoundary);
    const SweepCtx ctx = c.ctx(cellStartFluid, cellStartBoundary);
    const int n = (int)fluids->size();
    {
        ScopedKernel t("density_pressure");
        OpFluidProps<false, false, true> op{ctx, nullptr, nullptr, nullptr, fluids->getDensityPtr(), fluids->getPressurePtr(),
                                            c.pterm.addr(), rho0, 0.0f, 0.0f, dt, stiff};
        launch_op(op, n);
    }
    {
        ScopedKernel t("pressure_force");
        OpPressureForce op{ctx, c.pterm.addr(), fluids->getVelPtr(), dt, true};
        launch_op(op, n);
    }
}

// BasicSPHSolver::step, BasicSPHSolver.cu:237-260
void BasicSPHSolver::step(std::shared_ptr<SPHParticles>& fluids, const std::shared_ptr<SPHParticles>& boundaries,
                          const DArray<int>& cellStartFluid, const DArray<int>& cellStartBoundary, float3 spaceSize,
                          int3 cellSize, float cellLength, float radius, float dt, float rho0, float rhoB, float stiff,
                          float visc, float3 G, float surfaceTensionIntensity, float airPressure)
{
    invalidatePositions();   // the caller has just re-sorted the particles
    SweepCache& c = cache();
    const bool surface = surfaceTensionIntensity > EPSILON || airPressure > EPSILON;
    if (!c.fused() || typeid(*this) != typeid(BasicSPHSolver)) {
        force(fluids, dt, G);
        diffuse(fluids, cellStartFluid, cellSize, cellLength, rho0, radius, visc, dt);
        if (surface)
            handleSurface(fluids, boundaries, cellStartFluid, cellStartBoundary, rho0, rhoB, cellSize, cellLength, radius,
                          dt, surfaceTensionIntensity, airPressure);
        project(fluids, boundaries, cellStartFluid, cellStartBoundary, rho0, stiff, cellSize, cellLength, radius, dt);
        advect(fluids, dt, spaceSize);
        return;
    }
    const int fusedTail = SPHX_PH_W_SURFACE_PRESSURE;       // surface sweep + pressure force in one row walk
    const bool fuseTail = surface;
    for (int ph : {(int)SPHX_PH_W_SEARCH, (int)SPHX_PH_W_PROPS, fuseTail ? fusedTail : (int)SPHX_PH_W_SURFACE,
                   fuseTail ? -1 : (int)SPHX_PH_W_PRESSURE, (int)SPHX_PH_ADVECT})
        if (ph >= 0) runWcsphPhase(ph, fluids, boundaries, cellStartFluid, cellStartBoundary, spaceSize, cellSize, cellLength, radius, dt,
                      rho0, rhoB, stiff, visc, G, surfaceTensionIntensity, airPressure);
}

// One stage of the fused schedule (also what SPHSystem::phase runs for distributed drivers).
void BasicSPHSolver::runWcsphPhase(int phase, std::shared_ptr<SPHParticles>& fluids,
                                   const std::shared_ptr<SPHParticles>& boundaries, const DArray<int>& cellStartFluid,
                                   const DArray<int>& cellStartBoundary, float3 spaceSize, int3 cellSize, float cellLength,
                                   float radius, float dt, float rho0, float rhoB, float stiff, float visc, float3 G,
                                   float surfaceTensionIntensity, float airPressure)
{
    SweepCache& c = cache();
    const bool surface = surfaceTensionIntensity > EPSILON || airPressure > EPSILON;
    const int n = (int)fluids->size();
    if (phase == SPHX_PH_W_SEARCH) {
        invalidatePositions();
        c.setup(cellSize, cellLength, radius);
        c.packFluidKick(*fluids, make_float3(dt * G.x, dt * G.y, dt * G.z));
        c.packBoundary(*boundaries);
        c.ensureList(cellStartFluid, cellStartBoundary);
        return;
    }
    if (phase == SPHX_PH_ADVECT) { advect(fluids, dt, spaceSize); return; }
    c.setup(cellSize, cellLength, radius);
    c.packFluid(*fluids);
    c.packBoundary(*boundaries);
    c.ensureList(cellStartFluid, cellStartBoundary);
    const SweepCtx ctx = c.ctx(cellStartFluid, cellStartBoundary);
    if (phase == SPHX_PH_W_PROPS) {
        if (surface) {
            ScopedKernel t("visc_color_density");
            OpFluidProps<true, true, true> op{ctx, fluids->getVelPtr(), c.aux3.addr(), bufferFloat3.addr(),
                                              fluids->getDensityPtr(), fluids->getPressurePtr(), c.pterm.addr(),
                                              rho0, rhoB, visc, dt, stiff};
            launch_op(op, n);
        } else {
            ScopedKernel t("visc_density");
            OpFluidProps<true, false, true> op{ctx, fluids->getVelPtr(), c.aux3.addr(), nullptr, fluids->getDensityPtr(),
                                               fluids->getPressurePtr(), c.pterm.addr(), rho0, rhoB, visc, dt, stiff};
            launch_op(op, n);
        }
        return;
    }
    if (phase == SPHX_PH_W_SURFACE) {
        if (surface) {
            ScopedKernel t("surface_tension");
            OpSurface op{ctx, bufferFloat3.addr(), fluids->getVelPtr(), c.aux3.addr(), fluids->getVelPtr(), rho0,
                         surfaceTensionIntensity, airPressure, dt};
            launch_op(op, n);
        } else {
            ScopedKernel t("add_delta_v");
            const int lo = c.rangeLo >= 0 ? std::min(c.rangeLo, n) : 0, hi = c.rangeLo >= 0 ? std::min(c.rangeHi, n) : n;
            launch_add3(fluids->getVelPtr() + lo, c.vel4w() + lo, c.aux3.addr() + lo, hi - lo);
            if (c.rangeLo >= 0 && c.rangeLo2 >= 0) {       // the second range of a two-range stage
                const int lo2 = std::min(std::max(c.rangeLo2, hi), n), hi2 = std::min(std::max(c.rangeHi2, lo2), n);
                launch_add3(fluids->getVelPtr() + lo2, c.vel4w() + lo2, c.aux3.addr() + lo2, hi2 - lo2);
            }
        }
        return;
    }
    if (phase == SPHX_PH_W_SURFACE_PRESSURE) {
        if (!surface) throw "BasicSPHSolver::runWcsphPhase: the fused surface stage needs surface effects enabled";
        ScopedKernel t("surface_pressure_force");
        // posf.w holds the pressure term since W_PROPS (slab drivers refresh it with the pterm halo)
        launch_op(OpSurfaceThen<2>{ctx, bufferFloat3.addr(), fluids->getVelPtr(), c.aux3.addr(), fluids->getVelPtr(), c.pterm.addr(), rho0,
                                   surfaceTensionIntensity, airPressure, dt, true}, n);
        return;
    }
    if (phase == SPHX_PH_W_PRESSURE) {
        ScopedKernel t("pressure_force");
        OpPressureForce op{ctx, c.pterm.addr(), fluids->getVelPtr(), dt, true};
        launch_op(op, n);
        return;
    }
    throw "BasicSPHSolver::runWcsphPhase: unknown stage";
}
