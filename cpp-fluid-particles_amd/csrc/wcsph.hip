// wcsph.hip — BasicSPHSolver (weakly compressible SPH) as hand-written HIP kernels for gfx950.
//
// One lane per fluid particle; each lane walks its 27-cell neighbourhood in the reference order
// and accumulates into registers (see sph_device.hpp::sweep27).  Reference behaviour restated from
// src/BasicSPHSolver.cu:32-381 (kernel-by-kernel citations below); no code is shared with it.
#include "BasicSPHSolver.h"
#include "engine.hpp"
#include "sweep_ops.hpp"

using namespace sphx;

BasicSPHSolver::BasicSPHSolver(int num) : bufferFloat3((unsigned)num), _cache(new SweepCache(num)) {}
BasicSPHSolver::~BasicSPHSolver() noexcept {}

void BasicSPHSolver::invalidatePositions() { _cache->fluidValid = false; }

// BasicSPHSolver::force, BasicSPHSolver.cu:227-235: vel += dt * G
void BasicSPHSolver::force(std::shared_ptr<SPHParticles>& fluids, float dt, float3 G)
{
    const int n = (int)fluids->size();
    const float3 dv = make_float3(dt * G.x, dt * G.y, dt * G.z);
    ScopedKernel t("force");
    launch_add_const3(fluids->getVelPtr(), dv, n);
}

// BasicSPHSolver::advect, BasicSPHSolver.cu:98-101 (+ Particles::advect): pos += dt*vel, then the
// box clamp with velocity correction, fused into one pass.
void BasicSPHSolver::advect(std::shared_ptr<SPHParticles>& fluids, float dt, float3 spaceSize)
{
    const int n = (int)fluids->size();
    ScopedKernel t("advect_clamp");
    launch_advect_clamp(fluids->getPosPtr(), fluids->getVelPtr(), dt, spaceSize, n);
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
    {
        ScopedKernel t("viscosity");
        OpViscosity op{c.g, c.k, cellStartFluid.addr(), c.fluid4(), fluids->getVelPtr(), bufferFloat3.addr(), rho0, visc, dt};
        launch_op(op, n);
    }
    {
        ScopedKernel t("add_delta_v");
        launch_add3(fluids->getVelPtr(), bufferFloat3.addr(), n);
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
    const int n = (int)fluids->size();
    {
        ScopedKernel t("color_grad");
        OpColorGrad op{c.g, c.k, cellStartFluid.addr(), c.fluid4(), cellStartBoundary.addr(), c.boundary4(),
                       bufferFloat3.addr(), rho0, rhoB};
        launch_op(op, n);
    }
    {
        ScopedKernel t("surface_tension");
        OpSurface op{c.g, c.k, cellStartFluid.addr(), c.fluid4(), bufferFloat3.addr(), fluids->getVelPtr(), rho0,
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
    const int n = (int)fluids->size();
    {
        ScopedKernel t("density_pressure");
        OpDensityPressure op{c.g, c.k, cellStartFluid.addr(), c.fluid4(), cellStartBoundary.addr(), c.boundary4(),
                             fluids->getDensityPtr(), fluids->getPressurePtr(), c.pterm.addr(), rho0, stiff};
        launch_op(op, n);
    }
    {
        ScopedKernel t("pressure_force");
        OpPressureForce op{c.g, c.k, cellStartFluid.addr(), c.fluid4(), cellStartBoundary.addr(), c.boundary4(),
                           c.pterm.addr(), fluids->getVelPtr(), dt};
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
    force(fluids, dt, G);
    diffuse(fluids, cellStartFluid, cellSize, cellLength, rho0, radius, visc, dt);
    if (surfaceTensionIntensity > EPSILON || airPressure > EPSILON)
        handleSurface(fluids, boundaries, cellStartFluid, cellStartBoundary, rho0, rhoB, cellSize, cellLength, radius, dt,
                      surfaceTensionIntensity, airPressure);
    project(fluids, boundaries, cellStartFluid, cellStartBoundary, rho0, stiff, cellSize, cellLength, radius, dt);
    advect(fluids, dt, spaceSize);
}
