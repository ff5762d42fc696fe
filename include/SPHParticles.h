// SPHParticles.h — per-particle SPH state of the drop-in API (reference: src/SPHParticles.h:20-60).
//
// Adds pressure, density, mass and the particle->cell lookup key to Particles, with the
// reference accessors.  getParticle2Cell() holds the cell ids in PRE-sort order after a neighbour
// search.  Engine extensions (not in the reference): the permutation the last
// neighbour search applied (getSortPerm: perm[new] = old) so solvers can carry their own
// persistent arrays along without re-sorting, and an original-index channel (getIdPtr).
#pragma once

#include "Particles.h"

class SPHParticles final : public Particles {
public:
    explicit SPHParticles(const std::vector<float3>& p);
    explicit SPHParticles(Uninitialised u);             // engine extension (SPHSystem's persistent-order working set)

    SPHParticles(const SPHParticles&) = delete;
    SPHParticles& operator=(const SPHParticles&) = delete;

    float* getPressurePtr() const { return pressure.addr(); }
    const DArray<float>& getPressure() const { return pressure; }
    float* getDensityPtr() const { return density.addr(); }
    const DArray<float>& getDensity() const { return density; }
    int* getParticle2Cell() const { return particle2Cell.addr(); }
    float* getMassPtr() const { return mass.addr(); }

    // --- engine extensions -----------------------------------------------------------------
    int* getSortPerm() const { return sortPerm.addr(); }
    int* getIdPtr() const { return ids.addr(); }

    virtual ~SPHParticles() noexcept {}

protected:
    DArray<float> pressure;
    DArray<float> density;
    DArray<float> mass;
    DArray<int> particle2Cell;
    DArray<int> sortPerm;
    DArray<int> ids;
};
