// Particles.h — position/velocity state of the drop-in API (reference: src/Particles.h:20-50).
//
// Constructed from host positions (blocking upload, velocities start at zero); getPosPtr /
// getVelPtr are raw device pointers to AoS float3 arrays (12-byte stride) and stay valid and
// current after every SPHSystem::step().  advect(dt) is pos += dt * vel (Particles.cu:28-36).
#pragma once

#include <vector>
#include "DArray.h"

class Particles {
public:
    explicit Particles(const std::vector<float3>& p);
    struct Uninitialised { unsigned int count; };       // engine extension: `count` zero-filled slots, no upload
    explicit Particles(Uninitialised u);

    Particles(const Particles&) = delete;
    Particles& operator=(const Particles&) = delete;

    unsigned int size() const { return _active; }
    float3* getPosPtr() const { return pos.addr(); }
    float3* getVelPtr() const { return vel.addr(); }
    const DArray<float3>& getPos() const { return pos; }

    void advect(float dt);

    // --- engine extension: the arrays are allocated for capacity() particles; size() is the number
    // currently in use (distributed drivers move particles between processes every step)
    unsigned int capacity() const { return pos.length(); }
    void setActiveCount(unsigned int n) { _active = n <= pos.length() ? n : pos.length(); }

    virtual ~Particles() noexcept {}

protected:
    DArray<float3> pos;
    DArray<float3> vel;
    unsigned int _active;
};
