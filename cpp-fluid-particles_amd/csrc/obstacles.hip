// obstacles.hip — boundary-particle samplers for static obstacles: the step upstream of the hot path.
//
// The reference samples exactly one shape, the shell of the unit box (main.cpp:89-116), and turns whatever
// boundary particles it is given into masses with computeBoundaryMass_CUDA (SPHSystem.cu:79-112: rhoB over the
// sum of W over the boundary neighbourhood).  That second step is shape-agnostic and is what sphx_create runs
// on the whole boundary set; the samplers here produce particle layers for other shapes at a given spacing so
// that obstacles can be appended to the shell.  Host code, deterministic (no RNG), two-call protocol.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <unordered_set>
#include <vector>

#include "capi_internal.hpp"

namespace {

struct Emitter {
    float* out; int capacity; int count = 0;
    void put(float x, float y, float z)
    {
        if (out && count < capacity) { out[3 * (size_t)count] = x; out[3 * (size_t)count + 1] = y; out[3 * (size_t)count + 2] = z; }
        ++count;
    }
};

// number of intervals so that the step along an edge of length L is <= spacing (at least one)
int intervals(float length, float spacing) { return std::max(1, (int)std::ceil(length / spacing - 1e-4f)); }

int finish(const Emitter& e, int* count, const char* who)
{
    *count = e.count;
    if (e.out && e.count > e.capacity) return sphx_fail(SPHX_ERR_INVALID, std::string(who) + ": output capacity too small");
    return SPHX_OK;
}

}  // namespace

extern "C" {

// Surface of the axis-aligned box [lo, hi]: a regular lattice on every face with step <= spacing, edges and
// corners emitted once (the construction of the reference's shell, main.cpp:89-116, for any box).
int sphx_sample_box(const float lo[3], const float hi[3], float spacing, float* out_xyz, int capacity, int* count)
{
    if (!lo || !hi || !count || !(spacing > 0.0f) || hi[0] < lo[0] || hi[1] < lo[1] || hi[2] < lo[2])
        return sphx_fail(SPHX_ERR_INVALID, "sphx_sample_box: bad argument");
    const int nx = intervals(hi[0] - lo[0], spacing), ny = intervals(hi[1] - lo[1], spacing), nz = intervals(hi[2] - lo[2], spacing);
    auto at = [&](int axis, int k, int n) { return lo[axis] + (hi[axis] - lo[axis]) * ((float)k / (float)n); };
    Emitter e{out_xyz, capacity};
    for (int a = 0; a <= nx; ++a)                       // the two z faces
        for (int b = 0; b <= ny; ++b) { e.put(at(0, a, nx), at(1, b, ny), lo[2]); e.put(at(0, a, nx), at(1, b, ny), hi[2]); }
    for (int a = 0; a <= nx; ++a)                       // the two y faces without their z edges
        for (int c = 1; c < nz; ++c) { e.put(at(0, a, nx), lo[1], at(2, c, nz)); e.put(at(0, a, nx), hi[1], at(2, c, nz)); }
    for (int b = 1; b < ny; ++b)                        // the two x faces without y and z edges
        for (int c = 1; c < nz; ++c) { e.put(lo[0], at(1, b, ny), at(2, c, nz)); e.put(hi[0], at(1, b, ny), at(2, c, nz)); }
    return finish(e, count, "sphx_sample_box");
}

// Sphere surface: latitude rings spaced <= spacing apart along the meridian, each ring with points <= spacing
// apart along its circumference (poles: one point each).
int sphx_sample_sphere(const float center[3], float radius, float spacing, float* out_xyz, int capacity, int* count)
{
    if (!center || !count || !(spacing > 0.0f) || !(radius > 0.0f)) return sphx_fail(SPHX_ERR_INVALID, "sphx_sample_sphere: bad argument");
    const float pi = 3.14159265358979323846f;
    const int rings = intervals(pi * radius, spacing);
    Emitter e{out_xyz, capacity};
    for (int k = 0; k <= rings; ++k) {
        const float theta = pi * (float)k / (float)rings;
        const float y = center[1] + radius * std::cos(theta), rr = radius * std::sin(theta);
        if (k == 0 || k == rings) { e.put(center[0], y, center[2]); continue; }
        const int m = std::max(3, intervals(2.0f * pi * rr, spacing));
        for (int t = 0; t < m; ++t) {
            const float phi = 2.0f * pi * (float)t / (float)m;
            e.put(center[0] + rr * std::cos(phi), y, center[2] + rr * std::sin(phi));
        }
    }
    return finish(e, count, "sphx_sample_sphere");
}

// Triangle soup (9 floats per triangle): barycentric lattice per triangle with step <= spacing along every edge;
// points that coincide with an already emitted one to within spacing/8 (shared edges and vertices) are dropped.
int sphx_sample_triangles(const float* tri_xyz, int n_triangles, float spacing, float* out_xyz, int capacity, int* count)
{
    if (!count || n_triangles < 0 || (n_triangles && !tri_xyz) || !(spacing > 0.0f)) return sphx_fail(SPHX_ERR_INVALID, "sphx_sample_triangles: bad argument");
    Emitter e{out_xyz, capacity};
    const float q = spacing * 0.125f;
    std::unordered_set<uint64_t> seen;
    auto key = [&](float x, float y, float z) {
        const int64_t a = (int64_t)std::llround(x / q), b = (int64_t)std::llround(y / q), c = (int64_t)std::llround(z / q);
        return (uint64_t)((a & 0x1fffff) | ((b & 0x1fffff) << 21) | ((c & 0x1fffff) << 42));
    };
    for (int t = 0; t < n_triangles; ++t) {
        const float* A = tri_xyz + 9 * (size_t)t; const float* B = A + 3; const float* C = A + 6;
        auto len = [](const float* p, const float* r) { return std::sqrt((p[0] - r[0]) * (p[0] - r[0]) + (p[1] - r[1]) * (p[1] - r[1]) + (p[2] - r[2]) * (p[2] - r[2])); };
        const int n = intervals(std::max(len(A, B), std::max(len(B, C), len(C, A))), spacing);
        for (int i = 0; i <= n; ++i)
            for (int j = 0; i + j <= n; ++j) {
                const float u = (float)i / (float)n, v = (float)j / (float)n, w = 1.0f - u - v;
                const float x = w * A[0] + u * B[0] + v * C[0], y = w * A[1] + u * B[1] + v * C[1], z = w * A[2] + u * B[2] + v * C[2];
                if (seen.insert(key(x, y, z)).second) e.put(x, y, z);
            }
    }
    return finish(e, count, "sphx_sample_triangles");
}

}  // extern "C"
