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
#include <unordered_map>
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
// (bounded before the int cast: a tiny spacing must not overflow; kMaxIntervals^2 points per face is far beyond any capacity)
constexpr float kMaxIntervals = 1.0e6f;
int intervals(float length, float spacing) { return std::max(1, (int)std::min(kMaxIntervals, std::ceil(length / spacing - 1e-4f))); }

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
    // a box that is flat along an axis (lo == hi) has ONE face there and no intervals along it: a second face or a second
    // lattice line would put coincident particles
    const bool flatX = !(hi[0] > lo[0]), flatY = !(hi[1] > lo[1]), flatZ = !(hi[2] > lo[2]);
    const int nx = flatX ? 0 : intervals(hi[0] - lo[0], spacing), ny = flatY ? 0 : intervals(hi[1] - lo[1], spacing),
              nz = flatZ ? 0 : intervals(hi[2] - lo[2], spacing);
    auto at = [&](int axis, int k, int n) { return n > 0 ? lo[axis] + (hi[axis] - lo[axis]) * ((float)k / (float)n) : lo[axis]; };
    if (((double)nx + 1.0) * ((double)ny + 1.0) + ((double)nx + 1.0) * ((double)nz + 1.0) + ((double)ny + 1.0) * ((double)nz + 1.0) > 5.0e8)
        return sphx_fail(SPHX_ERR_INVALID, "sphx_sample_box: spacing too small for this box (more than 1e9 points)");
    Emitter e{out_xyz, capacity};
    for (int a = 0; a <= nx; ++a)                       // the two z faces
        for (int b = 0; b <= ny; ++b) { e.put(at(0, a, nx), at(1, b, ny), lo[2]); if (!flatZ) e.put(at(0, a, nx), at(1, b, ny), hi[2]); }
    for (int a = 0; a <= nx; ++a)                       // the two y faces without their z edges
        for (int c = 1; c < nz; ++c) { e.put(at(0, a, nx), lo[1], at(2, c, nz)); if (!flatY) e.put(at(0, a, nx), hi[1], at(2, c, nz)); }
    for (int b = 1; b < ny; ++b)                        // the two x faces without y and z edges
        for (int c = 1; c < nz; ++c) { e.put(lo[0], at(1, b, ny), at(2, c, nz)); if (!flatX) e.put(hi[0], at(1, b, ny), at(2, c, nz)); }
    return finish(e, count, "sphx_sample_box");
}

// Sphere surface: latitude rings spaced <= spacing apart along the meridian, each ring with points <= spacing
// apart along its circumference (poles: one point each).
int sphx_sample_sphere(const float center[3], float radius, float spacing, float* out_xyz, int capacity, int* count)
{
    if (!center || !count || !(spacing > 0.0f) || !(radius > 0.0f)) return sphx_fail(SPHX_ERR_INVALID, "sphx_sample_sphere: bad argument");
    const float pi = 3.14159265358979323846f;
    const int rings = intervals(pi * radius, spacing);
    if ((double)rings * (double)rings > 5.0e8) return sphx_fail(SPHX_ERR_INVALID, "sphx_sample_sphere: spacing too small for this sphere (more than 1e9 points)");
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
    // Points already emitted, binned on a lattice of step q = spacing / 8 with the FULL three 64-bit cell coordinates as key
    // (no masking: far-away coordinates cannot alias).  A new point is a duplicate when an emitted point lies within q of it,
    // which may sit in any of the 27 lattice cells around its own: all of them are looked at.
    const float q = spacing * 0.125f;
    struct Cell { int64_t a, b, c; bool operator==(const Cell& o) const { return a == o.a && b == o.b && c == o.c; } };
    struct CellHash { size_t operator()(const Cell& k) const { uint64_t h = (uint64_t)k.a * 0x9E3779B97F4A7C15ull; h ^= (uint64_t)k.b + 0x7F4A7C15u + (h << 6) + (h >> 2); h ^= (uint64_t)k.c * 0xC2B2AE3D27D4EB4Full + (h << 6) + (h >> 2); return (size_t)h; } };
    std::unordered_multimap<Cell, int, CellHash> seen;      // lattice cell -> index of an emitted point
    std::vector<float> kept;                                // emitted points (kept even when out_xyz is null / too small)
    auto cell_of = [&](float x, float y, float z) { return Cell{(int64_t)std::floor(x / q), (int64_t)std::floor(y / q), (int64_t)std::floor(z / q)}; };
    auto is_new = [&](float x, float y, float z) {
        const Cell c0 = cell_of(x, y, z);
        for (int64_t da = -1; da <= 1; ++da) for (int64_t db = -1; db <= 1; ++db) for (int64_t dc = -1; dc <= 1; ++dc) {
            const auto range = seen.equal_range(Cell{c0.a + da, c0.b + db, c0.c + dc});
            for (auto it = range.first; it != range.second; ++it) {
                const float* p = &kept[3 * (size_t)it->second];
                const float dx = p[0] - x, dy = p[1] - y, dz = p[2] - z;
                if (dx * dx + dy * dy + dz * dz <= q * q) return false;
            }
        }
        seen.emplace(c0, (int)(kept.size() / 3));
        kept.push_back(x); kept.push_back(y); kept.push_back(z);
        return true;
    };
    for (int t = 0; t < n_triangles; ++t) {
        const float* A = tri_xyz + 9 * (size_t)t; const float* B = A + 3; const float* C = A + 6;
        auto len = [](const float* p, const float* r) { return std::sqrt((p[0] - r[0]) * (p[0] - r[0]) + (p[1] - r[1]) * (p[1] - r[1]) + (p[2] - r[2]) * (p[2] - r[2])); };
        const int n = intervals(std::max(len(A, B), std::max(len(B, C), len(C, A))), spacing);
        if (n > 30000) return sphx_fail(SPHX_ERR_INVALID, "sphx_sample_triangles: spacing too small for a triangle (more than 4.5e8 points)");
        for (int i = 0; i <= n; ++i)
            for (int j = 0; i + j <= n; ++j) {
                const float u = (float)i / (float)n, v = (float)j / (float)n, w = 1.0f - u - v;
                const float x = w * A[0] + u * B[0] + v * C[0], y = w * A[1] + u * B[1] + v * C[1], z = w * A[2] + u * B[2] + v * C[2];
                if (is_new(x, y, z)) e.put(x, y, z);
            }
    }
    return finish(e, count, "sphx_sample_triangles");
}

}  // extern "C"
