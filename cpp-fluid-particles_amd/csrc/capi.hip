// capi.hip — the extern "C" boundary declared in include/sphx_c.h, a thin layer over the C++
// drop-in classes (SPHParticles, the three solvers, SPHSystem).
#include <algorithm>
#include <cmath>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "BasicSPHSolver.h"
#include "DFSPHSolver.h"
#include "PBDSolver.h"
#include "SPHSystem.h"
#include "engine.hpp"
#include "sphx_c.h"

namespace sphx {
const std::string& last_error_text();
void set_error_text(const std::string& s);
}  // namespace sphx
using namespace sphx;

#include "capi_internal.hpp"

int sphx_fail(int code, const std::string& msg)
{
    set_error_text(msg);
    return code;
}
static int fail(int code, const std::string& msg) { return sphx_fail(code, msg); }

// No C++ exception may cross the C boundary: the classes signal state errors with `throw "text"`
// like the reference (PBDSolver.cu:45-49), the runtime may throw std::bad_alloc.
template <class F>
static int guarded(const char* where, F&& body)
{
    try {
        return body();
    } catch (const char* msg) {
        return fail(SPHX_ERR_STATE, msg);
    } catch (const sphx::DeviceAllocError& e) {
        return fail(SPHX_ERR_HIP, std::string(where) + ": " + e.what());
    } catch (const std::bad_alloc&) {
        return fail(SPHX_ERR_HIP, std::string(where) + ": out of host memory");
    } catch (const std::exception& e) {
        return fail(SPHX_ERR_STATE, std::string(where) + ": " + e.what());
    } catch (...) {
        return fail(SPHX_ERR_STATE, std::string(where) + ": unknown exception");
    }
}


extern "C" {

const char* sphx_last_error(void) { return last_error_text().c_str(); }

int sphx_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return n;
}

int sphx_set_device(int ordinal)
{
    if (hipSetDevice(ordinal) != hipSuccess) return fail(SPHX_ERR_HIP, "hipSetDevice failed");
    return SPHX_OK;
}

int sphx_sizeof_params(void) { return (int)sizeof(sphx_params); }

int sphx_device_pci_id(int ordinal, char* out, int capacity)
{
    if (!out || capacity < 16) return fail(SPHX_ERR_INVALID, "sphx_device_pci_id: bad argument");
    if (hipDeviceGetPCIBusId(out, capacity, ordinal) != hipSuccess) { (void)hipGetLastError(); return fail(SPHX_ERR_HIP, "sphx_device_pci_id: hipDeviceGetPCIBusId failed"); }
    return SPHX_OK;
}

// ------------------------------------------------------------------------------------ tuning block (sphx_tuning)
int sphx_tuning_defaults(sphx_tuning* out)
{
    if (!out) return fail(SPHX_ERR_INVALID, "sphx_tuning_defaults: null block");
    *out = sphx::default_tuning();
    return SPHX_OK;
}

int sphx_set_tuning(const sphx_tuning* t)
{
    if (!t) { sphx::install_tuning(sphx::default_tuning()); return SPHX_OK; }
    if (t->struct_size != (int)sizeof(sphx_tuning)) return fail(SPHX_ERR_INVALID, "sphx_set_tuning: struct_size does not match this library's sphx_tuning");
    if ((t->row_capacity != 0 && (t->row_capacity < 8 || t->row_capacity > 1024)) || t->slab_comm_priority < 0 || t->slab_comm_priority > 2)
        return fail(SPHX_ERR_INVALID, "sphx_set_tuning: a field is out of range (row_capacity 0 or 8..1024, slab_comm_priority 0..2)");
    sphx::install_tuning(*t);
    return SPHX_OK;
}

int sphx_get_tuning(sphx_tuning* out)
{
    if (!out) return fail(SPHX_ERR_INVALID, "sphx_get_tuning: null block");
    *out = sphx::tuning();
    return SPHX_OK;
}

// the instantiation the most recent density / divergence error sweep was launched as (bench.py names it in its roofline block)
int sphx_last_rate_kernel(char* out, int capacity)
{
    if (!out || capacity < 48) return fail(SPHX_ERR_INVALID, "sphx_last_rate_kernel: bad argument");
    static const char* const names[] = {"none", "k_rate_quad<DENSITY_MODE, WARM, 0> (strict quad walk)", "k_rate_quad<DENSITY_MODE, WARM, 1> (tolerance quad walk)",
                                        "k_rate_quad<DENSITY_MODE, WARM, 0> (strict quad walk serving a tolerance-mode step)",
                                        "k_rate_duo<DENSITY_MODE, WARM>", "k_rate<DENSITY_MODE, WARM, false> (lane per particle)",
                                        "k_rate<DENSITY_MODE, WARM, true> (LDS tiles)", "k_rate_brick<DENSITY_MODE, WARM>"};
    const int v = std::min(std::max(sphx::g_lastRateVariant, 0), 7);
    std::strncpy(out, names[v], (size_t)capacity - 1); out[capacity - 1] = 0;
    return v;
}

// ------------------------------------------------------------------------------------ scene
// Constants of main.cpp:54-67; block and shell samplers of main.cpp:73-117; scaling rule of
// BASELINE.md §4 (s = nx/24; nx = 24 reproduces the reference scene bit-for-bit).
int sphx_scene_params(int nx, sphx_params* P)
{
    if (!P || nx < 2 || (nx & 1)) return fail(SPHX_ERR_INVALID, "scene: nx must be even and >= 2");
    std::memset(P, 0, sizeof(*P));
    const float scale = (float)nx / 24.0f;
    const float spacing = 0.02f;
    for (int a = 0; a < 3; ++a) P->space[a] = scale;
    P->radius = 2.0f * spacing;
    P->cell_length = 1.01f * P->radius;
    for (int a = 0; a < 3; ++a) P->cells[a] = (int)std::ceil(P->space[a] / P->cell_length);
    P->dt = 0.002f;
    P->rho0 = 1.0f;
    P->rho_boundary = 1.4f * P->rho0;
    P->m0 = 76.596750762082e-6f;
    P->stiff = 10.0f;
    P->gravity[0] = 0.0f; P->gravity[1] = -9.8f; P->gravity[2] = 0.0f;
    P->visc = 5e-4f;
    P->surface_tension = 0.0001f;
    P->air_pressure = 0.0001f;
    P->solver = SPHX_WCSPH;
    P->dfsph_density_thr = 1e-3f; P->dfsph_divergence_thr = 1e-3f; P->dfsph_max_iter = 20;
    P->dfsph_fixed_div = -1; P->dfsph_fixed_den = -1;
    P->pbd_iters = 20; P->pbd_xsph_c = 0.05f; P->pbd_relaxation = 0.75f;
    return SPHX_OK;
}

int sphx_scene_counts(int nx, int* n_fluid, int* n_boundary)
{
    sphx_params P;
    const int rc = sphx_scene_params(nx, &P);
    if (rc) return rc;
    const long long sx = 2LL * P.cells[0], sy = 2LL * P.cells[1], sz = 2LL * P.cells[2];
    const long long nf = (long long)nx * (3LL * nx / 2) * nx;
    const long long nbnd = 2 * sx * sy + 2 * sx * (sz - 2) + 2 * (sy - 2) * (sz - 2);
    if (nf > 2000000000LL || nbnd > 2000000000LL) return fail(SPHX_ERR_INVALID, "scene too large for int32 indices");
    *n_fluid = (int)nf;
    *n_boundary = (int)nbnd;
    return SPHX_OK;
}

int sphx_scene_fill(int nx, float* fluid, float* boundary)
{
    sphx_params P;
    const int rc = sphx_scene_params(nx, &P);
    if (rc) return rc;
    const float spacing = 0.02f, scale = P.space[0];
    const float x0 = 0.27f * scale, y0 = 0.10f * scale, z0 = 0.27f * scale;
    float* w = fluid;
    for (int iy = 0; iy < 3 * nx / 2; ++iy)
        for (int ix = 0; ix < nx; ++ix)
            for (int iz = 0; iz < nx; ++iz) {
                *w++ = x0 + spacing * ix;
                *w++ = y0 + spacing * iy;
                *w++ = z0 + spacing * iz;
            }
    const int shell[3] = {2 * P.cells[0], 2 * P.cells[1], 2 * P.cells[2]};
    w = boundary;
    auto emit = [&](int a, int b, int c) {
        const int idx[3] = {a, b, c};
        for (int d = 0; d < 3; ++d) {
            const float t = (float)idx[d] / (float)(shell[d] - 1) * P.space[d];
            *w++ = 0.99f * t + 0.005f * P.space[d];
        }
    };
    for (int a = 0; a < shell[0]; ++a)              // the two z faces
        for (int b = 0; b < shell[1]; ++b) { emit(a, b, 0); emit(a, b, shell[2] - 1); }
    for (int a = 0; a < shell[0]; ++a)              // the two y faces, without the z edges
        for (int c = 1; c < shell[2] - 1; ++c) { emit(a, 0, c); emit(a, shell[1] - 1, c); }
    for (int b = 1; b < shell[1] - 1; ++b)          // the two x faces, without y and z edges
        for (int c = 1; c < shell[2] - 1; ++c) { emit(0, b, c); emit(shell[0] - 1, b, c); }
    return SPHX_OK;
}

// ------------------------------------------------------------------------------------ lifetime
int sphx_create(const sphx_params* P, const float* fluid, int n, const float* boundary, int nb, int run_ctor_step,
                sphx_system** out)
{
    // (the internal mode 2 = "restored: no constructor step, no initial sort" belongs to sphx_snapshot_load alone: a caller
    // that passes 2 as "true" gets the ordinary constructor step)
    return guarded("sphx_create", [&] { return sphx_create_impl(P, fluid, n, boundary, nb, run_ctor_step ? 1 : 0, out); });
}

}  // extern "C"

int sphx_create_impl(const sphx_params* P, const float* fluid, int n, const float* boundary, int nb, int run_ctor_step,
                     sphx_system** out)
{
    if (!P || !out || n < 0 || nb < 0 || (n && !fluid) || (nb && !boundary)) return fail(SPHX_ERR_INVALID, "sphx_create: bad argument");
    if (P->pow7_mode != 0 || P->xsph_mode != 0) return fail(SPHX_ERR_INVALID, "sphx_create: pow7_mode and xsph_mode must be 0");
    if (P->cells[0] <= 0 || P->cells[1] <= 0 || P->cells[2] <= 0) return fail(SPHX_ERR_INVALID, "sphx_create: bad grid");
    if (sphx_device_count() <= 0) return fail(SPHX_ERR_NO_DEVICE, "sphx_create: no HIP device (the engine has no CPU path)");
    *out = nullptr;
    std::unique_ptr<sphx_system> h(new sphx_system());
    h->params = *P;
    h->n = n; h->nb = nb; h->cells = P->cells[0] * P->cells[1] * P->cells[2];
    std::vector<float3> fp((size_t)n), bp((size_t)nb);
    for (int i = 0; i < n; ++i) fp[i] = make_float3(fluid[3 * i], fluid[3 * i + 1], fluid[3 * i + 2]);
    for (int i = 0; i < nb; ++i) bp[i] = make_float3(boundary[3 * i], boundary[3 * i + 1], boundary[3 * i + 2]);
    auto fluids = std::make_shared<SPHParticles>(fp);
    auto walls = std::make_shared<SPHParticles>(bp);
    std::shared_ptr<BaseSolver> solver;
    switch (P->solver) {
    case SPHX_DFSPH: {
        auto s = std::make_shared<DFSPHSolver>(n, P->dfsph_density_thr, P->dfsph_divergence_thr, P->dfsph_max_iter);
        if (P->dfsph_fixed_div >= 0 || P->dfsph_fixed_den >= 0) {
            if (P->dfsph_fixed_div < 0 || P->dfsph_fixed_den < 0)
                return fail(SPHX_ERR_INVALID, "sphx_create: fix both DFSPH iteration counts or neither");
            s->setFixedIterations(P->dfsph_fixed_div, P->dfsph_fixed_den);
        }
        h->dfsph = s.get(); h->wcsph = s.get(); solver = s;
        break;
    }
    case SPHX_PBD: {
        auto s = std::make_shared<PBDSolver>(n, P->pbd_iters, P->pbd_xsph_c, P->pbd_relaxation);
        h->pbd = s.get(); h->wcsph = s.get(); solver = s;
        break;
    }
    case SPHX_WCSPH: {
        auto s = std::make_shared<BasicSPHSolver>(n);
        h->wcsph = s.get(); solver = s;
        break;
    }
    default: return fail(SPHX_ERR_INVALID, "sphx_create: unknown solver");
    }
    if (P->reserved[0]) h->wcsph->setEngineFlags(P->reserved[0]);
    if (P->reserved[3] < 0 || P->reserved[3] > 2)
        return fail(SPHX_ERR_INVALID, "sphx_create: reserved[3] (arithmetic) must be 0 (strict), 1 (tolerance) or 2 (tolerance with persistent rows)");
    if (P->reserved[3] >= 1) h->wcsph->setToleranceArithmetic(true);
    if (P->reserved[3] == 2 && (P->reserved[1] != 0 || P->reserved[2] != 0))
        return fail(SPHX_ERR_INVALID, "sphx_create: persistent rows (reserved[3] = 2) are a whole-domain mode, not available to slab systems");
    const float3 space = make_float3(P->space[0], P->space[1], P->space[2]);
    const float3 G = make_float3(P->gravity[0], P->gravity[1], P->gravity[2]);
    const int3 cells = make_int3(P->cells[0], P->cells[1], P->cells[2]);
    if (P->reserved[1] != 0 || P->reserved[2] != 0) {
        if (run_ctor_step) return fail(SPHX_ERR_INVALID, "sphx_create: slab systems take run_ctor_step = 0");
        h->system.reset(new SPHSystem(SPHSystem::Slab{P->reserved[1]}, fluids, walls, solver, space, P->cell_length, P->radius,
                                      P->dt, P->m0, P->rho0, P->rho_boundary, P->stiff, P->visc, P->surface_tension,
                                      P->air_pressure, G, cells));
    } else if (run_ctor_step == 2) {
        h->system.reset(new SPHSystem(SPHSystem::Restored{}, fluids, walls, solver, space, P->cell_length, P->radius, P->dt,
                                      P->m0, P->rho0, P->rho_boundary, P->stiff, P->visc, P->surface_tension,
                                      P->air_pressure, G, cells));
    } else if (run_ctor_step)
        h->system.reset(new SPHSystem(fluids, walls, solver, space, P->cell_length, P->radius, P->dt, P->m0, P->rho0,
                                      P->rho_boundary, P->stiff, P->visc, P->surface_tension, P->air_pressure, G, cells));
    else
        h->system.reset(new SPHSystem(SPHSystem::NoInitialStep{}, fluids, walls, solver, space, P->cell_length, P->radius,
                                      P->dt, P->m0, P->rho0, P->rho_boundary, P->stiff, P->visc, P->surface_tension,
                                      P->air_pressure, G, cells));
    // persistent rows: WCSPH / DFSPH (PBD moves positions inside the step and keeps its own skin scheme: plain tolerance mode there)
    if (P->reserved[3] == 2 && !h->pbd) h->system->setPersistentRows(true);
    if (hipStreamSynchronize(sphx::stream()) != hipSuccess) return fail(SPHX_ERR_HIP, last_error_text());
    *out = h.release();
    return SPHX_OK;
}

extern "C" {

int sphx_destroy(sphx_system* h)
{
    if (!h) return SPHX_OK;
    (void)hipStreamSynchronize(sphx::stream());
    delete h;
    return SPHX_OK;
}

// ------------------------------------------------------------------------------------ stepping
int sphx_step(sphx_system* h, float* ms)
{
    if (!h) return fail(SPHX_ERR_INVALID, "sphx_step: null system");
    return guarded("sphx_step", [&] {
        const float t = h->system->step();
        if (ms) *ms = t;
        return (int)SPHX_OK;
    });
}

int sphx_step_n(sphx_system* h, int n, float* ms_total)
{
    if (!h || n < 0) return fail(SPHX_ERR_INVALID, "sphx_step_n: bad argument");
    return guarded("sphx_step_n", [&] {
        const float t = h->system->stepN(n);
        if (ms_total) *ms_total = t;
        return (int)SPHX_OK;
    });
}

int sphx_counts(const sphx_system* h, int* n, int* nb, int* cells)
{
    if (!h) return fail(SPHX_ERR_INVALID, "null system");
    if (n) *n = h->n;
    if (nb) *nb = h->nb;
    if (cells) *cells = h->cells;
    return SPHX_OK;
}

int sphx_get_params(const sphx_system* h, sphx_params* out)
{
    if (!h || !out) return fail(SPHX_ERR_INVALID, "sphx_get_params: bad argument");
    *out = h->params;
    return SPHX_OK;
}

int sphx_row_stats(const sphx_system* h, long long* total, int* longest, int* hist128)
{
    if (!h || !h->wcsph || !total || !longest) return fail(SPHX_ERR_INVALID, "sphx_row_stats: bad argument");
    const int n = (int)h->system->getFluids()->size();
    std::vector<int> cnt((size_t)std::max(n, 1));
    if (n > 0 && (hipMemcpyAsync(cnt.data(), h->wcsph->engineRowCounts(), sizeof(int) * (size_t)n, hipMemcpyDeviceToHost, sphx::stream()) != hipSuccess ||
                  hipStreamSynchronize(sphx::stream()) != hipSuccess))
        return fail(SPHX_ERR_HIP, "sphx_row_stats: copy failed");
    long long t = 0; int mx = 0;
    if (hist128) std::memset(hist128, 0, sizeof(int) * 128);
    for (int i = 0; i < n; ++i) {
        t += cnt[i]; mx = std::max(mx, cnt[i]);
        if (hist128) hist128[std::min(std::max(cnt[i], 0), 127)]++;
    }
    *total = t; *longest = mx;
    return SPHX_OK;
}

// What the quad-per-particle row walk pays for ragged rows, from the row lengths of the most recent build (host arithmetic on a copy
// of the counts): a wave holds 16 particles x 4 lanes and runs to the longest row among its 16 (walk_row_quad, sph_device.hpp).
//   out[0] waves (16 particles each)            out[1] chunk steps as walked: sum over waves of max_i ceil(len_i / 4)
//   out[2] chunk steps of perfectly even rows: ceil(sum_i ceil(len_i / 4) / 16) summed per tile-quarter = total chunks / 16
//   out[3] chunk steps if every walk stopped at `cut` entries   out[4] chunk steps a compact second launch would need for the
//   tails beyond `cut` (16 tails per wave)      out[5] particles longer than `cut`
int sphx_row_walk_stats(const sphx_system* h, int cut, long long out[6])
{
    if (!h || !h->wcsph || !out || cut < 4) return fail(SPHX_ERR_INVALID, "sphx_row_walk_stats: bad argument");
    const int n = (int)h->system->getFluids()->size();
    std::vector<int> cnt((size_t)std::max(n, 1));
    if (n > 0 && (hipMemcpyAsync(cnt.data(), h->wcsph->engineRowCounts(), sizeof(int) * (size_t)n, hipMemcpyDeviceToHost, sphx::stream()) != hipSuccess ||
                  hipStreamSynchronize(sphx::stream()) != hipSuccess))
        return fail(SPHX_ERR_HIP, "sphx_row_walk_stats: copy failed");
    long long waves = 0, walked = 0, chunks = 0, cutWalk = 0, over = 0, tailChunks = 0;
    const int cutChunks = (cut + 3) / 4;
    for (int w0 = 0; w0 < n; w0 += 16) {
        int mx = 0, mxCut = 0;
        for (int i = w0; i < std::min(w0 + 16, n); ++i) {
            const int c = (std::max(cnt[i], 0) + 3) / 4;
            mx = std::max(mx, c); mxCut = std::max(mxCut, std::min(c, cutChunks));
            chunks += c;
            if (c > cutChunks) { ++over; tailChunks += c - cutChunks; }
        }
        ++waves; walked += mx; cutWalk += mxCut;
    }
    out[0] = waves; out[1] = walked; out[2] = (chunks + 15) / 16; out[3] = cutWalk;
    // tails packed 16 to a wave, sorted by nothing: a lower bound is tailChunks / 16, an upper bound one wave per 16 tails at the longest tail
    out[4] = (tailChunks + 15) / 16; out[5] = over;
    return SPHX_OK;
}

int sphx_row_capacity(const sphx_system* h, int* capacity)
{
    if (!h || !h->wcsph || !capacity) return fail(SPHX_ERR_INVALID, "sphx_row_capacity: bad argument");
    *capacity = h->wcsph->engineRowCapacity();
    return SPHX_OK;
}

int sphx_rows_stale(const sphx_system* h, int* stale)
{
    if (!h || !h->wcsph || !stale) return fail(SPHX_ERR_INVALID, "sphx_rows_stale: bad argument");
    *stale = 0;
    const int* flag = h->wcsph->engineStaleFlag();
    if (!flag) return SPHX_OK;
    if (hipMemcpyAsync(stale, flag, sizeof(int), hipMemcpyDeviceToHost, sphx::stream()) != hipSuccess ||
        hipStreamSynchronize(sphx::stream()) != hipSuccess)
        return fail(SPHX_ERR_HIP, "sphx_rows_stale: copy failed");
    return SPHX_OK;
}

int sphx_rows_partial(const sphx_system* h, int* partial)
{
    if (!h || !h->wcsph || !partial) return fail(SPHX_ERR_INVALID, "sphx_rows_partial: bad argument");
    *partial = 0;
    const int* flag = h->wcsph->engineStaleFlag();
    if (!flag) return SPHX_OK;
    if (hipMemcpyAsync(partial, flag + 3, sizeof(int), hipMemcpyDeviceToHost, sphx::stream()) != hipSuccess ||
        hipStreamSynchronize(sphx::stream()) != hipSuccess)
        return fail(SPHX_ERR_HIP, "sphx_rows_partial: copy failed");
    return SPHX_OK;
}

int sphx_persistent_stats(const sphx_system* h, int* in_use, int* row_builds, int* steps)
{
    if (!h || !h->wcsph) return fail(SPHX_ERR_INVALID, "sphx_persistent_stats: bad argument");
    int words[4] = {0, 0, 0, 0};
    const int* flags = h->system->persistentRows() ? h->wcsph->enginePersistFlags() : nullptr;
    if (flags && (hipMemcpyAsync(words, flags, sizeof(words), hipMemcpyDeviceToHost, sphx::stream()) != hipSuccess ||
                  hipStreamSynchronize(sphx::stream()) != hipSuccess))
        return fail(SPHX_ERR_HIP, "sphx_persistent_stats: copy failed");
    if (in_use) *in_use = flags ? 1 : 0;
    if (row_builds) *row_builds = words[2];
    if (steps) *steps = words[3];
    return SPHX_OK;
}

int sphx_iters(const sphx_system* h, int* div, int* den)
{
    if (!h) return fail(SPHX_ERR_INVALID, "null system");
    return guarded("sphx_iters", [&] {       // (reading the counts of device-decided loops also reports a failed loop-tail launch)
        if (div) *div = h->dfsph ? h->dfsph->lastDivergenceIterations() : 0;
        if (den) *den = h->dfsph ? h->dfsph->lastDensityIterations() : 0;
        return (int)SPHX_OK;
    });
}

// ------------------------------------------------------------------------------------ fields
// API fields: current in the reference's order after every step, whatever the mode.  Everything else is a per-particle array of the
// solver, which lives in the WORKING order while persistent rows are in use.
static bool api_field(int field)
{
    switch (field) {
    case SPHX_F_POS: case SPHX_F_VEL: case SPHX_F_DENSITY: case SPHX_F_PRESSURE: case SPHX_F_MASS: case SPHX_F_CELL:
    case SPHX_F_CELLSTART_F: case SPHX_F_CELLSTART_B: case SPHX_F_ID: case SPHX_F_BPOS: case SPHX_F_BMASS: return true;
    default: return false;
    }
}
static int locate_noflush(const sphx_system* h, int field, void** ptr, size_t* bytes);

// pointer + size of a field for callers that read or write it IN PLACE: with persistent rows the solver's own arrays are first
// brought into the API order (the next step re-primes the working copy and rebuilds the rows); may throw
int sphx_locate(const sphx_system* h, int field, void** ptr, size_t* bytes)
{
    if (!api_field(field)) h->system->invalidatePersistentOrder();
    return locate_noflush(h, field, ptr, bytes);
}

static int locate_noflush(const sphx_system* h, int field, void** ptr, size_t* bytes)
{
    const auto f = h->system->getFluids();
    const auto b = h->system->getBoundaries();
    const size_t n = (size_t)h->n, nb = (size_t)h->nb;
    void* p = nullptr; size_t sz = 0; bool known = true;
    switch (field) {
    case SPHX_F_POS: p = f->getPosPtr(); sz = 12 * n; break;
    case SPHX_F_VEL: p = f->getVelPtr(); sz = 12 * n; break;
    case SPHX_F_DENSITY: p = f->getDensityPtr(); sz = 4 * n; break;
    case SPHX_F_PRESSURE: p = f->getPressurePtr(); sz = 4 * n; break;
    case SPHX_F_MASS: p = f->getMassPtr(); sz = 4 * n; break;
    case SPHX_F_CELL: p = f->getParticle2Cell(); sz = 4 * n; break;
    case SPHX_F_CELLSTART_F: p = h->system->getCellStartFluid().addr(); sz = 4 * ((size_t)h->cells + 1); break;
    case SPHX_F_CELLSTART_B: p = h->system->getCellStartBoundary().addr(); sz = 4 * ((size_t)h->cells + 1); break;
    case SPHX_F_ID: p = f->getIdPtr(); sz = 4 * n; break;
    case SPHX_F_BPOS: p = b->getPosPtr(); sz = 12 * nb; break;
    case SPHX_F_BMASS: p = b->getMassPtr(); sz = 4 * nb; break;
    case SPHX_F_ALPHA: if (h->dfsph) { p = h->dfsph->getAlpha().addr(); sz = 4 * n; } else known = false; break;
    case SPHX_F_KAPPA: if (h->dfsph) { p = h->dfsph->getStiffness().addr(); sz = 4 * n; } else known = false; break;
    case SPHX_F_ERROR: if (h->dfsph) { p = h->dfsph->getError().addr(); sz = 4 * n; } else known = false; break;
    case SPHX_F_WARM: if (h->dfsph) { p = h->dfsph->getWarmStiffness().addr(); sz = 4 * n; } else known = false; break;
    case SPHX_F_POS_LAST: if (h->pbd) { p = h->pbd->getPosLast().addr(); sz = 12 * n; } else known = false; break;
    case SPHX_F_LAMBDA: if (h->pbd) { p = h->pbd->getLambda().addr(); sz = 4 * n; } else known = false; break;
    case SPHX_F_BUF3: if (h->wcsph) { p = h->wcsph->getColorGradient().addr(); sz = 12 * n; } else known = false; break;
    case SPHX_F_VEL4: if (h->wcsph) { p = h->wcsph->engineVel4(); sz = 16 * n; } else known = false; break;
    case SPHX_F_CG4: if (h->wcsph) { p = h->wcsph->engineCg4(); sz = 16 * n; } else known = false; break;
    case SPHX_F_PTERM: if (h->wcsph) { p = h->wcsph->enginePterm(); sz = 4 * n; } else known = false; break;
    case SPHX_F_POS4: if (h->wcsph) { p = h->wcsph->enginePos4(); sz = 16 * n; } else known = false; break;
    case SPHX_F_POSF: if (h->wcsph) { p = h->wcsph->enginePosf(); sz = 16 * n; } else known = false; break;
    default: known = false; break;
    }
    if (!known) return SPHX_ERR_INVALID;
    *ptr = p; *bytes = sz;
    return SPHX_OK;
}

int sphx_field_bytes(const sphx_system* h, int field, size_t* bytes)
{
    void* p;
    if (!h || !bytes || locate_noflush(h, field, &p, bytes)) return fail(SPHX_ERR_INVALID, "sphx_field_bytes: unknown field for this solver");
    return SPHX_OK;
}

// Raw device pointer of a field.  With persistent rows (reserved[3] = 2): the API fields (POS ... BMASS) are exported by every step and
// may be READ freely; a caller that WRITES one of them through the pointer must call sphx_invalidate_order() before the next step (the
// solver steps a working copy: the write would otherwise be overwritten by the next export).  Asking for a solver-internal field
// flushes the mode (the arrays are permuted into API order, the next step rebuilds its rows): use sphx_get for per-step diagnostics.
int sphx_device_ptr(const sphx_system* h, int field, void** out)
{
    if (!h || !out) return fail(SPHX_ERR_INVALID, "sphx_device_ptr: bad argument");
    return guarded("sphx_device_ptr", [&] {
        size_t sz;
        if (sphx_locate(h, field, out, &sz)) return fail(SPHX_ERR_INVALID, "sphx_device_ptr: unknown field for this solver");
        return (int)SPHX_OK;
    });
}

int sphx_invalidate_order(sphx_system* h)
{
    if (!h) return fail(SPHX_ERR_INVALID, "sphx_invalidate_order: null system");
    return guarded("sphx_invalidate_order", [&] { h->system->invalidatePersistentOrder(); return (int)SPHX_OK; });
}

int sphx_get(const sphx_system* h, int field, void* dst, size_t bytes)
{
    if (!h || !dst) return fail(SPHX_ERR_INVALID, "sphx_get: bad argument");
    return guarded("sphx_get", [&] {
        void* p; size_t sz;
        if (locate_noflush(h, field, &p, &sz)) return fail(SPHX_ERR_INVALID, "sphx_get: unknown field for this solver");
        if (bytes != sz) return fail(SPHX_ERR_INVALID, "sphx_get: size mismatch");
        if (!sz) return (int)SPHX_OK;
        // persistent rows: a solver-internal array is in the working order.  4- and 12-byte fields are read THROUGH the slot map into
        // a scratch buffer (the mode stays as it is: a per-step diagnostic read costs one gather); the 16-byte engine mirrors flush it.
        const int* slotMap = api_field(field) ? nullptr : h->system->persistentSlotMap();
        const size_t n = (size_t)h->n;
        std::unique_ptr<DArray<float>> scratch;
        if (slotMap && n > 0 && (sz == 4 * n || sz == 12 * n)) {
            const int live = (int)h->system->getFluids()->size();
            scratch.reset(new DArray<float>((unsigned)(sz / 4)));      // zero-filled by DArray: slots past a lowered active count read 0
            if (sz == 4 * n) ew_gather_float(scratch->addr(), static_cast<const float*>(p), slotMap, live);
            else ew_gather_float3(reinterpret_cast<float3*>(scratch->addr()), static_cast<const float3*>(p), slotMap, live);
            p = scratch->addr();
        } else if (slotMap) {
            if (sphx_locate(h, field, &p, &sz)) return fail(SPHX_ERR_INVALID, "sphx_get: unknown field for this solver");
        }
        if (hipMemcpyAsync(dst, p, sz, hipMemcpyDeviceToHost, sphx::stream()) != hipSuccess ||
            hipStreamSynchronize(sphx::stream()) != hipSuccess)
            return fail(SPHX_ERR_HIP, "sphx_get: copy failed");
        return (int)SPHX_OK;
    });
}

int sphx_set(sphx_system* h, int field, const void* src, size_t bytes)
{
    if (field != SPHX_F_POS && field != SPHX_F_VEL && field != SPHX_F_WARM && field != SPHX_F_BMASS && field != SPHX_F_POS_LAST &&
        field != SPHX_F_ID)
        return fail(SPHX_ERR_INVALID, "sphx_set: field is read-only");
    if (!h || !src) return fail(SPHX_ERR_INVALID, "sphx_set: bad argument");
    return guarded("sphx_set", [&] {
        void* p; size_t sz;
        h->system->invalidatePersistentOrder();       // the working copy is re-primed from the API arrays by the next step
        if (sphx_locate(h, field, &p, &sz)) return fail(SPHX_ERR_INVALID, "sphx_set: unknown field for this solver");
        if (bytes != sz) return fail(SPHX_ERR_INVALID, "sphx_set: size mismatch");
        if (!sz) return (int)SPHX_OK;
        if (hipMemcpyAsync(p, src, sz, hipMemcpyHostToDevice, sphx::stream()) != hipSuccess ||
            hipStreamSynchronize(sphx::stream()) != hipSuccess)
            return fail(SPHX_ERR_HIP, "sphx_set: copy failed");
        if (field == SPHX_F_BMASS && h->wcsph) h->wcsph->invalidateBoundary();
        if (field == SPHX_F_POS_LAST && h->pbd) h->pbd->markPosLastInitialized();
        return (int)SPHX_OK;
    });
}

int sphx_run_phase(sphx_system* h, int phase)
{
    if (!h) return fail(SPHX_ERR_INVALID, "sphx_run_phase: null system");
    return guarded("sphx_run_phase", [&] { h->system->phase(phase); return (int)SPHX_OK; });
}

int sphx_run_phase_reduce(sphx_system* h, int phase, int lo, int hi)
{
    if (!h) return fail(SPHX_ERR_INVALID, "sphx_run_phase_reduce: null system");
    return guarded("sphx_run_phase_reduce", [&] { h->system->phaseReduce(phase, lo, hi); return (int)SPHX_OK; });
}

int sphx_error_total_fixed(sphx_system* h, long long* total)
{
    if (!h || !total) return fail(SPHX_ERR_INVALID, "sphx_error_total_fixed: bad argument");
    return guarded("sphx_error_total_fixed", [&] { *total = h->system->errorTotalFixed(); return (int)SPHX_OK; });
}

int sphx_set_count(sphx_system* h, int n_fluid)
{
    if (!h || n_fluid < 0 || n_fluid > h->n) return fail(SPHX_ERR_INVALID, "sphx_set_count: count exceeds the capacity given to sphx_create");
    return guarded("sphx_set_count", [&] {
        h->system->invalidatePersistentOrder();
        h->system->getFluids()->setActiveCount((unsigned)n_fluid);
        return (int)SPHX_OK;
    });
}

int sphx_use_stream(void* hip_stream)
{
    sphx::use_external_stream(reinterpret_cast<hipStream_t>(hip_stream));
    return SPHX_OK;
}

int sphx_sync(void)
{
    if (hipStreamSynchronize(sphx::stream()) != hipSuccess) return fail(SPHX_ERR_HIP, "sphx_sync failed");
    return SPHX_OK;
}

// ------------------------------------------------------------------------------------ profiling
int sphx_profile_step(sphx_system* h, int cap, char (*names)[48], float* ms, int* count)
{
    if (!h || !names || !ms || !count) return fail(SPHX_ERR_INVALID, "sphx_profile_step: bad argument");
    KernelTimer::reset();
    KernelTimer::enabled = true;
    (void)h->system->step();
    KernelTimer::enabled = false;
    std::vector<std::string> nm; std::vector<float> t;
    KernelTimer::collect(nm, t);
    // merge spans of the same name, keep first-seen order
    std::vector<std::string> un; std::vector<float> ut;
    for (size_t i = 0; i < nm.size(); ++i) {
        size_t k = 0;
        while (k < un.size() && un[k] != nm[i]) ++k;
        if (k == un.size()) { un.push_back(nm[i]); ut.push_back(0.0f); }
        ut[k] += t[i];
    }
    *count = (int)std::min((size_t)cap, un.size());
    for (int i = 0; i < *count; ++i) {
        std::strncpy(names[i], un[i].c_str(), 47); names[i][47] = 0;
        ms[i] = ut[i];
    }
    KernelTimer::reset();
    return SPHX_OK;
}

// live kernel timing for bench.py: hipEvents on the engine stream around every launch whose span
// name equals `filter` (empty: all); while enabled, sphx_step_n launches eagerly (no graph).
int sphx_kernel_timer(int enable, const char* filter)
{
    KernelTimer::reset();
    KernelTimer::filter = (filter && enable) ? filter : "";
    KernelTimer::enabled = enable != 0;
    return SPHX_OK;
}

int sphx_kernel_timer_collect(int cap, char (*names)[48], float* total_ms, int* launches, int* count)
{
    if (!names || !total_ms || !launches || !count) return fail(SPHX_ERR_INVALID, "sphx_kernel_timer_collect: bad argument");
    std::vector<std::string> nm; std::vector<float> t;
    KernelTimer::collect(nm, t);
    std::vector<std::string> un; std::vector<float> ut; std::vector<int> uc;
    for (size_t i = 0; i < nm.size(); ++i) {
        size_t k = 0;
        while (k < un.size() && un[k] != nm[i]) ++k;
        if (k == un.size()) { un.push_back(nm[i]); ut.push_back(0.0f); uc.push_back(0); }
        ut[k] += t[i]; uc[k] += 1;
    }
    *count = (int)std::min((size_t)cap, un.size());
    for (int i = 0; i < *count; ++i) {
        std::strncpy(names[i], un[i].c_str(), 47); names[i][47] = 0;
        total_ms[i] = ut[i]; launches[i] = uc[i];
    }
    KernelTimer::reset();
    return SPHX_OK;
}

}  // extern "C"

// ------------------------------------------------------------------------------------ probes
__global__ void k_eval_kernels(const float3* __restrict__ r3, int n, KernelConsts k, float* __restrict__ W,
                               float3* __restrict__ G, float* __restrict__ V, float3* __restrict__ S)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float3 d = r3[i];
    const float r = len3(d);
    // plain operators here; the fast instantiations are compared with them by sphx_fastmath_selftest
    const float q = q_of<false>(r, k);
    W[i] = kW<false>(q, k);
    G[i] = kGradW<false>(d, q, k);
    V[i] = kViscLap<false>(r, k);
    S[i] = kSurfGrad<false>(d, r, k);
}

// global cell column of each position: exactly the expression of cell_of() (true fp32 division,
// truncation), for drivers that must agree with the engine on which column a particle is in
__global__ void k_cell_columns(const float3* __restrict__ pos, int n, float cellLength, int* __restrict__ out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (int)(pos[i].x / cellLength);
}

__global__ void k_ieee_probe(const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ c, int n,
                             float* __restrict__ quot, float* __restrict__ root, int* __restrict__ trunc,
                             float* __restrict__ muladd)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    quot[i] = a[i] / b[i];
    root[i] = sqrtf(fabsf(a[i]));
    trunc[i] = (int)(a[i] / b[i]);
    muladd[i] = a[i] * b[i] + c[i];
}

// generate_dots_CUDA, vbo.cu:26-44: copy positions, map density to the blue-white-pink ramp
__global__ void k_generate_dots(float3* __restrict__ dot, float3* __restrict__ color, const float3* __restrict__ pos,
                                const float* __restrict__ density, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    dot[i] = pos[i];
    const float rho = density[i];
    const float3 water = v3(0.34f, 0.46f, 0.7f), foam = v3(0.9f, 0.9f, 0.9f), dense = v3(1.0f, 0.4f, 0.7f);
    float3 out;
    if (rho < 0.75f) {
        out = water;
    } else if (rho < 1.0f) {
        const float w = (rho - 0.75f) * 4.0f;
        out = add3(smul3(w, foam), smul3(1 - w, water));
    } else {
        float w = (rho * rho - 1.0f) * 4.0f;
        w = fminf(w, 1.0f);
        out = add3(smul3(1 - w, foam), smul3(w, dense));
    }
    color[i] = out;
}

namespace {
template <class T>
struct DeviceTemp {
    T* p = nullptr;
    explicit DeviceTemp(size_t count) { if (hipMalloc((void**)&p, sizeof(T) * (count ? count : 1)) != hipSuccess) p = nullptr; }
    ~DeviceTemp() { if (p) (void)hipFree(p); }
};
}  // namespace

extern "C" {

int sphx_eval_kernels(const float* r3, int n, float radius, float* W, float* G, float* V, float* S)
{
    if (n <= 0) return SPHX_OK;
    if (sphx_device_count() <= 0) return fail(SPHX_ERR_NO_DEVICE, "no HIP device");
    DeviceTemp<float3> dr(n), dG(n), dS(n);
    DeviceTemp<float> dW(n), dV(n);
    if (!dr.p || !dG.p || !dS.p || !dW.p || !dV.p) return fail(SPHX_ERR_HIP, "hipMalloc failed");
    hipStream_t st = sphx::stream();
    HIP_CALL(hipMemcpyAsync(dr.p, r3, 12 * (size_t)n, hipMemcpyHostToDevice, st));
    k_eval_kernels<<<blocks_for(n), 256, 0, st>>>(dr.p, n, make_kernel_consts(radius), dW.p, dG.p, dV.p, dS.p);
    HIP_CALL(hipMemcpyAsync(W, dW.p, 4 * (size_t)n, hipMemcpyDeviceToHost, st));
    HIP_CALL(hipMemcpyAsync(G, dG.p, 12 * (size_t)n, hipMemcpyDeviceToHost, st));
    HIP_CALL(hipMemcpyAsync(V, dV.p, 4 * (size_t)n, hipMemcpyDeviceToHost, st));
    HIP_CALL(hipMemcpyAsync(S, dS.p, 12 * (size_t)n, hipMemcpyDeviceToHost, st));
    if (hipStreamSynchronize(st) != hipSuccess) return fail(SPHX_ERR_HIP, "sphx_eval_kernels failed");
    return SPHX_OK;
}

int sphx_ieee_probe(const float* a, const float* b, const float* c, int n, float* quot, float* root, int* trunc, float* muladd)
{
    if (n <= 0) return SPHX_OK;
    if (sphx_device_count() <= 0) return fail(SPHX_ERR_NO_DEVICE, "no HIP device");
    DeviceTemp<float> da(n), db(n), dc(n), dq(n), dr(n), dm(n);
    DeviceTemp<int> dt(n);
    if (!da.p || !db.p || !dc.p || !dq.p || !dr.p || !dm.p || !dt.p) return fail(SPHX_ERR_HIP, "hipMalloc failed");
    hipStream_t st = sphx::stream();
    HIP_CALL(hipMemcpyAsync(da.p, a, 4 * (size_t)n, hipMemcpyHostToDevice, st));
    HIP_CALL(hipMemcpyAsync(db.p, b, 4 * (size_t)n, hipMemcpyHostToDevice, st));
    HIP_CALL(hipMemcpyAsync(dc.p, c, 4 * (size_t)n, hipMemcpyHostToDevice, st));
    k_ieee_probe<<<blocks_for(n), 256, 0, st>>>(da.p, db.p, dc.p, n, dq.p, dr.p, dt.p, dm.p);
    HIP_CALL(hipMemcpyAsync(quot, dq.p, 4 * (size_t)n, hipMemcpyDeviceToHost, st));
    HIP_CALL(hipMemcpyAsync(root, dr.p, 4 * (size_t)n, hipMemcpyDeviceToHost, st));
    HIP_CALL(hipMemcpyAsync(trunc, dt.p, 4 * (size_t)n, hipMemcpyDeviceToHost, st));
    HIP_CALL(hipMemcpyAsync(muladd, dm.p, 4 * (size_t)n, hipMemcpyDeviceToHost, st));
    if (hipStreamSynchronize(st) != hipSuccess) return fail(SPHX_ERR_HIP, "sphx_ieee_probe failed");
    return SPHX_OK;
}

int sphx_fastmath_selftest(float radius, unsigned long long samples, unsigned int* mismatches3, int* enabled2)
{
    if (!mismatches3 || !enabled2) return fail(SPHX_ERR_INVALID, "sphx_fastmath_selftest: bad argument");
    if (sphx_device_count() <= 0) return fail(SPHX_ERR_NO_DEVICE, "no HIP device");
    fastmath_selftest(radius, samples, mismatches3, enabled2);
    return SPHX_OK;
}

int sphx_cell_columns(const float* device_xyz, int n, float cell_length, int* device_out)
{
    if (n <= 0) return SPHX_OK;
    if (!device_xyz || !device_out) return fail(SPHX_ERR_INVALID, "sphx_cell_columns: bad argument");
    k_cell_columns<<<blocks_for(n), 256, 0, sphx::stream()>>>(reinterpret_cast<const float3*>(device_xyz), n, cell_length, device_out);
    return SPHX_OK;
}

int sphx_generate_dots(const sphx_system* h, float* device_dot, float* device_color)
{
    if (!h || !device_dot || !device_color) return fail(SPHX_ERR_INVALID, "sphx_generate_dots: bad argument");
    const auto f = h->system->getFluids();
    const int n = h->n;
    if (n > 0)
        k_generate_dots<<<blocks_for(n), 256, 0, sphx::stream()>>>(reinterpret_cast<float3*>(device_dot),
                                                                   reinterpret_cast<float3*>(device_color), f->getPosPtr(),
                                                                   f->getDensityPtr(), n);
    if (hipStreamSynchronize(sphx::stream()) != hipSuccess) return fail(SPHX_ERR_HIP, "sphx_generate_dots failed");
    return SPHX_OK;
}

}  // extern "C"

// ------------------------------------------------------------------------------------ snapshots
// Container (little endian): "SPHXSNAP" | u32 version | u32 sizeof(sphx_params) | i32 n | i32 nb |
// u32 field count | sphx_params | then per field { i32 sphx_field id, u64 bytes, payload }.
// Saved: the fluid state in its CURRENT array order (pos, vel, id, density, pressure), the boundary
// set (sorted positions + masses) and the solver's persistent array (DFSPH warm stiffness / PBD last
// positions).  The reference has no checkpoint facility; this is §8(f)-2.
namespace {
constexpr unsigned int kSnapVersion = 2;       // 2: + active fluid count, + "PBD last positions initialised" (r03); version-1 files still load
struct FileCloser { void operator()(FILE* f) const { if (f) fclose(f); } };

std::vector<int> snapshot_fields(const sphx_system* h)
{
    std::vector<int> f = {SPHX_F_POS, SPHX_F_VEL, SPHX_F_ID, SPHX_F_DENSITY, SPHX_F_PRESSURE, SPHX_F_BPOS, SPHX_F_BMASS};
    if (h->dfsph) f.push_back(SPHX_F_WARM);
    if (h->pbd) f.push_back(SPHX_F_POS_LAST);
    return f;
}
}  // namespace

extern "C" {

int sphx_snapshot_save(const sphx_system* h, const char* path)
{
    if (!h || !path) return fail(SPHX_ERR_INVALID, "sphx_snapshot_save: bad argument");
    return guarded("sphx_snapshot_save", [&] {
        std::unique_ptr<FILE, FileCloser> fp(fopen(path, "wb"));
        if (!fp) return fail(SPHX_ERR_INVALID, std::string("sphx_snapshot_save: cannot open ") + path);
        const std::vector<int> fields = snapshot_fields(h);
        const unsigned int head[3] = {kSnapVersion, (unsigned int)sizeof(sphx_params), (unsigned int)fields.size()};
        const int counts[2] = {h->n, h->nb};
        // state that is not a field: the ACTIVE fluid count (sphx_set_count may have lowered it below the capacity h->n)
        // and whether the PBD solver has recorded last positions yet (its first step only does that, PBDSolver.cu:45-49)
        const int extra[2] = {(int)h->system->getFluids()->size(), (h->pbd && h->pbd->graphSafe()) ? 1 : 0};
        bool ok = fwrite("SPHXSNAP", 1, 8, fp.get()) == 8 && fwrite(head, 4, 2, fp.get()) == 2 &&
                  fwrite(counts, 4, 2, fp.get()) == 2 && fwrite(head + 2, 4, 1, fp.get()) == 1 &&
                  fwrite(extra, 4, 2, fp.get()) == 2 && fwrite(&h->params, sizeof(sphx_params), 1, fp.get()) == 1;
        std::vector<char> buf;
        for (int f : fields) {
            void* p; size_t sz;
            if (sphx_locate(h, f, &p, &sz)) return fail(SPHX_ERR_STATE, "sphx_snapshot_save: field missing");
            buf.resize(sz);
            const int rc = sz ? sphx_get(h, f, buf.data(), sz) : (int)SPHX_OK;
            if (rc) return rc;
            const unsigned long long bytes = sz;
            ok = ok && fwrite(&f, 4, 1, fp.get()) == 1 && fwrite(&bytes, 8, 1, fp.get()) == 1 &&
                 (sz == 0 || fwrite(buf.data(), 1, sz, fp.get()) == sz);
        }
        if (!ok) return fail(SPHX_ERR_INVALID, "sphx_snapshot_save: short write");
        return (int)SPHX_OK;
    });
}

int sphx_snapshot_load(const char* path, sphx_system** out)
{
    if (!path || !out) return fail(SPHX_ERR_INVALID, "sphx_snapshot_load: bad argument");
    return guarded("sphx_snapshot_load", [&] {
        *out = nullptr;
        std::unique_ptr<FILE, FileCloser> fp(fopen(path, "rb"));
        if (!fp) return fail(SPHX_ERR_INVALID, std::string("sphx_snapshot_load: cannot open ") + path);
        char magic[8]; unsigned int head[2]; int counts[2]; unsigned int nfields = 0;
        sphx_params P;
        int extra[2] = {-1, -1};      // version 1: every slot active, last positions initialised iff the field is there
        if (fread(magic, 1, 8, fp.get()) != 8 || std::memcmp(magic, "SPHXSNAP", 8) != 0 || fread(head, 4, 2, fp.get()) != 2 ||
            (head[0] != 1u && head[0] != kSnapVersion) || head[1] != sizeof(sphx_params) || fread(counts, 4, 2, fp.get()) != 2 ||
            fread(&nfields, 4, 1, fp.get()) != 1 || (head[0] >= 2u && fread(extra, 4, 2, fp.get()) != 2) ||
            fread(&P, sizeof(P), 1, fp.get()) != 1 || counts[0] < 0 || counts[1] < 0 || nfields > 64 || extra[0] > counts[0])
            return fail(SPHX_ERR_INVALID, "sphx_snapshot_load: not a version-1/2 sphx snapshot");
        std::vector<std::pair<int, std::vector<char>>> blobs(nfields);
        for (auto& b : blobs) {
            unsigned long long bytes = 0;
            if (fread(&b.first, 4, 1, fp.get()) != 1 || fread(&bytes, 8, 1, fp.get()) != 1 ||
                bytes > 16ull * (unsigned long long)std::max(counts[0], counts[1]) + 64ull)
                return fail(SPHX_ERR_INVALID, "sphx_snapshot_load: truncated field header");
            b.second.resize((size_t)bytes);
            if (bytes && fread(b.second.data(), 1, (size_t)bytes, fp.get()) != bytes)
                return fail(SPHX_ERR_INVALID, "sphx_snapshot_load: truncated payload");
        }
        auto find = [&](int id) -> std::vector<char>* {
            for (auto& b : blobs) if (b.first == id) return &b.second;
            return nullptr;
        };
        std::vector<char>* pos = find(SPHX_F_POS); std::vector<char>* bpos = find(SPHX_F_BPOS);
        if (!pos || !bpos || pos->size() != 12u * (size_t)counts[0] || bpos->size() != 12u * (size_t)counts[1])
            return fail(SPHX_ERR_INVALID, "sphx_snapshot_load: position fields missing or mis-sized");
        sphx_system* h = nullptr;
        // mode 2: no constructor step and NO initial fluid sort — the arrays continue in the saved order
        int rc = sphx_create_impl(&P, reinterpret_cast<const float*>(pos->data()), counts[0],
                             reinterpret_cast<const float*>(bpos->data()), counts[1], 2, &h);
        if (rc) return rc;
        for (auto& b : blobs) {
            if (b.first == SPHX_F_POS || b.first == SPHX_F_BPOS || b.second.empty()) continue;
            void* p; size_t sz;
            if (sphx_locate(h, b.first, &p, &sz) || sz != b.second.size()) { sphx_destroy(h); return fail(SPHX_ERR_INVALID, "sphx_snapshot_load: field does not fit this solver"); }
            if (hipMemcpyAsync(p, b.second.data(), sz, hipMemcpyHostToDevice, sphx::stream()) != hipSuccess ||
                hipStreamSynchronize(sphx::stream()) != hipSuccess) { sphx_destroy(h); return fail(SPHX_ERR_HIP, "sphx_snapshot_load: upload failed"); }
            if (b.first == SPHX_F_BMASS && h->wcsph) h->wcsph->invalidateBoundary();
            if (b.first == SPHX_F_POS_LAST && h->pbd && extra[1] != 0) h->pbd->markPosLastInitialized();
        }
        if (extra[0] >= 0 && extra[0] < counts[0]) h->system->getFluids()->setActiveCount((unsigned)extra[0]);
        *out = h;
        return (int)SPHX_OK;
    });
}

}  // extern "C"

// generate_dots with the reference's own signature (vbo.cu:46-51, declared `extern "C"` at
// main.cpp:268): a main.cpp that links against this symbol links against libsphx.so unchanged.
// dot / color are DEVICE pointers (the reference maps its VBOs first, main.cpp:270-289).
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wreturn-type-c-linkage"
extern "C" void generate_dots(float3* dot, float3* color, const std::shared_ptr<SPHParticles> particles)
{
    const int n = particles ? (int)particles->size() : 0;
    if (n <= 0 || !dot || !color) return;
    k_generate_dots<<<blocks_for(n), 256, 0, sphx::stream()>>>(dot, color, particles->getPosPtr(), particles->getDensityPtr(), n);
    HIP_CALL(hipStreamSynchronize(sphx::stream()));   // the caller unmaps its buffers right after (main.cpp:291-292)
}
#pragma clang diagnostic pop

