// sphx_demo — headless driver on the drop-in C++ API, used the way the reference's interactive
// driver uses it: particle sets -> solver plugin -> SPHSystem -> one step() per frame with a running
// timing report (what src/main.cpp:73-135 and :300-306 do, minus GLUT/GL).
//
//   sphx_demo [--solver wcsph|dfsph|pbd] [--nx 24] [--steps 100] [--restart-with wcsph|dfsph|pbd]
//             [--dump file.bin] [--dots file.bin] [--save snap.bin] [--load snap.bin] [--advect-check file.bin]
//             [--arith strict|tolerance|persistent]
//
// The scene (constants of main.cpp:54-67, block and shell samplers of :73-117, scaled by nx/24) comes
// from the library's scene generator (sphx_scene_params / sphx_scene_fill), the same one the tests
// and bench.py use.  --restart-with rebuilds everything with another solver in the same process —
// what the reference's keys '1' '2' '3' trigger (main.cpp:225-239).  --save writes a snapshot after
// the run; --load continues a saved run for --steps more frames (through the C ABI, which owns the
// snapshot format).  --dots calls the reference-signature generate_dots() and writes n, dot[n*3],
// color[n*3].  --dump writes n, pos[n*3], density[n] (current array order) for the parity tests.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <memory>
#include <string>
#include <vector>

#include "sphx_math.h"
#include "DArray.h"
#include "Particles.h"
#include "SPHParticles.h"
#include "BaseSolver.h"
#include "BasicSPHSolver.h"
#include "DFSPHSolver.h"
#include "PBDSolver.h"
#include "SPHSystem.h"
#include "sphx_c.h"

// the render-side consumer, declared the way the reference's driver declares it (main.cpp:268)
extern "C" void generate_dots(float3* dot, float3* color, const std::shared_ptr<SPHParticles> particles);

static int solver_from_name(const std::string& v) { return v == "wcsph" ? SPHX_WCSPH : (v == "dfsph" ? SPHX_DFSPH : SPHX_PBD); }

static std::vector<float3> as_float3(const std::vector<float>& xyz)
{
    std::vector<float3> out(xyz.size() / 3);
    for (size_t i = 0; i < out.size(); ++i) out[i] = make_float3(xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]);
    return out;
}

static void report(int frame, int frames, float ms, float total)
{
    if (frame % 10 == 0 || frame == frames)
        printf("step %d: %.2f ms   mean %.2f ms/step   %.1f steps/s\n", frame, ms, total / (float)frame, 1000.0f * (float)frame / total);
}

static int write_dump(const std::string& path, int n, const float3* dPos, const float* dDensity)
{
    std::vector<float3> hp(n);
    std::vector<float> hd(n);
    (void)hipMemcpy(hp.data(), dPos, sizeof(float3) * n, hipMemcpyDeviceToHost);
    (void)hipMemcpy(hd.data(), dDensity, sizeof(float) * n, hipMemcpyDeviceToHost);
    FILE* fp = fopen(path.c_str(), "wb");
    if (!fp) return 3;
    fwrite(&n, sizeof(int), 1, fp);
    fwrite(hp.data(), sizeof(float3), n, fp);
    fwrite(hd.data(), sizeof(float), n, fp);
    fclose(fp);
    return 0;
}

int main(int argc, char** argv)
{
    int solverKind = SPHX_PBD, nx = 24, steps = 100, restartKind = -1, arith = 0;
    std::string dump, savePath, loadPath, dotsPath, advectPath;
    for (int a = 1; a < argc; ++a) {
        const std::string k = argv[a];
        const bool more = a + 1 < argc;
        if (k == "--solver" && more) solverKind = solver_from_name(argv[++a]);
        else if (k == "--restart-with" && more) restartKind = solver_from_name(argv[++a]);
        else if (k == "--nx" && more) nx = atoi(argv[++a]);
        else if (k == "--steps" && more) steps = atoi(argv[++a]);
        else if (k == "--dump" && more) dump = argv[++a];
        else if (k == "--dots" && more) dotsPath = argv[++a];
        else if (k == "--save" && more) savePath = argv[++a];
        else if (k == "--load" && more) loadPath = argv[++a];
        else if (k == "--advect-check" && more) advectPath = argv[++a];
        else if (k == "--arith" && more) { const std::string v = argv[++a]; arith = v == "persistent" ? 2 : (v == "tolerance" ? 1 : 0); }
    }
    if (sphx_device_count() < 1) {
        fprintf(stderr, "sphx_demo: no HIP device (the engine has no CPU path)\n");
        return 2;
    }

    if (!advectPath.empty()) {          // Particles::advect on its own (Particles.cu:28-36): n, dt, pos before, vel, pos after
        const int n = 4099;
        std::vector<float3> p0((size_t)n), v((size_t)n), p1((size_t)n);
        unsigned int rng = 2463534242u;
        auto next = [&]() { rng = rng * 1664525u + 1013904223u; return (float)(rng >> 8) * (1.0f / 16777216.0f); };
        for (int i = 0; i < n; ++i) {
            p0[i] = make_float3(next() * 1.7f, next() * 1e-3f, next() * 40.0f - 20.0f);
            v[i] = make_float3(next() * 6.0f - 3.0f, -next() * 9.8f, (i % 7 == 0) ? 0.0f : next() * 1e-4f);
        }
        Particles set(p0);
        (void)hipMemcpy(set.getVelPtr(), v.data(), sizeof(float3) * n, hipMemcpyHostToDevice);
        const float dt = 0.002f;
        set.advect(dt);
        (void)hipDeviceSynchronize();
        (void)hipMemcpy(p1.data(), set.getPosPtr(), sizeof(float3) * n, hipMemcpyDeviceToHost);
        FILE* fp = fopen(advectPath.c_str(), "wb");
        if (!fp) return 3;
        fwrite(&n, sizeof(int), 1, fp); fwrite(&dt, sizeof(float), 1, fp);
        fwrite(p0.data(), sizeof(float3), n, fp); fwrite(v.data(), sizeof(float3), n, fp); fwrite(p1.data(), sizeof(float3), n, fp);
        fclose(fp);
        return 0;
    }

    if (!loadPath.empty()) {            // continue a saved run
        sphx_system* sys = nullptr;
        if (sphx_snapshot_load(loadPath.c_str(), &sys) != SPHX_OK) { fprintf(stderr, "sphx_demo: %s\n", sphx_last_error()); return 4; }
        float total = 0.0f;
        for (int f = 1; f <= steps; ++f) { float ms = 0.0f; sphx_step(sys, &ms); total += ms; report(f, steps, ms, total); }
        int n = 0; sphx_counts(sys, &n, nullptr, nullptr);
        void *dp = nullptr, *dd = nullptr;
        sphx_device_ptr(sys, SPHX_F_POS, &dp); sphx_device_ptr(sys, SPHX_F_DENSITY, &dd);
        int rc = dump.empty() ? 0 : write_dump(dump, n, (const float3*)dp, (const float*)dd);
        if (!savePath.empty() && sphx_snapshot_save(sys, savePath.c_str()) != SPHX_OK) rc = 5;
        sphx_destroy(sys);
        return rc;
    }

    sphx_params sc;
    int nFluid = 0, nWall = 0;
    if (sphx_scene_params(nx, &sc) != SPHX_OK || sphx_scene_counts(nx, &nFluid, &nWall) != SPHX_OK) {
        fprintf(stderr, "sphx_demo: %s\n", sphx_last_error());
        return 2;
    }
    std::vector<float> fluidXyz(3 * (size_t)nFluid), wallXyz(3 * (size_t)nWall);
    sphx_scene_fill(nx, fluidXyz.data(), wallXyz.data());

    // everything is rebuilt from scratch per solver, as the reference's initSPHSystem does
    auto build = [&](const int kind) {
        auto fluidParticles = std::make_shared<SPHParticles>(as_float3(fluidXyz));
        auto boundaryParticles = std::make_shared<SPHParticles>(as_float3(wallXyz));
        std::shared_ptr<BaseSolver> plugin;
        if (kind == SPHX_PBD) plugin = std::make_shared<PBDSolver>(fluidParticles->size());
        else if (kind == SPHX_DFSPH) plugin = std::make_shared<DFSPHSolver>(fluidParticles->size());
        else plugin = std::make_shared<BasicSPHSolver>(fluidParticles->size());
        // engine extensions of the drop-in API (none changes a reference signature): the arithmetic contract of the neighbour sweeps
        if (arith >= 1) static_cast<BasicSPHSolver*>(plugin.get())->setToleranceArithmetic(true);
        auto system = std::make_shared<SPHSystem>(
            fluidParticles, boundaryParticles, plugin, make_float3(sc.space[0], sc.space[1], sc.space[2]), sc.cell_length,
            sc.radius, sc.dt, sc.m0, sc.rho0, sc.rho_boundary, sc.stiff, sc.visc, sc.surface_tension, sc.air_pressure,
            make_float3(sc.gravity[0], sc.gravity[1], sc.gravity[2]), make_int3(sc.cells[0], sc.cells[1], sc.cells[2]));
        printf("scene: %d fluid + %d wall particles in a %d x %d x %d grid\n", system->fluidSize(), system->boundarySize(), sc.cells[0],
               sc.cells[1], sc.cells[2]);
        if (arith == 2 && !system->setPersistentRows(true)) printf("persistent rows are not available for this solver: tolerance arithmetic only\n");
        return system;
    };
    auto run = [&](const std::shared_ptr<SPHSystem>& system) {
        float total = 0.0f;
        for (int f = 1; f <= steps; ++f) {
            const float ms = system->step();
            total += ms;
            report(f, steps, ms, total);
        }
    };

    auto system = build(solverKind);
    run(system);
    if (restartKind >= 0) {
        system.reset();
        system = build(restartKind);
        run(system);
    }
    if (!savePath.empty()) fprintf(stderr, "sphx_demo: --save needs a run started with --load or the C ABI (snapshots belong to sphx_c.h)\n");
    if (!dotsPath.empty()) {            // what the reference's renderer does once per frame (main.cpp:270-292)
        const int n = system->fluidSize();
        float3 *dDot = nullptr, *dColor = nullptr;
        (void)hipMalloc((void**)&dDot, sizeof(float3) * n);
        (void)hipMalloc((void**)&dColor, sizeof(float3) * n);
        generate_dots(dDot, dColor, system->getFluids());
        std::vector<float3> h(2 * (size_t)n);
        (void)hipMemcpy(h.data(), dDot, sizeof(float3) * n, hipMemcpyDeviceToHost);
        (void)hipMemcpy(h.data() + n, dColor, sizeof(float3) * n, hipMemcpyDeviceToHost);
        (void)hipFree(dDot); (void)hipFree(dColor);
        if (FILE* fp = fopen(dotsPath.c_str(), "wb")) { fwrite(&n, sizeof(int), 1, fp); fwrite(h.data(), sizeof(float3), h.size(), fp); fclose(fp); }
    }
    if (!dump.empty()) {
        const auto f = system->getFluids();
        return write_dump(dump, system->fluidSize(), f->getPosPtr(), f->getDensityPtr());
    }
    return 0;
}
