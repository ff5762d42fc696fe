// sphx_demo — headless driver written against the drop-in C++ API exactly the way the reference's
// interactive driver uses it (scene constants, particle generation, solver selection, SPHSystem
// construction, one step per "frame" with the running-average report).  What the reference does in
// src/main.cpp:54-135 and :300-306, without the GLUT/GL parts.
//
//   sphx_demo [--solver wcsph|dfsph|pbd] [--nx 24] [--steps 100] [--restart-with wcsph|dfsph|pbd] [--dump file.bin]
//
// --restart-with re-initialises the whole scene with another solver after the first run, in the same
// process — what the reference's keys '1' '2' '3' do (src/main.cpp:225-239 -> initSPHSystem).
// Status: added after the round's GPU budget was spent; compiles and links, NOT yet run on a GPU.
// tests/test_gpu_parity.py::test_cpp_api_driver_matches_oracle covers the default path only.
// --dump writes n, then pos[n*3], density[n] (cell-sorted order) for the parity test.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstring>
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

namespace fluid_solver { enum { SPH, DFSPH, PBD }; }

int main(int argc, char** argv)
{
    int solverKind = fluid_solver::PBD, nx = 24, steps = 100, restartKind = -1;
    std::string dump;
    auto parseSolver = [](const std::string& v) {
        return v == "wcsph" ? (int)fluid_solver::SPH : v == "dfsph" ? (int)fluid_solver::DFSPH : (int)fluid_solver::PBD;
    };
    for (int a = 1; a < argc; ++a) {
        const std::string k = argv[a];
        if (k == "--solver" && a + 1 < argc) {
            solverKind = parseSolver(argv[++a]);
        } else if (k == "--restart-with" && a + 1 < argc) {
            restartKind = parseSolver(argv[++a]);
        } else if (k == "--nx" && a + 1 < argc) nx = atoi(argv[++a]);
        else if (k == "--steps" && a + 1 < argc) steps = atoi(argv[++a]);
        else if (k == "--dump" && a + 1 < argc) dump = argv[++a];
    }
    int devices = 0;
    if (hipGetDeviceCount(&devices) != hipSuccess || devices < 1) {
        fprintf(stderr, "sphx_demo: no HIP device (the engine has no CPU path)\n");
        return 2;
    }

    // scene constants (scaled by nx/24; nx = 24 is the reference scene)
    const float scale = (float)nx / 24.0f;
    const float3 spaceSize = make_float3(scale);
    const float sphSpacing = 0.02f;
    const float sphSmoothingRadius = 2.0f * sphSpacing;
    const float sphCellLength = 1.01f * sphSmoothingRadius;
    const float dt = 0.002f;
    const float sphRho0 = 1.0f;
    const float sphRhoBoundary = 1.4f * sphRho0;
    const float sphM0 = 76.596750762082e-6f;
    const float sphStiff = 10.0f;
    const float3 sphG = make_float3(0.0f, -9.8f, 0.0f);
    const float sphVisc = 5e-4f;
    const float sphSurfaceTensionIntensity = 0.0001f;
    const float sphAirPressure = 0.0001f;
    const int3 cellSize = make_int3((int)ceilf(spaceSize.x / sphCellLength), (int)ceilf(spaceSize.y / sphCellLength),
                                    (int)ceilf(spaceSize.z / sphCellLength));

    // initSPHSystem(solver), src/main.cpp:73-135: particles, solver and system are rebuilt from scratch
    auto initSPHSystem = [&](const int kind) -> std::shared_ptr<SPHSystem> {
    // fluid block
    std::vector<float3> pos;
    const float3 origin = make_float3(0.27f * scale, 0.10f * scale, 0.27f * scale);
    for (int iy = 0; iy < 3 * nx / 2; ++iy)
        for (int ix = 0; ix < nx; ++ix)
            for (int iz = 0; iz < nx; ++iz)
                pos.push_back(make_float3(origin.x + sphSpacing * ix, origin.y + sphSpacing * iy, origin.z + sphSpacing * iz));
    auto fluidParticles = std::make_shared<SPHParticles>(pos);

    // boundary shell: six faces, edges counted once
    pos.clear();
    const int3 shell = make_int3(2 * cellSize.x, 2 * cellSize.y, 2 * cellSize.z);
    auto wall = [&](int a, int b, int c) {
        const float3 t = make_float3((float)a / (float)(shell.x - 1) * spaceSize.x, (float)b / (float)(shell.y - 1) * spaceSize.y,
                                     (float)c / (float)(shell.z - 1) * spaceSize.z);
        pos.push_back(make_float3(0.99f * t.x + 0.005f * spaceSize.x, 0.99f * t.y + 0.005f * spaceSize.y,
                                  0.99f * t.z + 0.005f * spaceSize.z));
    };
    for (int a = 0; a < shell.x; ++a) for (int b = 0; b < shell.y; ++b) { wall(a, b, 0); wall(a, b, shell.z - 1); }
    for (int a = 0; a < shell.x; ++a) for (int c = 1; c < shell.z - 1; ++c) { wall(a, 0, c); wall(a, shell.y - 1, c); }
    for (int b = 1; b < shell.y - 1; ++b) for (int c = 1; c < shell.z - 1; ++c) { wall(0, b, c); wall(shell.x - 1, b, c); }
    auto boundaryParticles = std::make_shared<SPHParticles>(pos);

    std::shared_ptr<BaseSolver> pSolver;
    switch (kind) {
    case fluid_solver::PBD: pSolver = std::make_shared<PBDSolver>(fluidParticles->size()); break;
    case fluid_solver::DFSPH: pSolver = std::make_shared<DFSPHSolver>(fluidParticles->size()); break;
    default: pSolver = std::make_shared<BasicSPHSolver>(fluidParticles->size()); break;
    }
    auto pSystem = std::make_shared<SPHSystem>(fluidParticles, boundaryParticles, pSolver, spaceSize, sphCellLength,
                                               sphSmoothingRadius, dt, sphM0, sphRho0, sphRhoBoundary, sphStiff, sphVisc,
                                               sphSurfaceTensionIntensity, sphAirPressure, sphG, cellSize);
    printf("particles: %d fluid + %d boundary, grid %dx%dx%d\n", pSystem->fluidSize(), pSystem->boundarySize(), cellSize.x,
           cellSize.y, cellSize.z);
    return pSystem;
    };

    // oneStep() + the running report of src/main.cpp:300-306
    auto run = [&](const std::shared_ptr<SPHSystem>& system) {
        float totalTime = 0.0f;
        for (int frameId = 1; frameId <= steps; ++frameId) {
            const float milliseconds = system->step();
            totalTime += milliseconds;
            if (frameId % 10 == 0 || frameId == steps)
                printf("Frame %d - %2.2f ms, avg time - %2.2f ms/frame (%3.2f FPS)\n", frameId, milliseconds,
                       totalTime / float(frameId), float(frameId) * 1000.0f / totalTime);
        }
    };

    auto pSystem = initSPHSystem(solverKind);
    run(pSystem);
    if (restartKind >= 0) {
        pSystem.reset();
        pSystem = initSPHSystem(restartKind);
        run(pSystem);
    }

    if (!dump.empty()) {
        const int n = pSystem->fluidSize();
        std::vector<float3> hp(n);
        std::vector<float> hd(n);
        const auto f = pSystem->getFluids();
        (void)hipMemcpy(hp.data(), f->getPosPtr(), sizeof(float3) * n, hipMemcpyDeviceToHost);
        (void)hipMemcpy(hd.data(), f->getDensityPtr(), sizeof(float) * n, hipMemcpyDeviceToHost);
        FILE* fp = fopen(dump.c_str(), "wb");
        if (!fp) return 3;
        fwrite(&n, sizeof(int), 1, fp);
        fwrite(hp.data(), sizeof(float3), n, fp);
        fwrite(hd.data(), sizeof(float), n, fp);
        fclose(fp);
    }
    return 0;
}
