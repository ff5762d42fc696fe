// Diagnostic only: headless driver restating main.cpp:54-135 for the stand-in-header CPU build of the reference sources
// (tools/ref_oracle/README.md).  argv: solver(0|1|2) steps dt [dumpdir [dump_every [scene.bin]]]
#include <cuda_runtime.h>
#include <helper_math.h>
#include <thrust/execution_policy.h>
#include "CUDAFunctions.cuh"
#include "DArray.h"
#include "Particles.h"
#include "SPHParticles.h"
#include "BaseSolver.h"
#include "BasicSPHSolver.h"
#include "DFSPHSolver.h"
#include "PBDSolver.h"
#include "SPHSystem.h"
thread_local uint3s threadIdx, blockIdx, blockDim;
int g_it_div = -1, g_it_den = -1;
int main(int argc, char** argv) {
  int solver = atoi(argv[1]); int steps = atoi(argv[2]); float dt = atof(argv[3]); const char* dumpdir = argc > 4 ? argv[4] : nullptr;
  int dump_every = argc > 5 ? atoi(argv[5]) : 10;
  const char* sceneFile = argc > 6 ? argv[6] : nullptr;
  float sceneScale = 1.0f; int nSceneF = 0, nSceneB = 0; std::vector<float3> scenePos, sceneVel, sceneB;
  if (sceneFile) { FILE* f = fopen(sceneFile, "rb"); fread(&sceneScale, 4, 1, f); fread(&nSceneF, 4, 1, f); fread(&nSceneB, 4, 1, f);
    scenePos.resize(nSceneF); sceneVel.resize(nSceneF); sceneB.resize(nSceneB);
    fread(scenePos.data(), 12, nSceneF, f); fread(sceneVel.data(), 12, nSceneF, f); fread(sceneB.data(), 12, nSceneB, f); fclose(f); }
  const float3 spaceSize = make_float3(sceneScale);
  const float sphSpacing = 0.02f, R = 2.0f * sphSpacing, cellLen = 1.01f * R, rho0 = 1.0f, rhoB = 1.4f * rho0;
  const float m0 = 76.596750762082e-6f, stiff = 10.0f, visc = 5e-4f, sti = 0.0001f, airP = 0.0001f;
  const float3 G = make_float3(0.0f, -9.8f, 0.0f);
  const int3 cellSize = make_int3(ceil(spaceSize.x / cellLen), ceil(spaceSize.y / cellLen), ceil(spaceSize.z / cellLen));
  std::vector<float3> pos;
  for (int i = 0; i < 36; ++i) for (int j = 0; j < 24; ++j) for (int k = 0; k < 24; ++k)
    pos.push_back(make_float3(0.27f + sphSpacing * j, 0.10f + sphSpacing * i, 0.27f + sphSpacing * k));
  std::vector<float3> vel0;
  if (sceneFile) { pos = scenePos; vel0 = sceneVel; }
  // pre-sort by cell id (stable) so ids[q] = q after the ctor's searches
  { std::vector<int> key(pos.size()); for (size_t q = 0; q < pos.size(); ++q) { int3 c = make_int3(pos[q] / cellLen); key[q] = (c.x * cellSize.y + c.y) * cellSize.z + c.z; }
    std::vector<int> key2 = key;
    thrust::sort_by_key(thrust::device, key.data(), key.data() + key.size(), pos.data());
    if (!vel0.empty()) thrust::sort_by_key(thrust::device, key2.data(), key2.data() + key2.size(), vel0.data()); }
  std::vector<float3> pos0 = pos;
  auto fluid = std::make_shared<SPHParticles>(pos);
  if (!vel0.empty()) memcpy(fluid->getVelPtr(), vel0.data(), 12 * vel0.size());
  pos.clear();
  const int3 cs = 2 * cellSize;
  for (int i = 0; i < cs.x; ++i) for (int j = 0; j < cs.y; ++j) {
    float3 x = make_float3(i, j, 0) / make_float3(cs - make_int3(1)) * spaceSize; pos.push_back(0.99f * x + 0.005f * spaceSize);
    x = make_float3(i, j, cs.z - 1) / make_float3(cs - make_int3(1)) * spaceSize; pos.push_back(0.99f * x + 0.005f * spaceSize); }
  for (int i = 0; i < cs.x; ++i) for (int j = 0; j < cs.z - 2; ++j) {
    float3 x = make_float3(i, 0, j + 1) / make_float3(cs - make_int3(1)) * spaceSize; pos.push_back(0.99f * x + 0.005f * spaceSize);
    x = make_float3(i, cs.y - 1, j + 1) / make_float3(cs - make_int3(1)) * spaceSize; pos.push_back(0.99f * x + 0.005f * spaceSize); }
  for (int i = 0; i < cs.y - 2; ++i) for (int j = 0; j < cs.z - 2; ++j) {
    float3 x = make_float3(0, i + 1, j + 1) / make_float3(cs - make_int3(1)) * spaceSize; pos.push_back(0.99f * x + 0.005f * spaceSize);
    x = make_float3(cs.x - 1, i + 1, j + 1) / make_float3(cs - make_int3(1)) * spaceSize; pos.push_back(0.99f * x + 0.005f * spaceSize); }
  if (sceneFile) pos = sceneB;
  auto boundary = std::make_shared<SPHParticles>(pos);
  std::shared_ptr<BaseSolver> s;
  int n = fluid->size();
  if (solver == 2) s = std::make_shared<PBDSolver>(n); else if (solver == 1) s = std::make_shared<DFSPHSolver>(n); else s = std::make_shared<BasicSPHSolver>(n);
  if (dumpdir) { char fn[256]; snprintf(fn, 256, "%s/scene.bin", dumpdir); FILE* f = fopen(fn, "wb"); int nb = pos.size(); fwrite(&n, 4, 1, f); fwrite(&nb, 4, 1, f); fwrite(pos0.data(), 12, n, f); fwrite(pos.data(), 12, nb, f); fclose(f); }
  SPHSystem sys(fluid, boundary, s, spaceSize, cellLen, R, dt, m0, rho0, rhoB, stiff, visc, sti, airP, G, cellSize);
  std::vector<int> ids(n); for (int q = 0; q < n; ++q) ids[q] = q;
  auto report = [&](int step) {
    auto fl = sys.getFluids(); const float3* p = fl->getPosPtr(); const float3* v = fl->getVelPtr(); const float* d = fl->getDensityPtr();
    double sr = 0, sy = 0, vm = 0; float rmin = 1e30f, rmax = -1e30f;
    for (int q = 0; q < n; ++q) { sr += d[q]; sy += p[q].y; double l = sqrt((double)v[q].x * v[q].x + (double)v[q].y * v[q].y + (double)v[q].z * v[q].z); if (l > vm) vm = l; rmin = std::min(rmin, d[q]); rmax = std::max(rmax, d[q]); }
    printf("step %d rho_mean %.6f rho_min %.4f rho_max %.4f mean_y %.6f vmax %.4f iters (%d,%d)\n", step, sr / n, rmin, rmax, sy / n, vm, g_it_div, g_it_den);
    if (dumpdir && step % dump_every == 0) { char fn[256]; snprintf(fn, 256, "%s/s%d_%04d.bin", dumpdir, solver, step); FILE* f = fopen(fn, "wb");
      std::vector<float3> P(n), V(n); std::vector<float> D(n);
      for (int q = 0; q < n; ++q) { P[ids[q]] = p[q]; V[ids[q]] = v[q]; D[ids[q]] = d[q]; }
      fwrite(P.data(), 12, n, f); fwrite(V.data(), 12, n, f); fwrite(D.data(), 4, n, f); fclose(f); }
  };
  if (dumpdir) { char fn[256]; snprintf(fn, 256, "%s/bmass.bin", dumpdir); FILE* f = fopen(fn, "wb"); int nbb = sys.getBoundaries()->size();
    fwrite(sys.getBoundaries()->getMassPtr(), 4, nbb, f); fwrite(sys.getBoundaries()->getPosPtr(), 12, nbb, f); fclose(f); }
  report(0);
  for (int st = 1; st <= steps; ++st) {
    sys.step();
    std::vector<int> key(sys.getFluids()->getParticle2Cell(), sys.getFluids()->getParticle2Cell() + n);
    thrust::sort_by_key(thrust::device, key.data(), key.data() + n, ids.data());
    if (st % 10 == 0 || st <= 2) report(st);
  }
  return 0;
}
