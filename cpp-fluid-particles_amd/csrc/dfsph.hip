// dfsph.hip — DFSPHSolver (divergence-free SPH, Bender & Koschier 2015) as HIP kernels for gfx950.
//
// Reference behaviour restated from src/DFSPHSolver.cu:33-363: step order, the two
// solver loops and their warm-start bookkeeping (Q9).  Differences in mechanism, not in result:
// the warm-stiffness array follows the particles through the published sort permutation instead
// of a second key sort; the warm reset / accumulate passes and the |error| reduction are fused
// into the error kernel; the termination sum is an exact fixed-point integer, read
// back only in adaptive mode.
#include <algorithm>
#include <limits>

#include "DFSPHSolver.h"
#include "engine.hpp"
#include "sweep_ops.hpp"
#include "sphx_c.h"

using namespace sphx;

DFSPHSolver::DFSPHSolver(int num, float defaultDensityErrorThreshold, float defaultDivergenceErrorThreshold,
                         int defaultMaxIter)
    : BasicSPHSolver(num), alpha((unsigned)num), bufferFloat((unsigned)num), error((unsigned)num),
      denWarmStiff((unsigned)num), scratch((unsigned)num), errorAccum(2u * kErrorSlots * kErrorSlotStride),
      densityErrorThreshold(defaultDensityErrorThreshold), divergenceErrorThreshold(defaultDivergenceErrorThreshold),
      maxIter(defaultMaxIter), loopState((unsigned)kLoopWords)
{
    HIP_CALL(hipHostMalloc((void**)&hostIters, 3 * sizeof(int), hipHostMallocDefault));
    if (hostIters) hostIters[0] = hostIters[1] = hostIters[2] = 0;
}
DFSPHSolver::~DFSPHSolver() noexcept { if (hostIters) (void)hipHostFree(hostIters); }

// ---- adaptive loops decided on the device ------------------------------------------------------------------------------
// The reference's two solver loops (DFSPHSolver.cu:160-210, :331-363) test a reduced |error| total on the host after every
// iteration.  Here the total is an exact integer in device memory (D2): a one-block kernel after every error sweep adds the
// partial sums up, counts the iteration and raises `done` by the reference's own rule; the sweeps of all maxIter possible
// iterations are enqueued up front and leave at their first instruction once `done` is up (SweepCtx::gate).  No host round
// trip, same iteration counts, and the whole adaptive step can be captured into a hipGraph.
namespace {
__global__ void __launch_bounds__(kErrorSlots) k_loop_reset(int* __restrict__ st, unsigned long long* __restrict__ accum)
{
    accum[(size_t)threadIdx.x * kErrorSlotStride] = 0ull;
    if (threadIdx.x == 0) { st[kLoopDone] = 0; st[kLoopIter] = 0; st[kLoopBarrier] = 0; }
    for (int w = kLoopXcd + (int)threadIdx.x; w < kLoopWords; w += kErrorSlots) st[w] = 0;      // the XCD barrier's counters
}
__global__ void __launch_bounds__(kErrorSlots) k_loop_decide(int* __restrict__ st, unsigned long long* __restrict__ accum, float threshold,
                                                             int minIter, int maxIter, int which)
{
    if (st[kLoopDone] != 0) return;
    __shared__ unsigned long long part[kErrorSlots / 64];
    unsigned long long v = accum[(size_t)threadIdx.x * kErrorSlotStride];
    accum[(size_t)threadIdx.x * kErrorSlotStride] = 0ull;          // ready for the next error sweep
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x != 0) return;
    unsigned long long total = 0;
    for (int w = 0; w < kErrorSlots / 64; ++w) total += part[w];
    const float totalError = (float)((double)(long long)total * (1.0 / 4294967296.0));      // DFSPHSolver::readErrorTotal
    const int iter = st[kLoopIter] + 1;
    st[kLoopIter] = iter;
    st[which] = iter;
    if (!((iter < minIter || totalError > threshold) && iter < maxIter)) st[kLoopDone] = 1;
}
}  // namespace

namespace {
__global__ void k_set_posf_w(float4* __restrict__ posf, const float* __restrict__ src, int lo, int hi)
{
    const int i = lo + (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (i < hi) posf[i].w = src[i];
}
}  // namespace

void DFSPHSolver::packWarmIntoPosf(int lo, int hi)
{
    SweepCache& c = cache();
    lo = std::max(lo, 0); hi = std::min(hi, c.n);
    if (hi > lo) k_set_posf_w<<<blocks_for(hi - lo), 256, 0, sphx::stream()>>>(c.posfw(), denWarmStiff.addr(), lo, hi);
    warmInPosfGhosts = true;
}

bool DFSPHSolver::deviceLoops() const
{
    const bool hostLoop = tuning().dfsph_host_loop != 0;
    const SweepCache& c = const_cast<DFSPHSolver*>(this)->cache();
    return fixedDiv < 0 && fixedDen < 0 && !hostLoop && !c.isSlab && c.fused() && !c.brickWanted && maxIter >= 1;
}
bool DFSPHSolver::graphSafe() const { return (fixedDiv >= 0 && fixedDen >= 0) || deviceLoops(); }
void DFSPHSolver::fetchIterations()
{
    // (a replayed step graph updates the pinned words without passing through step(): while the device decides the loops the counts
    // are always taken from there)
    if (!itersPending) return;
    HIP_CALL(hipStreamSynchronize(sphx::stream()));
    lastDiv = hostIters[0]; lastDen = hostIters[1];
    if (hostIters[2] != 0) {       // k_dfsph_loop_tail: a grid barrier timed out -- the step that reported it is NOT valid
        hostIters[2] = 0;
        tailFailed = true;
        ++cache().generation;
        HIP_CALL(hipMemsetAsync(loopState.addr(kLoopFault), 0, sizeof(int), sphx::stream()));
        throw "DFSPHSolver: the loop tail's grid barrier timed out (its blocks were not resident at once). The step that reported it and the steps enqueued behind it did NOT move any particle (positions are those of the last valid step) but their velocities are only partly corrected: restore velocities from a checkpoint or accept them; the solver continues with gated launches";
    }
}

long long DFSPHSolver::readErrorTotalFixed()
{
    // kErrorSlots partial sums, one cache line apart (accumulate_error, sweep_ops.hpp): exact integers, any order
    static_assert(sizeof(unsigned long long) == 8, "64-bit accumulators");
    std::vector<unsigned long long> slots((size_t)kErrorSlots * kErrorSlotStride);
    HIP_CALL(hipMemcpyAsync(slots.data(), errorAccum.addr(), sizeof(unsigned long long) * slots.size(), hipMemcpyDeviceToHost, sphx::stream()));
    HIP_CALL(hipStreamSynchronize(sphx::stream()));
    unsigned long long acc = 0;
    for (int k = 0; k < kErrorSlots; ++k) acc += slots[(size_t)k * kErrorSlotStride];
    return (long long)acc;
}

void DFSPHSolver::resetErrorTotal()
{
    HIP_CALL(hipMemsetAsync(errorAccum.addr(), 0, sizeof(unsigned long long) * kErrorSlots * kErrorSlotStride, sphx::stream()));
}

void DFSPHSolver::permuteState(const int* perm, int n)
{
    BasicSPHSolver::permuteState(perm, n);
    for (DArray<float>* a : {&alpha, &bufferFloat, &error, &denWarmStiff}) {
        ew_gather_float(scratch.addr(), a->addr(), perm, n);
        ew_copy(a->addr(), scratch.addr(), sizeof(float) * (size_t)n);
    }
}

float DFSPHSolver::readErrorTotal()
{
    return (float)((double)readErrorTotalFixed() * (1.0 / 4294967296.0));
}

namespace {
template <bool DENSITY_MODE, int WARM>
void launch_rate(const OpRate& op, int n, bool reduce, bool keepAccum = false)
{
    if (n <= 0) return;
    OpRate o = op;
    if (!reduce) o.out.accum = nullptr;
    else if (!keepAccum) HIP_CALL(hipMemsetAsync(o.out.accum, 0, sizeof(unsigned long long) * kErrorSlots * kErrorSlotStride, sphx::stream()));
    launch_rate_kernel<DENSITY_MODE, WARM>(o, n);
}
}  // namespace

// computeDensityAlpha, DFSPHSolver.cu:251-259
void DFSPHSolver::computeDensityAlpha(std::shared_ptr<SPHParticles>& fluids, const std::shared_ptr<SPHParticles>& boundaries,
                                      const DArray<int>& cellStartFluid, const DArray<int>& cellStartBoundary,
                                      int3 cellSize, float cellLength, float radius)
{
    SweepCache& c = cache();
    c.setup(cellSize, cellLength, radius);
    c.packFluid(*fluids);
    c.packBoundary(*boundaries);
    c.ensureList(cellStartFluid, cellStartBoundary);
    const int num = (int)fluids->size();
    if (num <= 0) return;
    ScopedKernel t("density_alpha");
    OpDfsphHeadT<false> op{c.ctx(cellStartFluid, cellStartBoundary), nullptr, fluids->getDensityPtr(), alpha.addr(), RateOut{nullptr, nullptr, nullptr, nullptr, 0.0f, 0.0f, nullptr, 0, 0}};
    launch_dfsph_head<false>(op, num);
}

// correctDivergenceError, DFSPHSolver.cu:331-363.  `firstErrorDone`: the fused head sweep has
// already produced error/stiffness for the current velocities (DFSPHSolver.cu:341).
int DFSPHSolver::correctDivergenceError(std::shared_ptr<SPHParticles>& fluids, const std::shared_ptr<SPHParticles>& boundaries,
                                        const DArray<int>& cellStartFluid, const DArray<int>& cellStartBoundary, float rho0,
                                        int3 cellSize, float cellLength, float radius, float dt, float errorThreshold,
                                        int maxIterations)
{
    SweepCache& c = cache();
    c.setup(cellSize, cellLength, radius);
    c.packFluid(*fluids);
    c.packBoundary(*boundaries);
    c.ensureList(cellStartFluid, cellStartBoundary);
    const SweepCtx ctx = c.ctx(cellStartFluid, cellStartBoundary);
    const int num = (int)fluids->size();
    const bool adaptive = fixedDiv < 0;
    auto totalError = std::numeric_limits<float>::max();
    auto iter = 0;
    const OpRate rate{ctx, fluids->getVelPtr(), fluids->getDensityPtr(), alpha.addr(),
                      RateOut{error.addr(), bufferFloat.addr(), nullptr,
                              reinterpret_cast<unsigned long long*>(errorAccum.addr()), dt, rho0, c.posfw(), sumLo, sumHi}};
    const OpCorrect<false> correct{ctx, bufferFloat.addr(), fluids->getVelPtr(), dt, true};
    if (!headDidFirstError) {
        ScopedKernel t("divergence_error");
        launch_rate<false, 0>(rate, num, false);
    }
    headDidFirstError = false;
    while (adaptive ? ((iter < 1 || totalError > errorThreshold * num * rho0) && iter < maxIterations) : (iter < fixedDiv)) {
        {
            ScopedKernel t("divergence_correct");
            launch_op(correct, num);
        }
        if (adaptive || iter + 1 < fixedDiv) {      // (fixed counts: the sweep behind the last correction has no reader, see step())
            ScopedKernel t("divergence_error");
            launch_rate<false, 0>(rate, num, adaptive);
        }
        ++iter;
        if (adaptive) totalError = readErrorTotal();
    }
    return iter;
}

// DFSPHSolver::project, DFSPHSolver.cu:160-210
int DFSPHSolver::project(std::shared_ptr<SPHParticles>& fluids, const std::shared_ptr<SPHParticles>& boundaries,
                         const DArray<int>& cellStartFluid, const DArray<int>& cellStartBoundary, float rho0, int3 cellSize,
                         float cellLength, float radius, float dt, float errorThreshold, int maxIterations)
{
    SweepCache& c = cache();
    c.setup(cellSize, cellLength, radius);
    c.packFluid(*fluids);
    c.packBoundary(*boundaries);
    c.ensureList(cellStartFluid, cellStartBoundary);
    const SweepCtx ctx = c.ctx(cellStartFluid, cellStartBoundary);
    const int num = (int)fluids->size();
    const bool adaptive = fixedDen < 0;
    auto totalError = std::numeric_limits<float>::max();
    auto iter = 0;
    // carry last step's warm stiffness through this step's sort (DFSPHSolver.cu:170-171)
    {
        ScopedKernel t("warm_permute");
        ew_gather_float(scratch.addr(), denWarmStiff.addr(), fluids->getSortPerm(), num);
        ew_copy(denWarmStiff.addr(), scratch.addr(), sizeof(float) * num);
    }
    const OpRate rate{ctx, fluids->getVelPtr(), fluids->getDensityPtr(), alpha.addr(),
                      RateOut{error.addr(), bufferFloat.addr(), denWarmStiff.addr(),
                              reinterpret_cast<unsigned long long*>(errorAccum.addr()), dt, rho0, c.posfw(), sumLo, sumHi}};
    OpCorrect<true> correct{ctx, denWarmStiff.addr(), fluids->getVelPtr(), dt, false};   // warm start: posf.w holds kappa, not the warm array
    {
        ScopedKernel t("density_correct");   // warm start
        launch_op(correct, num);
    }
    {
        ScopedKernel t("density_error");     // also resets the warm stiffness to this stiffness
        launch_rate<true, 1>(rate, num, false);
    }
    correct.kappa = bufferFloat.addr();
    correct.packedScalar = true;
    while (adaptive ? ((iter < 2 || totalError > errorThreshold * num * rho0) && iter < maxIterations) : (iter < fixedDen)) {
        {
            ScopedKernel t("density_correct");
            launch_op(correct, num);
        }
        ++iter;
        const bool needTotal = adaptive && iter >= 2;
        {
            ScopedKernel t("density_error");   // accumulates the warm stiffness
            launch_rate<true, 2>(rate, num, needTotal);
        }
        if (needTotal) totalError = readErrorTotal();
    }
    return iter;
}

// DFSPHSolver::step, DFSPHSolver.cu:33-72
void DFSPHSolver::step(std::shared_ptr<SPHParticles>& fluids, const std::shared_ptr<SPHParticles>& boundaries,
                       const DArray<int>& cellStartFluid, const DArray<int>& cellStartBoundary, float3 spaceSize,
                       int3 cellSize, float cellLength, float radius, float dt, float rho0, float rhoB, float stiff,
                       float visc, float3 G, float surfaceTensionIntensity, float airPressure)
{
    (void)stiff;
    invalidatePositions();
    SweepCache& c = cache();
    const bool surface = surfaceTensionIntensity > EPSILON || airPressure > EPSILON;
    const int num = (int)fluids->size();
    if (!c.fused()) {
        itersPending = false;
        computeDensityAlpha(fluids, boundaries, cellStartFluid, cellStartBoundary, cellSize, cellLength, radius);
        lastDiv = correctDivergenceError(fluids, boundaries, cellStartFluid, cellStartBoundary, rho0, cellSize, cellLength,
                                         radius, dt, divergenceErrorThreshold, maxIter);
        force(fluids, dt, G);
        BasicSPHSolver::diffuse(fluids, cellStartFluid, cellSize, cellLength, rho0, radius, visc, dt);
        if (surface)
            handleSurface(fluids, boundaries, cellStartFluid, cellStartBoundary, rho0, rhoB, cellSize, cellLength, radius,
                          dt, surfaceTensionIntensity, airPressure);
        lastDen = project(fluids, boundaries, cellStartFluid, cellStartBoundary, rho0, cellSize, cellLength, radius, dt,
                          densityErrorThreshold, maxIter);
        advect(fluids, dt, spaceSize);
        return;
    }
    // fused schedule, stage by stage (the same stages a distributed driver runs through
    // SPHSystem::phase with halo refreshes in between):
    //   prepare -> [density+alpha+first divergence error] -> divergence loop -> gravity ->
    //   [viscosity+colour gradient] -> [surface (+ vel += deltaV)] -> density solve -> advect
    auto run = [&](int ph, bool reduce = false) {
        runPhase(ph, fluids, boundaries, cellStartFluid, cellStartBoundary, spaceSize, cellSize, cellLength, radius, dt, rho0,
                 rhoB, visc, G, surfaceTensionIntensity, airPressure, reduce);
    };
    const bool onDevice = deviceLoops();
    if (onDevice && tuning().dfsph_window >= 0) adaptWindows();
    // fixed counts with at least one divergence correction: the gravity kick rides in the last correction's store
    const bool kickFused = !onDevice && fixedDiv >= 1 && !c.isSlab && num > 0 && !tuning().no_kick_fusion;
    kickDv = make_float3(dt * G.x, dt * G.y, dt * G.z);
    unsigned long long* accum = reinterpret_cast<unsigned long long*>(errorAccum.addr());
    // `iterations` possible iterations of one loop, all enqueued: body(k) launches the sweeps of iteration k
    auto deviceLoop = [&](float threshold, int minIter, int which, auto&& body) {
        k_loop_reset<<<1, kErrorSlots, 0, sphx::stream()>>>(loopState.addr(), accum);
        c.gate = loopState.addr(kLoopDone);
        c.keepErrorAccum = true;                       // the decision kernel leaves the accumulators zeroed
        for (int k = 0; k < maxIter; ++k) {
            // The iterations the last steps needed (+ 2) are ordinary launches; whatever is left of the loop is ONE persistent launch
            // (where the configuration has one).  A loop that has terminated then costs a few launches that leave at once, not 3 per
            // possible iteration; one that runs longer than predicted is finished by the tail (slower per iteration: its grid barriers
            // cost more than kernel boundaries do -- profiles/r04_dfsph_loop_tail.txt -- so the window follows the counts, adaptWindows).
            if (k == std::max(minIter, which == kLoopDen ? windowDen : windowDiv) &&
                runLoopTail(which == kLoopDen, fluids, cellStartFluid, cellStartBoundary, dt, rho0, threshold, minIter, which)) break;
            body(k);
            k_loop_decide<<<1, kErrorSlots, 0, sphx::stream()>>>(loopState.addr(), accum, threshold, minIter, maxIter, which);
        }
        c.gate = nullptr;
        c.keepErrorAccum = false;
    };
    run(SPHX_PH_SEARCH);
    run(SPHX_PH_HEAD);
    if (onDevice) {
        deviceLoop(divergenceErrorThreshold * num * rho0, 1, kLoopDiv, [&](int) { run(SPHX_PH_DIV_CORRECT); run(SPHX_PH_DIV_ERROR, true); });
    } else {
        const bool adaptive = fixedDiv < 0;
        auto totalError = std::numeric_limits<float>::max();
        int iter = 0;
        while (adaptive ? ((iter < 1 || totalError > divergenceErrorThreshold * num * rho0) && iter < maxIter) : (iter < fixedDiv)) {
            kickInCorrect = kickFused && iter + 1 == fixedDiv;      // (no error sweep reads the velocities behind the last correction)
            run(SPHX_PH_DIV_CORRECT);
            kickInCorrect = false;
            // The error sweep behind a correction feeds the NEXT correction and the termination test (DFSPHSolver.cu:347-361).
            // With fixed counts nothing reads the one behind the last correction: error / stiffness (and posf.w) are rewritten by
            // the density solve before anybody looks at them, so every field of the finished step is unchanged without it.
            if (adaptive || iter + 1 < fixedDiv) run(SPHX_PH_DIV_ERROR, adaptive);
            ++iter;
            if (adaptive) totalError = readErrorTotal();
        }
        lastDiv = iter;
    }
    if (!kickFused) run(SPHX_PH_FORCE);
    run(SPHX_PH_VISC_COLOR);
    if (surface) {
        run(SPHX_PH_SURFACE_WARM);     // one row walk for the surface sweep and the warm-start correction
    } else {
        run(SPHX_PH_SURFACE);
        run(SPHX_PH_WARM_CORRECT);
    }
    run(SPHX_PH_DEN_ERROR_SET);
    if (onDevice) {
        deviceLoop(densityErrorThreshold * num * rho0, 2, kLoopDen, [&](int k) { run(SPHX_PH_DEN_CORRECT); run(SPHX_PH_DEN_ERROR_ACC, k + 1 >= 2); });
    } else {
        const bool adaptive = fixedDen < 0;
        auto totalError = std::numeric_limits<float>::max();
        int iter = 0;
        while (adaptive ? ((iter < 2 || totalError > densityErrorThreshold * num * rho0) && iter < maxIter) : (iter < fixedDen)) {
            run(SPHX_PH_DEN_CORRECT);
            ++iter;
            const bool needTotal = adaptive && iter >= 2;
            run(SPHX_PH_DEN_ERROR_ACC, needTotal);
            if (needTotal) totalError = readErrorTotal();
        }
        lastDen = iter;
    }
    // a step whose loop tail reported a fault (its grid barrier timed out: velocities only partly corrected) is NOT advected: positions,
    // cell indices and the exported arrays stay those of the last valid step until the host has reported it (fetchIterations)
    c.advectSkipIf = onDevice ? loopState.addr(kLoopFault) : nullptr;
    run(SPHX_PH_ADVECT);
    c.advectSkipIf = nullptr;
    if (onDevice) {          // the counts of this step: device words -> pinned host memory, read on demand (fetchIterations)
        HIP_CALL(hipMemcpyAsync(hostIters, loopState.addr(kLoopDiv), 3 * sizeof(int), hipMemcpyDeviceToHost, sphx::stream()));      // counts + the tail's fault word
    }
    itersPending = onDevice;
}

// A loop that outruns its window is finished by the tail at 2-3 times the cost per iteration (4 waves per SIMD, grid barriers):
// on a large scene that is milliseconds per step until the window has followed (1,022,208 particles, the impact: 25 steps at
// 6.2 ms instead of 3.5 with 16 steps between two looks, profiles/r04_dfsph_loop_tail.txt), so there the counts are looked at
// every 4 steps -- one stream synchronisation per 4 steps of >= 1 ms.  Small scenes keep the batches long.
int DFSPHSolver::tuneInterval() const
{
    return (deviceLoops() && alpha.length() >= 100000u) ? 4 : 16;
}
void DFSPHSolver::tune(int stepsSinceLastCall)
{
    BasicSPHSolver::tune(stepsSinceLastCall);
    if (deviceLoops()) { fetchIterations(); adaptWindows(); }
}
// grow at once (to twice the need, so that a rising count does not re-capture every few steps), shrink only when far too wide
void DFSPHSolver::adaptWindows()
{
    // sphx_tuning.dfsph_window = k: fixed windows (tests: 0 leaves every iteration beyond the reference's minimum to the tail kernel)
    if (tuning().dfsph_window >= 0) {
        const int w = tuning().dfsph_window;
        if (w != windowDiv || w != windowDen) { windowDiv = windowDen = w; ++cache().generation; }
        return;
    }
    auto adapt = [&](int& window, int seen, int minIter) {
        const int want = std::min(maxIter, std::max(minIter, seen) + 2);
        int next = window;
        if (seen + 1 > window) next = std::min(maxIter, std::max(want, 2 * window));
        else if (want + 6 < window) next = want;
        if (next == window) return;
        window = next;
        ++cache().generation;          // a captured step holds the old number of launches
    };
    adapt(windowDiv, lastDiv, 1);
    adapt(windowDen, lastDen, 2);
}

bool DFSPHSolver::runLoopTail(bool densityLoop, std::shared_ptr<SPHParticles>& fluids, const DArray<int>& cellStartFluid,
                              const DArray<int>& cellStartBoundary, float dt, float rho0, float threshold, int minIter, int which)
{
    if (tailFailed || tuning().dfsph_no_tail) return false;
    SweepCache& c = cache();
    const int num = (int)fluids->size();
    const SweepCtx ctx = c.ctx(cellStartFluid, cellStartBoundary);
    unsigned long long* accum = reinterpret_cast<unsigned long long*>(errorAccum.addr());
    const LoopTail tail{loopState.addr(), accum, threshold, minIter, maxIter, which, tuning().dfsph_tail_flat};
    ScopedKernel t(densityLoop ? "density_loop_tail" : "divergence_loop_tail");
    if (densityLoop)
        return launch_dfsph_loop_tail<true, 2>(OpCorrect<true>{ctx, bufferFloat.addr(), fluids->getVelPtr(), dt, true},
                                               OpRate{ctx, fluids->getVelPtr(), fluids->getDensityPtr(), alpha.addr(),
                                                      RateOut{error.addr(), bufferFloat.addr(), denWarmStiff.addr(), accum, dt, rho0, c.posfw(), sumLo, sumHi}},
                                               tail, num);
    return launch_dfsph_loop_tail<false, 0>(OpCorrect<false>{ctx, bufferFloat.addr(), fluids->getVelPtr(), dt, true},
                                            OpRate{ctx, fluids->getVelPtr(), fluids->getDensityPtr(), alpha.addr(),
                                                   RateOut{error.addr(), bufferFloat.addr(), nullptr, accum, dt, rho0, c.posfw(), sumLo, sumHi}},
                                            tail, num);
}

// One stage of the fused schedule.  Reference stages: DFSPHSolver.cu:33-72 (order), :160-210 and
// :331-363 (the two solver loops whose bodies are the CORRECT / ERROR stages).
void DFSPHSolver::runPhase(int phase, std::shared_ptr<SPHParticles>& fluids, const std::shared_ptr<SPHParticles>& boundaries,
                           const DArray<int>& cellStartFluid, const DArray<int>& cellStartBoundary, float3 spaceSize,
                           int3 cellSize, float cellLength, float radius, float dt, float rho0, float rhoB, float visc,
                           float3 G, float surfaceTensionIntensity, float airPressure, bool reduce)
{
    SweepCache& c = cache();
    const int num = (int)fluids->size();
    const bool surface = surfaceTensionIntensity > EPSILON || airPressure > EPSILON;
    if (phase == SPHX_PH_SEARCH) {
        warmInPosfAll = warmInPosfGhosts = false;
        invalidatePositions();
        c.setup(cellSize, cellLength, radius);
        c.packFluid(*fluids);
        c.packBoundary(*boundaries);
        c.ensureList(cellStartFluid, cellStartBoundary);
        ScopedKernel t("warm_permute");   // DFSPHSolver.cu:170-171
        if (warmSorted) { warmSorted = false; return; }      // (a staged slab sort delivered it in sorted order)
        if (c.persistRows) {              // the arrays were re-sorted in this step only if the rows are being rebuilt (device flag)
            ew_gather_float_if(scratch.addr(), denWarmStiff.addr(), fluids->getSortPerm(), num, c.persistFlags.addr(0));
            ew_copy_float_if(denWarmStiff.addr(), scratch.addr(), num, c.persistFlags.addr(0));
            return;
        }
        ew_gather_float(scratch.addr(), denWarmStiff.addr(), fluids->getSortPerm(), num);
        ew_copy(denWarmStiff.addr(), scratch.addr(), sizeof(float) * num);
        return;
    }
    if (phase == SPHX_PH_FORCE) { force(fluids, dt, G); return; }
    if (phase == SPHX_PH_ADVECT) { advect(fluids, dt, spaceSize); return; }
    c.setup(cellSize, cellLength, radius);
    c.packFluid(*fluids);
    c.packBoundary(*boundaries);
    c.ensureList(cellStartFluid, cellStartBoundary);
    const SweepCtx ctx = c.ctx(cellStartFluid, cellStartBoundary);
    unsigned long long* accum = reduce ? reinterpret_cast<unsigned long long*>(errorAccum.addr()) : nullptr;
    DArray<float3>& cg = colorGradientBuffer();
    switch (phase) {
    case SPHX_PH_HEAD: {
        ScopedKernel t("density_alpha_diverr");
        OpDfsphHeadT<true> op{ctx, fluids->getVelPtr(), fluids->getDensityPtr(), alpha.addr(),
                       RateOut{error.addr(), bufferFloat.addr(), nullptr, nullptr, dt, rho0, c.posfw(), sumLo, sumHi}};
        launch_dfsph_head<true>(op, num);
        break;
    }
    case SPHX_PH_DIV_CORRECT: {
        ScopedKernel t("divergence_correct");
        launch_op(OpCorrect<false>{ctx, bufferFloat.addr(), fluids->getVelPtr(), dt, true, kickInCorrect, kickDv.x, kickDv.y, kickDv.z}, num);
        break;
    }
    case SPHX_PH_DIV_ERROR: {
        ScopedKernel t("divergence_error");
        launch_rate<false, 0>(OpRate{ctx, fluids->getVelPtr(), fluids->getDensityPtr(), alpha.addr(),
                                     RateOut{error.addr(), bufferFloat.addr(), nullptr, accum, dt, rho0, c.posfw(), sumLo, sumHi}}, num, reduce,
                              c.keepErrorAccum);
        break;
    }
    case SPHX_PH_VISC_COLOR: {
        if (surface) {
            ScopedKernel t("visc_color");
            // the warm stiffness rides in posf.w for the fused surface + warm-start sweep that follows (slab drivers, which sweep their
            // owned particles only, add the ghosts' with packWarmIntoPosf)
            launch_op(OpFluidProps<true, true, false>{ctx, fluids->getVelPtr(), c.aux3.addr(), cg.addr(), nullptr, nullptr,
                                                      nullptr, rho0, rhoB, visc, dt, 0.0f, denWarmStiff.addr()}, num);
            if (c.rangeLo < 0) warmInPosfAll = true;
        } else {
            ScopedKernel t("viscosity");
            launch_op(OpFluidProps<true, false, false>{ctx, fluids->getVelPtr(), c.aux3.addr(), nullptr, nullptr, nullptr,
                                                       nullptr, rho0, rhoB, visc, dt, 0.0f}, num);
        }
        break;
    }
    case SPHX_PH_SURFACE: {
        if (surface) {
            ScopedKernel t("surface_tension");
            launch_op(OpSurface{ctx, cg.addr(), fluids->getVelPtr(), c.aux3.addr(), fluids->getVelPtr(), rho0,
                                surfaceTensionIntensity, airPressure, dt}, num);
        } else {
            ScopedKernel t("add_delta_v");
            const int lo = c.rangeLo >= 0 ? std::min(c.rangeLo, num) : 0, hi = c.rangeLo >= 0 ? std::min(c.rangeHi, num) : num;
            launch_add3(fluids->getVelPtr() + lo, c.vel4w() + lo, c.aux3.addr() + lo, hi - lo);
            if (c.rangeLo >= 0 && c.rangeLo2 >= 0) {       // the second range of a two-range stage
                const int lo2 = std::min(std::max(c.rangeLo2, hi), num), hi2 = std::min(std::max(c.rangeHi2, lo2), num);
                launch_add3(fluids->getVelPtr() + lo2, c.vel4w() + lo2, c.aux3.addr() + lo2, hi2 - lo2);
            }
        }
        break;
    }
    case SPHX_PH_SURFACE_WARM: {
        if (!surface) throw "DFSPHSolver::runPhase: the fused surface stage needs surface effects enabled";
        ScopedKernel t("surface_warm_correct");
        launch_op(OpSurfaceThen<1>{ctx, cg.addr(), fluids->getVelPtr(), c.aux3.addr(), fluids->getVelPtr(), denWarmStiff.addr(), rho0,
                                   surfaceTensionIntensity, airPressure, dt, !c.isSlab || warmInPosfAll || warmInPosfGhosts}, num);
        break;
    }
    case SPHX_PH_WARM_CORRECT: {
        ScopedKernel t("density_correct");
        launch_op(OpCorrect<true>{ctx, denWarmStiff.addr(), fluids->getVelPtr(), dt, false}, num);
        break;
    }
    case SPHX_PH_DEN_CORRECT: {
        ScopedKernel t("density_correct");
        launch_op(OpCorrect<true>{ctx, bufferFloat.addr(), fluids->getVelPtr(), dt, true}, num);
        break;
    }
    case SPHX_PH_DEN_ERROR_SET:
    case SPHX_PH_DEN_ERROR_ACC: {
        ScopedKernel t("density_error");
        const OpRate rate{ctx, fluids->getVelPtr(), fluids->getDensityPtr(), alpha.addr(),
                          RateOut{error.addr(), bufferFloat.addr(), denWarmStiff.addr(), accum, dt, rho0, c.posfw(), sumLo, sumHi}};
        if (phase == SPHX_PH_DEN_ERROR_SET) launch_rate<true, 1>(rate, num, reduce, c.keepErrorAccum);
        else launch_rate<true, 2>(rate, num, reduce, c.keepErrorAccum);
        break;
    }
    default: throw "DFSPHSolver::runPhase: unknown stage";
    }
}
