// sweep_ops.hpp — the per-particle neighbour-sweep operators of all three solvers, plus the small
// element-wise passes, as device functors launched one lane per fluid particle.
//
// Each operator cites the reference kernel(s) whose arithmetic it restates (association order kept,
// see sph_device.hpp).  Three kinds of rewrites are used, none of which changes a bit:
//   * hoisting: sub-expressions depending only on particle i leave the pair loop; sub-expressions
//     depending only on particle j are read from per-particle arrays (`pterm`);
//   * fusion: sweeps that read the same frozen inputs run as one walk with independent
//     accumulators (template flags select the reference's unfused building blocks);
//   * boundary neighbours go through the fluid formula with the absent field set to +0
//     (x - 0 == x, x + 0 == x), which is what the reference's separate boundary helpers compute.
#pragma once

#include "engine.hpp"

namespace sphx {

// ---- launch: 256-thread blocks, one 64-particle tile per wave, optional per-wave LDS slab ------------
template <class Op, bool STREAM>
struct BlockLds {
    float4 pos[STREAM ? kWideBlock / kTile : 1][STREAM ? kGroupSlots : 1];
    typename Op::Field field[STREAM ? kWideBlock / kTile : 1][STREAM ? kGroupSlots : 1];
};

#ifndef SPHX_MINWAVES
#define SPHX_MINWAVES 1
#endif
// which sweeps have a quad-per-particle variant switched on (SweepCtx::quad; SweepCache picks the mask)
enum QuadBits { kQuadRate = 1, kQuadHead = 2, kQuadProps = 4, kQuadCorrect = 8, kQuadPressure = 16, kQuadLambda = 32,
                kQuadDelta = 64, kQuadXsph = 128, kQuadSurface = 256 };
template <class Op> __host__ __device__ constexpr auto op_quad_bit_impl(int) -> decltype(Op::kQuadBit) { return Op::kQuadBit; }
template <class Op> __host__ __device__ constexpr int op_quad_bit_impl(long) { return 0; }
template <class Op> __host__ __device__ constexpr int op_quad_bit() { return op_quad_bit_impl<Op>(0); }

// MODE: 0 lane per particle, 1 quad per particle, 2 duo (two lanes per particle)
// TOLK (quad / duo launches): 0 strict arithmetic, 1 tolerance arithmetic -- the two walks live in separate kernels so that
// neither pays for the other's registers (sharing one kernel cost the strict rate sweep 40 bytes of scratch and 50 % of its
// speed when the tolerance walk changed, r03); -1: decided at run time from SweepCtx::k.tol (lane-per-particle launches)
template <int TOLK> __device__ __forceinline__ void assume_arith(const SweepCtx& c)
{
    if constexpr (TOLK == 0) __builtin_assume(c.k.tol == 0);
    if constexpr (TOLK == 1) __builtin_assume(c.k.tol != 0);
}
// (occupancy of the tolerance quad walks: experiments with -DSPHX_RUNOP_TOL_WAVES / -DSPHX_HEAD_TOL_WAVES)
#ifndef SPHX_RUNOP_TOL_WAVES
#define SPHX_RUNOP_TOL_WAVES SPHX_MINWAVES
#endif
#ifndef SPHX_HEAD_TOL_WAVES
#define SPHX_HEAD_TOL_WAVES SPHX_MINWAVES
#endif
template <class Op, bool STREAM, int MODE = 0, int TOLK = -1>
__global__ void __launch_bounds__(kWideBlock, (MODE == 1 && TOLK == 1) ? SPHX_RUNOP_TOL_WAVES : SPHX_MINWAVES) k_run_op(const Op op, int n)
{
    (void)n;
    assume_arith<TOLK>(op.c);
    if constexpr (MODE != 0) {
        const int i = MODE == 1 ? quad_particle(op.c) : duo_particle(op.c);
        if (i < 0) return;
        op.template operator()<MODE>(i, in_range(op.c, i), nullptr, nullptr);
    } else {
        __shared__ BlockLds<Op, STREAM> lds;
        const int wave = threadIdx.x >> 6;
        const int tile = wave_tile(op.c);
        if (tile < 0) return;                  // whole wave past the end (wave-uniform)
        const int i = tile * kTile + (int)(threadIdx.x & 63);
        op(i, in_range(op.c, i), STREAM ? lds.pos[wave] : nullptr, STREAM ? lds.field[wave] : nullptr);
    }
}
// the sweep of one particle in either launch shape; in a quad launch only lane 0 of the quad stores the results
template <int MODE, bool WANT_BOUNDARY, class Op, class Body>
__device__ __forceinline__ void sweep_any(const Op& op, const SweepCtx& c, float4* lp, typename Op::Field* lf, int i, bool valid,
                                          float3 pi, Body& body)
{
    if constexpr (MODE == 1) sweep_quad<WANT_BOUNDARY>(op, c, i, valid, pi, body);
    else if constexpr (MODE == 2) sweep_duo<WANT_BOUNDARY>(op, c, i, valid, pi, body);
    else if constexpr (MODE == 3) sweep_brick<WANT_BOUNDARY>(op, c, lp, lf, i, valid, pi, body);      // rows of LDS slots (compact brick)
    else sweep<WANT_BOUNDARY>(op, c, lp, lf, i, valid, pi, body);
}
// quad: every lane holds the sums, lane 0 stores; duo: the sums end in lane 1 of the pair
template <int MODE> __device__ __forceinline__ bool stores_results(bool valid)
{
    return valid && (MODE == 0 || MODE == 3 || (MODE == 1 && (threadIdx.x & 3) == 0) || (MODE == 2 && (threadIdx.x & 1) == 1));
}
// grid of a sweep launch: one wave per tile of the launch's range, padded to a multiple of 8 blocks
inline unsigned int sweep_grid(const SweepCtx& c) { return xcd_grid(c.numTiles * kTile, kWideBlock); }
inline unsigned int quad_grid(const SweepCtx& c) { return xcd_grid(c.numTiles * kWideBlock, kWideBlock); }   // one block per tile
inline unsigned int duo_grid(const SweepCtx& c) { return xcd_grid(((c.numTiles + 1) / 2) * kWideBlock, kWideBlock); }   // one block per two tiles
// ---- compact-brick launches (sph_device.hpp "Compact-brick LDS stage"): one block per brick, all sweeps of the tolerance path ----
// The block's dynamic LDS: [kBrickSlots + 1 positions | kBrickSlots + 1 field records].
template <class Field> inline size_t brick_lds_bytes() { return (size_t)(kBrickSlots + 1) * (sizeof(float4) + sizeof(Field)); }
// Runs `each(i, valid, lp, lf)` for every own particle of this block's brick, slice by slice; all threads of the block take part in
// every round (valid = false past the end), so `each` may use wave-wide operations.  Returns false when the brick cannot be staged.
template <class Field, class Stage, class Each>
__device__ __forceinline__ void brick_run(const SweepCtx& c, Stage&& stage, Each&& each)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char brickLds[];
    __shared__ __attribute__((aligned(16))) BrickTables T;
    float4* lp = reinterpret_cast<float4*>(brickLds);
    Field* lf = reinterpret_cast<Field*>(lp + kBrickSlots + 1);
    brick_for_each_slice(c, T, [&](BrickTables& tab) {
        brick_stage(c, tab, lp, lf, stage);
        for (int p0 = 0; p0 < tab.own; p0 += kBrickThreads) {
            const int p = p0 + (int)threadIdx.x;
            const bool valid = p < tab.own;
            each(valid ? brick_own_index(tab, p) : 0, valid, lp, lf);
        }
        __syncthreads();                                                // the stage and the tables are reused
    });
}
template <class Op>
__global__ void __launch_bounds__(kBrickThreads, 4) k_brick_op(const Op op, int n)
{
    (void)n;
    assume_arith<1>(op.c);
    using BF = brick_field_t<Op>;
    brick_run<BF>(op.c, [&](bool isB, int u) { return brick_pack(op, op.stage(isB, u)); },
                  [&](int i, bool valid, float4* lp, BF* lf) { op.template operator()<3>(i, valid, lp, reinterpret_cast<typename Op::Field*>(lf)); });
}
template <class Kernel>
inline void brick_launch_prepare(Kernel kernel, size_t ldsBytes)
{
    // more than 64 KB of dynamic LDS must be asked for once per kernel
    static thread_local const void* seen[64]; static thread_local int nSeen = 0;
    const void* fn = reinterpret_cast<const void*>(kernel);
    for (int k = 0; k < nSeen; ++k) if (seen[k] == fn) return;
    HIP_CALL(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsBytes));
    if (nSeen < 64) seen[nSeen++] = fn;
}
inline unsigned int brick_grid(const SweepCtx& c) { return xcd_grid(c.brickBlocks * kBrickThreads, kBrickThreads); }

template <class Op>
inline void launch_op(const Op& op, int n)
{
    if (n <= 0 || op.c.numTiles <= 0) return;
    if (op.c.brick) {
        const size_t lds = brick_lds_bytes<brick_field_t<Op>>();
        brick_launch_prepare(k_brick_op<Op>, lds);
        k_brick_op<Op><<<brick_grid(op.c), kBrickThreads, lds, stream()>>>(op, n);
        return;
    }
    if (op.c.nbr && op.c.tileFmt) k_run_op<Op, true><<<sweep_grid(op.c), kWideBlock, 0, stream()>>>(op, n);
    else if constexpr (op_quad_bit<Op>() != 0) {
        if (op.c.nbr && (op.c.quad & op_quad_bit<Op>())) {
            if (op.c.k.tol) k_run_op<Op, false, 1, 1><<<quad_grid(op.c), kWideBlock, 0, stream()>>>(op, n);
            else k_run_op<Op, false, 1, 0><<<quad_grid(op.c), kWideBlock, 0, stream()>>>(op, n);
        }
        else if (op.c.nbr && (op.c.duo & op_quad_bit<Op>())) k_run_op<Op, false, 2><<<duo_grid(op.c), kWideBlock, 0, stream()>>>(op, n);
        else k_run_op<Op, false><<<sweep_grid(op.c), kWideBlock, 0, stream()>>>(op, n);
    } else k_run_op<Op, false><<<sweep_grid(op.c), kWideBlock, 0, stream()>>>(op, n);
}

// wave-level sum of the fixed-point |error| terms, one atomic per wave.  The total is spread over
// kErrorSlots accumulators, one 128-byte line each (the host adds them up: integer sums, any order): atomics on ONE
// address serialise in the L2 — 643 k waves x ~6 ns made a post-impact rate sweep 4.7 ms instead of 0.9.
constexpr int kErrorSlots = 256;
constexpr int kErrorSlotStride = 16;       // in 64-bit words
__device__ __forceinline__ void accumulate_error(long long fixed, unsigned long long* accum)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) fixed += __shfl_down(fixed, off, 64);
    if ((threadIdx.x & 63) == 0 && fixed != 0) {
        const unsigned int slot = (blockIdx.x * (kWideBlock / kTile) + (threadIdx.x >> 6)) & (kErrorSlots - 1);
        atomicAdd(accum + (size_t)slot * kErrorSlotStride, (unsigned long long)fixed);
    }
}

__device__ __forceinline__ float4 f4(const float3 v) { return make_float4(v.x, v.y, v.z, 0.0f); }
__device__ __forceinline__ float4 f4zero() { return make_float4(0.0f, 0.0f, 0.0f, 0.0f); }
// stage(isBoundary, j): j indexes the unified neighbour space [fluid | boundary].
// float4 fields (vel4, cg4) have a +0 boundary tail, so their load needs no test at all; scalar
// fields live in solver arrays of fluid length: boundary entries read slot 0 and discard it.  Either
// way the load is unconditional — a guarded load compiles to an exec-mask branch with
// `s_waitcnt vmcnt(0)` inside, which serialises the kAhead gathers a lane has in flight.
__device__ __forceinline__ float4 field4(const float4* __restrict__ a, int j) { return gather16(a, (unsigned int)j << 4); }
__device__ __forceinline__ float fluid_only(const float* __restrict__ a, bool isB, int j)
{
    const float v = a[isB ? 0 : j];
    return isB ? 0.0f : v;
}
// the lane's own position (lanes past the end of the array still take part in wave-wide staging)
__device__ __forceinline__ float3 own_pos(const SweepCtx& c, int i, bool valid) { return valid ? xyz(c.posm[i]) : v3(0, 0, 0); }

// =================================================================================== shared sweeps
// Per-particle properties that depend on positions (and the current velocities) only:
//   VISC    viscosity_CUDA, BasicSPHSolver.cu:183-209            -> deltaV
//   COLOR   computeColorGrad_CUDA, BasicSPHSolver.cu:277-318     -> colorGrad
//   DENS    computeDensity_CUDA + computePressure_CUDA, :32-83, :103-111 -> density, pressure, pterm
template <bool VISC, bool COLOR, bool DENS>
struct OpFluidProps {
    SweepCtx c;
    const float3* vel; float3* deltaV; float3* colorGrad;
    float* density; float* pressure; float* pterm;
    float rho0, rhoB, visc, dt, stiff;
    const float* nextScalar = nullptr;   // when set: copied into posf.w, so that the NEXT sweep reads it with the position
    using Field = float4;   // neighbour velocity
    __device__ __forceinline__ Field stage(bool, int j) const { return VISC ? field4(c.vel4, j) : f4zero(); }
    struct Body {
        const OpFluidProps& o; float3 vi; float3 a; float3 cg; float cden; float den;
        float mRef, volRef;       // this particle's mass and mRef / rho0: the volume of every fluid neighbour of that mass
        template <bool FAST>
        __device__ __forceinline__ void pair(Field vj, bool isB, float3 d, float r2, float mj, int)
        {
            const float r = sqrt_sel<FAST>(r2);
            if (VISC && !isB) {
                const float3 dv = sub3(xyz(vj), vi);
                // x / 1.0f is x: the reference scene's rho0 = 1 needs no division (launch-uniform test)
                a = add3(a, mul3s(smul3(mj, (o.rho0 == 1.0f) ? dv : div3s(dv, o.rho0)), kViscLap<FAST>(r, o.c.k)));
            }
            if (COLOR || DENS) {
                const float q = q_of<FAST>(r, o.c.k);
                const float w = kW<FAST>(q, o.c.k);
                if (DENS) den += mj * w;
                if (COLOR) {
                    // same operands, same quotient: only neighbours of another mass (boundaries) divide
                    float vol = volRef;
                    if (__any(isB || mj != mRef)) vol = mj / (isB ? o.rhoB : o.rho0);
                    cg = add3(cg, smul3(vol, kGradW<FAST>(d, q, o.c.k)));
                    cden += vol * w;
                }
            }
        }
        __device__ __forceinline__ void pair_tol(Field vj, bool isB, float3 d, float r2, float mj)
        {
            const TolPair t = tol_pair(r2, o.c.k);
            if (VISC && !isB) {
                const float s = mj * __builtin_amdgcn_rcpf(o.rho0) * tol_viscLap(t, o.c.k);
                a = tol_axpy(a, v3(vj.x - vi.x, vj.y - vi.y, vj.z - vi.z), s);
            }
            if (COLOR || DENS) {
                const float w = tol_W(t, o.c.k);
                if (DENS) den = fmaf(mj, w, den);
                if (COLOR) {
                    const float vol = mj * __builtin_amdgcn_rcpf(isB ? o.rhoB : o.rho0);
                    const float s = vol * tol_gradW_scale(t, o.c.k);
                    cg = tol_axpy(cg, d, s);
                    cden = fmaf(vol, w, cden);
                }
            }
        }
        template <class F> __device__ __forceinline__ void each_acc(Body& other, F f)
        {
            if (VISC) { f(a.x, other.a.x); f(a.y, other.a.y); f(a.z, other.a.z); }
            if (COLOR) { f(cg.x, other.cg.x); f(cg.y, other.cg.y); f(cg.z, other.cg.z); f(cden, other.cden); }
            if (DENS) f(den, other.den);
        }
    };
    static constexpr int kQuadBit = kQuadProps;
    template <int QUAD = 0>
    __device__ void operator()(int i, bool valid, float4* lp, Field* lf) const
    {
        const float mRef = valid ? c.posm[i].w : 0.0f;
        Body b{*this, (VISC && valid) ? vel[i] : v3(0, 0, 0), v3(0, 0, 0), v3(0, 0, 0), 0.0f, 0.0f, mRef, mRef / rho0};
        sweep_any<QUAD, COLOR || DENS>(*this, c, lp, lf, i, valid, own_pos(c, i, valid), b);
        if (!stores_results<QUAD>(valid)) return;
        if (VISC) deltaV[i] = mul3s(smul3(visc, b.a), dt);
        if (COLOR) { const float3 g = div3s(b.cg, max_eps(b.cden)); colorGrad[i] = g; c.cg4[i] = f4(g); }
        if (nextScalar) c.posf[i].w = nextScalar[i];
        if (DENS) {
            density[i] = b.den;
            float p = stiff * (pow7(b.den / rho0) - 1.0f);
            if (p < 0.0f) p = 0.0f;
            pressure[i] = p;
            const float pt = p / max_eps(b.den * b.den);
            pterm[i] = pt;
            c.posf[i].w = pt;
        }
    }
};

// Per-particle constants of the surface sweep's pair term (BasicSPHSolver.cu:350-362).  The two coefficients depend on
// the neighbour only through its mass: for a neighbour of this particle's own mass they are computed once (same
// operands, same association, same bits).  The three divisions by ml = max(EPS, |colorGrad_i|) share one denominator:
// the refined-reciprocal division of sph_device.hpp gives the IEEE quotients when ml is in [2^-90, 2^16] and no
// numerator component is a non-zero value below 2^-101 (tested per pair; otherwise the plain operator).
struct SurfaceConsts {
    float dii, li, ml, mRef, tensionRef, airRef; bool mlFast;
    __device__ __forceinline__ float tension_coef(float mj, float rho0, float tension) const
    {
        float c = tensionRef;
        if (__any(mj != mRef)) c = 0.25f * mj / (rho0 * rho0) * tension;
        return c;
    }
    __device__ __forceinline__ float air_coef(float mj, float rho0, float airPressure) const
    {
        float c = airRef;
        if (__any(mj != mRef)) c = airPressure * mj / (rho0 * rho0);
        return c;
    }
    template <bool FAST>
    __device__ __forceinline__ float3 over_ml(float3 n) const
    {
        if (!FAST || __any(!mlFast || pair_needs_plain_ops(n, 1.0f))) return div3s(n, ml);
        return div3_sel<true>(n, ml);
    }
};
__device__ __forceinline__ SurfaceConsts surface_consts(float3 cgi, float mRef, float rho0, float tension, float airPressure)
{
    SurfaceConsts s;
    s.li = len3(cgi); s.dii = dot3(cgi, cgi); s.ml = max_eps(s.li); s.mRef = mRef;
    s.tensionRef = 0.25f * mRef / (rho0 * rho0) * tension;
    s.airRef = airPressure * mRef / (rho0 * rho0);
    s.mlFast = s.ml >= 8.0779357e-28f /* 2^-90 */ && s.ml <= 65536.0f;
    return s;
}

// surfaceTensionAndAirPressure_CUDA, BasicSPHSolver.cu:332-370.  `velIn` is the velocity the
// reference kernel would read for particle i (it only reads its own); when `addend` is given the
// pending element-wise update vel += addend (BasicSPHSolver.cu:219-224) is applied first.
struct OpSurface {
    SweepCtx c;
    const float3* colorGrad; const float3* velIn; const float3* addend; float3* velOut;
    float rho0, tension, airPressure, dt;
    using Field = float4;   // neighbour colour gradient
    __device__ __forceinline__ Field stage(bool, int j) const { return field4(c.cg4, j); }
    struct Body {
        const OpSurface& o; SurfaceConsts s; float dii, li, ml; float3 a;
        template <bool FAST>
        __device__ __forceinline__ void pair(Field cg4, bool, float3 d, float r2, float mj, int)
        {
            const float r = sqrt_sel<FAST>(r2);
            const float q = q_of<FAST>(r, o.c.k);
            const float3 cgj = xyz(cg4);
            a = add3(a, smul3(s.tension_coef(mj, o.rho0, o.tension) * (dii + dot3(cgj, cgj)), kSurfGrad<FAST>(d, r, o.c.k)));
            a = add3(a, s.over_ml<FAST>(mul3s(smul3(s.air_coef(mj, o.rho0, o.airPressure), kGradW<FAST>(d, q, o.c.k)), li)));
        }
        __device__ __forceinline__ void pair_tol(Field cg4, bool, float3 d, float r2, float mj)
        {
            const TolPair t = tol_pair(r2, o.c.k);
            const float m2 = mj * __builtin_amdgcn_rcpf(o.rho0 * o.rho0);
            const float air = o.airPressure * m2 * tol_gradW_scale(t, o.c.k) * li * __builtin_amdgcn_rcpf(ml);
            const float s = fmaf(0.25f * m2 * o.tension * (dii + tol_dot(cg4.x, cg4.y, cg4.z, v3(cg4.x, cg4.y, cg4.z))), tol_surf_scale(t, o.c.k), air);
            a = tol_axpy(a, d, s);
        }
        template <class F> __device__ __forceinline__ void each_acc(Body& other, F f) { f(a.x, other.a.x); f(a.y, other.a.y); f(a.z, other.a.z); }
    };
    // (no quad variant: strict arithmetic adds two terms per entry to one accumulator, which the ordered one-term-per-lane
    // accumulation cannot reproduce; under the tolerance arithmetic the quad kernel needs ~100 VGPRs + scratch and measured 5x slower)
    template <int MODE = 0>         // 0 lane-per-particle over global rows, 3 compact brick
    __device__ void operator()(int i, bool valid, float4* lp, Field* lf) const
    {
        const float3 cgi = valid ? colorGrad[i] : v3(0, 0, 0);
        const SurfaceConsts sc = surface_consts(cgi, valid ? c.posm[i].w : 0.0f, rho0, tension, airPressure);
        Body b{*this, sc, sc.dii, sc.li, sc.ml, v3(0, 0, 0)};
        sweep_any<MODE, false>(*this, c, lp, lf, i, valid, own_pos(c, i, valid), b);
        if (!valid) return;
        float3 v = velIn[i];
        if (addend) v = add3(v, addend[i]);
        const float3 vn = add3(v, mul3s(b.a, dt));
        velOut[i] = vn;
        c.vel4[i] = f4(vn);
    }
};

// The surface sweep fused with the sweep that follows it in the step when that one reads nothing the surface sweep
// writes for the NEIGHBOURS (both only update the particle's own velocity):
//   NEXT = 1  DFSPH: + the warm-start density correction (correctDensityError_CUDA with last step's stiffness,
//             DFSPHSolver.cu:138-158, :181-183)
//   NEXT = 2  WCSPH: + the pressure force (pressureForce_CUDA, BasicSPHSolver.cu:113-165)
// One row walk instead of two, and gradW of the pair is evaluated once for both.  The velocity is updated in the
// order of the separate sweeps (surface first), each with its own rounding, so the result has the same bits.
template <int NEXT>
struct OpSurfaceThen {
    SweepCtx c;
    const float3* colorGrad; const float3* velIn; const float3* addend; float3* velOut;
    const float* scalar;    // NEXT 1: warm stiffness kappa; NEXT 2: pressure term p / max(EPS, rho^2)
    float rho0, tension, airPressure, dt;
    bool packedScalar;      // posf.w holds `scalar` for every fluid particle: two gathers per pair instead of three
    struct Field { float4 cg; float s; };
    __device__ __forceinline__ Field stage(bool isB, int j) const { return Field{field4(c.cg4, j), fluid_only(scalar, isB, j)}; }
    __device__ __forceinline__ Field stage_packed(bool, int j, float s) const { return Field{field4(c.cg4, j), s}; }
    struct Body {
        const OpSurfaceThen& o; SurfaceConsts s; float dii, li, ml, si; float3 a; float3 b;
        template <bool FAST>
        __device__ __forceinline__ void pair(Field f, bool isB, float3 d, float r2, float mj, int)
        {
            const float r = sqrt_sel<FAST>(r2);
            const float q = q_of<FAST>(r, o.c.k);
            const float3 gw = kGradW<FAST>(d, q, o.c.k);
            if (!isB) {           // the surface sweep ignores boundary particles (BasicSPHSolver.cu:350-362)
                const float3 cgj = xyz(f.cg);
                a = add3(a, smul3(s.tension_coef(mj, o.rho0, o.tension) * (dii + dot3(cgj, cgj)), kSurfGrad<FAST>(d, r, o.c.k)));
                a = add3(a, s.template over_ml<FAST>(mul3s(smul3(s.air_coef(mj, o.rho0, o.airPressure), gw), li)));
            }
            if (NEXT == 1) b = add3(b, smul3(mj * (si + f.s), gw));
            else b = add3(b, smul3(-mj * (si + f.s), gw));
        }
        __device__ __forceinline__ void pair_tol(Field f, bool isB, float3 d, float r2, float mj)
        {
            const TolPair t = tol_pair(r2, o.c.k);
            const float g = tol_gradW_scale(t, o.c.k);
            if (!isB) {
                const float m2 = mj * __builtin_amdgcn_rcpf(o.rho0 * o.rho0);
                const float air = o.airPressure * m2 * g * li * __builtin_amdgcn_rcpf(ml);
                const float s = fmaf(0.25f * m2 * o.tension * (dii + tol_dot(f.cg.x, f.cg.y, f.cg.z, v3(f.cg.x, f.cg.y, f.cg.z))), tol_surf_scale(t, o.c.k), air);
                a = tol_axpy(a, d, s);
            }
            const float sb = (NEXT == 1 ? mj : -mj) * (si + f.s) * g;
            b = tol_axpy(b, d, sb);
        }
        template <class F> __device__ __forceinline__ void each_acc(Body& other, F f)
        {
            f(a.x, other.a.x); f(a.y, other.a.y); f(a.z, other.a.z); f(b.x, other.b.x); f(b.y, other.b.y); f(b.z, other.b.z);
        }
    };
    // the brick stage keeps (colour gradient, scalar) in ONE 16-byte record
    using BrickField = float4;
    __device__ __forceinline__ BrickField brick_pack(const Field& f) const { return make_float4(f.cg.x, f.cg.y, f.cg.z, f.s); }
    __device__ __forceinline__ Field brick_unpack(const BrickField& t) const { return Field{make_float4(t.x, t.y, t.z, 0.0f), t.w}; }
    template <int MODE = 0>         // 0 lane-per-particle over global rows (no quad variant, see OpSurface), 3 compact brick
    __device__ void operator()(int i, bool valid, float4* lp, Field* lf) const
    {
        const float3 cgi = valid ? colorGrad[i] : v3(0, 0, 0);
        const SurfaceConsts sc = surface_consts(cgi, valid ? c.posm[i].w : 0.0f, rho0, tension, airPressure);
        Body body{*this, sc, sc.dii, sc.li, sc.ml, valid ? scalar[i] : 0.0f, v3(0, 0, 0), v3(0, 0, 0)};
        sweep_any<MODE, true>(*this, c, lp, lf, i, valid, own_pos(c, i, valid), body);
        if (!valid) return;
        float3 v = velIn[i];
        if (addend) v = add3(v, addend[i]);
        const float3 vs = add3(v, mul3s(body.a, dt));                 // what the surface sweep alone would have stored
        float3 vn;
        if (NEXT == 1) {
            vn = add3(vs, div3s(body.b, dt));                         // OpCorrect<true>
        } else {
            float3 acc = body.b;                                      // OpPressureForce
            if (len3(acc) > kMaxA) acc = mul3s(mul3s(acc, 1.0f / sqrtf(dot3(acc, acc))), kMaxA);
            vn = add3(vs, mul3s(acc, dt));
        }
        velOut[i] = vn;
        c.vel4[i] = f4(vn);
    }
};

// pressureForce_CUDA, BasicSPHSolver.cu:113-165
struct OpPressureForce {
    SweepCtx c;
    const float* pterm; float3* vel;
    float dt;
    bool packedScalar;      // posf.w holds pterm
    using Field = float;    // neighbour p_j / max(EPS, rho_j^2)
    __device__ __forceinline__ Field stage(bool isB, int j) const { return fluid_only(pterm, isB, j); }
    struct Body {
        const OpPressureForce& o; int i; float pti; float3 a;
        template <bool FAST>
        __device__ __forceinline__ void pair(Field ptj, bool isB, float3 d, float r2, float mj, int idx)
        {
            if (!isB && idx == i) return;
            a = add3(a, smul3(-mj * (pti + ptj), kGradW<FAST>(d, q_of<FAST>(sqrt_sel<FAST>(r2), o.c.k), o.c.k)));
        }
        __device__ __forceinline__ void pair_tol(Field ptj, bool, float3 d, float r2, float mj)      // rows never hold the particle itself
        {
            const float s = -mj * (pti + ptj) * tol_gradW_scale(tol_pair(r2, o.c.k), o.c.k);
            a = tol_axpy(a, d, s);
        }
        static constexpr bool kPair2 = true;      // (rows never hold the particle itself, so no j == i test)
        __device__ __forceinline__ void pair2(Body& A, Body& B, Field ta, Field tb, bool, bool, float3 pi, float4 pa, float4 pb) const
        {
            const Pair2 p = pair2_geometry(pi, pa, pb, o.c.k);
            const f2x3 g = kGradW_fast2(p.d, p.q, o.c.k);
            const f2 s = -f2{pa.w, pb.w} * (pti + f2{ta, tb});
            const f2 cx = s * g.x, cy = s * g.y, cz = s * g.z;
            A.a = add3(A.a, v3(cx.x, cy.x, cz.x));
            B.a = add3(B.a, v3(cx.y, cy.y, cz.y));
        }
        template <class F> __device__ __forceinline__ void each_acc(Body& other, F f) { f(a.x, other.a.x); f(a.y, other.a.y); f(a.z, other.a.z); }
    };
    static constexpr int kQuadBit = kQuadPressure;
    template <int QUAD = 0>
    __device__ void operator()(int i, bool valid, float4* lp, Field* lf) const
    {
        Body b{*this, i, valid ? pterm[i] : 0.0f, v3(0, 0, 0)};
        sweep_any<QUAD, true>(*this, c, lp, lf, i, valid, own_pos(c, i, valid), b);
        if (!stores_results<QUAD>(valid)) return;
        float3 a = b.a;
        if (len3(a) > kMaxA) a = mul3s(mul3s(a, 1.0f / sqrtf(dot3(a, a))), kMaxA);
        vel[i] = add3(vel[i], mul3s(a, dt));
    }
};

// =================================================================================== DFSPH
// divergence / density error epilogue shared by the fused head and the stand-alone rate sweep:
// computeDivergenceError_CUDA (DFSPHSolver.cu:296-304) and computeDensityError_CUDA (:111-114),
// with the warm-stiffness bookkeeping of DFSPHSolver.cu:185,199-203 (WARM 0 none, 1 set, 2 add).
struct RateOut {
    float* error; float* kappa; float* warm; unsigned long long* accum;
    float dt, rho0;
    float4* posf;     // (x,y,z,kappa) records for the one-gather correction sweeps (may be nullptr)
    int sumLo, sumHi; // particles whose |error| enters the accumulated total (slab: the owned range)
};
template <bool DENSITY_MODE, int WARM>
__device__ __forceinline__ long long finish_rate(const RateOut& r, int i, float e, float den, float alpha)
{
    float err;
    if (DENSITY_MODE) {
        err = max0(r.dt * e + den - r.rho0);
    } else {
        err = max0(e);
        if (den + r.dt * err < r.rho0 && den <= r.rho0) err = 0.0f;
    }
    const float kap = err * alpha;
    r.error[i] = err;
    r.kappa[i] = kap;
    if (r.posf) r.posf[i].w = kap;
    if (WARM == 1) r.warm[i] = kap;
    if (WARM == 2) r.warm[i] = r.warm[i] + kap;
    return (i >= r.sumLo && i < r.sumHi) ? error_fixed(err) : 0;
}

// computeDensityAlpha_CUDA (DFSPHSolver.cu:212-249), optionally fused with the first
// computeDivergenceError_CUDA (:261-306), whose inputs (density_i, alpha_i) are this lane's own.
template <bool RATE>     // RATE: fused with the first divergence error (the velocity gather exists only then)
struct OpDfsphHeadT {
    SweepCtx c;
    const float3* vel; float* density; float* alpha;
    RateOut out;
    using Field = float4;   // neighbour velocity
    __device__ __forceinline__ Field stage(bool, int j) const { return RATE ? field4(c.vel4, j) : f4zero(); }
    struct Body {
        const OpDfsphHeadT& o; float vix, viy, viz;   // own velocity as scalars: a float3 member keeps the whole struct in memory
        float den, sl, e; float3 gs;
        static constexpr bool withRate = RATE;
        template <bool FAST>
        __device__ __forceinline__ void pair(Field vj, bool isB, float3 d, float r2, float mj, int)
        {
            const float q = q_of<FAST>(sqrt_sel<FAST>(r2), o.c.k);
            den += mj * kW<FAST>(q, o.c.k);
            const float3 gw = kGradW<FAST>(d, q, o.c.k);
            const float3 gr = smul3(mj, gw);
            gs = add3(gs, gr);
            if (!isB) sl += dot3(gr, gr);
            if (withRate) e += mj * dot3(sub3(v3(vix, viy, viz), xyz(vj)), gw);
        }
        __device__ __forceinline__ void pair_tol(Field vj, bool isB, float3 d, float r2, float mj)
        {
            const TolPair t = tol_pair(r2, o.c.k);
            den = fmaf(mj, tol_W(t, o.c.k), den);
            const float g = tol_gradW_scale(t, o.c.k), s = mj * g;
            const float3 gr = v3(d.x * s, d.y * s, d.z * s);
            gs = v3(gs.x + gr.x, gs.y + gr.y, gs.z + gr.z);
            if (!isB) sl += tol_dot(gr.x, gr.y, gr.z, gr);
            if (withRate) e = fmaf(s, tol_dot(vix - vj.x, viy - vj.y, viz - vj.z, d), e);
        }
        static constexpr bool kPair2 = true;
        __device__ __forceinline__ void pair2(Body& A, Body& B, Field va, Field vb, bool isBa, bool isBb, float3 pi, float4 pa, float4 pb) const
        {
            const Pair2 p = pair2_geometry(pi, pa, pb, o.c.k);
            const f2 m = f2{pa.w, pb.w};
            const f2 dw = m * kW_fast2(p.q, o.c.k);
            const f2x3 gw = kGradW_fast2(p.d, p.q, o.c.k);
            const f2 gx = m * gw.x, gy = m * gw.y, gz = m * gw.z;
            const f2 s2 = gx * gx + gy * gy + gz * gz;
            f2 r = splat2(0.0f);
            if (withRate) r = m * ((vix - f2{va.x, vb.x}) * gw.x + (viy - f2{va.y, vb.y}) * gw.y + (viz - f2{va.z, vb.z}) * gw.z);
            A.den += dw.x; A.gs = add3(A.gs, v3(gx.x, gy.x, gz.x)); if (!isBa) A.sl += s2.x; if (withRate) A.e += r.x;
            B.den += dw.y; B.gs = add3(B.gs, v3(gx.y, gy.y, gz.y)); if (!isBb) B.sl += s2.y; if (withRate) B.e += r.y;
        }
        template <class F> __device__ __forceinline__ void each_acc(Body& other, F f)
        {
            f(den, other.den); f(gs.x, other.gs.x); f(gs.y, other.gs.y); f(gs.z, other.gs.z); f(sl, other.sl);
            if (withRate) f(e, other.e);
        }
    };
};
template <bool WITH_RATE, bool STREAM>
__global__ void __launch_bounds__(kWideBlock, SPHX_MINWAVES) k_dfsph_head(const OpDfsphHeadT<WITH_RATE> o, int n)
{
    __shared__ BlockLds<OpDfsphHeadT<WITH_RATE>, STREAM> lds;
    const int wave = threadIdx.x >> 6;
    const int tile = wave_tile(o.c);
    if (tile < 0) return;
    const int i = tile * kTile + (int)(threadIdx.x & 63);
    (void)n;
    const bool valid = in_range(o.c, i);
    long long fixed = 0;
    const float3 own = (WITH_RATE && valid) ? o.vel[i] : v3(0, 0, 0);
    typename OpDfsphHeadT<WITH_RATE>::Body b{o, own.x, own.y, own.z, 0.0f, 0.0f, 0.0f, v3(0, 0, 0)};
    sweep<true>(o, o.c, STREAM ? lds.pos[wave] : nullptr, STREAM ? lds.field[wave] : nullptr, i, valid, own_pos(o.c, i, valid), b);
    if (valid) {
        const float al = -1.0f / max_eps(dot3(b.gs, b.gs) + b.sl);
        o.density[i] = b.den;
        o.alpha[i] = al;
        if (WITH_RATE) fixed = finish_rate<false, 0>(o.out, i, b.e, b.den, al);
    }
    if (WITH_RATE && o.out.accum) accumulate_error(fixed, o.out.accum);
}
template <bool WITH_RATE, int MODE, int TOLK = -1>
__global__ void __launch_bounds__(kWideBlock, (MODE == 1 && TOLK == 1) ? SPHX_HEAD_TOL_WAVES : SPHX_MINWAVES) k_dfsph_head_group(const OpDfsphHeadT<WITH_RATE> o, int n)
{
    assume_arith<TOLK>(o.c);
    const int i = MODE == 1 ? quad_particle(o.c) : duo_particle(o.c);
    if (i < 0) return;
    (void)n;
    const bool valid = in_range(o.c, i);
    long long fixed = 0;
    const float3 own = (WITH_RATE && valid) ? o.vel[i] : v3(0, 0, 0);
    typename OpDfsphHeadT<WITH_RATE>::Body b{o, own.x, own.y, own.z, 0.0f, 0.0f, 0.0f, v3(0, 0, 0)};
    sweep_any<MODE, true>(o, o.c, nullptr, nullptr, i, valid, own_pos(o.c, i, valid), b);
    if (stores_results<MODE>(valid)) {
        const float al = -1.0f / max_eps(dot3(b.gs, b.gs) + b.sl);
        o.density[i] = b.den;
        o.alpha[i] = al;
        if (WITH_RATE) fixed = finish_rate<false, 0>(o.out, i, b.e, b.den, al);
    }
    if (WITH_RATE && o.out.accum) accumulate_error(fixed, o.out.accum);
}
template <bool WITH_RATE>
__global__ void __launch_bounds__(kBrickThreads, 4) k_dfsph_head_brick(const OpDfsphHeadT<WITH_RATE> o, int n)
{
    (void)n;
    assume_arith<1>(o.c);
    using Op = OpDfsphHeadT<WITH_RATE>;
    brick_run<typename Op::Field>(o.c, [&](bool isB, int u) { return o.stage(isB, u); },
        [&](int i, bool valid, float4* lp, typename Op::Field* lf) {
            long long fixed = 0;
            const float3 own = (WITH_RATE && valid) ? o.vel[i] : v3(0, 0, 0);
            typename Op::Body b{o, own.x, own.y, own.z, 0.0f, 0.0f, 0.0f, v3(0, 0, 0)};
            sweep_brick<true>(o, o.c, lp, lf, i, valid, own_pos(o.c, i, valid), b);
            if (valid) {
                const float al = -1.0f / max_eps(dot3(b.gs, b.gs) + b.sl);
                o.density[i] = b.den;
                o.alpha[i] = al;
                if (WITH_RATE) fixed = finish_rate<false, 0>(o.out, i, b.e, b.den, al);
            }
            if (WITH_RATE && o.out.accum) accumulate_error(fixed, o.out.accum);
        });
}
template <bool WITH_RATE>
inline void launch_dfsph_head(const OpDfsphHeadT<WITH_RATE>& o, int n)
{
    if (n <= 0 || o.c.numTiles <= 0) return;
    if (o.c.brick) {
        const size_t lds = brick_lds_bytes<typename OpDfsphHeadT<WITH_RATE>::Field>();
        brick_launch_prepare(k_dfsph_head_brick<WITH_RATE>, lds);
        k_dfsph_head_brick<WITH_RATE><<<brick_grid(o.c), kBrickThreads, lds, stream()>>>(o, n);
        return;
    }
    if (o.c.nbr && o.c.tileFmt) k_dfsph_head<WITH_RATE, true><<<sweep_grid(o.c), kWideBlock, 0, stream()>>>(o, n);
    else if (o.c.nbr && (o.c.quad & kQuadHead)) {
        if (o.c.k.tol) k_dfsph_head_group<WITH_RATE, 1, 1><<<quad_grid(o.c), kWideBlock, 0, stream()>>>(o, n);
        else k_dfsph_head_group<WITH_RATE, 1, 0><<<quad_grid(o.c), kWideBlock, 0, stream()>>>(o, n);
    }
    else if (o.c.nbr && (o.c.duo & kQuadHead)) k_dfsph_head_group<WITH_RATE, 2><<<duo_grid(o.c), kWideBlock, 0, stream()>>>(o, n);
    else k_dfsph_head<WITH_RATE, false><<<sweep_grid(o.c), kWideBlock, 0, stream()>>>(o, n);
}

// stand-alone rate sweep: e = sum_f m_j (v_i - v_j).gradW + sum_b m_j v_i.gradW
// (boundaries are staged with v_j = +0: v_i - 0 == v_i)
struct OpRate {
    SweepCtx c;
    const float3* vel; const float* density; const float* alpha;
    RateOut out;
    using Field = float4;   // neighbour velocity
    __device__ __forceinline__ Field stage(bool, int j) const { return field4(c.vel4, j); }
    struct Body {
        const OpRate& o; float vix, viy, viz; float e;
        template <bool FAST>
        __device__ __forceinline__ void pair(Field vj, bool, float3 d, float r2, float mj, int)
        {
            e += mj * dot3(sub3(v3(vix, viy, viz), xyz(vj)), kGradW<FAST>(d, q_of<FAST>(sqrt_sel<FAST>(r2), o.c.k), o.c.k));
        }
        __device__ __forceinline__ void pair_tol(Field vj, bool, float3 d, float r2, float mj)
        {
            const float s = mj * tol_gradW_scale(tol_pair(r2, o.c.k), o.c.k);
            e = fmaf(s, tol_dot(vix - vj.x, viy - vj.y, viz - vj.z, d), e);
        }
        static constexpr bool kPair2 = true;
        __device__ __forceinline__ void pair2(Body& A, Body& B, Field va, Field vb, bool, bool, float3 pi, float4 pa, float4 pb) const
        {
            const Pair2 p = pair2_geometry(pi, pa, pb, o.c.k);
            const f2x3 g = kGradW_fast2(p.d, p.q, o.c.k);
            const f2 t = f2{pa.w, pb.w} * ((vix - f2{va.x, vb.x}) * g.x + (viy - f2{va.y, vb.y}) * g.y + (viz - f2{va.z, vb.z}) * g.z);
            A.e += t.x; B.e += t.y;
        }
        template <class F> __device__ __forceinline__ void each_acc(Body& other, F f) { f(e, other.e); }
    };
};
// quad-per-particle variant (walk_row_quad): two divergent gathers per pair make this sweep the one that gains most
#ifndef SPHX_QUAD_WAVES
#define SPHX_QUAD_WAVES 8      // waves per SIMD the register budget is cut for (64 VGPRs, 16 bytes of scratch): measured 4 % faster than 7
#endif
#ifndef SPHX_QUAD_WAVES_TOL
#define SPHX_QUAD_WAVES_TOL 6  // the tolerance walk keeps 4 chunks of partial state live: 6 waves per SIMD (80 VGPRs) avoid its scratch
#endif
template <bool DENSITY_MODE, int WARM, int TOLK>
__global__ void __launch_bounds__(kWideBlock, TOLK == 1 ? SPHX_QUAD_WAVES_TOL : SPHX_QUAD_WAVES) k_rate_quad(const OpRate o, int n)
{
    assume_arith<TOLK>(o.c);
    const int i = quad_particle(o.c);
    if (i < 0) return;
    (void)n;
    const bool valid = in_range(o.c, i);
    long long fixed = 0;
    const float3 own = valid ? o.vel[i] : v3(0, 0, 0);
    OpRate::Body b{o, own.x, own.y, own.z, 0.0f};
    sweep_quad<true>(o, o.c, i, valid, own_pos(o.c, i, valid), b);
    if (stores_results<1>(valid)) fixed = finish_rate<DENSITY_MODE, WARM>(o.out, i, b.e, o.density[i], o.alpha[i]);
    if (o.out.accum) accumulate_error(fixed, o.out.accum);
}
template <bool DENSITY_MODE, int WARM>
__global__ void __launch_bounds__(kWideBlock, SPHX_MINWAVES) k_rate_duo(const OpRate o, int n)
{
    const int i = duo_particle(o.c);
    if (i < 0) return;
    (void)n;
    const bool valid = in_range(o.c, i);
    long long fixed = 0;
    const float3 own = valid ? o.vel[i] : v3(0, 0, 0);
    OpRate::Body b{o, own.x, own.y, own.z, 0.0f};
    sweep_duo<true>(o, o.c, i, valid, own_pos(o.c, i, valid), b);
    if (stores_results<2>(valid)) fixed = finish_rate<DENSITY_MODE, WARM>(o.out, i, b.e, o.density[i], o.alpha[i]);
    if (o.out.accum) accumulate_error(fixed, o.out.accum);
}
template <bool DENSITY_MODE, int WARM, bool STREAM>
__global__ void __launch_bounds__(kWideBlock, SPHX_MINWAVES) k_rate(const OpRate o, int n)
{
    __shared__ BlockLds<OpRate, STREAM> lds;
    const int wave = threadIdx.x >> 6;
    const int tile = wave_tile(o.c);
    if (tile < 0) return;
    const int i = tile * kTile + (int)(threadIdx.x & 63);
    (void)n;
    const bool valid = in_range(o.c, i);
    long long fixed = 0;
    const float3 own = valid ? o.vel[i] : v3(0, 0, 0);
    OpRate::Body b{o, own.x, own.y, own.z, 0.0f};
    sweep<true>(o, o.c, STREAM ? lds.pos[wave] : nullptr, STREAM ? lds.field[wave] : nullptr, i, valid, own_pos(o.c, i, valid), b);
    if (valid) fixed = finish_rate<DENSITY_MODE, WARM>(o.out, i, b.e, o.density[i], o.alpha[i]);
    if (o.out.accum) accumulate_error(fixed, o.out.accum);
}
template <bool DENSITY_MODE, int WARM>
__global__ void __launch_bounds__(kBrickThreads, 4) k_rate_brick(const OpRate o, int n)
{
    (void)n;
    assume_arith<1>(o.c);
    brick_run<OpRate::Field>(o.c, [&](bool isB, int u) { return o.stage(isB, u); },
        [&](int i, bool valid, float4* lp, OpRate::Field* lf) {
            long long fixed = 0;
            const float3 own = valid ? o.vel[i] : v3(0, 0, 0);
            OpRate::Body b{o, own.x, own.y, own.z, 0.0f};
            sweep_brick<true>(o, o.c, lp, lf, i, valid, own_pos(o.c, i, valid), b);
            if (valid) fixed = finish_rate<DENSITY_MODE, WARM>(o.out, i, b.e, o.density[i], o.alpha[i]);
            if (o.out.accum) accumulate_error(fixed, o.out.accum);
        });
}
template <bool DENSITY_MODE, int WARM>
inline void launch_rate_kernel(const OpRate& o, int n)
{
    if (n <= 0 || o.c.numTiles <= 0) return;
    if (o.c.brick) {
        const size_t lds = brick_lds_bytes<OpRate::Field>();
        brick_launch_prepare(k_rate_brick<DENSITY_MODE, WARM>, lds);
        k_rate_brick<DENSITY_MODE, WARM><<<brick_grid(o.c), kBrickThreads, lds, stream()>>>(o, n);
        g_lastRateVariant = kRateBrick;
        return;
    }
    if (o.c.nbr && o.c.tileFmt) { k_rate<DENSITY_MODE, WARM, true><<<sweep_grid(o.c), kWideBlock, 0, stream()>>>(o, n); g_lastRateVariant = kRateLdsTiles; }
    else if (o.c.nbr && (o.c.quad & kQuadRate)) {
        if (o.c.k.tol && o.c.plainBits && !o.c.brick) {      // tolerance mode at size: the strict kernel is the faster one (SweepCache::ctx)
            OpRate s = o;
            s.c.k.tol = 0;
            k_rate_quad<DENSITY_MODE, WARM, 0><<<quad_grid(s.c), kWideBlock, 0, stream()>>>(s, n);
            g_lastRateVariant = kRateQuadStrictInTol;
        }
        else if (o.c.k.tol) { k_rate_quad<DENSITY_MODE, WARM, 1><<<quad_grid(o.c), kWideBlock, 0, stream()>>>(o, n); g_lastRateVariant = kRateQuadTol; }
        else { k_rate_quad<DENSITY_MODE, WARM, 0><<<quad_grid(o.c), kWideBlock, 0, stream()>>>(o, n); g_lastRateVariant = kRateQuadStrict; }
    }
    else if (o.c.nbr && (o.c.duo & kQuadRate)) { k_rate_duo<DENSITY_MODE, WARM><<<duo_grid(o.c), kWideBlock, 0, stream()>>>(o, n); g_lastRateVariant = kRateDuo; }
    else { k_rate<DENSITY_MODE, WARM, false><<<sweep_grid(o.c), kWideBlock, 0, stream()>>>(o, n); g_lastRateVariant = kRateLane; }
}

// correctDivergenceError_CUDA (DFSPHSolver.cu:308-329) / correctDensityError_CUDA (:138-158)
// (boundaries are staged with kappa_j = +0: kappa_i + 0 == kappa_i)
template <bool DIVIDE_BY_DT>
struct OpCorrect {
    SweepCtx c;
    const float* kappa; float3* vel;
    float dt;
    bool packedScalar;      // posf.w currently holds THIS kappa array for every fluid particle
    // fixed iteration counts, whole-domain steps: the gravity kick of BasicSPHSolver::force (vel += dt G, BasicSPHSolver.cu:227-235) that
    // follows the LAST divergence correction is applied by that correction's store -- the same two rounded additions, one pass less
    bool addKick = false; float kx = 0.0f, ky = 0.0f, kz = 0.0f;
    using Field = float;    // neighbour stiffness
    __device__ __forceinline__ Field stage(bool isB, int j) const { return fluid_only(kappa, isB, j); }
    struct Body {
        const OpCorrect& o; float ki; float3 a;
        template <bool FAST>
        __device__ __forceinline__ void pair(Field kj, bool, float3 d, float r2, float mj, int)
        {
            a = add3(a, smul3(mj * (ki + kj), kGradW<FAST>(d, q_of<FAST>(sqrt_sel<FAST>(r2), o.c.k), o.c.k)));
        }
        __device__ __forceinline__ void pair_tol(Field kj, bool, float3 d, float r2, float mj)
        {
            const float s = mj * (ki + kj) * tol_gradW_scale(tol_pair(r2, o.c.k), o.c.k);
            a = tol_axpy(a, d, s);
        }
        static constexpr bool kPair2 = true;
        __device__ __forceinline__ void pair2(Body& A, Body& B, Field ka, Field kb, bool, bool, float3 pi, float4 pa, float4 pb) const
        {
            const Pair2 p = pair2_geometry(pi, pa, pb, o.c.k);
            const f2x3 g = kGradW_fast2(p.d, p.q, o.c.k);
            const f2 s = f2{pa.w, pb.w} * (ki + f2{ka, kb});
            const f2 cx = s * g.x, cy = s * g.y, cz = s * g.z;
            A.a = add3(A.a, v3(cx.x, cy.x, cz.x));
            B.a = add3(B.a, v3(cx.y, cy.y, cz.y));
        }
        template <class F> __device__ __forceinline__ void each_acc(Body& other, F f) { f(a.x, other.a.x); f(a.y, other.a.y); f(a.z, other.a.z); }
    };
    static constexpr int kQuadBit = kQuadCorrect;
    template <int QUAD = 0>
    __device__ void operator()(int i, bool valid, float4* lp, Field* lf) const
    {
        Body b{*this, valid ? kappa[i] : 0.0f, v3(0, 0, 0)};
        sweep_any<QUAD, true>(*this, c, lp, lf, i, valid, own_pos(c, i, valid), b);
        if (!stores_results<QUAD>(valid)) return;
        float3 vn = add3(vel[i], DIVIDE_BY_DT ? div3s(b.a, dt) : b.a);
        if (addKick) vn = add3(vn, v3(kx, ky, kz));
        vel[i] = vn;
        c.vel4[i] = f4(vn);
    }
};

// ---- the tail of an adaptive DFSPH solver loop as ONE launch (r04) --------------------------------------------------------
// The device-decided loops (dfsph.hip) enqueue the launches of every possible iteration and gate them off once the loop has
// terminated; at 20 possible iterations that left ~100 launches per step that leave at their first instruction (1.5 us each on
// the reference scene: 0.16 of its 0.38 ms per step).  Here the first iteration(s) stay ordinary launches; every further
// iteration runs inside one persistent kernel: correction sweep, grid barrier, error sweep (its |error| terms go to the
// exact integer accumulators), grid barrier, then EVERY block forms the total from the same accumulators and applies the
// reference's rule (DFSPHSolver.cu:187-208, :347-361) -- the same decision everywhere, no third barrier.  The sweeps are the
// bodies of k_run_op<OpCorrect, quad> and k_rate_quad, so every per-particle sum is the one the separate launches form.
// The blocks spin on a counter in device memory: the grid never exceeds what the device holds at once (launch_dfsph_loop_tail).
struct LoopTail {
    int* st;                          // DFSPHSolver::loopState: done flag, iterations so far, divergence count, density count, fault, barrier word
    unsigned long long* accum;        // kErrorSlots partial totals (zero on entry; cumulative inside the tail)
    float threshold; int minIter, maxIter, which;
    int flatBarrier;                  // 1: the r04 barrier (one counter, a fence pair per block) -- sphx_tuning.dfsph_tail_flat, for measurements
};
// kLoopXcd: the state of the XCD-hierarchical barrier, one word per 128-byte line: blocks per XCD [0,8), arrivals per XCD [8,16),
// generation per XCD [16,24), arrivals of the XCD leaders [24]
enum { kLoopDone = 0, kLoopIter = 1, kLoopDiv = 2, kLoopDen = 3, kLoopFault = 4, kLoopBarrier = 5, kLoopXcd = 32, kLoopXcdLine = 32,
       kLoopWords = 32 + 25 * 32 };
// false: the other blocks did not arrive within seconds (the grid was not resident at once after all -- a profiler or a partition
// the occupancy query does not know of): the block raises st[kLoopFault] and leaves; the host reports it with the iteration counts
// (DFSPHSolver::fetchIterations) and goes back to gated launches.  A hang here would take the whole device with it.
__device__ __forceinline__ bool grid_barrier(unsigned int* word, unsigned int& target, int* fault)
{
    __shared__ int gaveUp;
    if (threadIdx.x == 0) gaveUp = 0;
    __syncthreads();
#ifndef SPHX_TAIL_FENCE
#define SPHX_TAIL_FENCE 3          // experiments: bit 0 release fence, bit 1 acquire fence (anything but 3 computes wrong results)
#endif
#ifndef SPHX_TAIL_SLEEP
#define SPHX_TAIL_SLEEP 2
#endif
    if (threadIdx.x == 0) {
        target += gridDim.x;
        if (SPHX_TAIL_FENCE & 1) __threadfence();              // release: what this block wrote (agent scope: L2 write-back across XCDs)
        atomicAdd(word, 1u);
        unsigned int spins = 0;
        while (__hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(SPHX_TAIL_SLEEP);
            if (++spins > (1u << 24)) { gaveUp = 1; *fault = 1; __threadfence(); break; }       // 2^24 x (uncached load + 128 clocks): seconds; a legitimate wait is a few ms
        }
        if (SPHX_TAIL_FENCE & 2) __threadfence();              // acquire: drop what the caches hold of the other blocks' arrays
    }
    __syncthreads();
    return gaveUp == 0;
}
// r06: the barrier between the sweeps of an iteration, XCD-hierarchical (MI355X_MICROARCH.md "barrier-xcd": 4-10 us where the flat counter
// with a release / acquire pair per BLOCK costs ~35 us at this kernel's 4+ blocks per CU).  The blocks of one XCD share its L2: a block's
// stores are there once its waves have passed __syncthreads (vmcnt(0)), so a block only ARRIVES on its XCD's counter; the last block of
// an XCD writes that L2 back ONCE (release, agent scope), arrives on the top counter, waits for the other XCDs' leaders, and publishes the
// XCD's generation; every block ends with an agent-scope acquire (its CU's L1 and the non-local lines of the L2).  Which XCD a block
// runs on is read from the hardware (HW_REG_XCC_ID), how many blocks each XCD got is counted at the start of the launch behind ONE flat
// barrier: nothing is assumed about the dispatcher's placement.  Spins are bounded and report like grid_barrier's.
struct XcdBarrier { int* base; int xcc; unsigned int g; unsigned int mine, groups; };
__device__ __forceinline__ bool xcd_barrier_begin(XcdBarrier& B, int* st, unsigned int* flatWord, unsigned int& flatTarget)
{
    B.base = st + kLoopXcd; B.g = 0u;
    B.xcc = (int)(__builtin_amdgcn_s_getreg(6164) & 7u);                // hwreg(HW_REG_XCC_ID, 0, 4)
    if (threadIdx.x == 0) atomicAdd(reinterpret_cast<unsigned int*>(B.base) + B.xcc * kLoopXcdLine, 1u);
    if (!grid_barrier(flatWord, flatTarget, st + kLoopFault)) return false;
    B.mine = 0u; B.groups = 0u;
    if (threadIdx.x == 0) {
        for (int x = 0; x < 8; ++x) {
            const unsigned int cnt = (unsigned int)__hip_atomic_load(B.base + x * kLoopXcdLine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (x == B.xcc) B.mine = cnt;
            B.groups += cnt ? 1u : 0u;
        }
    }
    return true;
}
__device__ __forceinline__ bool xcd_barrier(XcdBarrier& B, int* fault)
{
    __shared__ int gaveUpX;
    if (threadIdx.x == 0) gaveUpX = 0;
    __syncthreads();                                                     // every wave's stores have left for the L2
    if (threadIdx.x == 0) {
        ++B.g;
        unsigned int* const arrive = reinterpret_cast<unsigned int*>(B.base) + (8 + B.xcc) * kLoopXcdLine;
        int* const gen = B.base + (16 + B.xcc) * kLoopXcdLine;
        unsigned int* const top = reinterpret_cast<unsigned int*>(B.base) + 24 * kLoopXcdLine;
        unsigned int spins = 0;
        const unsigned int old = atomicAdd(arrive, 1u);
        if (old + 1u == B.mine * B.g) {                                  // the last block of this XCD
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");           // buffer_wbl2 sc1: this XCD's dirty lines, once
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            atomicAdd(top, 1u);
            while ((unsigned int)__hip_atomic_load(reinterpret_cast<int*>(top), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < B.groups * B.g) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > (1u << 24)) { gaveUpX = 1; *fault = 1; __threadfence(); break; }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            __hip_atomic_store(gen, (int)B.g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            while ((unsigned int)__hip_atomic_load(gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < B.g) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > (1u << 24)) { gaveUpX = 1; *fault = 1; __threadfence(); break; }
                if ((spins & 1023u) == 0u && __hip_atomic_load(fault, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) { gaveUpX = 1; break; }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
    }
    __syncthreads();
    return gaveUpX == 0;
}
// CM: the correction sweep's launch shape, 0 lane per particle (a block takes four tiles), 1 quad per particle (one tile)
template <bool DENSITY_MODE, int WARM, int CM, int TOLC, int TOLR>
__global__ void __launch_bounds__(kWideBlock) k_dfsph_loop_tail(const OpCorrect<DENSITY_MODE> corr, const OpRate rate, const LoopTail t)
{
    static_assert(kErrorSlots == kWideBlock, "one accumulator per thread of a block");
    if (t.st[kLoopDone] != 0) return;                          // (written by the launch before this one: the same for every block)
    if (__hip_atomic_load(t.st + kLoopFault, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return;      // a block that started late: the launch has failed already
    __shared__ unsigned long long part[kWideBlock / 64];
    unsigned int* const word = reinterpret_cast<unsigned int*>(t.st + kLoopBarrier);
    unsigned int target = 0;
    int iter = t.st[kLoopIter];
    unsigned long long before = 0;
    XcdBarrier xb;
    const bool flat = t.flatBarrier != 0;
    if (!flat && !xcd_barrier_begin(xb, t.st, word, target)) return;
    for (;;) {
        if constexpr (CM == 1) {
            for (int lt = (int)blockIdx.x; lt < corr.c.numTiles; lt += (int)gridDim.x) {
                assume_arith<TOLC>(corr.c);
                const int i = quad_particle_of(corr.c, lt);
                if (i >= 0) corr.template operator()<1>(i, in_range(corr.c, i), nullptr, nullptr);
            }
        } else {
            for (int lt = (int)blockIdx.x * (kWideBlock / kTile) + (int)(threadIdx.x >> 6); lt < corr.c.numTiles; lt += (int)gridDim.x * (kWideBlock / kTile)) {
                const int tile = wave_tile_of(corr.c, lt);
                if (tile < 0) continue;
                const int i = tile * kTile + (int)(threadIdx.x & 63);
                corr(i, in_range(corr.c, i), nullptr, nullptr);
            }
        }
        if (!(flat ? grid_barrier(word, target, t.st + kLoopFault) : xcd_barrier(xb, t.st + kLoopFault))) return;
        for (int lt = (int)blockIdx.x; lt < rate.c.numTiles; lt += (int)gridDim.x) {
            assume_arith<TOLR>(rate.c);
            const int i = quad_particle_of(rate.c, lt);
            if (i < 0) continue;
            const bool valid = in_range(rate.c, i);
            long long fixed = 0;
            const float3 own = valid ? rate.vel[i] : v3(0, 0, 0);
            OpRate::Body b{rate, own.x, own.y, own.z, 0.0f};
            sweep_quad<true>(rate, rate.c, i, valid, own_pos(rate.c, i, valid), b);
            if (stores_results<1>(valid)) fixed = finish_rate<DENSITY_MODE, WARM>(rate.out, i, b.e, rate.density[i], rate.alpha[i]);
            accumulate_error(fixed, rate.out.accum);
        }
        if (!(flat ? grid_barrier(word, target, t.st + kLoopFault) : xcd_barrier(xb, t.st + kLoopFault))) return;
        unsigned long long v = t.accum[(size_t)threadIdx.x * kErrorSlotStride];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
        if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = v;
        __syncthreads();
        unsigned long long sum = 0;
        for (int w = 0; w < kWideBlock / 64; ++w) sum += part[w];
        const unsigned long long total = sum - before;         // the accumulators are not zeroed between iterations here
        before = sum;
        ++iter;
        const float totalError = (float)((double)(long long)total * (1.0 / 4294967296.0));      // DFSPHSolver::readErrorTotal
        if (!((iter < t.minIter || totalError > t.threshold) && iter < t.maxIter)) break;
        // (part[] is rewritten only behind the next two barriers)
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) { t.st[kLoopIter] = iter; t.st[t.which] = iter; t.st[kLoopDone] = 1; }
}
template <bool DENSITY_MODE, int WARM, int CM, int TOLC, int TOLR>
inline bool launch_tail_kernel(const OpCorrect<DENSITY_MODE>& corr, const OpRate& rate, const LoopTail& t)
{
    // blocks of THIS kernel a compute unit of THIS device holds at once (queried once per device and instantiation; the host side of
    // the engine is single-threaded per process)
    constexpr int kMaxDevices = 64;
    static int perCUof[kMaxDevices], cusOf[kMaxDevices];
    static bool known[kMaxDevices];
    int dev = 0;
    HIP_CALL(hipGetDevice(&dev));
    if (dev < 0 || dev >= kMaxDevices) return false;
    if (!known[dev]) {
        hipDeviceProp_t prop;
        int blocks = 0;
        if (hipGetDeviceProperties(&prop, dev) != hipSuccess ||
            hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, k_dfsph_loop_tail<DENSITY_MODE, WARM, CM, TOLC, TOLR>, kWideBlock, 0) != hipSuccess) {
            (void)hipGetLastError();
            blocks = 0; prop.multiProcessorCount = 0;
        }
        perCUof[dev] = blocks; cusOf[dev] = prop.multiProcessorCount; known[dev] = true;
    }
    const int perCU = perCUof[dev], cus = cusOf[dev];
    if (perCU < 1 || cus < 1) return false;
    int grid = std::min(corr.c.numTiles, perCU * cus);         // (the rate sweep takes one tile per block; a lane-walk correction four)
#ifdef SPHX_TEST_HOOKS
    if (getenv("SPHX_DFSPH_TAIL_OVERSUBSCRIBE")) grid = std::max(corr.c.numTiles, 16 * perCU * cus);      // more blocks than fit: the barrier must time out and report
#endif
    k_dfsph_loop_tail<DENSITY_MODE, WARM, CM, TOLC, TOLR><<<grid, kWideBlock, 0, stream()>>>(corr, rate, t);
    // a launch that was refused (not one that failed later) leaves the loop to the caller's gated launches; inside a stream capture
    // the error, if any, surfaces when the capture ends and the step then runs eagerly (SPHSystem::stepN)
    if (hipGetLastError() != hipSuccess) return false;
    return true;
}
// false: this configuration has no tail kernel (the caller enqueues gated launches instead)
template <bool DENSITY_MODE, int WARM>
inline bool launch_dfsph_loop_tail(const OpCorrect<DENSITY_MODE>& corrIn, const OpRate& rateIn, const LoopTail& t, int n)
{
    const SweepCtx& c = corrIn.c;
    if (n <= 0 || c.numTiles <= 0 || !c.nbr || c.tileFmt || c.brick || !(c.quad & kQuadRate)) return false;
    const bool quadCorrect = (c.quad & kQuadCorrect) != 0;
    if (!quadCorrect && (c.duo & kQuadCorrect)) return false;  // (launch_op: the two-lane walk has no tail form)
    OpCorrect<DENSITY_MODE> corr = corrIn;
    OpRate rate = rateIn;
    corr.c.gate = nullptr; rate.c.gate = nullptr;
    const bool tolC = c.k.tol != 0;
    const bool tolR = tolC && !c.plainBits;                    // (launch_rate_kernel: the strict walk serves the rate sweeps of large tolerance runs)
    if (tolC && !tolR) rate.c.k.tol = 0;
    if (!quadCorrect) {                                        // (lane walks decide their arithmetic at run time: k_run_op<Op, false, 0, -1>)
        if (tolR) return launch_tail_kernel<DENSITY_MODE, WARM, 0, -1, 1>(corr, rate, t);
        return launch_tail_kernel<DENSITY_MODE, WARM, 0, -1, 0>(corr, rate, t);
    }
    if (tolC && tolR) return launch_tail_kernel<DENSITY_MODE, WARM, 1, 1, 1>(corr, rate, t);
    if (tolC) return launch_tail_kernel<DENSITY_MODE, WARM, 1, 1, 0>(corr, rate, t);
    return launch_tail_kernel<DENSITY_MODE, WARM, 1, 0, 0>(corr, rate, t);
}

// =================================================================================== PBD
// computeDensityLambda_CUDA, PBDSolver.cu:127-168.  `rb` is (float)(bool)rho0;
// dividing by 1.0f is the identity, so the division is only performed when rb != 1.
struct OpLambda {
    SweepCtx c;
    float* density; float* lambda;
    float rho0, rb, relaxation;
    using Field = float;    // unused
    __device__ __forceinline__ Field stage(bool, int) const { return 0.0f; }
    struct Body {
        const OpLambda& o; float den, sl; float3 gs;
        template <bool FAST>
        __device__ __forceinline__ void pair(Field, bool, float3 d, float r2, float mj, int)
        {
            const float q = q_of<FAST>(sqrt_sel<FAST>(r2), o.c.k);
            den += mj * kW<FAST>(q, o.c.k);
            float3 gr = smul3(-mj, kGradW<FAST>(d, q, o.c.k));
            if (o.rb != 1.0f) gr = div3s(gr, o.rb);
            gs = sub3(gs, gr);
            sl += dot3(gr, gr);
        }
        __device__ __forceinline__ void pair_tol(Field, bool, float3 d, float r2, float mj)
        {
            const TolPair t = tol_pair(r2, o.c.k);
            den = fmaf(mj, tol_W(t, o.c.k), den);
            float s = -mj * tol_gradW_scale(t, o.c.k);
            if (o.rb != 1.0f) s = s / o.rb;
            const float3 gr = v3(d.x * s, d.y * s, d.z * s);
            gs = v3(gs.x - gr.x, gs.y - gr.y, gs.z - gr.z);
            sl += tol_dot(gr.x, gr.y, gr.z, gr);
        }
        template <class F> __device__ __forceinline__ void each_acc(Body& other, F f)
        {
            f(den, other.den); f(gs.x, other.gs.x); f(gs.y, other.gs.y); f(gs.z, other.gs.z); f(sl, other.sl);
        }
    };
    static constexpr int kQuadBit = kQuadLambda;
    template <int QUAD = 0>
    __device__ void operator()(int i, bool valid, float4* lp, Field* lf) const
    {
        Body b{*this, 0.0f, 0.0f, v3(0, 0, 0)};
        sweep_any<QUAD, true>(*this, c, lp, lf, i, valid, own_pos(c, i, valid), b);
        if (!stores_results<QUAD>(valid)) return;
        density[i] = b.den;
        float lam = (b.den > rho0) ? (-(b.den / rho0 - 1.0f) / (dot3(b.gs, b.gs) + b.sl + kEps)) : 0.0f;
        lam *= relaxation;
        lambda[i] = lam;
        c.posf[i].w = lam;
    }
};

// computeDeltaPos_CUDA, PBDSolver.cu:170-210 (boundaries: lambda_j = +0)
// r05: the sweep can also APPLY its delta-p (PBDSolver.cu:247-253: pos += delta-p, box clamp) in its store.  A Jacobi iteration may
// not move a position before every delta-p has been formed from the old ones, and the sweep gathers neighbour positions
// from posm / posf -- so the moved positions go into the OTHER half of a double-buffered pair (SweepCache::posmAlt / posfAlt), which
// becomes the live one behind the launch.  The API position array is only ever read for the particle's own entry: updated in place.
struct DeltaApply {
    float3* pos = nullptr; float4* posmNext = nullptr; float4* posfNext = nullptr; float3 space = {0.0f, 0.0f, 0.0f};
    SkinWatch watch{};      // skin rows: k_apply_delta_clamp's watch
};
struct OpDeltaPos {
    SweepCtx c;
    const float* lambda; float3* deltaPos;
    float rho0;
    bool packedScalar;      // posf.w holds lambda
    DeltaApply apply{};
    using Field = float;    // neighbour lambda
    __device__ __forceinline__ Field stage(bool isB, int j) const { return fluid_only(lambda, isB, j); }
    struct Body {
        const OpDeltaPos& o; float li; float3 a;
        template <bool FAST>
        __device__ __forceinline__ void pair(Field lj, bool, float3 d, float r2, float mj, int)
        {
            a = add3(a, smul3(mj * (li + lj), kGradW<FAST>(d, q_of<FAST>(sqrt_sel<FAST>(r2), o.c.k), o.c.k)));
        }
        __device__ __forceinline__ void pair_tol(Field lj, bool, float3 d, float r2, float mj)
        {
            const float s = mj * (li + lj) * tol_gradW_scale(tol_pair(r2, o.c.k), o.c.k);
            a = tol_axpy(a, d, s);
        }
        template <class F> __device__ __forceinline__ void each_acc(Body& other, F f) { f(a.x, other.a.x); f(a.y, other.a.y); f(a.z, other.a.z); }
    };
    static constexpr int kQuadBit = kQuadDelta;
    template <int QUAD = 0>
    __device__ void operator()(int i, bool valid, float4* lp, Field* lf) const
    {
        Body b{*this, valid ? lambda[i] : 0.0f, v3(0, 0, 0)};
        sweep_any<QUAD, true>(*this, c, lp, lf, i, valid, own_pos(c, i, valid), b);
        if (!stores_results<QUAD>(valid)) return;
        const float3 dp = div3s(b.a, rho0);
        if (!apply.pos) { deltaPos[i] = dp; return; }
        float3 p = add3(apply.pos[i], dp);                      // exactly k_apply_delta_clamp
        float3 unused = v3(0, 0, 0);
        clamp_box<false>(p, unused, apply.space);
        apply.pos[i] = p;
        apply.posmNext[i] = make_float4(p.x, p.y, p.z, c.posm[i].w);
        apply.posfNext[i] = make_float4(p.x, p.y, p.z, 0.0f);
        if (apply.watch.posBuild) skin_watch(apply.watch, p, i, c.g);
    }
};

// XSPHViscosity_CUDA (PBDSolver.cu:89-115), Jacobi form (reads vel, writes velOut),
// optionally fused with the colour gradient (BasicSPHSolver.cu:277-318).
template <bool COLOR>
struct OpXsph {
    SweepCtx c;
    const float3* vel; float3* velOut; float3* colorGrad;
    float xsphC, rho0, rhoB;
    using Field = float4;   // neighbour velocity
    __device__ __forceinline__ Field stage(bool, int j) const { return field4(c.vel4, j); }
    struct Body {
        const OpXsph& o; float3 vi; float3 a; float3 cg; float cden;
        float mRef, volRef;
        template <bool FAST>
        __device__ __forceinline__ void pair(Field vj, bool isB, float3 d, float r2, float mj, int)
        {
            const float q = q_of<FAST>(sqrt_sel<FAST>(r2), o.c.k);
            const float w = kW<FAST>(q, o.c.k);
            if (!isB) a = add3(a, mul3s(smul3(mj, sub3(xyz(vj), vi)), w));
            if (COLOR) {
                float vol = volRef;
                if (__any(isB || mj != mRef)) vol = mj / (isB ? o.rhoB : o.rho0);
                cg = add3(cg, smul3(vol, kGradW<FAST>(d, q, o.c.k)));
                cden += vol * w;
            }
        }
        __device__ __forceinline__ void pair_tol(Field vj, bool isB, float3 d, float r2, float mj)
        {
            const TolPair t = tol_pair(r2, o.c.k);
            const float w = tol_W(t, o.c.k);
            if (!isB) a = tol_axpy(a, v3(vj.x - vi.x, vj.y - vi.y, vj.z - vi.z), mj * w);
            if (COLOR) {
                const float vol = mj * __builtin_amdgcn_rcpf(isB ? o.rhoB : o.rho0);
                const float s = vol * tol_gradW_scale(t, o.c.k);
                cg = tol_axpy(cg, d, s);
                cden = fmaf(vol, w, cden);
            }
        }
        template <class F> __device__ __forceinline__ void each_acc(Body& other, F f)
        {
            f(a.x, other.a.x); f(a.y, other.a.y); f(a.z, other.a.z);
            if (COLOR) { f(cg.x, other.cg.x); f(cg.y, other.cg.y); f(cg.z, other.cg.z); f(cden, other.cden); }
        }
    };
    static constexpr int kQuadBit = kQuadXsph;
    template <int QUAD = 0>
    __device__ void operator()(int i, bool valid, float4* lp, Field* lf) const
    {
        const float mRef = valid ? c.posm[i].w : 0.0f;
        Body b{*this, valid ? vel[i] : v3(0, 0, 0), v3(0, 0, 0), v3(0, 0, 0), 0.0f, mRef, mRef / rho0};
        sweep_any<QUAD, COLOR>(*this, c, lp, lf, i, valid, own_pos(c, i, valid), b);
        if (!stores_results<QUAD>(valid)) return;
        velOut[i] = add3(b.vi, div3s(smul3(xsphC, b.a), rho0));   // not the live velocity yet: vel4 untouched
        if (COLOR) { const float3 g = div3s(b.cg, max_eps(b.cden)); colorGrad[i] = g; c.cg4[i] = f4(g); }
    }
};

// =================================================================================== element-wise
// velocity writers also refresh the aligned mirror vel4 (may be nullptr for non-velocity arrays)
static __global__ void k_add_const3(float3* __restrict__ v, float4* __restrict__ v4, float3 c, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    v[i] = add3(v[i], c);
    // the mirror is advanced from the mirror: for an owned particle both hold the same bits; for a
    // slab's ghost particle only the mirror is kept current (halo refresh), and it must stay so
    if (v4) { const float4 q = v4[i]; v4[i] = make_float4(q.x + c.x, q.y + c.y, q.z + c.z, 0.0f); }
}
static __global__ void k_add3(float3* __restrict__ v, float4* __restrict__ v4, const float3* __restrict__ w, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float3 r = add3(v[i], w[i]);
    v[i] = r;
    if (v4) v4[i] = make_float4(r.x, r.y, r.z, 0.0f);
}
static __global__ void k_copy3_mirror(float3* __restrict__ v, float4* __restrict__ v4, const float3* __restrict__ src, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float3 r = src[i];
    v[i] = r;
    v4[i] = make_float4(r.x, r.y, r.z, 0.0f);
}
// pack (x,y,z,mass) and apply the gravity kick vel += dv in the same pass (BasicSPHSolver.cu:227-235)
static __global__ void k_pack_kick(float4* __restrict__ posm, const float3* __restrict__ pos, const float* __restrict__ mass,
                                   float3* __restrict__ vel, float3 dv, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float3 p = pos[i];
    posm[i] = make_float4(p.x, p.y, p.z, mass[i]);
    vel[i] = add3(vel[i], dv);
}
// Particles::advect + enforceBoundary_CUDA(pos, vel): Particles.cu:28-36, BasicSPHSolver.cu:85-101
// skipIf: a device word that, when raised, keeps the positions where they are (DFSPH's loop tail reported a fault in this step:
// velocities that are only partly corrected must not move anything; the host reports the step as invalid)
static __global__ void k_advect_clamp(float3* __restrict__ pos, float3* __restrict__ vel, float dt, float3 space, int n, const int* __restrict__ skipIf)
{
    if (skipIf && *skipIf != 0) return;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float3 v = vel[i];
    float3 p = add3(pos[i], smul3(dt, v));
    clamp_box<true>(p, v, space);
    pos[i] = p;
    vel[i] = v;
}
// PBD tail: vel += dv (gravity), posLast = pos, then advect + clamp (PBDSolver.cu:69-79)
static __global__ void k_kick_remember_advect(float3* __restrict__ pos, float3* __restrict__ vel, float3* __restrict__ posLast,
                                              float3 dv, float dt, float3 space, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float3 v = add3(vel[i], dv);
    const float3 p0 = pos[i];
    posLast[i] = p0;
    float3 p = add3(p0, smul3(dt, v));
    clamp_box<true>(p, v, space);
    pos[i] = p;
    vel[i] = v;
}
// pos += deltaPos; enforceBoundary_CUDA(pos): PBDSolver.cu:212-223, :247-253 (also refreshes posm)
// With skin rows (watch.posBuild != nullptr) the update also checks how far the particle is from where its row was built and
// whether it is still in the cell the row was built around (skin_watch).
static __global__ void k_apply_delta_clamp(float3* __restrict__ pos, float4* __restrict__ posm, float4* __restrict__ posf,
                                           const float3* __restrict__ dpos, float3 space, int n, SkinWatch watch, GridDesc g)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float3 p = add3(pos[i], dpos[i]);
    float3 unused = v3(0, 0, 0);
    clamp_box<false>(p, unused, space);
    pos[i] = p;
    posm[i] = make_float4(p.x, p.y, p.z, posm[i].w);
    posf[i] = make_float4(p.x, p.y, p.z, 0.0f);
    if (watch.posBuild) skin_watch(watch, p, i, g);
}
// vel = (pos - posLast) / dt, PBDSolver.cu:55-60
static __global__ void k_velocity_from_displacement(float3* __restrict__ vel, float4* __restrict__ vel4,
                                                    const float3* __restrict__ pos, const float3* __restrict__ posLast,
                                                    float dt, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float3 r = div3s(sub3(pos[i], posLast[i]), dt);
    vel[i] = r;
    vel4[i] = make_float4(r.x, r.y, r.z, 0.0f);
}

inline void launch_add_const3(float3* v, float4* v4, float3 c, int n)
{
    if (n > 0) k_add_const3<<<blocks_for(n), 256, 0, stream()>>>(v, v4, c, n);
}
inline void launch_add3(float3* v, float4* v4, const float3* w, int n)
{
    if (n > 0) k_add3<<<blocks_for(n), 256, 0, stream()>>>(v, v4, w, n);
}
inline void launch_copy3_mirror(float3* v, float4* v4, const float3* src, int n)
{
    if (n > 0) k_copy3_mirror<<<blocks_for(n), 256, 0, stream()>>>(v, v4, src, n);
}
inline void launch_pack_kick(float4* posm, const float3* pos, const float* mass, float3* vel, float3 dv, int n)
{
    if (n > 0) k_pack_kick<<<blocks_for(n), 256, 0, stream()>>>(posm, pos, mass, vel, dv, n);
}
inline void launch_advect_clamp(float3* pos, float3* vel, float dt, float3 space, int n, const int* skipIf = nullptr)
{
    if (n > 0) k_advect_clamp<<<blocks_for(n), 256, 0, stream()>>>(pos, vel, dt, space, n, skipIf);
}
inline void launch_kick_remember_advect(float3* pos, float3* vel, float3* posLast, float3 dv, float dt, float3 space, int n)
{
    if (n > 0) k_kick_remember_advect<<<blocks_for(n), 256, 0, stream()>>>(pos, vel, posLast, dv, dt, space, n);
}
inline void launch_apply_delta_clamp(float3* pos, float4* posm, float4* posf, const float3* dpos, float3 space, int n,
                                     const SkinWatch& watch = SkinWatch{}, GridDesc g = GridDesc{})
{
    if (n > 0) k_apply_delta_clamp<<<blocks_for(n), 256, 0, stream()>>>(pos, posm, posf, dpos, space, n, watch, g);
}
inline void launch_velocity_from_displacement(float3* vel, float4* vel4, const float3* pos, const float3* posLast, float dt, int n)
{
    if (n > 0) k_velocity_from_displacement<<<blocks_for(n), 256, 0, stream()>>>(vel, vel4, pos, posLast, dt, n);
}

}  // namespace sphx
