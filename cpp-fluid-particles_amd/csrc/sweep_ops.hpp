// sweep_ops.hpp — the per-particle neighbour-sweep operators of all three solvers, plus the small
// element-wise passes, as device functors launched one lane per fluid particle.
//
// Each operator cites the reference kernel whose arithmetic it restates (association order kept,
// see sph_device.hpp).  Quantities that depend only on particle i are hoisted out of the pair loop
// and quantities that depend only on particle j are read from per-particle arrays (e.g. `pterm`);
// both are pure-function hoists and change no bit.
#pragma once

#include "engine.hpp"

namespace sphx {

template <class Op>
__global__ void __launch_bounds__(256) k_run_op(const Op op, int n)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) op(i);
}
template <class Op>
inline void launch_op(const Op& op, int n)
{
    if (n > 0) k_run_op<Op><<<blocks_for(n), 256, 0, stream()>>>(op, n);
}

// =================================================================================== WCSPH
// viscosity_CUDA, BasicSPHSolver.cu:183-209
struct OpViscosity {
    GridDesc g; KernelConsts k;
    const int* csF; const float4* posm; const float3* vel; float3* deltaV;
    float rho0, visc, dt;
    struct Body {
        const OpViscosity& o; float3 vi; float3 a;
        __device__ __forceinline__ void fluid(int j, float3, float r2, float mj)
        {
            a = add3(a, mul3s(smul3(mj, div3s(sub3(o.vel[j], vi), o.rho0)), kViscLap(sqrtf(r2), o.k)));
        }
        __device__ __forceinline__ void boundary(int, float3, float, float) {}
    };
    __device__ void operator()(int i) const
    {
        Body b{*this, vel[i], v3(0, 0, 0)};
        sweep27<true, false>(g, k, csF, posm, nullptr, nullptr, xyz(posm[i]), b);
        deltaV[i] = mul3s(smul3(visc, b.a), dt);
    }
};

// computeColorGrad_CUDA, BasicSPHSolver.cu:277-318
struct OpColorGrad {
    GridDesc g; KernelConsts k;
    const int* csF; const float4* posm; const int* csB; const float4* bposm;
    float3* colorGrad;
    float rho0, rhoB;
    struct Body {
        const OpColorGrad& o; float3 cg; float den;
        __device__ __forceinline__ void term(float vol, float3 d, float r2)
        {
            const float q = q_of(sqrtf(r2), o.k);
            cg = add3(cg, smul3(vol, kGradW(d, q, o.k)));
            den += vol * kW(q, o.k);
        }
        __device__ __forceinline__ void fluid(int, float3 d, float r2, float mj) { term(mj / o.rho0, d, r2); }
        __device__ __forceinline__ void boundary(int, float3 d, float r2, float mj) { term(mj / o.rhoB, d, r2); }
    };
    __device__ void operator()(int i) const
    {
        Body b{*this, v3(0, 0, 0), 0.0f};
        sweep27<true, true>(g, k, csF, posm, csB, bposm, xyz(posm[i]), b);
        colorGrad[i] = div3s(b.cg, max_eps(b.den));
    }
};

// surfaceTensionAndAirPressure_CUDA, BasicSPHSolver.cu:332-370
struct OpSurface {
    GridDesc g; KernelConsts k;
    const int* csF; const float4* posm; const float3* colorGrad; float3* vel;
    float rho0, tension, airPressure, dt;
    struct Body {
        const OpSurface& o; float dii, li, ml; float3 a;
        __device__ __forceinline__ void fluid(int j, float3 d, float r2, float mj)
        {
            const float r = sqrtf(r2);
            const float q = q_of(r, o.k);
            const float3 cgj = o.colorGrad[j];
            a = add3(a, smul3(0.25f * mj / (o.rho0 * o.rho0) * o.tension * (dii + dot3(cgj, cgj)), kSurfGrad(d, r, o.k)));
            a = add3(a, div3s(mul3s(smul3(o.airPressure * mj / (o.rho0 * o.rho0), kGradW(d, q, o.k)), li), ml));
        }
        __device__ __forceinline__ void boundary(int, float3, float, float) {}
    };
    __device__ void operator()(int i) const
    {
        const float3 cgi = colorGrad[i];
        const float li = len3(cgi);
        Body b{*this, dot3(cgi, cgi), li, max_eps(li), v3(0, 0, 0)};
        sweep27<true, false>(g, k, csF, posm, nullptr, nullptr, xyz(posm[i]), b);
        vel[i] = add3(vel[i], mul3s(b.a, dt));
    }
};

// computeDensity_CUDA + computePressure_CUDA, BasicSPHSolver.cu:32-83, :103-111 (EOS fused into the
// epilogue; also emits pterm = p / max(EPS, rho^2) for the pressure-force sweep)
struct OpDensityPressure {
    GridDesc g; KernelConsts k;
    const int* csF; const float4* posm; const int* csB; const float4* bposm;
    float* density; float* pressure; float* pterm;
    float rho0, stiff;
    struct Body {
        const OpDensityPressure& o; float den;
        __device__ __forceinline__ void fluid(int, float3, float r2, float mj) { den += mj * kW(q_of(sqrtf(r2), o.k), o.k); }
        __device__ __forceinline__ void boundary(int, float3, float r2, float mj) { den += mj * kW(q_of(sqrtf(r2), o.k), o.k); }
    };
    __device__ void operator()(int i) const
    {
        Body b{*this, 0.0f};
        sweep27<true, true>(g, k, csF, posm, csB, bposm, xyz(posm[i]), b);
        density[i] = b.den;
        float p = stiff * (pow7(b.den / rho0) - 1.0f);
        if (p < 0.0f) p = 0.0f;
        pressure[i] = p;
        pterm[i] = p / max_eps(b.den * b.den);
    }
};

// pressureForce_CUDA, BasicSPHSolver.cu:113-165
struct OpPressureForce {
    GridDesc g; KernelConsts k;
    const int* csF; const float4* posm; const int* csB; const float4* bposm;
    const float* pterm; float3* vel;
    float dt;
    struct Body {
        const OpPressureForce& o; int i; float pti; float3 a;
        __device__ __forceinline__ void fluid(int j, float3 d, float r2, float mj)
        {
            if (j == i) return;
            a = add3(a, smul3(-mj * (pti + o.pterm[j]), kGradW(d, q_of(sqrtf(r2), o.k), o.k)));
        }
        __device__ __forceinline__ void boundary(int, float3 d, float r2, float mj)
        {
            a = add3(a, smul3(-mj * pti, kGradW(d, q_of(sqrtf(r2), o.k), o.k)));
        }
    };
    __device__ void operator()(int i) const
    {
        Body b{*this, i, pterm[i], v3(0, 0, 0)};
        sweep27<true, true>(g, k, csF, posm, csB, bposm, xyz(posm[i]), b);
        float3 a = b.a;
        if (len3(a) > kMaxA) a = mul3s(mul3s(a, 1.0f / sqrtf(dot3(a, a))), kMaxA);
        vel[i] = add3(vel[i], mul3s(a, dt));
    }
};

// =================================================================================== DFSPH
// computeDensityAlpha_CUDA, DFSPHSolver.cu:212-249
struct OpDensityAlpha {
    GridDesc g; KernelConsts k;
    const int* csF; const float4* posm; const int* csB; const float4* bposm;
    float* density; float* alpha;
    struct Body {
        const OpDensityAlpha& o; float den, sl; float3 gs;
        __device__ __forceinline__ void fluid(int, float3 d, float r2, float mj)
        {
            const float q = q_of(sqrtf(r2), o.k);
            den += mj * kW(q, o.k);
            const float3 gr = smul3(mj, kGradW(d, q, o.k));
            gs = add3(gs, gr);
            sl += dot3(gr, gr);
        }
        __device__ __forceinline__ void boundary(int, float3 d, float r2, float mj)
        {
            const float q = q_of(sqrtf(r2), o.k);
            den += mj * kW(q, o.k);
            gs = add3(gs, smul3(mj, kGradW(d, q, o.k)));
        }
    };
    __device__ void operator()(int i) const
    {
        Body b{*this, 0.0f, 0.0f, v3(0, 0, 0)};
        sweep27<true, true>(g, k, csF, posm, csB, bposm, xyz(posm[i]), b);
        density[i] = b.den;
        alpha[i] = -1.0f / max_eps(dot3(b.gs, b.gs) + b.sl);
    }
};

// computeDivergenceError_CUDA (DFSPHSolver.cu:261-306) and computeDensityError_CUDA (:74-116):
// e = sum_f m_j (v_i - v_j).gradW + sum_b m_j v_i.gradW, then the mode-specific clamp.  The warm
// stiffness bookkeeping of DFSPHSolver.cu:185,199-203 and the |error| reduction of :206,:360 are
// fused into the epilogue (WARM: 0 none, 1 set, 2 accumulate).
struct OpRate {
    GridDesc g; KernelConsts k;
    const int* csF; const float4* posm; const int* csB; const float4* bposm;
    const float3* vel; const float* density; const float* alpha;
    float* error; float* kappa; float* warm;
    unsigned long long* accum;   // fixed-point sum of |error| (nullptr: no reduction)
    float dt, rho0;
    struct Body {
        const OpRate& o; float3 vi; float e;
        __device__ __forceinline__ void fluid(int j, float3 d, float r2, float mj)
        {
            e += mj * dot3(sub3(vi, o.vel[j]), kGradW(d, q_of(sqrtf(r2), o.k), o.k));
        }
        __device__ __forceinline__ void boundary(int, float3 d, float r2, float mj)
        {
            e += mj * dot3(vi, kGradW(d, q_of(sqrtf(r2), o.k), o.k));
        }
    };
};
template <bool DENSITY_MODE, int WARM>
__global__ void __launch_bounds__(256) k_rate(const OpRate o, int n)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    long long fixed = 0;
    if (i < n) {
        OpRate::Body b{o, o.vel[i], 0.0f};
        sweep27<true, true>(o.g, o.k, o.csF, o.posm, o.csB, o.bposm, xyz(o.posm[i]), b);
        float err;
        const float den = o.density[i];
        if (DENSITY_MODE) {
            err = max0(o.dt * b.e + den - o.rho0);
        } else {
            err = max0(b.e);
            if (den + o.dt * err < o.rho0 && den <= o.rho0) err = 0.0f;
        }
        const float kap = err * o.alpha[i];
        o.error[i] = err;
        o.kappa[i] = kap;
        if (WARM == 1) o.warm[i] = kap;
        if (WARM == 2) o.warm[i] = o.warm[i] + kap;
        fixed = error_fixed(err);
    }
    if (o.accum) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) fixed += __shfl_down(fixed, off, 64);
        if ((threadIdx.x & 63) == 0 && fixed != 0) atomicAdd(o.accum, (unsigned long long)fixed);
    }
}

// correctDivergenceError_CUDA (DFSPHSolver.cu:308-329) / correctDensityError_CUDA (:138-158)
template <bool DIVIDE_BY_DT>
struct OpCorrect {
    GridDesc g; KernelConsts k;
    const int* csF; const float4* posm; const int* csB; const float4* bposm;
    const float* kappa; float3* vel;
    float dt;
    struct Body {
        const OpCorrect& o; float ki; float3 a;
        __device__ __forceinline__ void fluid(int j, float3 d, float r2, float mj)
        {
            a = add3(a, smul3(mj * (ki + o.kappa[j]), kGradW(d, q_of(sqrtf(r2), o.k), o.k)));
        }
        __device__ __forceinline__ void boundary(int, float3 d, float r2, float mj)
        {
            a = add3(a, smul3(mj * ki, kGradW(d, q_of(sqrtf(r2), o.k), o.k)));
        }
    };
    __device__ void operator()(int i) const
    {
        Body b{*this, kappa[i], v3(0, 0, 0)};
        sweep27<true, true>(g, k, csF, posm, csB, bposm, xyz(posm[i]), b);
        vel[i] = add3(vel[i], DIVIDE_BY_DT ? div3s(b.a, dt) : b.a);
    }
};

// =================================================================================== PBD
// computeDensityLambda_CUDA, PBDSolver.cu:127-168.  `rb` is (float)(bool)rho0;
// dividing by 1.0f is the identity, so the division is only performed when rb != 1.
struct OpLambda {
    GridDesc g; KernelConsts k;
    const int* csF; const float4* posm; const int* csB; const float4* bposm;
    float* density; float* lambda;
    float rho0, rb, relaxation;
    struct Body {
        const OpLambda& o; float den, sl; float3 gs;
        __device__ __forceinline__ void term(float3 d, float r2, float mj)
        {
            const float q = q_of(sqrtf(r2), o.k);
            den += mj * kW(q, o.k);
            float3 gr = smul3(-mj, kGradW(d, q, o.k));
            if (o.rb != 1.0f) gr = div3s(gr, o.rb);
            gs = sub3(gs, gr);
            sl += dot3(gr, gr);
        }
        __device__ __forceinline__ void fluid(int, float3 d, float r2, float mj) { term(d, r2, mj); }
        __device__ __forceinline__ void boundary(int, float3 d, float r2, float mj) { term(d, r2, mj); }
    };
    __device__ void operator()(int i) const
    {
        Body b{*this, 0.0f, 0.0f, v3(0, 0, 0)};
        sweep27<true, true>(g, k, csF, posm, csB, bposm, xyz(posm[i]), b);
        density[i] = b.den;
        float lam = (b.den > rho0) ? (-(b.den / rho0 - 1.0f) / (dot3(b.gs, b.gs) + b.sl + kEps)) : 0.0f;
        lam *= relaxation;
        lambda[i] = lam;
    }
};

// computeDeltaPos_CUDA, PBDSolver.cu:170-210
struct OpDeltaPos {
    GridDesc g; KernelConsts k;
    const int* csF; const float4* posm; const int* csB; const float4* bposm;
    const float* lambda; float3* deltaPos;
    float rho0;
    struct Body {
        const OpDeltaPos& o; float li; float3 a;
        __device__ __forceinline__ void fluid(int j, float3 d, float r2, float mj)
        {
            a = add3(a, smul3(mj * (li + o.lambda[j]), kGradW(d, q_of(sqrtf(r2), o.k), o.k)));
        }
        __device__ __forceinline__ void boundary(int, float3 d, float r2, float mj)
        {
            a = add3(a, smul3(mj * li, kGradW(d, q_of(sqrtf(r2), o.k), o.k)));
        }
    };
    __device__ void operator()(int i) const
    {
        Body b{*this, lambda[i], v3(0, 0, 0)};
        sweep27<true, true>(g, k, csF, posm, csB, bposm, xyz(posm[i]), b);
        deltaPos[i] = div3s(b.a, rho0);
    }
};

// XSPHViscosity_CUDA, PBDSolver.cu:89-115, Jacobi form: reads vel, writes velOut
struct OpXsph {
    GridDesc g; KernelConsts k;
    const int* csF; const float4* posm; const float3* vel; float3* velOut;
    float c, rho0;
    struct Body {
        const OpXsph& o; float3 vi; float3 a;
        __device__ __forceinline__ void fluid(int j, float3, float r2, float mj)
        {
            a = add3(a, mul3s(smul3(mj, sub3(o.vel[j], vi)), kW(q_of(sqrtf(r2), o.k), o.k)));
        }
        __device__ __forceinline__ void boundary(int, float3, float, float) {}
    };
    __device__ void operator()(int i) const
    {
        Body b{*this, vel[i], v3(0, 0, 0)};
        sweep27<true, false>(g, k, csF, posm, nullptr, nullptr, xyz(posm[i]), b);
        velOut[i] = add3(b.vi, div3s(smul3(c, b.a), rho0));
    }
};

// =================================================================================== element-wise
static __global__ void k_add_const3(float3* __restrict__ v, float3 c, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) v[i] = add3(v[i], c);
}
static __global__ void k_add3(float3* __restrict__ v, const float3* __restrict__ w, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) v[i] = add3(v[i], w[i]);
}
// Particles::advect + enforceBoundary_CUDA(pos, vel): Particles.cu:28-36, BasicSPHSolver.cu:85-101
static __global__ void k_advect_clamp(float3* __restrict__ pos, float3* __restrict__ vel, float dt, float3 space, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float3 v = vel[i];
    float3 p = add3(pos[i], smul3(dt, v));
    clamp_box<true>(p, v, space);
    pos[i] = p;
    vel[i] = v;
}
// pos += deltaPos; enforceBoundary_CUDA(pos): PBDSolver.cu:212-223, :247-253 (also refreshes posm)
static __global__ void k_apply_delta_clamp(float3* __restrict__ pos, float4* __restrict__ posm,
                                           const float3* __restrict__ dpos, float3 space, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float3 p = add3(pos[i], dpos[i]);
    float3 unused = v3(0, 0, 0);
    clamp_box<false>(p, unused, space);
    pos[i] = p;
    posm[i] = make_float4(p.x, p.y, p.z, posm[i].w);
}
// vel = (pos - posLast) / dt, PBDSolver.cu:55-60
static __global__ void k_velocity_from_displacement(float3* __restrict__ vel, const float3* __restrict__ pos,
                                                    const float3* __restrict__ posLast, float dt, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) vel[i] = div3s(sub3(pos[i], posLast[i]), dt);
}

inline void launch_add_const3(float3* v, float3 c, int n)
{
    if (n > 0) k_add_const3<<<blocks_for(n), 256, 0, stream()>>>(v, c, n);
}
inline void launch_add3(float3* v, const float3* w, int n)
{
    if (n > 0) k_add3<<<blocks_for(n), 256, 0, stream()>>>(v, w, n);
}
inline void launch_advect_clamp(float3* pos, float3* vel, float dt, float3 space, int n)
{
    if (n > 0) k_advect_clamp<<<blocks_for(n), 256, 0, stream()>>>(pos, vel, dt, space, n);
}
inline void launch_apply_delta_clamp(float3* pos, float4* posm, const float3* dpos, float3 space, int n)
{
    if (n > 0) k_apply_delta_clamp<<<blocks_for(n), 256, 0, stream()>>>(pos, posm, dpos, space, n);
}
inline void launch_velocity_from_displacement(float3* vel, const float3* pos, const float3* posLast, float dt, int n)
{
    if (n > 0) k_velocity_from_displacement<<<blocks_for(n), 256, 0, stream()>>>(vel, pos, posLast, dt, n);
}

}  // namespace sphx
