/*
 * sph_oracle.c — CPU restatement of the reference SPH hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This file is the parity oracle for the HIP engine in cpp-fluid-particles_amd/.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may build, load or call it; the
 * product never links it and has no CPU fallback.
 *
 * PARITY PINNING: the reference (zhai-xiao/CPP-Fluid-Particles) ships no tests, fixtures or
 * golden vectors, and its CUDA sources cannot be built in this image (needs nvcc, thrust,
 * the un-vendored CUDA-Samples helper_math.h; shimming those is not allowed).  The oracle is
 * therefore "parity unpinned" by reference-owned vectors.  It is anchored instead on (1) a
 * line-by-line restatement of the reference arithmetic, each function citing the file:line it
 * follows, (2) the known-answer digits SURVEY.md 8(c) recorded from the survey's own CPU run
 * of the reference sources (tests/test_cpu_oracle.py), and (3) since r03, CRC-32 anchors of the
 * complete pos / vel / density arrays of a diagnostic CPU build of the reference's own sources
 * (stand-in CUDA headers, done once in a scratch directory, nothing of it committed) at every
 * 10th..50th step through the landing of the column, for all three solvers: this file was
 * bit-identical to that build on every value (tests/golden/refsrc_anchors.json, README.md there).
 * That build is evidence, not a sanctioned reference build: the status stays "parity unpinned".
 *
 * Arithmetic contract: IEEE-754 binary32 add/mul/div/sqrt, no FMA contraction
 * (build with -ffp-contract=off), summation in the reference's loop order.  Deliberate, documented
 * deviations from the literal source (all three are places where the reference's own result is
 * toolchain- or schedule-dependent):
 *   D1  x^7 in the Tait EOS (BasicSPHSolver.cu:108 uses powf) is a fixed fp64 multiply chain
 *       rounded once to fp32 (pow7_mode=0).  pow7_mode=1 uses libm powf like a literal CPU build.
 *   D2  The DFSPH termination sum (DFSPHSolver.cu:206,360, thrust::reduce, unspecified order) is
 *       an exact 2^-32 fixed-point integer sum, so it is order- and partition-independent.
 *   D3  PBD XSPH (PBDSolver.cu:89-115) is racy in the reference (reads vel[j] while writing
 *       vel[i]); xsph_mode=0 is Jacobi (read old, write new), xsph_mode=1 is the serial-order
 *       in-place result a one-thread CPU build would give.
 * The `bool rho0` quirk (PBDSolver.cu:127-133) is reproduced: division by 1.0f.
 *
 * Parallelism: OpenMP over particles only; every per-particle sum keeps its order, so results are
 * bit-identical for any thread count.
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORACLE_EPS 1e-6f                       /* global.h:21 */
#define ORACLE_PI 3.14159265358979323846f      /* global.h:22 */
#define ORACLE_MAX_A 1000.0f                   /* global.h:26 */

typedef struct { float x, y, z; } f3;

/* Mirrors include/sphx_c.h:sphx_params field-for-field (tests assert equal sizeof). */
typedef struct {
    float space[3];
    int   cells[3];
    float cell_length, radius, dt, m0, rho0, rho_boundary, stiff, visc;
    float surface_tension, air_pressure;
    float gravity[3];
    int   solver;            /* 0 WCSPH, 1 DFSPH, 2 PBD */
    float dfsph_density_thr, dfsph_divergence_thr;
    int   dfsph_max_iter;
    int   dfsph_fixed_div;   /* <0: adaptive (reference loop); >=0: exactly that many iterations */
    int   dfsph_fixed_den;
    int   pbd_iters;
    float pbd_xsph_c, pbd_relaxation;
    int   pow7_mode;         /* D1 */
    int   xsph_mode;         /* D3 */
    int   reserved[4];
} oracle_params;

typedef struct {
    oracle_params P;
    int n, nb, C, cap;
    f3 *pos, *vel;  float *pressure, *density, *mass;  int *p2c;  int *ids;
    f3 *bpos, *bvel; float *bmass; int *bp2c;
    int *csF, *csB;                 /* cellStart arrays, C+1 entries */
    f3 *buf3;                       /* BasicSPHSolver::bufferFloat3 (deltaV / colour gradient) */
    float *alpha, *kappa, *error, *warm;       /* DFSPHSolver.h:57-61 */
    f3 *pos_last, *dpos; float *lambda; int pos_last_init;   /* PBDSolver.h:76-84 */
    f3 *tmp3; float *tmp1; int *tmpi; int *fill;  /* sort scratch */
    int it_div, it_den;
    long long steps;
    float visc_r6;                  /* powf(R,6), hoisted: CUDAFunctions.cuh:53 */
} oracle_sys;

/* ------------------------------------------------------------------ helper_math semantics ---- */
static inline f3 mk3(float x, float y, float z) { f3 r = { x, y, z }; return r; }
static inline f3 add3(f3 a, f3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline f3 sub3(f3 a, f3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline f3 neg3(f3 a) { return mk3(-a.x, -a.y, -a.z); }
static inline f3 mul3s(f3 a, float s) { return mk3(a.x * s, a.y * s, a.z * s); }
static inline f3 smul3(float s, f3 a) { return mk3(s * a.x, s * a.y, s * a.z); }
static inline f3 div3s(f3 a, float s) { return mk3(a.x / s, a.y / s, a.z / s); }   /* true division */
static inline float dot3(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline float len3(f3 a) { return sqrtf(dot3(a, a)); }


/* fmaxf/fminf restated with explicit compares so that signed zeros and NaNs behave the same on
 * every toolchain (PTX max/min order -0 < +0; libm and v_max_f32 need not agree bit-for-bit):
 *   max_eps(x) = fmaxf(EPSILON, x);  max0(x) = fmaxf(x, 0.0f);  min0(x) = fminf(x, 0.0f).       */
static inline float max_eps(float x) { return (x > ORACLE_EPS) ? x : ORACLE_EPS; }
static inline float max0(float x) { return (x > 0.0f) ? x : 0.0f; }
static inline float min0(float x) { return (x < 0.0f) ? x : ((x == 0.0f) ? x : 0.0f); }

/* ------------------------------------------------------------- CUDAFunctions.cuh kernels ---- */
/* cubic_spline_kernel, CUDAFunctions.cuh:23-35 */
/* g_w_promote (oracle_set_w_promote, default 0): how `fabs(r)` at CUDAFunctions.cuh:25 resolves.
 *   0  float fabs(float) -- what nvcc's device headers (and MSVC's <cmath>) provide: q is fp32.  THE contract.
 *   1  C's double fabs(double) -- what a g++/clang host compile of the same text picks when only <cmath> is
 *      included: `const auto q` becomes a double and the whole polynomial is evaluated in fp64, rounded once
 *      on return.  SURVEY.md 8(c)'s known answers were recorded from such a host build; this mode exists
 *      only so that tests can reproduce those digits (DFSPH step 100: 0.892478) next to mode 0's (0.892446). */
static int g_w_promote = 0;
static inline float kW_promoted(float r, float R)
{
    const double q = 2.0f * fabs(r) / R;
    if (q > 2.0f || q < ORACLE_EPS) return 0.0f;
    const float a = 0.25f / (ORACLE_PI * R * R * R);
    return (float)(a * ((q > 1.0f) ? (2.0f - q) * (2.0f - q) * (2.0f - q) : ((3.0f * q - 6.0f) * q * q + 4.0f)));
}
static inline float kW(float r, float R)
{
    if (g_w_promote) return kW_promoted(r, R);
    const float q = 2.0f * fabsf(r) / R;
    if (q > 2.0f || q < ORACLE_EPS) return 0.0f;
    const float a = 0.25f / (ORACLE_PI * R * R * R);
    return a * ((q > 1.0f) ? (2.0f - q) * (2.0f - q) * (2.0f - q) : ((3.0f * q - 6.0f) * q * q + 4.0f));
}
/* cubic_spline_kernel_gradient, CUDAFunctions.cuh:37-50 */
static inline f3 kGradW(f3 r, float R)
{
    const float q = 2.0f * len3(r) / R;
    if (q > 2.0f) return mk3(0.0f, 0.0f, 0.0f);
    const f3 a = div3s(r, ORACLE_PI * (q + ORACLE_EPS) * R * R * R * R * R);
    return mul3s(a, (q > 1.0f) ? ((12.0f - 3.0f * q) * q - 12.0f) : ((9.0f * q - 12.0f) * q));
}
/* viscosity_kernel_laplacian, CUDAFunctions.cuh:52-54 (powf(R,6) hoisted into r6) */
static inline float kViscLap(float r, float R, float r6)
{
    return (r <= R) ? (45.0f * (R - r) / (ORACLE_PI * r6)) : 0.0f;
}
static inline float cube(float x) { return x * x * x; }
/* surface_tension_kernel_gradient, CUDAFunctions.cuh:82-98 */
static inline f3 kSurfGrad(f3 r, float R)
{
    const float x = len3(r);
    if (x > R || x < ORACLE_EPS) return mk3(0.0f, 0.0f, 0.0f);
    const f3 a = div3s(smul3(136.0241f, neg3(r)), ORACLE_PI * cube(R) * cube(R) * cube(R) * x);
    return mul3s(a, (2.0f * x <= R) ? (2.0f * cube(R - x) * cube(x) - 0.0156f * cube(R) * cube(R))
                                    : (cube(R - x) * cube(x)));
}
/* particlePos2cellIdx, CUDAFunctions.cuh:64-70 */
static inline int cell_id(int x, int y, int z, const int *cs)
{
    return (x >= 0 && x < cs[0] && y >= 0 && y < cs[1] && z >= 0 && z < cs[2])
               ? ((x * cs[1] + y) * cs[2] + z) : (cs[0] * cs[1] * cs[2]);
}
/* make_int3(pos / cellLength): fp32 division, truncation toward zero (helper_math.h).  g_xoff is the
 * global x index of local cell column 0 (0 for a whole domain; slab tests use x0-1). */
static int g_xoff = 0;
static inline void cell_of(f3 p, float cl, int *c)
{
    c[0] = (int)(p.x / cl) - g_xoff; c[1] = (int)(p.y / cl); c[2] = (int)(p.z / cl);
}

/* --------------------------------------------------------------------------- neighbour grid */
/* Stable sort-by-key of a payload (thrust::sort_by_key on int keys is a stable radix sort).  */
static void stable_perm(const int *keys, int n, int C, int *fill, int *perm)
{
    memset(fill, 0, sizeof(int) * (size_t)(C + 2));
    for (int i = 0; i < n; ++i) fill[keys[i] + 1]++;
    for (int c = 0; c <= C; ++c) fill[c + 1] += fill[c];
    for (int i = 0; i < n; ++i) perm[fill[keys[i]]++] = i;     /* perm[new] = old */
}
static void gather3(f3 *a, f3 *tmp, const int *perm, int n)
{
    for (int q = 0; q < n; ++q) tmp[q] = a[perm[q]];
    memcpy(a, tmp, sizeof(f3) * (size_t)n);
}
static void gather1(float *a, float *tmp, const int *perm, int n)
{
    for (int q = 0; q < n; ++q) tmp[q] = a[perm[q]];
    memcpy(a, tmp, sizeof(float) * (size_t)n);
}
static void gatheri(int *a, int *tmp, const int *perm, int n)
{
    for (int q = 0; q < n; ++q) tmp[q] = a[perm[q]];
    memcpy(a, tmp, sizeof(int) * (size_t)n);
}

/* SPHSystem::neighborSearch, SPHSystem.cu:114-127.  p2c stays in PRE-sort order. */
static void neighbor_search(oracle_sys *s, f3 *pos, f3 *vel, int *p2c, int *ids, int n, int *cellStart)
{
    const oracle_params *P = &s->P;
    for (int i = 0; i < n; ++i) {
        int c[3]; cell_of(pos[i], P->cell_length, c);
        p2c[i] = cell_id(c[0], c[1], c[2], P->cells);
    }
    stable_perm(p2c, n, s->C, s->fill, s->tmpi);
    gather3(pos, s->tmp3, s->tmpi, n);
    gather3(vel, s->tmp3, s->tmpi, n);
    if (ids) {
        int *t = (int *)s->tmp1;       /* tmp1 is sized >= n floats == n ints */
        gatheri(ids, t, s->tmpi, n);
    }
    memset(cellStart, 0, sizeof(int) * (size_t)(s->C + 1));
    for (int i = 0; i < n; ++i) cellStart[p2c[i]]++;
    int run = 0;
    for (int c = 0; c <= s->C; ++c) { int k = cellStart[c]; cellStart[c] = run; run += k; }
}
/* the "re-apply this step's permutation" trick: DFSPHSolver.cu:170-171, PBDSolver.cu:84-85 */
static void resort_float(oracle_sys *s, float *a)
{
    stable_perm(s->p2c, s->n, s->C, s->fill, s->tmpi);
    gather1(a, s->tmp1, s->tmpi, s->n);
}
static void resort_f3(oracle_sys *s, f3 *a)
{
    stable_perm(s->p2c, s->n, s->C, s->fill, s->tmpi);
    gather3(a, s->tmp3, s->tmpi, s->n);
}

/* Sweep skeleton: m -> (m/9-1, (m%9)/3-1, m%3-1), skip out-of-grid cells,
 * per cell the fluid range then the boundary range, j ascending.                         */
#define SWEEP_BEGIN(POS_I)                                                                   \
    {   int c0_[3]; cell_of((POS_I), P->cell_length, c0_);                                   \
        for (int m_ = 0; m_ < 27; ++m_) {                                                    \
            const int cid = cell_id(c0_[0] + m_ / 9 - 1, c0_[1] + (m_ % 9) / 3 - 1,          \
                                    c0_[2] + m_ % 3 - 1, P->cells);                          \
            if (cid == s->C) continue;
#define SWEEP_END }}

/* computeBoundaryMass_CUDA, SPHSystem.cu:79-105 */
static void boundary_mass(oracle_sys *s)
{
    const oracle_params *P = &s->P;
#pragma omp parallel for schedule(static)
    for (int i = 0; i < s->nb; ++i) {
        float sum = 0.0f;
        SWEEP_BEGIN(s->bpos[i])
            for (int j = s->csB[cid]; j < s->csB[cid + 1]; ++j)
                sum += kW(len3(sub3(s->bpos[i], s->bpos[j])), P->radius);
        SWEEP_END
        s->bmass[i] = P->rho_boundary / max_eps(sum);
    }
}

/* ------------------------------------------------------------------- BasicSPHSolver (WCSPH) */
/* force, BasicSPHSolver.cu:227-235 */
static void k_force(oracle_sys *s)
{
    const oracle_params *P = &s->P;
    const f3 dv = mk3(P->dt * P->gravity[0], P->dt * P->gravity[1], P->dt * P->gravity[2]);
#pragma omp parallel for schedule(static)
    for (int i = 0; i < s->n; ++i) s->vel[i] = add3(s->vel[i], dv);
}
/* viscosity_CUDA + add, BasicSPHSolver.cu:183-225 */
static void k_viscosity(oracle_sys *s)
{
    const oracle_params *P = &s->P;
#pragma omp parallel for schedule(static)
    for (int i = 0; i < s->n; ++i) {
        f3 a = mk3(0, 0, 0);
        SWEEP_BEGIN(s->pos[i])
            for (int j = s->csF[cid]; j < s->csF[cid + 1]; ++j)
                a = add3(a, mul3s(smul3(s->mass[j], div3s(sub3(s->vel[j], s->vel[i]), P->rho0)),
                                  kViscLap(len3(sub3(s->pos[i], s->pos[j])), P->radius, s->visc_r6)));
        SWEEP_END
        s->buf3[i] = mul3s(smul3(P->visc, a), P->dt);
    }
#pragma omp parallel for schedule(static)
    for (int i = 0; i < s->n; ++i) s->vel[i] = add3(s->vel[i], s->buf3[i]);
}
/* computeColorGrad_CUDA, BasicSPHSolver.cu:277-318 */
static void k_color_grad(oracle_sys *s)
{
    const oracle_params *P = &s->P;
#pragma omp parallel for schedule(static)
    for (int i = 0; i < s->n; ++i) {
        f3 cg = mk3(0, 0, 0); float den = 0.0f;
        const f3 pi = s->pos[i];
        SWEEP_BEGIN(pi)
            for (int j = s->csF[cid]; j < s->csF[cid + 1]; ++j) {
                const f3 d = sub3(pi, s->pos[j]);
                cg = add3(cg, smul3(s->mass[j] / P->rho0, kGradW(d, P->radius)));
                den += s->mass[j] / P->rho0 * kW(len3(d), P->radius);
            }
            for (int j = s->csB[cid]; j < s->csB[cid + 1]; ++j) {
                const f3 d = sub3(pi, s->bpos[j]);
                cg = add3(cg, smul3(s->bmass[j] / P->rho_boundary, kGradW(d, P->radius)));
                den += s->bmass[j] / P->rho_boundary * kW(len3(d), P->radius);
            }
        SWEEP_END
        s->buf3[i] = div3s(cg, max_eps(den));
    }
}
/* surfaceTensionAndAirPressure_CUDA, BasicSPHSolver.cu:332-370 */
static void k_surface(oracle_sys *s)
{
    const oracle_params *P = &s->P;
    const f3 *cgv = s->buf3;
#pragma omp parallel for schedule(static)
    for (int i = 0; i < s->n; ++i) {
        f3 a = mk3(0, 0, 0);
        const f3 pi = s->pos[i];
        SWEEP_BEGIN(pi)
            for (int j = s->csF[cid]; j < s->csF[cid + 1]; ++j) {
                const f3 d = sub3(pi, s->pos[j]);
                a = add3(a, smul3(0.25f * s->mass[j] / (P->rho0 * P->rho0) * P->surface_tension
                                      * (dot3(cgv[i], cgv[i]) + dot3(cgv[j], cgv[j])),
                                  kSurfGrad(d, P->radius)));
                a = add3(a, div3s(mul3s(smul3(P->air_pressure * s->mass[j] / (P->rho0 * P->rho0),
                                              kGradW(d, P->radius)),
                                        len3(cgv[i])),
                                  max_eps(len3(cgv[i]))));
            }
        SWEEP_END
        s->tmp3[i] = add3(s->vel[i], mul3s(a, P->dt));   /* no neighbour reads vel here */
    }
    memcpy(s->vel, s->tmp3, sizeof(f3) * (size_t)s->n);
}
static void k_handle_surface(oracle_sys *s)       /* BasicSPHSolver.cu:262-275 */
{
    k_color_grad(s);
    k_surface(s);
}
/* computeDensity_CUDA, BasicSPHSolver.cu:32-83 */
static void k_density(oracle_sys *s)
{
    const oracle_params *P = &s->P;
#pragma omp parallel for schedule(static)
    for (int i = 0; i < s->n; ++i) {
        float den = 0.0f;
        const f3 pi = s->pos[i];
        SWEEP_BEGIN(pi)
            for (int j = s->csF[cid]; j < s->csF[cid + 1]; ++j)
                den += s->mass[j] * kW(len3(sub3(pi, s->pos[j])), P->radius);
            for (int j = s->csB[cid]; j < s->csB[cid + 1]; ++j)
                den += s->bmass[j] * kW(len3(sub3(pi, s->bpos[j])), P->radius);
        SWEEP_END
        s->density[i] = den;
    }
}
static inline float pow7(float x, int mode)
{
    if (mode == 1) return powf(x, 7);
    const double d = (double)x, d2 = d * d, d4 = d2 * d2, d6 = d4 * d2;
    return (float)(d6 * d);
}
/* computePressure_CUDA, BasicSPHSolver.cu:103-111 */
static void k_pressure(oracle_sys *s)
{
    const oracle_params *P = &s->P;
#pragma omp parallel for schedule(static)
    for (int i = 0; i < s->n; ++i) {
        float p = P->stiff * (pow7(s->density[i] / P->rho0, P->pow7_mode) - 1.0f);
        if (p < 0.0f) p = 0.0f;
        s->pressure[i] = p;
    }
}
/* pressureForce_CUDA, BasicSPHSolver.cu:113-165 */
static void k_pressure_force(oracle_sys *s)
{
    const oracle_params *P = &s->P;
#pragma omp parallel for schedule(static)
    for (int i = 0; i < s->n; ++i) {
        f3 a = mk3(0, 0, 0);
        const f3 pi = s->pos[i];
        const float di = s->density[i], pri = s->pressure[i];
        SWEEP_BEGIN(pi)
            for (int j = s->csF[cid]; j < s->csF[cid + 1]; ++j) {
                if (i == j) continue;
                const float dj = s->density[j];
                a = add3(a, smul3(-s->mass[j] * (pri / max_eps(di * di)
                                                 + s->pressure[j] / max_eps(dj * dj)),
                                  kGradW(sub3(pi, s->pos[j]), P->radius)));
            }
            for (int j = s->csB[cid]; j < s->csB[cid + 1]; ++j)
                a = add3(a, smul3(-s->bmass[j] * (pri / max_eps(di * di)),
                                  kGradW(sub3(pi, s->bpos[j]), P->radius)));
        SWEEP_END
        if (len3(a) > ORACLE_MAX_A)   /* normalize(a) = a * (1/sqrt(dot)) (helper_math, IEEE) */
            a = mul3s(mul3s(a, 1.0f / sqrtf(dot3(a, a))), ORACLE_MAX_A);
        s->tmp3[i] = add3(s->vel[i], mul3s(a, P->dt));
    }
    memcpy(s->vel, s->tmp3, sizeof(f3) * (size_t)s->n);
}
/* Particles::advect + enforceBoundary_CUDA(pos,vel), Particles.cu:28-36, BasicSPHSolver.cu:85-101 */
static void k_advect(oracle_sys *s)
{
    const oracle_params *P = &s->P;
#pragma omp parallel for schedule(static)
    for (int i = 0; i < s->n; ++i) {
        f3 p = add3(s->pos[i], smul3(P->dt, s->vel[i]));
        f3 v = s->vel[i];
        const float lx = P->space[0] * .00f, hx = P->space[0] * .99f;
        const float ly = P->space[1] * .00f, hy = P->space[1] * .99f;
        const float lz = P->space[2] * .00f, hz = P->space[2] * .99f;
        if (p.x <= lx) { p.x = lx; v.x = max0(v.x); }
        if (p.x >= hx) { p.x = hx; v.x = min0(v.x); }
        if (p.y <= ly) { p.y = ly; v.y = max0(v.y); }
        if (p.y >= hy) { p.y = hy; v.y = min0(v.y); }
        if (p.z <= lz) { p.z = lz; v.z = max0(v.z); }
        if (p.z >= hz) { p.z = hz; v.z = min0(v.z); }
        s->pos[i] = p; s->vel[i] = v;
    }
}
/* BasicSPHSolver::step, BasicSPHSolver.cu:237-260 */
static void wcsph_step(oracle_sys *s)
{
    const oracle_params *P = &s->P;
    k_force(s);
    k_viscosity(s);
    if (P->surface_tension > ORACLE_EPS || P->air_pressure > ORACLE_EPS) k_handle_surface(s);
    k_density(s);
    k_pressure(s);
    k_pressure_force(s);
    k_advect(s);
}

/* ------------------------------------------------------------------------------ DFSPHSolver */
/* computeDensityAlpha_CUDA, DFSPHSolver.cu:212-249 */
static void k_density_alpha(oracle_sys *s)
{
    const oracle_params *P = &s->P;
#pragma omp parallel for schedule(static)
    for (int i = 0; i < s->n; ++i) {
        f3 gs = mk3(0, 0, 0); float sl = 0.0f, den = 0.0f;
        const f3 pi = s->pos[i];
        SWEEP_BEGIN(pi)
            for (int j = s->csF[cid]; j < s->csF[cid + 1]; ++j) {
                const f3 d = sub3(pi, s->pos[j]);
                den += s->mass[j] * kW(len3(d), P->radius);
                const f3 g = smul3(s->mass[j], kGradW(d, P->radius));
                gs = add3(gs, g);
                sl += dot3(g, g);
            }
            for (int j = s->csB[cid]; j < s->csB[cid + 1]; ++j) {
                const f3 d = sub3(pi, s->bpos[j]);
                den += s->bmass[j] * kW(len3(d), P->radius);
                gs = add3(gs, smul3(s->bmass[j], kGradW(d, P->radius)));
            }
        SWEEP_END
        s->density[i] = den;
        s->alpha[i] = -1.0f / max_eps(dot3(gs, gs) + sl);
    }
}
/* shared body of computeDivergenceError_CUDA (DFSPHSolver.cu:261-306) and
 * computeDensityError_CUDA (DFSPHSolver.cu:74-116): e = sum m_j (v_i - v_j).gradW           */
static inline float rate_sum(const oracle_sys *s, int i)
{
    const oracle_params *P = &s->P;
    float e = 0.0f;
    const f3 pi = s->pos[i], vi = s->vel[i];
    SWEEP_BEGIN(pi)
        for (int j = s->csF[cid]; j < s->csF[cid + 1]; ++j)
            e += s->mass[j] * dot3(sub3(vi, s->vel[j]), kGradW(sub3(pi, s->pos[j]), P->radius));
        for (int j = s->csB[cid]; j < s->csB[cid + 1]; ++j)
            e += s->bmass[j] * dot3(vi, kGradW(sub3(pi, s->bpos[j]), P->radius));
    SWEEP_END
    return e;
}
static void k_divergence_error(oracle_sys *s)
{
    const oracle_params *P = &s->P;
#pragma omp parallel for schedule(static)
    for (int i = 0; i < s->n; ++i) {
        float err = max0(rate_sum(s, i));
        if (s->density[i] + P->dt * err < P->rho0 && s->density[i] <= P->rho0) err = 0.0f;
        s->error[i] = err;
        s->kappa[i] = err * s->alpha[i];
    }
}
static void k_density_error(oracle_sys *s)
{
    const oracle_params *P = &s->P;
#pragma omp parallel for schedule(static)
    for (int i = 0; i < s->n; ++i) {
        const float err = max0(P->dt * rate_sum(s, i) + s->density[i] - P->rho0);
        s->error[i] = err;
        s->kappa[i] = err * s->alpha[i];
    }
}
/* correctDivergenceError_CUDA (DFSPHSolver.cu:308-329) / correctDensityError_CUDA (:138-158) */
static void k_correct(oracle_sys *s, const float *kap, int divide_by_dt)
{
    const oracle_params *P = &s->P;
#pragma omp parallel for schedule(static)
    for (int i = 0; i < s->n; ++i) {
        f3 a = mk3(0, 0, 0);
        const f3 pi = s->pos[i];
        const float ki = kap[i];
        SWEEP_BEGIN(pi)
            for (int j = s->csF[cid]; j < s->csF[cid + 1]; ++j)
                a = add3(a, smul3(s->mass[j] * (ki + kap[j]), kGradW(sub3(pi, s->pos[j]), P->radius)));
            for (int j = s->csB[cid]; j < s->csB[cid + 1]; ++j)
                a = add3(a, smul3(s->bmass[j] * ki, kGradW(sub3(pi, s->bpos[j]), P->radius)));
        SWEEP_END
        s->vel[i] = add3(s->vel[i], divide_by_dt ? div3s(a, P->dt) : a);
    }
}
/* D2: exact fixed-point |error| sum replacing thrust::reduce(abs_plus) */
static float error_total(const oracle_sys *s)
{
    long long acc = 0;
    for (int i = 0; i < s->n; ++i) {
        float e = fabsf(s->error[i]) * 4294967296.0f;
        if (!(e < 4.0e18f)) e = 4.0e18f;          /* saturate (also catches NaN/inf) */
        acc += (long long)e;
    }
    return (float)((double)acc * (1.0 / 4294967296.0));
}
/* DFSPHSolver::correctDivergenceError, DFSPHSolver.cu:331-363 */
static int dfsph_divergence_solve(oracle_sys *s)
{
    const oracle_params *P = &s->P;
    float total = 3.402823466e+38f;
    int iter = 0;
    k_divergence_error(s);
    for (;;) {
        const int go = (P->dfsph_fixed_div >= 0)
            ? (iter < P->dfsph_fixed_div)
            : ((iter < 1 || total > P->dfsph_divergence_thr * s->n * P->rho0) && iter < P->dfsph_max_iter);
        if (!go) break;
        k_correct(s, s->kappa, 0);
        k_divergence_error(s);
        ++iter;
        if (P->dfsph_fixed_div < 0) total = error_total(s);
    }
    return iter;
}
/* DFSPHSolver::project, DFSPHSolver.cu:160-210 */
static int dfsph_density_solve(oracle_sys *s)
{
    const oracle_params *P = &s->P;
    float total = 3.402823466e+38f;
    int iter = 0;
    resort_float(s, s->warm);
    k_correct(s, s->warm, 1);                      /* warm start */
    k_density_error(s);
    memcpy(s->warm, s->kappa, sizeof(float) * (size_t)s->n);
    for (;;) {
        const int go = (P->dfsph_fixed_den >= 0)
            ? (iter < P->dfsph_fixed_den)
            : ((iter < 2 || total > P->dfsph_density_thr * s->n * P->rho0) && iter < P->dfsph_max_iter);
        if (!go) break;
        k_correct(s, s->kappa, 1);
        k_density_error(s);
#pragma omp parallel for schedule(static)
        for (int i = 0; i < s->n; ++i) s->warm[i] = s->warm[i] + s->kappa[i];
        ++iter;
        if (P->dfsph_fixed_den < 0 && iter >= 2) total = error_total(s);
    }
    return iter;
}
/* DFSPHSolver::step, DFSPHSolver.cu:33-72 */
static void dfsph_step(oracle_sys *s)
{
    const oracle_params *P = &s->P;
    k_density_alpha(s);
    s->it_div = dfsph_divergence_solve(s);
    k_force(s);
    k_viscosity(s);
    if (P->surface_tension > ORACLE_EPS || P->air_pressure > ORACLE_EPS) k_handle_surface(s);
    s->it_den = dfsph_density_solve(s);
    k_advect(s);
}

/* -------------------------------------------------------------------------------- PBDSolver */
/* computeDensityLambda_CUDA, PBDSolver.cu:127-168 (Q11: `/ rho0` is `/ (float)(bool)rho0`) */
static void k_density_lambda(oracle_sys *s)
{
    const oracle_params *P = &s->P;
    const float rb = (P->rho0 != 0.0f) ? 1.0f : 0.0f;
#pragma omp parallel for schedule(static)
    for (int i = 0; i < s->n; ++i) {
        f3 gs = mk3(0, 0, 0); float sl = 0.0f, den = 0.0f;
        const f3 pi = s->pos[i];
        SWEEP_BEGIN(pi)
            for (int j = s->csF[cid]; j < s->csF[cid + 1]; ++j) {
                const f3 d = sub3(pi, s->pos[j]);
                den += s->mass[j] * kW(len3(d), P->radius);
                const f3 g = div3s(smul3(-s->mass[j], kGradW(d, P->radius)), rb);
                gs = sub3(gs, g);
                sl += dot3(g, g);
            }
            for (int j = s->csB[cid]; j < s->csB[cid + 1]; ++j) {
                const f3 d = sub3(pi, s->bpos[j]);
                den += s->bmass[j] * kW(len3(d), P->radius);
                const f3 g = div3s(smul3(-s->bmass[j], kGradW(d, P->radius)), rb);
                gs = sub3(gs, g);
                sl += dot3(g, g);
            }
        SWEEP_END
        s->density[i] = den;
        float lam = (den > P->rho0) ? (-(den / P->rho0 - 1.0f) / (dot3(gs, gs) + sl + ORACLE_EPS)) : 0.0f;
        lam *= P->pbd_relaxation;
        s->lambda[i] = lam;
    }
}
/* computeDeltaPos_CUDA, PBDSolver.cu:170-210 */
static void k_delta_pos(oracle_sys *s)
{
    const oracle_params *P = &s->P;
#pragma omp parallel for schedule(static)
    for (int i = 0; i < s->n; ++i) {
        f3 a = mk3(0, 0, 0);
        const f3 pi = s->pos[i];
        const float li = s->lambda[i];
        SWEEP_BEGIN(pi)
            for (int j = s->csF[cid]; j < s->csF[cid + 1]; ++j)
                a = add3(a, smul3(s->mass[j] * (li + s->lambda[j]), kGradW(sub3(pi, s->pos[j]), P->radius)));
            for (int j = s->csB[cid]; j < s->csB[cid + 1]; ++j)
                a = add3(a, smul3(s->bmass[j] * li, kGradW(sub3(pi, s->bpos[j]), P->radius)));
        SWEEP_END
        s->dpos[i] = div3s(a, P->rho0);
    }
}
/* pos += dpos; enforceBoundary_CUDA(pos): PBDSolver.cu:212-223, :247-253 */
static void k_apply_dpos(oracle_sys *s)
{
    const oracle_params *P = &s->P;
#pragma omp parallel for schedule(static)
    for (int i = 0; i < s->n; ++i) {
        f3 p = add3(s->pos[i], s->dpos[i]);
        const float lx = P->space[0] * .00f, hx = P->space[0] * .99f;
        const float ly = P->space[1] * .00f, hy = P->space[1] * .99f;
        const float lz = P->space[2] * .00f, hz = P->space[2] * .99f;
        if (p.x <= lx) p.x = lx;
        if (p.x >= hx) p.x = hx;
        if (p.y <= ly) p.y = ly;
        if (p.y >= hy) p.y = hy;
        if (p.z <= lz) p.z = lz;
        if (p.z >= hz) p.z = hz;
        s->pos[i] = p;
    }
}
/* XSPHViscosity_CUDA, PBDSolver.cu:89-115 (D3) */
static void k_xsph(oracle_sys *s)
{
    const oracle_params *P = &s->P;
    if (P->xsph_mode == 1) {                       /* serial in-place (Gauss-Seidel) */
        for (int i = 0; i < s->n; ++i) {
            f3 a = mk3(0, 0, 0);
            const f3 pi = s->pos[i];
            SWEEP_BEGIN(pi)
                for (int j = s->csF[cid]; j < s->csF[cid + 1]; ++j)
                    a = add3(a, mul3s(smul3(s->mass[j], sub3(s->vel[j], s->vel[i])),
                                      kW(len3(sub3(pi, s->pos[j])), P->radius)));
            SWEEP_END
            s->vel[i] = add3(s->vel[i], div3s(smul3(P->pbd_xsph_c, a), P->rho0));
        }
        return;
    }
#pragma omp parallel for schedule(static)
    for (int i = 0; i < s->n; ++i) {
        f3 a = mk3(0, 0, 0);
        const f3 pi = s->pos[i];
        SWEEP_BEGIN(pi)
            for (int j = s->csF[cid]; j < s->csF[cid + 1]; ++j)
                a = add3(a, mul3s(smul3(s->mass[j], sub3(s->vel[j], s->vel[i])),
                                  kW(len3(sub3(pi, s->pos[j])), P->radius)));
        SWEEP_END
        s->tmp3[i] = add3(s->vel[i], div3s(smul3(P->pbd_xsph_c, a), P->rho0));
    }
    memcpy(s->vel, s->tmp3, sizeof(f3) * (size_t)s->n);
}
/* PBDSolver::step, PBDSolver.cu:34-79.  Returns 1 if it "threw" (first call). */
static int pbd_step(oracle_sys *s)
{
    const oracle_params *P = &s->P;
    if (!s->pos_last_init) {
        memcpy(s->pos_last, s->pos, sizeof(f3) * (size_t)s->n);
        s->pos_last_init = 1;
        return 1;
    }
    resort_f3(s, s->pos_last);
    for (int it = 0; it < P->pbd_iters; ++it) {
        k_density_lambda(s);
        k_delta_pos(s);
        k_apply_dpos(s);
    }
#pragma omp parallel for schedule(static)
    for (int i = 0; i < s->n; ++i) s->vel[i] = div3s(sub3(s->pos[i], s->pos_last[i]), P->dt);
    k_xsph(s);
    if (P->surface_tension > ORACLE_EPS || P->air_pressure > ORACLE_EPS) k_handle_surface(s);
    k_force(s);
    memcpy(s->pos_last, s->pos, sizeof(f3) * (size_t)s->n);
    k_advect(s);
    return 0;
}

/* ------------------------------------------------------------------------------- public API */
static double now_ms(void)
{
    struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}

/* SPHSystem::step, SPHSystem.cu:129-158 */
float oracle_step(oracle_sys *s)
{
    const double t0 = now_ms();
    g_xoff = s->P.reserved[1];
    neighbor_search(s, s->pos, s->vel, s->p2c, s->ids, s->n, s->csF);
    switch (s->P.solver) {
    case 1: dfsph_step(s); break;
    case 2: pbd_step(s); break;
    default: wcsph_step(s); break;
    }
    s->steps++;
    return (float)(now_ms() - t0);
}

/* Stage-wise DFSPH step (same stage numbering as sphx_phase in include/sphx_c.h), used by the
 * world_size-2 gloo tests of the slab driver with this oracle standing in for the HIP engine. */
enum { PH_SEARCH = 0, PH_HEAD, PH_DIV_CORRECT, PH_DIV_ERROR, PH_FORCE, PH_VISC_COLOR, PH_SURFACE, PH_WARM_CORRECT,
       PH_DEN_ERROR_SET, PH_DEN_CORRECT, PH_DEN_ERROR_ACC, PH_ADVECT, PH_W_SEARCH, PH_W_PROPS, PH_W_SURFACE, PH_W_PRESSURE,
       PH_P_SEARCH, PH_P_LAMBDA, PH_P_DELTA, PH_P_VELOCITY, PH_P_XSPH, PH_P_SURFACE, PH_P_TAIL };
int oracle_run_phase(oracle_sys *s, int phase)
{
    const oracle_params *P = &s->P;
    const int surface = P->surface_tension > ORACLE_EPS || P->air_pressure > ORACLE_EPS;
    g_xoff = P->reserved[1];
    switch (phase) {
    case PH_SEARCH:
        neighbor_search(s, s->pos, s->vel, s->p2c, s->ids, s->n, s->csF);
        resort_float(s, s->warm);
        break;
    case PH_HEAD: k_density_alpha(s); k_divergence_error(s); break;
    case PH_DIV_CORRECT: k_correct(s, s->kappa, 0); break;
    case PH_DIV_ERROR: k_divergence_error(s); break;
    case PH_FORCE: k_force(s); break;
    case PH_VISC_COLOR: k_viscosity(s); if (surface) k_color_grad(s); break;
    case PH_SURFACE: if (surface) k_surface(s); break;
    case PH_WARM_CORRECT: k_correct(s, s->warm, 1); break;
    case PH_DEN_ERROR_SET: k_density_error(s); memcpy(s->warm, s->kappa, sizeof(float) * (size_t)s->n); break;
    case PH_DEN_CORRECT: k_correct(s, s->kappa, 1); break;
    case PH_DEN_ERROR_ACC:
        k_density_error(s);
        for (int i = 0; i < s->n; ++i) s->warm[i] = s->warm[i] + s->kappa[i];
        break;
    case PH_ADVECT: k_advect(s); s->steps++; break;
    case PH_W_SEARCH:
        neighbor_search(s, s->pos, s->vel, s->p2c, s->ids, s->n, s->csF);
        k_force(s);
        break;
    case PH_W_PROPS: k_viscosity(s); if (surface) k_color_grad(s); k_density(s); k_pressure(s); break;
    case PH_W_SURFACE: if (surface) k_surface(s); break;
    case PH_W_PRESSURE: k_pressure_force(s); break;
    /* PBDSolver::step (PBDSolver.cu:34-79) cut at the points where a sweep reads what a previous
     * one wrote for the neighbours */
    case PH_P_SEARCH:
        neighbor_search(s, s->pos, s->vel, s->p2c, s->ids, s->n, s->csF);
        resort_f3(s, s->pos_last);
        s->pos_last_init = 1;
        break;
    case PH_P_LAMBDA: k_density_lambda(s); break;
    case PH_P_DELTA: k_delta_pos(s); k_apply_dpos(s); break;
    case PH_P_VELOCITY:
        for (int i = 0; i < s->n; ++i) s->vel[i] = div3s(sub3(s->pos[i], s->pos_last[i]), P->dt);
        break;
    case PH_P_XSPH: k_xsph(s); if (surface) k_color_grad(s); break;
    case PH_P_SURFACE: if (surface) k_surface(s); break;
    case PH_P_TAIL:
        k_force(s);
        memcpy(s->pos_last, s->pos, sizeof(f3) * (size_t)s->n);
        k_advect(s);
        s->steps++;
        break;
    default: return -1;
    }
    return 0;
}

/* exact fixed-point |error| sum over particles [lo, hi) (a slab's owned range), D2 */
long long oracle_error_total_fixed(const oracle_sys *s, int lo, int hi)
{
    long long acc = 0;
    for (int i = lo; i < hi && i < s->n; ++i) {
        float e = fabsf(s->error[i]) * 4294967296.0f;
        if (!(e < 4.0e18f)) e = 4.0e18f;
        acc += (long long)e;
    }
    return acc;
}

void oracle_destroy(oracle_sys *s)
{
    if (!s) return;
    free(s->pos); free(s->vel); free(s->pressure); free(s->density); free(s->mass); free(s->p2c);
    free(s->ids); free(s->bpos); free(s->bvel); free(s->bmass); free(s->bp2c); free(s->csF);
    free(s->csB); free(s->buf3); free(s->alpha); free(s->kappa); free(s->error); free(s->warm);
    free(s->pos_last); free(s->dpos); free(s->lambda); free(s->tmp3); free(s->tmp1); free(s->tmpi);
    free(s->fill); free(s);
}

/* SPHSystem::SPHSystem, SPHSystem.cu:33-77: boundary search -> boundary mass ->
 * fluid mass fill -> fluid search -> one full step().                                          */
oracle_sys *oracle_create(const oracle_params *P, const float *fluid_xyz, int n,
                          const float *boundary_xyz, int nb, int run_ctor_step)
{
    oracle_sys *s = (oracle_sys *)calloc(1, sizeof(oracle_sys));
    s->P = *P; s->n = n; s->nb = nb; s->cap = n;
    s->C = P->cells[0] * P->cells[1] * P->cells[2];
    const size_t nm = (size_t)(n > nb ? n : nb) + 1;
    s->pos = calloc((size_t)n + 1, sizeof(f3)); s->vel = calloc((size_t)n + 1, sizeof(f3));
    s->pressure = calloc((size_t)n + 1, 4); s->density = calloc((size_t)n + 1, 4);
    s->mass = calloc((size_t)n + 1, 4); s->p2c = calloc((size_t)n + 1, 4); s->ids = calloc((size_t)n + 1, 4);
    s->bpos = calloc((size_t)nb + 1, sizeof(f3)); s->bvel = calloc((size_t)nb + 1, sizeof(f3));
    s->bmass = calloc((size_t)nb + 1, 4); s->bp2c = calloc((size_t)nb + 1, 4);
    s->csF = calloc((size_t)s->C + 2, 4); s->csB = calloc((size_t)s->C + 2, 4);
    s->buf3 = calloc((size_t)n + 1, sizeof(f3));
    s->alpha = calloc((size_t)n + 1, 4); s->kappa = calloc((size_t)n + 1, 4);
    s->error = calloc((size_t)n + 1, 4); s->warm = calloc((size_t)n + 1, 4);
    s->pos_last = calloc((size_t)n + 1, sizeof(f3)); s->dpos = calloc((size_t)n + 1, sizeof(f3));
    s->lambda = calloc((size_t)n + 1, 4);
    s->tmp3 = calloc(nm, sizeof(f3)); s->tmp1 = calloc(nm, 4); s->tmpi = calloc(nm, 4);
    s->fill = calloc((size_t)s->C + 3, 4);
    memcpy(s->pos, fluid_xyz, sizeof(f3) * (size_t)n);
    memcpy(s->bpos, boundary_xyz, sizeof(f3) * (size_t)nb);
    for (int i = 0; i < n; ++i) s->ids[i] = i;
    s->visc_r6 = powf(P->radius, 6);
    g_xoff = P->reserved[1];

    neighbor_search(s, s->bpos, s->bvel, s->bp2c, NULL, nb, s->csB);
    boundary_mass(s);
    for (int i = 0; i < n; ++i) s->mass[i] = P->m0;
    neighbor_search(s, s->pos, s->vel, s->p2c, s->ids, n, s->csF);
    if (run_ctor_step) oracle_step(s);
    return s;
}

enum { OF_POS = 0, OF_VEL, OF_DENSITY, OF_PRESSURE, OF_MASS, OF_CELL, OF_CELLSTART_F, OF_CELLSTART_B,
       OF_ID, OF_BPOS, OF_BMASS, OF_ALPHA, OF_KAPPA, OF_ERROR, OF_WARM, OF_POS_LAST, OF_LAMBDA, OF_BUF3 };

/* copies a field to dst; returns bytes written or -1 */
long long oracle_get(const oracle_sys *s, int field, void *dst, long long cap)
{
    const void *src = NULL; long long bytes = 0;
    switch (field) {
    case OF_POS: src = s->pos; bytes = 12LL * s->n; break;
    case OF_VEL: src = s->vel; bytes = 12LL * s->n; break;
    case OF_DENSITY: src = s->density; bytes = 4LL * s->n; break;
    case OF_PRESSURE: src = s->pressure; bytes = 4LL * s->n; break;
    case OF_MASS: src = s->mass; bytes = 4LL * s->n; break;
    case OF_CELL: src = s->p2c; bytes = 4LL * s->n; break;
    case OF_CELLSTART_F: src = s->csF; bytes = 4LL * (s->C + 1); break;
    case OF_CELLSTART_B: src = s->csB; bytes = 4LL * (s->C + 1); break;
    case OF_ID: src = s->ids; bytes = 4LL * s->n; break;
    case OF_BPOS: src = s->bpos; bytes = 12LL * s->nb; break;
    case OF_BMASS: src = s->bmass; bytes = 4LL * s->nb; break;
    case OF_ALPHA: src = s->alpha; bytes = 4LL * s->n; break;
    case OF_KAPPA: src = s->kappa; bytes = 4LL * s->n; break;
    case OF_ERROR: src = s->error; bytes = 4LL * s->n; break;
    case OF_WARM: src = s->warm; bytes = 4LL * s->n; break;
    case OF_POS_LAST: src = s->pos_last; bytes = 12LL * s->n; break;
    case OF_LAMBDA: src = s->lambda; bytes = 4LL * s->n; break;
    case OF_BUF3: src = s->buf3; bytes = 12LL * s->n; break;
    default: return -1;
    }
    if (bytes > cap) return -1;
    memcpy(dst, src, (size_t)bytes);
    return bytes;
}
/* overwrite pos / vel (for per-kernel tests on perturbed states) */
int oracle_set(oracle_sys *s, int field, const void *src, long long bytes)
{
    void *dst = NULL; long long want = 0;
    switch (field) {
    case OF_POS: dst = s->pos; want = 12LL * s->n; break;
    case OF_VEL: dst = s->vel; want = 12LL * s->n; break;
    case OF_WARM: dst = s->warm; want = 4LL * s->n; break;
    case OF_POS_LAST: dst = s->pos_last; want = 12LL * s->n; s->pos_last_init = 1; break;
    case OF_LAMBDA: dst = s->lambda; want = 4LL * s->n; break;
    case OF_ID: dst = s->ids; want = 4LL * s->n; break;
    case OF_KAPPA: dst = s->kappa; want = 4LL * s->n; break;
    case OF_BUF3: dst = s->buf3; want = 12LL * s->n; break;
    case OF_BMASS: dst = s->bmass; want = 4LL * s->nb; break;
    case OF_DENSITY: dst = s->density; want = 4LL * s->n; break;
    case OF_PRESSURE: dst = s->pressure; want = 4LL * s->n; break;
    default: return -1;
    }
    if (bytes != want) return -1;
    memcpy(dst, src, (size_t)bytes);
    return 0;
}
int oracle_set_count(oracle_sys *s, int n) { if (n < 0 || n > s->cap) return -1; s->n = n; return 0; }
void oracle_iters(const oracle_sys *s, int *div, int *den) { *div = s->it_div; *den = s->it_den; }
int oracle_sizeof_params(void) { return (int)sizeof(oracle_params); }
void oracle_set_threads(int t)
{
#ifdef _OPENMP
    if (t > 0) omp_set_num_threads(t);
#else
    (void)t;
#endif
}
void oracle_set_w_promote(int on) { g_w_promote = on ? 1 : 0; }
int oracle_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* Pointwise kernel evaluation for the device-function parity test (n samples). */
void oracle_eval_kernels(const float *r3, int n, float R, float *W_out, float *grad_out,
                         float *visc_out, float *surf_out)
{
    const float r6 = powf(R, 6);
    for (int i = 0; i < n; ++i) {
        const f3 r = mk3(r3[3 * i], r3[3 * i + 1], r3[3 * i + 2]);
        const float l = len3(r);
        W_out[i] = kW(l, R);
        const f3 g = kGradW(r, R);
        grad_out[3 * i] = g.x; grad_out[3 * i + 1] = g.y; grad_out[3 * i + 2] = g.z;
        visc_out[i] = kViscLap(l, R, r6);
        const f3 t = kSurfGrad(r, R);
        surf_out[3 * i] = t.x; surf_out[3 * i + 1] = t.y; surf_out[3 * i + 2] = t.z;
    }
}

/* ------------------------------------------------------------------------ scene generator --
 * main.cpp:54-117 restated, with the BASELINE.md §4 scaling rule: s = nx/24, box (s,s,s),
 * block nx x (3nx/2) x nx at (0.27s, 0.10s, 0.27s).  nx = 24 is the reference scene verbatim. */
void oracle_scene_params(int nx, oracle_params *P)
{
    memset(P, 0, sizeof(*P));
    const float s = (float)nx / 24.0f;
    const float spacing = 0.02f;
    P->space[0] = P->space[1] = P->space[2] = s;
    P->radius = 2.0f * spacing;
    P->cell_length = 1.01f * P->radius;
    for (int a = 0; a < 3; ++a) P->cells[a] = (int)ceilf(P->space[a] / P->cell_length);
    P->dt = 0.002f; P->m0 = 76.596750762082e-6f; P->rho0 = 1.0f; P->rho_boundary = 1.4f * P->rho0;
    P->stiff = 10.0f; P->visc = 5e-4f; P->surface_tension = 0.0001f; P->air_pressure = 0.0001f;
    P->gravity[0] = 0.0f; P->gravity[1] = -9.8f; P->gravity[2] = 0.0f;
    P->solver = 0;
    P->dfsph_density_thr = 1e-3f; P->dfsph_divergence_thr = 1e-3f; P->dfsph_max_iter = 20;
    P->dfsph_fixed_div = -1; P->dfsph_fixed_den = -1;
    P->pbd_iters = 20; P->pbd_xsph_c = 0.05f; P->pbd_relaxation = 0.75f;
}
int oracle_scene_counts(int nx, int *n_fluid, int *n_boundary)
{
    oracle_params P; oracle_scene_params(nx, &P);
    const int cx = 2 * P.cells[0], cy = 2 * P.cells[1], cz = 2 * P.cells[2];
    *n_fluid = nx * (3 * nx / 2) * nx;
    *n_boundary = 2 * cx * cy + 2 * cx * (cz - 2) + 2 * (cy - 2) * (cz - 2);
    return 0;
}
static inline void shell_pt(float *out, int i, int j, int k, const int *cs, const float *sp)
{
    const float a[3] = { (float)i / (float)(cs[0] - 1) * sp[0], (float)j / (float)(cs[1] - 1) * sp[1],
                         (float)k / (float)(cs[2] - 1) * sp[2] };
    for (int d = 0; d < 3; ++d) out[d] = 0.99f * a[d] + 0.005f * sp[d];
}
void oracle_scene_fill(int nx, float *fluid_xyz, float *boundary_xyz)
{
    oracle_params P; oracle_scene_params(nx, &P);
    const float s = P.space[0], spacing = 0.02f;
    const float ox = 0.27f * s, oy = 0.10f * s, oz = 0.27f * s;
    const int ny = 3 * nx / 2, nz = nx;
    size_t w = 0;
    for (int i = 0; i < ny; ++i) for (int j = 0; j < nx; ++j) for (int k = 0; k < nz; ++k) {
        fluid_xyz[w++] = ox + spacing * j; fluid_xyz[w++] = oy + spacing * i; fluid_xyz[w++] = oz + spacing * k;
    }
    const int cs[3] = { 2 * P.cells[0], 2 * P.cells[1], 2 * P.cells[2] };
    w = 0;
    for (int i = 0; i < cs[0]; ++i) for (int j = 0; j < cs[1]; ++j) {
        shell_pt(boundary_xyz + w, i, j, 0, cs, P.space); w += 3;
        shell_pt(boundary_xyz + w, i, j, cs[2] - 1, cs, P.space); w += 3;
    }
    for (int i = 0; i < cs[0]; ++i) for (int j = 0; j < cs[2] - 2; ++j) {
        shell_pt(boundary_xyz + w, i, 0, j + 1, cs, P.space); w += 3;
        shell_pt(boundary_xyz + w, i, cs[1] - 1, j + 1, cs, P.space); w += 3;
    }
    for (int i = 0; i < cs[1] - 2; ++i) for (int j = 0; j < cs[2] - 2; ++j) {
        shell_pt(boundary_xyz + w, 0, i + 1, j + 1, cs, P.space); w += 3;
        shell_pt(boundary_xyz + w, cs[0] - 1, i + 1, j + 1, cs, P.space); w += 3;
    }
}
