// sph_device.hpp — device-side arithmetic of the SPH hot path (gfx950).
//
// Everything here is written so that the bits produced are a pure function of the IEEE-754
// binary32 operations add / mul / div / sqrt applied in the reference's association order
// (SURVEY.md §2c); the translation units including this header are compiled with
// -ffp-contract=off and correctly-rounded division / square root.  Hoisting a sub-expression that
// depends only on constants (wA, viscDen, stK, stC) does not change any bit.
//
// Smoothing kernels follow the reference's src/CUDAFunctions.cuh:23-98; `q`, `r` and `d` are
// shared between W and gradW of the same pair because they are the same expressions there.
#pragma once

#include <hip/hip_runtime.h>

#include <type_traits>

namespace sphx {

constexpr float kEps = 1e-6f;                    // global.h:21
constexpr float kPi = 3.14159265358979323846f;   // global.h:22
constexpr float kMaxA = 1000.0f;                 // global.h:26

struct KernelConsts {
    float R;        // smoothing radius (support of every kernel)
    float wA;       // 0.25f / (PI*R*R*R)                      CUDAFunctions.cuh:32
    float viscDen;  // PI * powf(R, 6)                         CUDAFunctions.cuh:53
    float stK;      // PI*cube(R)*cube(R)*cube(R)              CUDAFunctions.cuh:93
    float stC;      // 0.0156f*cube(R)*cube(R)                 CUDAFunctions.cuh:95
    float tCut;     // largest squared distance at which ANY kernel can be non-zero (exact)
    // --- exact fast paths (validated on the device by validate_fast_math, else 0 = plain IEEE ops)
    float rcpR;     // RN(1/R)
    int fastQ;      // 1: x/R == fma-refined x*rcpR for EVERY float x in [0, 2.2R] (checked exhaustively)
    int fastDiv;    // 1: the denominators of gradW / surface gradient stay inside [2^-90, 2^16]
    int q2Free;     // 1: r2 <= tCut implies q <= 2 and r <= R, i.e. row entries never fail a support test
    float rcpViscDen;   // RN(1 / viscDen)
    int fastVisc;   // 1: x / viscDen == fma-refined x * rcpViscDen for EVERY float x in {0} U [2^-40, 46 R] (checked exhaustively)
    // --- tolerance arithmetic (sphx_params.reserved[3] = 1): hardware rsq / rcp, fused multiply-adds, constants folded
    int tol;        // 1: row walks use the pair_tol bodies
    float twoOverR;     // 2 / R
    float gradScale;    // 1 / (PI R^5)
    float viscScale;    // 45 / (PI R^6)
    float stScale;      // 136.0241 / (PI R^9)
};

struct GridDesc {
    int gx, gy, gz, C;   // cells per axis, C = gx*gy*gz (also the out-of-grid sentinel id)
    float cellLength;
    int xOff;            // global x index of local cell column 0 (0 for a whole domain; slabs: x0-1)
};

// ---- float3 helpers with helper_math.h semantics (explicit association) ----------------------
__device__ __forceinline__ float3 v3(float x, float y, float z) { return make_float3(x, y, z); }
__device__ __forceinline__ float3 add3(float3 a, float3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ float3 sub3(float3 a, float3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ float3 neg3(float3 a) { return v3(-a.x, -a.y, -a.z); }
__device__ __forceinline__ float3 mul3s(float3 a, float s) { return v3(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ float3 smul3(float s, float3 a) { return v3(s * a.x, s * a.y, s * a.z); }
__device__ __forceinline__ float3 div3s(float3 a, float s) { return v3(a.x / s, a.y / s, a.z / s); }
__device__ __forceinline__ float dot3(float3 a, float3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ float len3(float3 a) { return sqrtf(dot3(a, a)); }
__device__ __forceinline__ float cube(float x) { return x * x * x; }
__device__ __forceinline__ float3 xyz4(const float4 v) { return make_float3(v.x, v.y, v.z); }

// fmaxf/fminf with explicit compares (identical to the oracle's helpers; see oracle header)
__device__ __forceinline__ float max_eps(float x) { return (x > kEps) ? x : kEps; }
__device__ __forceinline__ float max0(float x) { return (x > 0.0f) ? x : 0.0f; }
__device__ __forceinline__ float min0(float x) { return (x < 0.0f) ? x : ((x == 0.0f) ? x : 0.0f); }

// ---- exact fast paths for sqrt and division ----------------------------------------------------
// hipcc's correctly rounded sqrtf / `/` are v_sqrt / v_rcp followed by FMA refinement, wrapped in
// range scaling (v_div_scale, 2^+-32 pre-scaling) and special-case fix-ups (v_div_fmas,
// v_div_fixup, v_cmp_class).  When the operands are known to be in the range where the scaling is
// the identity, the refinement alone yields the same correctly rounded bits with a third of the
// instructions.  Every pair evaluates ONE predicate (pair_needs_plain_ops) and the whole wave
// branches uniformly: the <true> instantiations below are branch-free, the <false> ones are the
// plain operators.  sphx_fastmath_selftest compares both bit for bit (x/R over every float the
// sweeps can produce, sqrt over every non-negative float, the shared-denominator division over
// 2^28 operand triples), and every parity test runs through them.

// correctly rounded sqrt for x == 0 or x >= 2^-96: the compiler's own refinement without its
// denormal pre-scaling
template <bool FAST>
__device__ __forceinline__ float sqrt_sel(float x)
{
    if (!FAST) return sqrtf(x);
    const float s = __builtin_amdgcn_sqrtf(x);
    const float sm = __uint_as_float(__float_as_uint(s) - 1u);
    const float sp = __uint_as_float(__float_as_uint(s) + 1u);
    const float rm = __builtin_fmaf(-sm, s, x);
    const float rp = __builtin_fmaf(-sp, s, x);
    float out = (rm <= 0.0f) ? sm : s;
    out = (rp > 0.0f) ? sp : out;
    return out;
}

// n / den for three numerators sharing one denominator: one reciprocal refinement, then the
// compiler's quotient refinement per component.  Requires den in [2^-90, 2^16] (k.fastDiv) and
// every numerator zero or >= 2^-101 in magnitude.
template <bool FAST>
__device__ __forceinline__ float3 div3_sel(float3 n, float den)
{
    if (!FAST) return div3s(n, den);
    float rc = __builtin_amdgcn_rcpf(den);
    rc = __builtin_fmaf(__builtin_fmaf(-den, rc, 1.0f), rc, rc);
    float3 q;
    {
        float q0 = n.x * rc; q0 = __builtin_fmaf(__builtin_fmaf(-den, q0, n.x), rc, q0);
        q.x = __builtin_fmaf(__builtin_fmaf(-den, q0, n.x), rc, q0);
    }
    {
        float q0 = n.y * rc; q0 = __builtin_fmaf(__builtin_fmaf(-den, q0, n.y), rc, q0);
        q.y = __builtin_fmaf(__builtin_fmaf(-den, q0, n.y), rc, q0);
    }
    {
        float q0 = n.z * rc; q0 = __builtin_fmaf(__builtin_fmaf(-den, q0, n.z), rc, q0);
        q.z = __builtin_fmaf(__builtin_fmaf(-den, q0, n.z), rc, q0);
    }
    return q;
}

// x / R for the constant R: with y = RN(1/R), two FMA refinement steps of x*y equal RN(x/R) for
// x == 0 or x >= 2^-47.  That is not a theorem for arbitrary R, so validate_fast_math checks EVERY
// float in {0} U [2^-47, 2.2R] before k.fastQ is set.
template <bool FAST>
__device__ __forceinline__ float div_by_radius(float x, const KernelConsts& k)
{
    if (!FAST) return x / k.R;
    float q0 = x * k.rcpR;
    q0 = __builtin_fmaf(__builtin_fmaf(-k.R, q0, x), k.rcpR, q0);
    return __builtin_fmaf(__builtin_fmaf(-k.R, q0, x), k.rcpR, q0);
}

// x / den for a constant den with y = RN(1/den): the same two refinement steps; valid only where validate_fast_math
// has compared it with the plain operator for every float of the range in use
__device__ __forceinline__ float div_by_const_refined(float x, float den, float y)
{
    float q0 = x * y;
    q0 = __builtin_fmaf(__builtin_fmaf(-den, q0, x), y, q0);
    return __builtin_fmaf(__builtin_fmaf(-den, q0, x), y, q0);
}

// true when this pair must use the plain operators: a positive squared distance below 2^-96 (then
// r < 2^-48 as well), a non-zero displacement component below 2^-101, or fast paths not validated
// (positions are frozen while a row is valid, so the row builder evaluates this once per pair and
// stores it as a flag bit of the entry; the sweeps only test the bit)
__device__ __forceinline__ bool pair_needs_plain_ops(float3 d, float r2)
{
    const int e = min(min(__builtin_amdgcn_frexp_expf(d.x), __builtin_amdgcn_frexp_expf(d.y)), __builtin_amdgcn_frexp_expf(d.z));
    return e < -100 || (__float_as_uint(r2) - 1u) < (0x0f800000u - 1u);
}
__device__ __forceinline__ bool fast_paths_enabled(const KernelConsts& k) { return (k.fastQ & k.fastDiv & k.q2Free) != 0; }

// ---- smoothing kernels ------------------------------------------------------------------------
// q = 2*|r|/R as computed by both cubic-spline functions
template <bool FAST>
__device__ __forceinline__ float q_of(float r, const KernelConsts& k) { return div_by_radius<FAST>(2.0f * r, k); }

// cubic_spline_kernel, CUDAFunctions.cuh:23-35
// (FAST: the pair comes from a row, so q <= 2 already holds — KernelConsts::q2Free)
template <bool FAST>
__device__ __forceinline__ float kW(float q, const KernelConsts& k)
{
    const float w = k.wA * ((q > 1.0f) ? (2.0f - q) * (2.0f - q) * (2.0f - q) : ((3.0f * q - 6.0f) * q * q + 4.0f));
    return ((!FAST && q > 2.0f) || q < kEps) ? 0.0f : w;
}
// cubic_spline_kernel_gradient, CUDAFunctions.cuh:37-50
template <bool FAST>
__device__ __forceinline__ float3 kGradW(float3 d, float q, const KernelConsts& k)
{
    const float3 a = div3_sel<FAST>(d, kPi * (q + kEps) * k.R * k.R * k.R * k.R * k.R);
    const float3 g = mul3s(a, (q > 1.0f) ? ((12.0f - 3.0f * q) * q - 12.0f) : ((9.0f * q - 12.0f) * q));
    if (FAST) return g;
    return (q > 2.0f) ? v3(0.0f, 0.0f, 0.0f) : g;      // select instead of an early return (same value)
}
// viscosity_kernel_laplacian, CUDAFunctions.cuh:52-54
template <bool FAST>
__device__ __forceinline__ float kViscLap(float r, const KernelConsts& k)
{
    const float x = 45.0f * (k.R - r);
    // (row entries have r <= R, so x is 0 or at least 45 ulp(R) > 2^-40: inside the validated range)
    const float l = (FAST && k.fastVisc) ? div_by_const_refined(x, k.viscDen, k.rcpViscDen) : x / k.viscDen;
    return (FAST || r <= k.R) ? l : 0.0f;
}
// surface_tension_kernel_gradient, CUDAFunctions.cuh:82-98
template <bool FAST>
__device__ __forceinline__ float3 kSurfGrad(float3 d, float x, const KernelConsts& k)
{
    const bool outside = (!FAST && x > k.R) || x < kEps;
    // (outside the support the quotient is discarded; a harmless denominator keeps it finite)
    const float3 a = div3_sel<FAST>(smul3(136.0241f, neg3(d)), outside ? 1.0f : k.stK * x);
    const float3 g = mul3s(a, (2.0f * x <= k.R) ? (2.0f * cube(k.R - x) * cube(x) - k.stC) : (cube(k.R - x) * cube(x)));
    return outside ? v3(0.0f, 0.0f, 0.0f) : g;
}

// ---- the exact fast paths for TWO pairs at a time (packed fp32) ------------------------------------------------------
// v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32 apply the IEEE operation to each half independently, so evaluating two
// row entries in the halves of one register pair yields exactly the bits of two scalar evaluations at half the
// instruction count (the scalar chain is VALU-issue-bound at <= 1 M particles: profiles/r02_ubench_sweep_structure.txt).
// Accumulation into the per-particle sums stays scalar and in row order.  Only the <FAST> forms exist: a wave whose
// pair needs the plain operators takes the scalar path for both entries.
typedef float f2 __attribute__((ext_vector_type(2)));
typedef int i2 __attribute__((ext_vector_type(2)));
struct f2x3 { f2 x, y, z; };
__device__ __forceinline__ f2 fma2(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f2 sel2(i2 m, f2 a, f2 b) { return f2{m.x ? a.x : b.x, m.y ? a.y : b.y}; }
__device__ __forceinline__ f2 splat2(float v) { return f2{v, v}; }
// sqrt_sel<true> per half
__device__ __forceinline__ f2 sqrt_fast2(f2 x)
{
    const f2 s = f2{__builtin_amdgcn_sqrtf(x.x), __builtin_amdgcn_sqrtf(x.y)};
    const i2 si = __builtin_bit_cast(i2, s);
    const f2 sm = __builtin_bit_cast(f2, si - 1), sp = __builtin_bit_cast(f2, si + 1);
    const f2 rm = fma2(-sm, s, x), rp = fma2(-sp, s, x);
    f2 out = sel2(rm <= 0.0f, sm, s);
    out = sel2(rp > 0.0f, sp, out);
    return out;
}
// q_of<true> per half: div_by_radius<true>(2 r)
__device__ __forceinline__ f2 q_fast2(f2 r, const KernelConsts& k)
{
    const f2 x = 2.0f * r, R = splat2(k.R), y = splat2(k.rcpR);
    f2 q0 = x * y;
    q0 = fma2(fma2(-R, q0, x), y, q0);
    return fma2(fma2(-R, q0, x), y, q0);
}
// kW<true> per half
__device__ __forceinline__ f2 kW_fast2(f2 q, const KernelConsts& k)
{
    const f2 w = k.wA * sel2(q > 1.0f, (2.0f - q) * (2.0f - q) * (2.0f - q), ((3.0f * q - 6.0f) * q * q + 4.0f));
    return sel2(q < kEps, splat2(0.0f), w);
}
// kGradW<true> per half (div3_sel<true> inlined)
__device__ __forceinline__ f2x3 kGradW_fast2(const f2x3& d, f2 q, const KernelConsts& k)
{
    const f2 den = kPi * (q + kEps) * k.R * k.R * k.R * k.R * k.R;
    f2 rc = f2{__builtin_amdgcn_rcpf(den.x), __builtin_amdgcn_rcpf(den.y)};
    rc = fma2(fma2(-den, rc, splat2(1.0f)), rc, rc);
    const f2 poly = sel2(q > 1.0f, ((12.0f - 3.0f * q) * q - 12.0f), ((9.0f * q - 12.0f) * q));
    f2x3 g;
    { f2 q0 = d.x * rc; q0 = fma2(fma2(-den, q0, d.x), rc, q0); g.x = fma2(fma2(-den, q0, d.x), rc, q0) * poly; }
    { f2 q0 = d.y * rc; q0 = fma2(fma2(-den, q0, d.y), rc, q0); g.y = fma2(fma2(-den, q0, d.y), rc, q0) * poly; }
    { f2 q0 = d.z * rc; q0 = fma2(fma2(-den, q0, d.z), rc, q0); g.z = fma2(fma2(-den, q0, d.z), rc, q0) * poly; }
    return g;
}
// displacement of two neighbours from pi, their squared lengths, and gradW: what every pair2 body starts from
struct Pair2 { f2x3 d; f2 r2, q; };
__device__ __forceinline__ Pair2 pair2_geometry(const float3 pi, const float4 a, const float4 b, const KernelConsts& k)
{
    Pair2 p;
    p.d.x = pi.x - f2{a.x, b.x}; p.d.y = pi.y - f2{a.y, b.y}; p.d.z = pi.z - f2{a.z, b.z};
    p.r2 = p.d.x * p.d.x + p.d.y * p.d.y + p.d.z * p.d.z;
    p.q = q_fast2(sqrt_fast2(p.r2), k);
    return p;
}
template <class B> __device__ __forceinline__ constexpr auto has_pair2_impl(int) -> decltype(B::kPair2) { return B::kPair2; }
template <class B> __device__ __forceinline__ constexpr bool has_pair2_impl(long) { return false; }
template <class B> __device__ __forceinline__ constexpr bool has_pair2() { return has_pair2_impl<B>(0); }

// ---- tolerance arithmetic ---------------------------------------------------------------------------------------
// The same formulas with v_rsq_f32 / v_rcp_f32 (1 ulp), fused multiply-adds and folded constants: relative
// deviations of a few 1e-7 per pair term from the strict path (the reference binary itself is built with
// -use_fast_math, src/CMakeLists.txt:43).  Only the row walks use it; support tests are unchanged because the rows
// were built with the exact threshold.  tests/test_gpu_tolerance.py bounds the deviation from the oracle.
// r05: every fused multiply-add is WRITTEN (fmaf), nothing is left to the compiler's contraction.  With `#pragma clang fp
// contract(fast)` the choice of which product of `a*b + c*d + e*f` is fused differed between instantiations of the same
// source (the 4-chunk body and the 1-chunk remainder of walk_row_quad), so a pair term depended on how long the OTHER rows
// of its wave were -- i.e. on how particles are grouped into waves: a slab run and the single-device run agreed to 1e-7,
// not bit for bit (tools/slab_tol_diag.py).  Now a tolerance-mode result is a function of the row and its inputs alone.
struct TolPair { float r, q, rcpq; };      // |d|, 2|d|/R, 1/(q + EPS)
__device__ __forceinline__ TolPair tol_pair(float r2, const KernelConsts& k)
{
    TolPair t;
    t.r = r2 * __builtin_amdgcn_rsqf(fmaxf(r2, 1.0e-36f));
    t.q = t.r * k.twoOverR;
    t.rcpq = __builtin_amdgcn_rcpf(t.q + kEps);
    return t;
}
// W (CUDAFunctions.cuh:23-35)
__device__ __forceinline__ float tol_W(const TolPair& t, const KernelConsts& k)
{
    const float a = 2.0f - t.q;
    const float near = fmaf(fmaf(3.0f, t.q, -6.0f) * t.q, t.q, 4.0f);      // (3q - 6) q^2 + 4
    const float w = k.wA * ((t.q > 1.0f) ? a * a * a : near);
    return (t.q < kEps) ? 0.0f : w;
}
// gradW = d * tol_gradW_scale (CUDAFunctions.cuh:37-50)
__device__ __forceinline__ float tol_gradW_scale(const TolPair& t, const KernelConsts& k)
{
    const float poly = (t.q > 1.0f) ? fmaf(fmaf(-3.0f, t.q, 12.0f), t.q, -12.0f) : (fmaf(9.0f, t.q, -12.0f) * t.q);
    return poly * k.gradScale * t.rcpq;
}
// viscosity laplacian (CUDAFunctions.cuh:52-54)
__device__ __forceinline__ float tol_viscLap(const TolPair& t, const KernelConsts& k)
{
    return (k.R - t.r) * k.viscScale;
}
// surface-tension gradient = d * tol_surf_scale (CUDAFunctions.cuh:82-98)
__device__ __forceinline__ float tol_surf_scale(const TolPair& t, const KernelConsts& k)
{
    const float x = t.r;
    const float c3 = cube(k.R - x) * cube(x);
    const float poly = (2.0f * x <= k.R) ? fmaf(2.0f, c3, -k.stC) : c3;
    const float s = -k.stScale * __builtin_amdgcn_rcpf(fmaxf(x, kEps)) * poly;
    return (x < kEps) ? 0.0f : s;
}
// a + d * s per component, and the dot products of the pair terms, each as the one chain of fused multiply-adds written here
__device__ __forceinline__ float3 tol_axpy(const float3 a, const float3 d, const float s) { return v3(fmaf(d.x, s, a.x), fmaf(d.y, s, a.y), fmaf(d.z, s, a.z)); }
__device__ __forceinline__ float tol_dot(const float ax, const float ay, const float az, const float3 b) { return fmaf(az, b.z, fmaf(ay, b.y, ax * b.x)); }

// x^7 of the Tait equation of state (BasicSPHSolver.cu:108): fp64 multiply chain, one rounding
__device__ __forceinline__ float pow7(float x)
{
    const double d = (double)x, d2 = d * d, d4 = d2 * d2, d6 = d4 * d2;
    return (float)(d6 * d);
}

// particlePos2cellIdx + make_int3(pos / cellLength), CUDAFunctions.cuh:64-78
// (local coordinates: the global x index minus g.xOff, so a slab's sub-grid sees the same cells)
__device__ __forceinline__ int3 cell_of(float3 p, const GridDesc& g)
{
    return make_int3((int)(p.x / g.cellLength) - g.xOff, (int)(p.y / g.cellLength), (int)(p.z / g.cellLength));
}
__device__ __forceinline__ int cell_id(int x, int y, int z, const GridDesc& g)
{
    return (x >= 0 && x < g.gx && y >= 0 && y < g.gy && z >= 0 && z < g.gz) ? ((x * g.gy + y) * g.gz + z) : g.C;
}

// box clamp of enforceBoundary_CUDA (BasicSPHSolver.cu:85-96 with velocity, PBDSolver.cu:212-223
// without): [0, 0.99*size] per axis, <= / >= tests, outward velocity component removed.
template <bool WITH_VEL>
__device__ __forceinline__ void clamp_box(float3& p, float3& v, const float3 space)
{
    const float lx = space.x * .00f, hx = space.x * .99f;
    const float ly = space.y * .00f, hy = space.y * .99f;
    const float lz = space.z * .00f, hz = space.z * .99f;
    if (p.x <= lx) { p.x = lx; if (WITH_VEL) v.x = max0(v.x); }
    if (p.x >= hx) { p.x = hx; if (WITH_VEL) v.x = min0(v.x); }
    if (p.y <= ly) { p.y = ly; if (WITH_VEL) v.y = max0(v.y); }
    if (p.y >= hy) { p.y = hy; if (WITH_VEL) v.y = min0(v.y); }
    if (p.z <= lz) { p.z = lz; if (WITH_VEL) v.z = max0(v.z); }
    if (p.z >= hz) { p.z = hz; if (WITH_VEL) v.z = min0(v.z); }
}

// (ops whose Field is neither never run one-gather sweeps; the generic form only keeps their instantiation well-formed)
template <class F> __device__ __forceinline__ F scalar_field(float) { return F{}; }
template <> __device__ __forceinline__ float scalar_field<float>(float v) { return v; }
template <> __device__ __forceinline__ float4 scalar_field<float4>(float v) { return make_float4(v, 0.0f, 0.0f, 0.0f); }

// ---- neighbour sweep skeleton -------------------------------------------------------------------
// Visits the 27 cells around the cell of `pi` in the reference order: dx outer, dy,
// dz inner; per cell the fluid range then the boundary range, j ascending.  Candidates farther
// than the common support are skipped: beyond tCut every kernel returns exactly +0 (or the zero
// vector), and adding that to a non-negative-zero-initialised accumulator changes no bit.
// Body::fluid / Body::boundary receive (j, d = pi - pj, r2 = |d|^2, w = pj.w).
template <bool FLUID, bool BOUNDARY, class Body>
__device__ __forceinline__ void sweep27(const GridDesc& g, const KernelConsts& k, const int* __restrict__ csF,
                                        const float4* __restrict__ posmF, const int* __restrict__ csB,
                                        const float4* __restrict__ posmB, const float3 pi, Body& body)
{
    const int3 c0 = cell_of(pi, g);
    for (int dx = -1; dx <= 1; ++dx) {
        const int X = c0.x + dx;
        if (X < 0 || X >= g.gx) continue;
        for (int dy = -1; dy <= 1; ++dy) {
            const int Y = c0.y + dy;
            if (Y < 0 || Y >= g.gy) continue;
            for (int dz = -1; dz <= 1; ++dz) {
                const int Z = c0.z + dz;
                if (Z < 0 || Z >= g.gz) continue;
                const int c = (X * g.gy + Y) * g.gz + Z;
                if (FLUID) {
                    const int e = csF[c + 1];
                    for (int j = csF[c]; j < e; ++j) {
                        const float4 pj = posmF[j];
                        const float3 d = sub3(pi, v3(pj.x, pj.y, pj.z));
                        const float r2 = dot3(d, d);
                        if (r2 > k.tCut) continue;
                        body.fluid(j, d, r2, pj.w);
                    }
                }
                if (BOUNDARY) {
                    const int e = csB[c + 1];
                    for (int j = csB[c]; j < e; ++j) {
                        const float4 pj = posmB[j];
                        const float3 d = sub3(pi, v3(pj.x, pj.y, pj.z));
                        const float r2 = dot3(d, d);
                        if (r2 > k.tCut) continue;
                        body.boundary(j, d, r2, pj.w);
                    }
                }
            }
        }
    }
}

// ---- per-step compact neighbour rows + LDS-streamed tiles -----------------------------------------
// While positions are frozen (all sweeps of a WCSPH/DFSPH step; the two sweeps of one PBD
// iteration) every sweep of particle i meets the same candidates and rejects the same ones, and a
// rejected candidate contributes exactly +0.  The first pass therefore records, per particle, the
// candidates with r2 <= tCut IN VISIT ORDER (self excluded: its terms are exactly zero); later
// sweeps walk that row.  Order is preserved, so every accumulated bit is.
//
// Row layout: wave-interleaved — entry k of particle i lives at ((i>>6)*cap + k)*64 + (i&63), so
// the 64 lanes of a wave read 256 contiguous bytes per k.  count > cap means the row overflowed:
// that lane falls back to the direct 27-cell walk.
//
// Entry formats (per 64-particle tile, chosen by the row builder, recorded in tileFmt):
//   fmt 0  bit31 = boundary, bit30 = pair needs the plain operators, bits 0..29 = global index: sweeps gather from global memory.  Each
//          divergent gather costs ~64 cycles of the CU's single texture-address pipe per
//          wave-instruction, which is what bounds these sweeps.
//   fmt 2  bits 30..29 = dx group g (0,1,2 <-> dx = -1,0,+1), bit 28 = boundary, bit 27 = plain ops, bits 0..26 = slot
//          in the LDS stage of that group.  A tile is one wave of consecutive cell-sorted particles;
//          because the cell id runs z fastest, the neighbour cells of the whole tile for one
//          (dx,dy) are ONE contiguous cell range [first+off-1, last+off+1], off = (dx*gy+dy)*gz,
//          i.e. one contiguous particle range of the fluid array and one of the boundary array.
//          The sweep streams through the three dx groups: the wave copies the group's 3+3 ranges
//          (position+mass and the one per-neighbour field of the sweep) into its private LDS slab
//          with coalesced loads, then every lane walks the entries of that group from LDS.  Rows
//          are in visit order (dx outermost), so the entries of a group are contiguous.
//          Used when positions are the binned ones (not PBD), the tile is in-grid and every group
//          fits kGroupSlots; otherwise the tile keeps fmt 0.
constexpr int kTile = 64;
constexpr int kWideBlock = 256;     // threads per block: 4 waves = 4 adjacent tiles share a CU
constexpr int kGroupSlots = 384;    // LDS slots per wave and dx group
// fmt 0: bit 31 boundary, bit 30 plain-ops, bits 29..28 zero, 0..27 index into the unified neighbour
// space [fluid slots | boundary slots] (so `entry << 4` is the byte offset of a float4 record)
constexpr unsigned int kBoundaryBit = 0x80000000u;
constexpr unsigned int kPlainBit = 0x40000000u;
constexpr unsigned int kIndexMask = 0x0fffffffu;
constexpr unsigned int kStreamBoundaryBit = 0x10000000u;  // fmt 2: 30..29 group, 28 boundary, 27 plain-ops, 0..26 slot
constexpr unsigned int kStreamPlainBit = 0x08000000u;
constexpr unsigned int kStreamSlotMask = 0x07ffffffu;

// Row storage.  Entry k of particle i lives at
//     ((i >> 6) * cap/4 + (k >> 2)) * 256 + (i & 63) * 4 + (k & 3)          (32-bit words)
// i.e. rows are cut into chunks of 4 entries = 16 bytes and chunk s of the 64 particles of a tile is one contiguous KB.
// Two ways to walk it, both fully coalesced:
//   lane per particle : a lane reads its own chunk with one 16-byte load (the wave reads the whole KB);
//   quad per particle : the 4 lanes of a quad read the 4 entries of one chunk (a wave = 16 particles reads 256 bytes);
//                       the 4 neighbours of a chunk are consecutive accepted candidates, i.e. (almost) adjacent
//                       records, so the quad's gathers share cache lines (walk_row_quad).
constexpr int kRowChunk = 4;
// (Non-temporal row loads / stores were measured and dropped: -3 % per sweep in the micro-benchmark at 10.3 M particles, +14..19 % at
// 1 M, nothing in the engine's sweeps and a 2x slower row builder -- profiles/r04_ubench_tiles.txt, DESIGN.md section 5.  Routing the row
// loads through a helper function also cost the strict rate kernel 16 %: the compiler then waits for each row load before issuing
// the next one.  Keep them as plain indexed loads through the __restrict__ pointer.)
__device__ __forceinline__ size_t row_base_offset(int i, int cap) { return ((size_t)(i >> 6) * (size_t)cap) * 64u + (size_t)(i & 63) * kRowChunk; }
__device__ __forceinline__ size_t row_entry_offset(int k) { return (size_t)(k >> 2) * 256u + (unsigned)(k & 3); }

// XCD-aware block order: the dispatcher places block b on XCD b % 8 (MI355X_MICROARCH.md), each XCD
// has its own L2.  Logical block = (b % 8) * chunk + b / 8 gives every XCD one contiguous run of
// logical blocks = one spatial slab of the cell-sorted particles, so the neighbour data an XCD
// re-reads stays in ITS L2 instead of being fetched into all eight.  Grids are launched with
// 8 * chunk blocks; logical blocks past the end exit.  (Speed only: any placement is correct.)
__device__ __forceinline__ int logical_block() { return (int)(blockIdx.x & 7u) * (int)(gridDim.x >> 3) + (int)(blockIdx.x >> 3); }
inline unsigned int xcd_grid(int n, int block) { const int nb = n > 0 ? (n - 1) / block + 1 : 1; return (unsigned int)(((nb + 7) / 8) * 8); }

struct SweepCtx {
    GridDesc g; KernelConsts k;
    const int* csF; const float4* posm;     // fluid cell starts, packed (x,y,z,mass)
    const int* csB; const float4* bposm;    // boundary cell starts, packed (x,y,z,mass) = posm + bOff
    int bOff;                               // unified index of boundary particle 0 (= fluid capacity)
    const unsigned int* nbr; const int* nbrCount; int cap;   // nbr == nullptr: direct sweeps only
    const int* tileFmt;                     // per tile entry format (nullptr: all tiles fmt 0)
    float4* vel4;                           // 16-byte aligned mirror of the fluid velocities (one gather)
    float4* cg4;                            // 16-byte aligned mirror of the colour gradient
    float4* posf;                           // (x, y, z, scalar field): position AND the neighbour scalar in one gather
    const int* massUniform;                 // device flag: 1 when every fluid particle has the mass of particle 0
    const int* tileOrder;                   // schedule: the tile each launched wave works on (nullptr: identity)
    // Skin rows (PBD): the rows were built once per step with the enlarged cutoff `buildCut`, positions have moved
    // since (Jacobi iterations on a fixed cell table, PBDSolver.cu:225-258).  Every pair re-tests its CURRENT squared
    // distance against tCut (beyond it every kernel is exactly +0, so skipping is exact) and re-derives its plain-ops
    // predicate.  `stale` (device flag) is raised when some particle moved farther than half the skin since the
    // build: the rows may then miss a pair, and every sweep walks the cells directly until the next build.
    const int* stale;                       // nullptr: ordinary rows (positions frozen since the build)
    const int* rowCell;                     // skin rows: the cell each particle's row was built around.  The reference walks the
                                            // 27 cells around the cell of the CURRENT position (PBDSolver.cu:139-141 on moved
                                            // positions), so a row is only valid while its particle stays in that cell
    float buildCut;                         // squared cutoff the row builder accepts candidates with
    // Persistent rows (tolerance arithmetic, WCSPH / DFSPH; SPHSystem's persistent mode): the particle arrays keep the order of
    // the last row build across steps (the API arrays are exported in the reference's order separately), rows were built with
    // the skin, every pair re-tests its CURRENT distance.  A row stays valid while no particle has moved more than 0.45 skin (PBD skin rows; 0.49 skin for persistent rows)
    // RELATIVE to the others since the build (checked on the device at the start of every step), whatever cells the particles
    // are in by now: the neighbour set of the reference is {r <= R}, and the order of a particle's sum is free under this
    // contract.  Particles without a row (overflow) walk the cells around the cell their row was built around, in the cell
    // table of the build (csF then points at that copy).
    int persist;
    // Adaptive solver loops without a host round trip per iteration (DFSPH): the launches of every possible iteration are enqueued
    // at once; once the device-side test of the |error| total has raised *gate the remaining ones leave at their first instruction.
    const int* gate;
    int plainBits;                          // tolerance launches: row entries carry the plain-operator flag anyway (a sweep may take the strict kernel)
    int quad;                               // QuadBits: sweeps that run quad-per-particle (walk_row_quad)
    int duo;                                // QuadBits: sweeps that run two lanes per particle (walk_row_duo); quad wins where both are set
    int numTiles;                           // tiles this launch covers
    int tile0;                              // first tile of a range-restricted launch (0 otherwise; no schedule then)
    int lo, hi;                             // particles [lo, hi) are processed; lanes outside only take part in wave-wide staging
    int lo2, hi2;                           // ... and [lo2, hi2), a second range behind the first (empty unless a two-range launch)
    int tileSplit, tile1;                   // launch tiles >= tileSplit belong to the second range: tile = lt + tile1
    int n;
    int* overflowMax;                       // row builder only: longest row that did not fit `cap` (atomicMax; nullptr otherwise)
    int brick;                              // 1: rows hold 16-bit slots of the compact-brick LDS stage (tolerance arithmetic, see "brick" below)
    int* brickFault;                        // device flag: a one-cell slice of some brick exceeded the stage (the host leaves brick mode)
    const struct BrickTables* brickTab;     // tables of the non-empty bricks (whole-brick slice), built once per step by k_brick_list
    const int* brickCount;                  // ... and how many there are (device word)
    int brickBlocks;                        // blocks a brick launch uses (they stride over the list)
};

// The tile (64 consecutive particles) this wave works on.  Launch order is a free choice — results
// do not depend on it — so tiles are ordered by (y-chunk, x) rather than the array's x-major order:
// each XCD then walks along x inside one y-chunk and the x+-1 neighbour layers of that chunk stay
// in its 4 MB L2 (at 10 M particles a full x-layer is ~3 MB per array and would not).
// a scheduled launch restricted to a particle range visits every tile of the schedule: tiles wholly outside the range leave at once
__device__ __forceinline__ bool tile_outside(const SweepCtx& c, int tile)
{
    const int a = tile * kTile, b = a + kTile;
    return (b <= c.lo || a >= c.hi) && (b <= c.lo2 || a >= c.hi2);
}
__device__ __forceinline__ int wave_tile_of(const SweepCtx& c, const int lt)      // lt: the launch-order tile number of this wave
{
    if (lt >= c.numTiles) return -1;
    if (!c.tileOrder) return lt + (lt < c.tileSplit ? c.tile0 : c.tile1);
    const int tile = c.tileOrder[lt];
    return tile_outside(c, tile) ? -1 : tile;
}
__device__ __forceinline__ int wave_tile(const SweepCtx& c)
{
    if (c.gate && *c.gate != 0) return -1;
    return wave_tile_of(c, logical_block() * (kWideBlock / kTile) + (int)(threadIdx.x >> 6));
}
__device__ __forceinline__ bool in_range(const SweepCtx& c, int i) { return (i >= c.lo && i < c.hi) || (i >= c.lo2 && i < c.hi2); }

// The 18 neighbour ranges of a tile, one per lane (lanes 0..8 fluid, 9..17 boundary; r = 3*(dx+1) +
// (dy+1)), with each range's offset inside the LDS stage of its dx group (fluid dy=-1,0,1 first,
// then boundary dy=-1,0,1).  Wave-synchronous: all 64 lanes call it, only shuffles inside.
struct WaveRanges { int start, len, off; bool ok; };

__device__ __forceinline__ WaveRanges wave_ranges(const SweepCtx& c, const int i0)
{
    const int lane = threadIdx.x & 63;
    WaveRanges w; w.start = 0; w.len = 0; w.off = 0;
    const int i1 = min(i0 + kTile, c.n);
    const int3 cf = cell_of(xyz4(c.posm[i0]), c.g);
    const int3 cl = cell_of(xyz4(c.posm[i1 - 1]), c.g);
    const int idF = cell_id(cf.x, cf.y, cf.z, c.g), idL = cell_id(cl.x, cl.y, cl.z, c.g);
    bool ok = idF < c.g.C && idL < c.g.C && idF <= idL;
    const int r = lane % 9;
    if (lane < 18) {
        const int off = ((r / 3 - 1) * c.g.gy + (r % 3 - 1)) * c.g.gz;
        const int lo = max(idF + off - 1, 0), hi = min(idL + off + 1, c.g.C - 1);
        const int* cs = lane < 9 ? c.csF : c.csB;
        if (ok && lo <= hi) { w.start = cs[lo]; w.len = cs[hi + 1] - w.start; }
    }
    // position q of this lane's range inside its group: fluid dy (0..2), then boundary dy (3..5)
    const int g = r / 3, q = (lane < 9 ? 0 : 3) + r % 3;
    int gsize = 0;
#pragma unroll
    for (int qq = 0; qq < 6; ++qq) {
        const int src = (qq < 3 ? 0 : 9) + g * 3 + (qq % 3);
        const int l = __shfl(w.len, src, 64);
        if (qq < q) w.off += l;
        gsize += l;
    }
    // every group must fit the stage (lanes 0,3,6 hold the three group sizes)
    const int g0 = __shfl(gsize, 0, 64), g1 = __shfl(gsize, 3, 64), g2 = __shfl(gsize, 6, 64);
    w.ok = ok && g0 <= kGroupSlots && g1 <= kGroupSlots && g2 <= kGroupSlots;
    return w;
}

// make this wave's LDS writes visible to its own later reads (and vice versa) without a block barrier
__device__ __forceinline__ void wave_lds_fence()
{
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// direct walk in reference order; `visit(j, isBoundary, d, r2, mass_j)`
template <bool WANT_BOUNDARY, class Visit>
__device__ __forceinline__ void walk_cells_around(const SweepCtx& c, const int3 c0, const float3 pi, Visit&& visit);
template <bool WANT_BOUNDARY, class Visit>
__device__ __forceinline__ void walk_cells(const SweepCtx& c, const float3 pi, Visit&& visit)
{
    walk_cells_around<WANT_BOUNDARY>(c, cell_of(pi, c.g), pi, visit);
}
// the fallback of a particle without a usable row: around the cell of its position, or (persistent rows) around the cell its row
// was built around -- the arrays are still sorted by THOSE cells, and the skin is smaller than the slack of the cell length, so
// every particle within R now was within the 27 cells then
template <bool WANT_BOUNDARY, class Visit>
__device__ __forceinline__ void walk_cells_fallback(const SweepCtx& c, const int i, const float3 pi, Visit&& visit)
{
    // (persistent rows exist under the tolerance arithmetic only: in the strict instantiations of the quad kernels, where
    // assume_arith pins k.tol to 0, this branch disappears -- its extra live registers made the compiler serialise the row
    // loads of the strict rate kernel, +16 % on the dominant launch: ISA diff in r04)
    int3 c0;
    if (c.k.tol != 0 && c.persist) {
        const int id = c.rowCell[i];
        c0 = make_int3(id / (c.g.gz * c.g.gy), (id / c.g.gz) % c.g.gy, id % c.g.gz);
        if (id >= c.g.C) c0.x = -4;          // built out of the grid (sentinel bucket): every cell of the walk is skipped
    } else c0 = cell_of(pi, c.g);
    walk_cells_around<WANT_BOUNDARY>(c, c0, pi, visit);     // ONE call site: the visit body is inlined once
}
template <bool WANT_BOUNDARY, class Visit>
__device__ __forceinline__ void walk_cells_around(const SweepCtx& c, const int3 c0, const float3 pi, Visit&& visit)
{
    for (int dx = -1; dx <= 1; ++dx) {
        const int X = c0.x + dx;
        if (X < 0 || X >= c.g.gx) continue;
        for (int dy = -1; dy <= 1; ++dy) {
            const int Y = c0.y + dy;
            if (Y < 0 || Y >= c.g.gy) continue;
            for (int dz = -1; dz <= 1; ++dz) {
                const int Z = c0.z + dz;
                if (Z < 0 || Z >= c.g.gz) continue;
                const int cell = (X * c.g.gy + Y) * c.g.gz + Z;
                {
                    const int e = c.csF[cell + 1];
                    for (int j = c.csF[cell]; j < e; ++j) {
                        const float4 pj = c.posm[j];
                        const float3 d = sub3(pi, v3(pj.x, pj.y, pj.z));
                        const float r2 = dot3(d, d);
                        if (r2 > c.k.tCut) continue;
                        visit(j, false, d, r2, pj.w);
                    }
                }
                if (WANT_BOUNDARY) {
                    const int e = c.csB[cell + 1];
                    for (int j = c.csB[cell]; j < e; ++j) {
                        const float4 pj = c.bposm[j];
                        const float3 d = sub3(pi, v3(pj.x, pj.y, pj.z));
                        const float r2 = dot3(d, d);
                        if (r2 > c.k.tCut) continue;
                        visit(j, true, d, r2, pj.w);
                    }
                }
            }
        }
    }
}

// Ops whose neighbour field is one scalar (kappa, pterm, lambda) may declare `bool packedScalar`:
// when all fluid masses are equal (checked on the device at pack time) position and field are read
// with ONE 16-byte gather from posf instead of two gathers — the sweeps are bound by the vector L1's
// line rate (~1 line per cycle per CU), so halving the gathers matters.  Same values, same order.
template <class Op> __device__ __forceinline__ auto op_packed_impl(const Op& op, int) -> decltype(op.packedScalar) { return op.packedScalar; }
template <class Op> __device__ __forceinline__ bool op_packed_impl(const Op&, long) { return false; }
template <class Op> __device__ __forceinline__ bool op_packed_scalar(const Op& op) { return op_packed_impl(op, 0); }

// the Field of a one-gather sweep: the scalar that came with the position record, plus whatever else the op reads per
// neighbour (ops with more than a scalar define stage_packed(isBoundary, j, scalar))
template <class Op> __device__ __forceinline__ auto packed_field(const Op& op, bool isB, int j, float s, int) -> decltype(op.stage_packed(isB, j, s)) { return op.stage_packed(isB, j, s); }
template <class Op> __device__ __forceinline__ typename Op::Field packed_field(const Op&, bool, int, float s, long) { return scalar_field<typename Op::Field>(s); }

// 16-byte record at byte offset `off` of a unified array: uniform base + 32-bit lane offset
__device__ __forceinline__ float4 gather16(const float4* __restrict__ base, unsigned int off)
{
    return *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(base) + off);
}

template <bool PACKED, class Op>
__device__ __forceinline__ void fetch_pair(const Op& op, const SweepCtx& c, float m0, unsigned int e, float4& pj,
                                           typename Op::Field& f)
{
    const bool isB = (e & kBoundaryBit) != 0u;
    const unsigned int off = e << 4;            // the four flag bits shift out
    if (PACKED) {
        const float4 r = gather16(c.posf, off); // fluid: (pos, field); boundary: (pos, mass)
        pj = make_float4(r.x, r.y, r.z, isB ? r.w : m0);
        f = packed_field(op, isB, (int)(e & kIndexMask), isB ? 0.0f : r.w, 0);
    } else {
        pj = gather16(c.posm, off);
        f = op.stage(isB, (int)(e & kIndexMask));
    }
}

// One pair term: the whole wave takes the branch-free fast arithmetic unless some lane's pair needs
// the plain operators (a wave-uniform branch, so no exec-mask bookkeeping per pair).
template <class Body, class Field>
__device__ __forceinline__ void pair_dispatch(Body& body, const bool plain, const Field& f, bool isB, float3 d, float r2,
                                              float mj, int idx)
{
    if (__builtin_expect(__any(plain), 0)) body.template pair<false>(f, isB, d, r2, mj, idx);
    else body.template pair<true>(f, isB, d, r2, mj, idx);
}

// One lane's row.  The chain "row entry -> gather -> arithmetic" is latency-bound when walked one
// entry at a time (the row streams from HBM, the gathers mostly from L2): kAhead entries and their
// gathers are issued together, the pair terms are then accumulated strictly in row order.
#ifndef SPHX_AHEAD
#define SPHX_AHEAD 4
#endif
template <bool PACKED, bool WANT_BOUNDARY, bool SKIN, bool TOL, class Op, class Body>
__device__ __forceinline__ void walk_row(const Op& op, const SweepCtx& c, const unsigned int* __restrict__ row, const int cnt,
                                         const float m0, const bool allPlain, const float3 pi, Body& body)
{
    constexpr int kAhead = SPHX_AHEAD;
    int t = 0;
    for (; t + kAhead <= cnt; t += kAhead) {
        unsigned int e[kAhead];
        if constexpr (kAhead == kRowChunk) {      // t is a multiple of 4: one 16-byte load
            const uint4 ch = *reinterpret_cast<const uint4*>(row + (size_t)(t >> 2) * 256u);
            e[0] = ch.x; e[1] = ch.y; e[2] = ch.z; e[3] = ch.w;
        } else {
#pragma unroll
            for (int u = 0; u < kAhead; ++u) e[u] = row[row_entry_offset(t + u)];
        }
        float4 pj[kAhead];
        typename Op::Field f[kAhead];
#pragma unroll
        for (int u = 0; u < kAhead; ++u) fetch_pair<PACKED, Op>(op, c, m0, e[u], pj[u], f[u]);
        if constexpr (has_pair2<Body>() && WANT_BOUNDARY && !SKIN && !TOL && (kAhead % 2 == 0)) {
            // two entries per packed evaluation; one wave-uniform test of the plain-operator flags per group
            unsigned int flags = 0u;
#pragma unroll
            for (int u = 0; u < kAhead; ++u) flags |= e[u];
            if (!__any(allPlain || (flags & kPlainBit) != 0u)) {
#pragma unroll
                for (int u = 0; u < kAhead; u += 2)
                    body.pair2(body, body, f[u], f[u + 1], (e[u] & kBoundaryBit) != 0u, (e[u + 1] & kBoundaryBit) != 0u, pi, pj[u], pj[u + 1]);
                continue;
            }
        }
#pragma unroll
        for (int u = 0; u < kAhead; ++u) {
            const bool isB = (e[u] & kBoundaryBit) != 0u;
            if (!WANT_BOUNDARY && isB) continue;
            const float3 d = sub3(pi, v3(pj[u].x, pj[u].y, pj[u].z));
            const float r2 = dot3(d, d);
            if (SKIN && r2 > c.k.tCut) continue;
            if (TOL) { body.pair_tol(f[u], isB, d, r2, pj[u].w); continue; }
            const bool plain = SKIN ? pair_needs_plain_ops(d, r2) : (e[u] & kPlainBit) != 0u;
            pair_dispatch(body, allPlain || plain, f[u], isB, d, r2, pj[u].w, (int)(e[u] & kIndexMask));
        }
    }
    for (; t < cnt; ++t) {
        const unsigned int e = row[row_entry_offset(t)];
        const bool isB = (e & kBoundaryBit) != 0u;
        if (!WANT_BOUNDARY && isB) continue;
        float4 pj; typename Op::Field fj;
        fetch_pair<PACKED, Op>(op, c, m0, e, pj, fj);
        const float3 d = sub3(pi, v3(pj.x, pj.y, pj.z));
        const float r2 = dot3(d, d);
        if (SKIN && r2 > c.k.tCut) continue;
        if (TOL) { body.pair_tol(fj, isB, d, r2, pj.w); continue; }
        const bool plain = SKIN ? pair_needs_plain_ops(d, r2) : (e & kPlainBit) != 0u;
        pair_dispatch(body, allPlain || plain, fj, isB, d, r2, pj.w, (int)(e & kIndexMask));
    }
}

// The sweep of one particle (all 64 lanes of a wave call it together; `valid` = lane has a particle).
// Op supplies `Field` (the per-neighbour value its pair term reads) and `stage(isBoundary, j)` (its
// global load; boundaries yield zeros); Body::pair(field, isBoundary, d, r2, mass_j, j_or_-1)
// accumulates.  ldsPos/ldsField: this wave's LDS slab (kGroupSlots entries) or nullptr.
template <bool WANT_BOUNDARY, class Op, class Body>
__device__ __forceinline__ void sweep(const Op& op, const SweepCtx& c, float4* ldsPos, typename Op::Field* ldsField,
                                      const int i, const bool valid, const float3 pi, Body& body)
{
    const int lane = threadIdx.x & 63;
    const bool skin = c.stale != nullptr || (c.k.tol != 0 && c.persist != 0);
    const bool rows = c.nbr != nullptr && !(c.stale != nullptr && *c.stale != 0);    // stale skin rows: direct walks (launch-uniform)
    const bool allPlain = !fast_paths_enabled(c.k);
    const int cnt = (rows && valid) ? c.nbrCount[i] : 0;
    bool useRow = rows && valid && cnt <= c.cap;
    if (c.stale != nullptr && useRow) {     // (normally never fails: the position update asks for a rebuild on a crossing)
        const int3 cNow = cell_of(pi, c.g);
        useRow = cell_id(cNow.x, cNow.y, cNow.z, c.g) == c.rowCell[i];
    }
    const unsigned int* row = rows ? c.nbr + row_base_offset(i, c.cap) : nullptr;
    // tile format is wave-uniform (i>>6 is the same for all lanes of the wave)
    const int fmt = (rows && ldsPos && c.tileFmt) ? c.tileFmt[__builtin_amdgcn_readfirstlane(i >> 6)] : 0;
    if (fmt == 2) {
        const WaveRanges w = wave_ranges(c, (i >> 6) << 6);
        int k = 0;
        unsigned int e = (useRow && cnt > 0) ? row[0] : 0xffffffffu;
#pragma unroll 1
        for (int g = 0; g < 3; ++g) {
#pragma unroll 1
            for (int qq = 0; qq < 6; ++qq) {
                const int src = (qq < 3 ? 0 : 9) + g * 3 + (qq % 3);
                const int s0 = __shfl(w.start, src, 64), ln = __shfl(w.len, src, 64), o = __shfl(w.off, src, 64);
                const bool isB = qq >= 3;
                if (!WANT_BOUNDARY && isB) continue;
                const float4* from = isB ? c.bposm : c.posm;
                for (int t = lane; t < ln; t += kTile) {
                    ldsPos[o + t] = from[s0 + t];
                    ldsField[o + t] = op.stage(isB, s0 + t + (isB ? c.bOff : 0));
                }
            }
            wave_lds_fence();
            while (useRow && k < cnt && (e >> 29) == (unsigned)g) {
                const bool isB = (e & kStreamBoundaryBit) != 0u;
                const int slot = (int)(e & kStreamSlotMask);
                ++k;
                const unsigned int next = (k < cnt) ? row[row_entry_offset(k)] : 0xffffffffu;
                if (WANT_BOUNDARY || !isB) {
                    const float4 pj = ldsPos[slot];
                    const float3 d = sub3(pi, v3(pj.x, pj.y, pj.z));
                    pair_dispatch(body, allPlain || (e & kStreamPlainBit) != 0u, ldsField[slot], isB, d, dot3(d, d), pj.w, -1);
                }
                e = next;
            }
            wave_lds_fence();
        }
        if (valid && !useRow)
            walk_cells<WANT_BOUNDARY>(c, pi, [&](int j, bool isB, float3 d, float r2, float mj) {
                body.template pair<false>(op.stage(isB, j + (isB ? c.bOff : 0)), isB, d, r2, mj, j);
            });
        return;
    }
    if (!valid) return;
    // one-gather mode (ops with a scalar neighbour field): uniform over the launch
    const bool packed = op_packed_scalar<Op>(op) && c.posf && c.massUniform && *c.massUniform != 0;
    const float m0 = packed ? c.posm[0].w : 0.0f;
    if (useRow) {
        // one-gather mode is uniform over the launch: two separate loops, so that each keeps its
        // single 16-byte gather per neighbour (a merged loop makes the compiler split the loads)
        // arithmetic mode, skin rows and one-gather mode are uniform over the launch: separate loops
        if (c.k.tol) {
            if (skin) {
                if (packed) walk_row<true, WANT_BOUNDARY, true, true>(op, c, row, cnt, m0, allPlain, pi, body);
                else walk_row<false, WANT_BOUNDARY, true, true>(op, c, row, cnt, m0, allPlain, pi, body);
            } else {
                if (packed) walk_row<true, WANT_BOUNDARY, false, true>(op, c, row, cnt, m0, allPlain, pi, body);
                else walk_row<false, WANT_BOUNDARY, false, true>(op, c, row, cnt, m0, allPlain, pi, body);
            }
        } else if (skin) {
            if (packed) walk_row<true, WANT_BOUNDARY, true, false>(op, c, row, cnt, m0, allPlain, pi, body);
            else walk_row<false, WANT_BOUNDARY, true, false>(op, c, row, cnt, m0, allPlain, pi, body);
        } else {
            if (packed) walk_row<true, WANT_BOUNDARY, false, false>(op, c, row, cnt, m0, allPlain, pi, body);
            else walk_row<false, WANT_BOUNDARY, false, false>(op, c, row, cnt, m0, allPlain, pi, body);
        }
        return;
    }
    walk_cells_fallback<WANT_BOUNDARY>(c, i, pi, [&](int j, bool isB, float3 d, float r2, float mj) {
        body.template pair<false>(op.stage(isB, j + (isB ? c.bOff : 0)), isB, d, r2, mj, j);
    });
}

// ------------------------------------------------------------------------------------------------------------
// Quad-per-particle walk.  The 4 lanes of a quad work on ONE particle: in a step, lane g evaluates entry 4s + g of its
// row.  The 4 entries of a chunk are consecutive accepted candidates of the cell walk, i.e. records that lie next to
// each other in the cell-sorted arrays, so the quad's gathers fall into one or two cache lines (lane-per-particle
// gathers touch ~50 lines per instruction and are bound by the L1's line rate: profiles/r02_ubench_sweep_structure.txt).
// The per-particle sums stay strictly in row order: each lane computes the term of its entry into a zeroed copy of the
// accumulators, then every lane of the quad adds the 4 terms one after the other (quad broadcasts), so all 4 lanes
// hold the same running sums as the lane-per-particle walk would.  (0 + t == t bit for bit, and a running sum that
// started at +0 is never -0, so routing a term through a zeroed accumulator changes nothing.)
// A Body opts in with  template <class F> void each_acc(Body& other, F f)  calling f(mine, others) per accumulator.
template <int G> __device__ __forceinline__ int quad_bcast_i(int v)
{
    constexpr int ctrl = G == 0 ? 0x00 : (G == 1 ? 0x55 : (G == 2 ? 0xAA : 0xFF));      // quad_perm:[G,G,G,G]
    return __builtin_amdgcn_mov_dpp(v, ctrl, 0xf, 0xf, true);
}
template <int G> __device__ __forceinline__ float quad_bcast_f(float v) { return __int_as_float(quad_bcast_i<G>(__float_as_int(v))); }

// adds the 4 lanes' terms to every lane's running sums, in lane order.  A dropped term (entry past the row's end,
// skipped boundary, pair beyond the support) is replaced by +0 first: x + (+0) == x for every x but -0, and a running
// sum that started at +0 cannot be -0.
template <class Body>
__device__ __forceinline__ void quad_accumulate(Body& body, Body& term, const bool use)
{
    body.each_acc(term, [&](float& a, float& t) {
        const float tz = use ? t : 0.0f;
        a += quad_bcast_f<0>(tz); a += quad_bcast_f<1>(tz); a += quad_bcast_f<2>(tz); a += quad_bcast_f<3>(tz);
    });
}

// tolerance walks: butterfly sum of the 4 lanes' partial sums; afterwards every lane of the quad holds the total
template <class Body>
__device__ __forceinline__ void quad_reduce(Body& body)
{
    body.each_acc(body, [](float& a, float&) {
        a += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(a), 0xB1, 0xf, 0xf, true));      // quad_perm:[1,0,3,2]
        a += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(a), 0x4E, 0xf, 0xf, true));      // quad_perm:[2,3,0,1]
    });
}

// U consecutive chunks [s, s + U) of the quad's row, straight-line: all row loads, then all gathers, then the terms in
// order.  (No early exit inside: a conditional between the chunks makes the compiler sink each chunk's loads next to
// their use, and the walk becomes load -> wait -> compute per chunk.)
#ifndef SPHX_QUAD_PAIR2
#define SPHX_QUAD_PAIR2 0      // packed two-chunk evaluation: -21 % VALU instructions but 90 instead of 72 VGPRs; measured slower
#endif
template <int U, bool PACKED, bool WANT_BOUNDARY, bool SKIN, bool TOL, class Op, class Body>
__device__ __forceinline__ void quad_chunks(const Op& op, const SweepCtx& c, const unsigned int* __restrict__ rowq, const int cnt,
                                            const int s, const float m0, const bool allPlain, const float3 pi, Body& body)
{
    const int g = threadIdx.x & 3;
    unsigned int e[U];
    bool ok[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        // unconditional load (chunks past this row's end hold stale entries), then: past the end -> record 0,
        // evaluated and dropped
        ok[u] = 4 * (s + u) + g < cnt;
        const unsigned int raw = rowq[(size_t)(s + u) * 256u];
        e[u] = ok[u] ? raw : 0u;
    }
    float4 pj[U];
    typename Op::Field f[U];
#pragma unroll
    for (int u = 0; u < U; ++u) fetch_pair<PACKED, Op>(op, c, m0, e[u], pj[u], f[u]);
    if constexpr (SPHX_QUAD_PAIR2 && has_pair2<Body>() && WANT_BOUNDARY && !SKIN && !TOL && U % 2 == 0) {
        // two chunks per packed evaluation: lane g holds entries 4s+g and 4(s+1)+g; their terms go to separate
        // zeroed accumulators and are added chunk by chunk, entry by entry
        bool fast = true;
#pragma unroll
        for (int u = 0; u < U; ++u) fast = fast && !(ok[u] && (e[u] & kPlainBit) != 0u);
        if (!__any(allPlain || !fast)) {
#pragma unroll
            for (int u = 0; u < U; u += 2) {
                Body ta = body, tb = body;
                ta.each_acc(ta, [](float& a, float&) { a = 0.0f; });
                tb.each_acc(tb, [](float& a, float&) { a = 0.0f; });
                body.pair2(ta, tb, f[u], f[u + 1], (e[u] & kBoundaryBit) != 0u, (e[u + 1] & kBoundaryBit) != 0u, pi, pj[u], pj[u + 1]);
                quad_accumulate(body, ta, ok[u]);
                quad_accumulate(body, tb, ok[u + 1]);
            }
            return;
        }
    }
    if constexpr (TOL) {
        // tolerance arithmetic: the order of a particle's sum is free, so every lane of the quad keeps its OWN partial sums
        // (entries 4s + g) and the quad adds them up once at the end (quad_reduce: two DPP steps per accumulator) instead of
        // the 4 ordered adds per entry of the strict walk.  A dropped entry enters with mass 0: every pair_tol term carries
        // the neighbour's mass as a factor.
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const bool isB = (e[u] & kBoundaryBit) != 0u;
            const float3 d = sub3(pi, v3(pj[u].x, pj[u].y, pj[u].z));
            const float r2 = dot3(d, d);
            bool use = ok[u];
            if (!WANT_BOUNDARY && isB) use = false;
            if (SKIN && r2 > c.k.tCut) use = false;
            body.pair_tol(f[u], isB, d, r2, use ? pj[u].w : 0.0f);
        }
        return;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const bool isB = (e[u] & kBoundaryBit) != 0u;
        const float3 d = sub3(pi, v3(pj[u].x, pj[u].y, pj[u].z));
        const float r2 = dot3(d, d);
        bool use = ok[u];
        if (!WANT_BOUNDARY && isB) use = false;
        if (SKIN && r2 > c.k.tCut) use = false;
        Body term = body;
        term.each_acc(term, [](float& a, float&) { a = 0.0f; });
        if (TOL) term.pair_tol(f[u], isB, d, r2, pj[u].w);
        else {
            const bool plain = use && (SKIN ? pair_needs_plain_ops(d, r2) : (e[u] & kPlainBit) != 0u);
            pair_dispatch(term, allPlain || plain, f[u], isB, d, r2, pj[u].w, (int)(e[u] & kIndexMask));
        }
        quad_accumulate(body, term, use);
    }
}

template <bool PACKED, bool WANT_BOUNDARY, bool SKIN, bool TOL, class Op, class Body>
__device__ __forceinline__ void walk_row_quad(const Op& op, const SweepCtx& c, const unsigned int* __restrict__ rowq, const int cnt,
                                              const float m0, const bool allPlain, const float3 pi, Body& body)
{
#ifndef SPHX_QUAD_U
#define SPHX_QUAD_U 4
#endif
    constexpr int U = SPHX_QUAD_U;             // chunks in flight
    int steps = (cnt + kRowChunk - 1) >> 2;    // the same in the 4 lanes of a quad; the wave runs to its longest row
#pragma unroll
    for (int off = 32; off >= 4; off >>= 1) steps = max(steps, __shfl_xor(steps, off, 64));
    int s = 0;
    for (; s + U <= steps; s += U) quad_chunks<U, PACKED, WANT_BOUNDARY, SKIN, TOL>(op, c, rowq, cnt, s, m0, allPlain, pi, body);
    for (; s < steps; ++s) quad_chunks<1, PACKED, WANT_BOUNDARY, SKIN, TOL>(op, c, rowq, cnt, s, m0, allPlain, pi, body);
    if constexpr (TOL) quad_reduce(body);
}

// The particle of this lane's quad in a quad-per-particle launch: one block of 4 waves per tile, wave w takes the
// particles [16 w, 16 w + 16) of the tile.  -1: the block is past the end.
// (lt: the launch-order tile number; a persistent block of the DFSPH loop tail walks several of them)
__device__ __forceinline__ int quad_particle_of(const SweepCtx& c, const int lt)
{
    if (lt >= c.numTiles) return -1;
    const int tile = c.tileOrder ? c.tileOrder[lt] : lt + (lt < c.tileSplit ? c.tile0 : c.tile1);
    if (c.tileOrder && tile_outside(c, tile)) return -1;
    // (r04: handing the particles of a tile to the waves sorted by row length -- counting sort with ballots in wave 0, a barrier --
    // was measured and removed: the quads of a wave then gather around non-adjacent particles, 61.1 -> 64.5 ms per post-impact
    // step at 10.3 M, +1 % strict and +6 % tolerance in free fall.)
    return tile * kTile + (int)(threadIdx.x >> 6) * 16 + (int)((threadIdx.x & 63) >> 2);
}
__device__ __forceinline__ int quad_particle(const SweepCtx& c)
{
    if (c.gate && *c.gate != 0) return -1;
    return quad_particle_of(c, logical_block());
}

// sweep() for a quad-per-particle launch (rows in global memory only; no LDS-streamed tiles)
template <bool WANT_BOUNDARY, class Op, class Body>
__device__ __forceinline__ void sweep_quad(const Op& op, const SweepCtx& c, const int i, const bool valid, const float3 pi, Body& body)
{
    const bool skin = c.stale != nullptr || (c.k.tol != 0 && c.persist != 0);
    const bool rows = c.nbr != nullptr && !(c.stale != nullptr && *c.stale != 0);
    const bool allPlain = !fast_paths_enabled(c.k);
    const int cnt = (rows && valid) ? c.nbrCount[i] : 0;
    bool useRow = rows && valid && cnt <= c.cap;
    if (c.stale != nullptr && useRow) {
        const int3 cNow = cell_of(pi, c.g);
        useRow = cell_id(cNow.x, cNow.y, cNow.z, c.g) == c.rowCell[i];
    }
    const bool packed = op_packed_scalar<Op>(op) && c.posf && c.massUniform && *c.massUniform != 0;
    const float m0 = packed ? c.posm[0].w : 0.0f;
    // every lane takes part in the row walk (wave-wide shuffles); lanes without a row run it with an empty one
    const unsigned int* rowq = rows ? c.nbr + row_base_offset(valid ? i : 0, c.cap) + (threadIdx.x & 3) : nullptr;
    const int len = useRow ? cnt : 0;
    if (rows) {
        if (c.k.tol) {
            if (skin) {
                if (packed) walk_row_quad<true, WANT_BOUNDARY, true, true>(op, c, rowq, len, m0, allPlain, pi, body);
                else walk_row_quad<false, WANT_BOUNDARY, true, true>(op, c, rowq, len, m0, allPlain, pi, body);
            } else {
                if (packed) walk_row_quad<true, WANT_BOUNDARY, false, true>(op, c, rowq, len, m0, allPlain, pi, body);
                else walk_row_quad<false, WANT_BOUNDARY, false, true>(op, c, rowq, len, m0, allPlain, pi, body);
            }
        } else if (skin) {
            if (packed) walk_row_quad<true, WANT_BOUNDARY, true, false>(op, c, rowq, len, m0, allPlain, pi, body);
            else walk_row_quad<false, WANT_BOUNDARY, true, false>(op, c, rowq, len, m0, allPlain, pi, body);
        } else {
            if (packed) walk_row_quad<true, WANT_BOUNDARY, false, false>(op, c, rowq, len, m0, allPlain, pi, body);
            else walk_row_quad<false, WANT_BOUNDARY, false, false>(op, c, rowq, len, m0, allPlain, pi, body);
        }
    }
    if (valid && !useRow)      // no rows, or this particle's row overflowed / went stale: the 4 lanes walk the cells alike
        walk_cells_fallback<WANT_BOUNDARY>(c, i, pi, [&](int j, bool isB, float3 d, float r2, float mj) {
            body.template pair<false>(op.stage(isB, j + (isB ? c.bOff : 0)), isB, d, r2, mj, j);
        });
}

// ------------------------------------------------------------------------------------------------------------
// Duo walk: TWO lanes per particle.  Lane h of a pair evaluates entries h and h + 2 of every chunk (so the two lanes
// gather adjacent records in each instruction and a quad touches 2 x ~1.2 lines instead of ~3.5), both with the packed
// two-entry arithmetic where the body has it.  The running sums are handed back and forth: entry order is
// lane0.A, lane1.A, lane0.B, lane1.B, and every hop is  a = swap(a) + term  executed by both lanes (the lane that does
// not own the slot computes a value nobody reads).  After every chunk the sums sit in lane 1, which stores the results.
__device__ __forceinline__ float duo_swap_f(float v) { return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0xB1, 0xf, 0xf, true)); }   // quad_perm:[1,0,3,2]

template <class Body>
__device__ __forceinline__ void duo_hops(Body& body, Body& ta, Body& tb, const bool okA, const bool okB)
{
    // dropped terms become +0 (see quad_accumulate)
    ta.each_acc(ta, [&](float& a, float&) { a = okA ? a : 0.0f; });
    tb.each_acc(tb, [&](float& a, float&) { a = okB ? a : 0.0f; });
    body.each_acc(ta, [&](float& a, float& t) { a = duo_swap_f(a) + t; a = duo_swap_f(a) + t; });
    body.each_acc(tb, [&](float& a, float& t) { a = duo_swap_f(a) + t; a = duo_swap_f(a) + t; });
}

template <int U, bool PACKED, bool WANT_BOUNDARY, bool SKIN, bool TOL, class Op, class Body>
__device__ __forceinline__ void duo_chunks(const Op& op, const SweepCtx& c, const unsigned int* __restrict__ rowd, const int cnt,
                                           const int s, const float m0, const bool allPlain, const float3 pi, Body& body)
{
    const int h = threadIdx.x & 1;
    unsigned int e[U][2];
    bool ok[U][2];
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
        for (int w = 0; w < 2; ++w) {
            ok[u][w] = 4 * (s + u) + h + 2 * w < cnt;
            const unsigned int raw = rowd[(size_t)(s + u) * 256u + 2 * w];
            e[u][w] = ok[u][w] ? raw : 0u;
        }
    float4 pj[U][2];
    typename Op::Field f[U][2];
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
        for (int w = 0; w < 2; ++w) fetch_pair<PACKED, Op>(op, c, m0, e[u][w], pj[u][w], f[u][w]);
#pragma unroll
    for (int u = 0; u < U; ++u) {
        Body ta = body, tb = body;
        ta.each_acc(ta, [](float& a, float&) { a = 0.0f; });
        tb.each_acc(tb, [](float& a, float&) { a = 0.0f; });
        bool useA = ok[u][0], useB = ok[u][1];
        const bool isBa = (e[u][0] & kBoundaryBit) != 0u, isBb = (e[u][1] & kBoundaryBit) != 0u;
        bool done = false;
        if constexpr (has_pair2<Body>() && WANT_BOUNDARY && !SKIN && !TOL) {
            const bool plain = (useA && (e[u][0] & kPlainBit) != 0u) || (useB && (e[u][1] & kPlainBit) != 0u);
            if (!__any(allPlain || plain)) {
                body.pair2(ta, tb, f[u][0], f[u][1], isBa, isBb, pi, pj[u][0], pj[u][1]);
                done = true;
            }
        }
        if (!done) {
#pragma unroll
            for (int w = 0; w < 2; ++w) {
                Body& t = w ? tb : ta;
                bool& use = w ? useB : useA;
                const bool isB = w ? isBb : isBa;
                const float3 d = sub3(pi, v3(pj[u][w].x, pj[u][w].y, pj[u][w].z));
                const float r2 = dot3(d, d);
                if (!WANT_BOUNDARY && isB) use = false;
                if (SKIN && r2 > c.k.tCut) use = false;
                if (TOL) t.pair_tol(f[u][w], isB, d, r2, pj[u][w].w);
                else {
                    const bool plain = use && (SKIN ? pair_needs_plain_ops(d, r2) : (e[u][w] & kPlainBit) != 0u);
                    pair_dispatch(t, allPlain || plain, f[u][w], isB, d, r2, pj[u][w].w, (int)(e[u][w] & kIndexMask));
                }
            }
        }
        duo_hops(body, ta, tb, useA, useB);
    }
}

template <bool PACKED, bool WANT_BOUNDARY, bool SKIN, bool TOL, class Op, class Body>
__device__ __forceinline__ void walk_row_duo(const Op& op, const SweepCtx& c, const unsigned int* __restrict__ rowd, const int cnt,
                                             const float m0, const bool allPlain, const float3 pi, Body& body)
{
    constexpr int U = 2;                       // chunks in flight = 4 entries per lane
    int steps = (cnt + kRowChunk - 1) >> 2;    // the same in both lanes of a pair; the wave runs to its longest row
#pragma unroll
    for (int off = 32; off >= 2; off >>= 1) steps = max(steps, __shfl_xor(steps, off, 64));
    int s = 0;
    for (; s + U <= steps; s += U) duo_chunks<U, PACKED, WANT_BOUNDARY, SKIN, TOL>(op, c, rowd, cnt, s, m0, allPlain, pi, body);
    for (; s < steps; ++s) duo_chunks<1, PACKED, WANT_BOUNDARY, SKIN, TOL>(op, c, rowd, cnt, s, m0, allPlain, pi, body);
}

// The particle of this lane's pair in a duo launch: a block of 4 waves covers two tiles, wave w takes the particles
// [32 (w & 1), +32) of tile 2 b + (w >> 1).  -1: the wave is past the end.
__device__ __forceinline__ int duo_particle(const SweepCtx& c)
{
    if (c.gate && *c.gate != 0) return -1;
    const int lt = logical_block() * 2 + (int)(threadIdx.x >> 7);
    if (lt >= c.numTiles) return -1;
    const int tile = c.tileOrder ? c.tileOrder[lt] : lt + (lt < c.tileSplit ? c.tile0 : c.tile1);
    if (c.tileOrder && tile_outside(c, tile)) return -1;
    return tile * kTile + (int)((threadIdx.x >> 6) & 1) * 32 + (int)((threadIdx.x & 63) >> 1);
}

template <bool WANT_BOUNDARY, class Op, class Body>
__device__ __forceinline__ void sweep_duo(const Op& op, const SweepCtx& c, const int i, const bool valid, const float3 pi, Body& body)
{
    const bool skin = c.stale != nullptr || (c.k.tol != 0 && c.persist != 0);
    const bool rows = c.nbr != nullptr && !(c.stale != nullptr && *c.stale != 0);
    const bool allPlain = !fast_paths_enabled(c.k);
    const int cnt = (rows && valid) ? c.nbrCount[i] : 0;
    bool useRow = rows && valid && cnt <= c.cap;
    if (c.stale != nullptr && useRow) {
        const int3 cNow = cell_of(pi, c.g);
        useRow = cell_id(cNow.x, cNow.y, cNow.z, c.g) == c.rowCell[i];
    }
    const bool packed = op_packed_scalar<Op>(op) && c.posf && c.massUniform && *c.massUniform != 0;
    const float m0 = packed ? c.posm[0].w : 0.0f;
    const unsigned int* rowd = rows ? c.nbr + row_base_offset(valid ? i : 0, c.cap) + (threadIdx.x & 1) : nullptr;
    const int len = useRow ? cnt : 0;
    if (rows) {
        if (c.k.tol) {
            if (skin) {
                if (packed) walk_row_duo<true, WANT_BOUNDARY, true, true>(op, c, rowd, len, m0, allPlain, pi, body);
                else walk_row_duo<false, WANT_BOUNDARY, true, true>(op, c, rowd, len, m0, allPlain, pi, body);
            } else {
                if (packed) walk_row_duo<true, WANT_BOUNDARY, false, true>(op, c, rowd, len, m0, allPlain, pi, body);
                else walk_row_duo<false, WANT_BOUNDARY, false, true>(op, c, rowd, len, m0, allPlain, pi, body);
            }
        } else if (skin) {
            if (packed) walk_row_duo<true, WANT_BOUNDARY, true, false>(op, c, rowd, len, m0, allPlain, pi, body);
            else walk_row_duo<false, WANT_BOUNDARY, true, false>(op, c, rowd, len, m0, allPlain, pi, body);
        } else {
            if (packed) walk_row_duo<true, WANT_BOUNDARY, false, false>(op, c, rowd, len, m0, allPlain, pi, body);
            else walk_row_duo<false, WANT_BOUNDARY, false, false>(op, c, rowd, len, m0, allPlain, pi, body);
        }
    }
    if (valid && !useRow)      // both lanes walk the cells alike: the sums end up in lane 1 as well
        walk_cells_fallback<WANT_BOUNDARY>(c, i, pi, [&](int j, bool isB, float3 d, float r2, float mj) {
            body.template pair<false>(op.stage(isB, j + (isB ? c.bOff : 0)), isB, d, r2, mj, j);
        });
}

// Row construction for one wave = one tile.  Same walk as walk_cells; the three z-adjacent cells of
// a (dx,dy) column are contiguous in memory, so when they hold no boundary particles the fluid
// ranges are visited as one run (identical order).  With `ldsPos` (streamed tile, fmt 2) the wave
// stages one dx group at a time, candidates are read from LDS and entries carry (group, slot).
#ifndef SPHX_BUILD_AHEAD
#define SPHX_BUILD_AHEAD 4
#endif
constexpr int kBuildAhead = SPHX_BUILD_AHEAD;
constexpr int kRowStage = 32;      // entries per lane staged in LDS by the row builder (longer rows continue in global memory)
#ifndef SPHX_BUILD_REGSTAGE
#define SPHX_BUILD_REGSTAGE 1      // a lane collects 4 entries in registers and stores its own 16-byte chunks; 0: rows staged in LDS,
                                   // one KB per chunk and tile stored at the end (2 % slower at 10 M, 13 % at 263 k: LDS caps the occupancy at 5)
#endif
__device__ __forceinline__ void put_entry(const SweepCtx& c, unsigned int* stage, unsigned int* row, int lane, int cnt, unsigned int e,
                                          uint4& pend)
{
    if (SPHX_BUILD_REGSTAGE && !stage) {
        const int w = cnt & 3;
        pend.x = w == 0 ? e : pend.x; pend.y = w == 1 ? e : pend.y; pend.z = w == 2 ? e : pend.z; pend.w = w == 3 ? e : pend.w;
        if (w == 3 && cnt < c.cap) *reinterpret_cast<uint4*>(row + (size_t)(cnt >> 2) * 256u) = pend;
        return;
    }
    if (stage && cnt < kRowStage) stage[cnt * 64 + lane] = e;
    else if (cnt < c.cap) row[row_entry_offset(cnt)] = e;
}
// `stage`: this wave's LDS staging area of kRowStage x 64 entries (SPHX_BUILD_REGSTAGE = 0), or nullptr.  A lane appends at its own count, so
// written straight to the wave-interleaved global layout the 64 lanes touch 64 different 256-byte lines at any
// moment and every line is completed by 64 separate 4-byte stores spread over the whole walk (measured: 4.2x write
// amplification at 10 M particles).  Staged, the rows are written once at the end, one full line per entry index.
__device__ __forceinline__ void build_neighbor_rows(const SweepCtx& c, float4* ldsPos, const bool streamed,
                                                    unsigned int* nbr, int* nbrCount, const int i, const bool valid,
                                                    unsigned int* stage = nullptr)
{
    const int lane = threadIdx.x & 63;
    const float4 self = valid ? c.posm[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    const float3 pi = v3(self.x, self.y, self.z);
    unsigned int* row = nbr + row_base_offset(i, c.cap);
    int cnt = 0;
    uint4 pend = make_uint4(0u, 0u, 0u, 0u);          // register-staged chunk (SPHX_BUILD_REGSTAGE)
    const int3 c0 = cell_of(pi, c.g);
    const int zlo = max(c0.z - 1, 0), zhi = min(c0.z + 1, c.g.gz - 1);
    // the plain-operator flag of an entry is read by the strict walks only (the tolerance arithmetic has no exact fast paths to
    // guard): launch-uniform
    const bool wantPlain = c.k.tol == 0 || c.plainBits != 0;
    WaveRanges w; w.start = w.len = w.off = 0; w.ok = false;
    if (streamed) w = wave_ranges(c, (i >> 6) << 6);
#pragma unroll 1
    for (int dx = -1; dx <= 1; ++dx) {
        const int g = dx + 1;
        if (streamed) {
#pragma unroll 1
            for (int qq = 0; qq < 6; ++qq) {
                const int src = (qq < 3 ? 0 : 9) + g * 3 + (qq % 3);
                const int s0 = __shfl(w.start, src, 64), ln = __shfl(w.len, src, 64), o = __shfl(w.off, src, 64);
                const float4* from = qq >= 3 ? c.bposm : c.posm;
                for (int t = lane; t < ln; t += kTile) ldsPos[o + t] = from[s0 + t];
            }
            wave_lds_fence();
        }
        // slot = j + shift for the three columns of this group (shuffles stay in uniform control flow)
        int fSh[3] = {0, 0, 0}, bSh[3] = {0, 0, 0};
        if (streamed) {
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                fSh[t] = __shfl(w.off, g * 3 + t, 64) - __shfl(w.start, g * 3 + t, 64);
                bSh[t] = __shfl(w.off, 9 + g * 3 + t, 64) - __shfl(w.start, 9 + g * 3 + t, 64);
            }
        }
        const int X = c0.x + dx;
        if (valid && X >= 0 && X < c.g.gx && zlo <= zhi) {
            for (int dy = -1; dy <= 1; ++dy) {
                const int Y = c0.y + dy;
                if (Y < 0 || Y >= c.g.gy) continue;
                const int base = (X * c.g.gy + Y) * c.g.gz;
                const int fShift = dy < 0 ? fSh[0] : (dy == 0 ? fSh[1] : fSh[2]);
                const int bShift = streamed ? (dy < 0 ? bSh[0] : (dy == 0 ? bSh[1] : bSh[2])) : c.bOff;
                const unsigned int plainBit = streamed ? kStreamPlainBit : kPlainBit;
                const unsigned int fTag = streamed ? ((unsigned)g << 29) : 0u;
                const unsigned int bTag = streamed ? (((unsigned)g << 29) | kStreamBoundaryBit) : kBoundaryBit;
                const bool noWall = c.csB[base + zlo] == c.csB[base + zhi + 1];
                const int step = noWall ? (zhi - zlo + 1) : 1;
                for (int z = zlo; z <= zhi; z += step) {
                    const int cell = base + z;
                    const int e = c.csF[cell + step];
                    int j = c.csF[cell];
                    // kBuildAhead candidates per trip: the loads are independent, the appends stay in order
                    for (; j + kBuildAhead <= e; j += kBuildAhead) {
                        float4 pj[kBuildAhead];
#pragma unroll
                        for (int u = 0; u < kBuildAhead; ++u) pj[u] = streamed ? ldsPos[j + u + fShift] : c.posm[j + u];
#pragma unroll
                        for (int u = 0; u < kBuildAhead; ++u) {
                            const float3 d = sub3(pi, v3(pj[u].x, pj[u].y, pj[u].z));
                            const float r2 = dot3(d, d);
                            if (r2 > c.buildCut || j + u == i) continue;
                            put_entry(c, stage, row, lane, cnt, (unsigned int)(j + u + fShift) | fTag | ((wantPlain && pair_needs_plain_ops(d, r2)) ? plainBit : 0u), pend);
                            ++cnt;
                        }
                    }
                    for (; j < e; ++j) {
                        const float4 pj = streamed ? ldsPos[j + fShift] : c.posm[j];
                        const float3 d = sub3(pi, v3(pj.x, pj.y, pj.z));
                        const float r2 = dot3(d, d);
                        if (r2 > c.buildCut || j == i) continue;
                        put_entry(c, stage, row, lane, cnt, (unsigned int)(j + fShift) | fTag | ((wantPlain && pair_needs_plain_ops(d, r2)) ? plainBit : 0u), pend);
                        ++cnt;
                    }
                    if (!noWall) {
                        const int eb = c.csB[cell + 1];
                        for (int j = c.csB[cell]; j < eb; ++j) {
                            const float4 pj = streamed ? ldsPos[j + bShift] : c.bposm[j];
                            const float3 d = sub3(pi, v3(pj.x, pj.y, pj.z));
                            const float r2 = dot3(d, d);
                            if (r2 > c.buildCut) continue;
                            put_entry(c, stage, row, lane, cnt, (unsigned int)(j + bShift) | bTag | ((wantPlain && pair_needs_plain_ops(d, r2)) ? plainBit : 0u), pend);   // bShift: + bOff (fmt 0)
                            ++cnt;
                        }
                    }
                }
            }
        }
        if (streamed) wave_lds_fence();
    }
    if (valid) nbrCount[i] = cnt;
    if (valid && cnt > c.cap && c.overflowMax) atomicMax(c.overflowMax, cnt);     // (rare: the host enlarges the rows, SweepCache::tuneRowCapacity)
    if (SPHX_BUILD_REGSTAGE && !stage && valid && (cnt & 3) != 0 && cnt < c.cap) *reinterpret_cast<uint4*>(row + (size_t)(cnt >> 2) * 256u) = pend;
    if (stage) {                                  // one coalesced 1 KB store per chunk index (slots past a row's end hold
        wave_lds_fence();                         // stale stage contents: readers never look past nbrCount)
        int top = valid ? min(min(cnt, c.cap), kRowStage) : 0;
        const int own = top;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) top = max(top, __shfl_xor(top, off, 64));
        for (int k = 0; k < top; k += kRowChunk)
            if (k < own)
                *reinterpret_cast<uint4*>(row + (size_t)(k >> 2) * 256u) =
                    make_uint4(stage[k * 64 + lane], stage[(k + 1) * 64 + lane], stage[(k + 2) * 64 + lane], stage[(k + 3) * 64 + lane]);
    }
}

// Skin rows: what a position update checks for the particle it moved (SweepCache::skinWatch).  Farther from where the rows were
// built than the skin allows (also for a NaN) -> every row is stale (`stale`).  In another cell than the one ITS row was built
// around -> only that row is: the particle goes onto the list of this update (r06; k_build_list_changed rebuilds those rows in
// front of the next sweep -- after the landing some particle crosses a cell face in nearly every Jacobi iteration, which used to
// cost a rebuild of ALL rows per iteration).  No list, or a full one: all rows stale, as before.
struct SkinWatch {
    const float4* posBuild = nullptr; const int* rowCell = nullptr; int* stale = nullptr; float limit2 = 0.0f;
    int* changedCount = nullptr; int* changedList = nullptr; int listCap = 0;
};
__device__ __forceinline__ void skin_watch(const SkinWatch& w, const float3 p, const int i, const GridDesc& g)
{
    const float3 d = sub3(p, xyz4(w.posBuild[i]));
    if (!(dot3(d, d) <= w.limit2)) { *w.stale = 1; return; }
    const int3 cNow = cell_of(p, g);
    if (cell_id(cNow.x, cNow.y, cNow.z, g) == w.rowCell[i]) return;
    if (!w.changedList) { *w.stale = 1; return; }
    const int at = atomicAdd(w.changedCount, 1);
    if (at < w.listCap) w.changedList[at] = i;
    else *w.stale = 1;
}

// Small scenes (r06): the same rows from 16 lanes per particle.  The lane-per-particle walk above is a chain of ~70 dependent
// load rounds (9 columns x (cell tables, then candidates four at a time)); with fewer waves than the device has SIMDs that latency
// IS the builder's time (42 us at the reference scene's 20,736 particles: 40 % of a WCSPH step, and what PBD pays per Jacobi
// iteration once the fluid has landed).  Here lane q < 9 of a group of 16 walks column q = 3 (dx + 1) + (dy + 1) of the particle's
// 3 x 3 x 3 cells -- the order of the single-lane walk -- keeping what it would append in LDS, and stores it behind a prefix sum
// over the group's counts.  Same entries in the same order (entry k of a row lives at row_entry_offset(k) whoever writes it).
template <int AHEAD, class Emit>
__device__ __forceinline__ void walk_build_column(const SweepCtx& c, const float4* __restrict__ cand, const float3 pi, const int i, const int X,
                                                  const int Y, const int zlo, const int zhi, const bool wantPlain, Emit emit)
{
    const int base = (X * c.g.gy + Y) * c.g.gz;
    const bool noWall = c.csB[base + zlo] == c.csB[base + zhi + 1];
    const int step = noWall ? (zhi - zlo + 1) : 1;
    for (int z = zlo; z <= zhi; z += step) {
        const int cell = base + z;
        const int e = c.csF[cell + step];
        int j = c.csF[cell];
        for (; j + AHEAD <= e; j += AHEAD) {
            float4 pj[AHEAD];
#pragma unroll
            for (int u = 0; u < AHEAD; ++u) pj[u] = cand[j + u];
#pragma unroll
            for (int u = 0; u < AHEAD; ++u) {
                const float3 d = sub3(pi, v3(pj[u].x, pj[u].y, pj[u].z));
                const float r2 = dot3(d, d);
                if (r2 > c.buildCut || j + u == i) continue;
                emit((unsigned int)(j + u) | ((wantPlain && pair_needs_plain_ops(d, r2)) ? kPlainBit : 0u));
            }
        }
        for (; j < e; ++j) {
            const float4 pj = cand[j];
            const float3 d = sub3(pi, v3(pj.x, pj.y, pj.z));
            const float r2 = dot3(d, d);
            if (r2 > c.buildCut || j == i) continue;
            emit((unsigned int)j | ((wantPlain && pair_needs_plain_ops(d, r2)) ? kPlainBit : 0u));
        }
        if (!noWall) {
            const int eb = c.csB[cell + 1];
            for (int jb = c.csB[cell]; jb < eb; ++jb) {
                const float4 pj = c.bposm[jb];
                const float3 d = sub3(pi, v3(pj.x, pj.y, pj.z));
                const float r2 = dot3(d, d);
                if (r2 > c.buildCut) continue;
                emit((unsigned int)(jb + c.bOff) | kBoundaryBit | ((wantPlain && pair_needs_plain_ops(d, r2)) ? kPlainBit : 0u));
            }
        }
    }
}
#ifndef SPHX_GROUP_AHEAD
#define SPHX_GROUP_AHEAD 2      // candidates in flight per lane: 2 keep the kernel at 56 VGPRs (8 waves per SIMD); 4 need 95
#endif
constexpr int kGroupAhead = SPHX_GROUP_AHEAD;
constexpr int kBuildGroup = 16;            // lanes per particle (9 walk a column each)
constexpr int kGroupStash = 16;            // entries a lane keeps in LDS between counting and storing (16 KB per block of 256: 8 blocks per CU)
// `cand`: the fluid positions the distances are taken from -- the live ones (c.posm) for a build of everything; the positions of
// the LAST such build for the rows of particles that changed their cell since (k_build_list_changed): the row around the new cell
// that build would have made, so the skin's allowance still counts from there.  The cell is always the one of the live position.
__device__ __forceinline__ int build_neighbor_rows_group(const SweepCtx& c, const float4* __restrict__ cand, unsigned int* nbr, int* nbrCount,
                                                         const int i, const bool valid, unsigned int* stash)
{
    const int q = (int)(threadIdx.x & (kBuildGroup - 1));
    const float4 self = valid ? cand[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    const float3 pi = v3(self.x, self.y, self.z);
    const int3 c0 = cell_of(valid ? xyz4(c.posm[i]) : pi, c.g);
    const int zlo = max(c0.z - 1, 0), zhi = min(c0.z + 1, c.g.gz - 1);
    const bool wantPlain = c.k.tol == 0 || c.plainBits != 0;
    const int X = c0.x + q / 3 - 1, Y = c0.y + q % 3 - 1;
    const bool column = valid && q < 9 && X >= 0 && X < c.g.gx && Y >= 0 && Y < c.g.gy && zlo <= zhi;
    unsigned int* row = nbr + row_base_offset(valid ? i : 0, c.cap);
    // a lane keeps the first kGroupStash entries of its column in LDS (slot k of thread t at stash[k * blockDim + t]: no bank
    // conflicts); a column with more -- compressed states -- is walked a second time for the rest
    int mine = 0;
    if (column) walk_build_column<kGroupAhead>(c, cand, pi, i, X, Y, zlo, zhi, wantPlain, [&](unsigned int e) {
        if (mine < kGroupStash) stash[mine * kWideBlock + (int)threadIdx.x] = e;
        ++mine;
    });
    int upTo = mine;                         // inclusive prefix over the group's lanes
#pragma unroll
    for (int off = 1; off < kBuildGroup; off <<= 1) {
        const int t = __shfl_up(upTo, off, kBuildGroup);
        if (q >= off) upTo += t;
    }
    const int total = __shfl(upTo, kBuildGroup - 1, kBuildGroup);
    const int first = upTo - mine;
    const int kept = min(mine, kGroupStash);
    for (int k = 0; k < kept; ++k)
        if (first + k < c.cap) row[row_entry_offset(first + k)] = stash[k * kWideBlock + (int)threadIdx.x];
    if (mine > kGroupStash) {
        int at = first;
        walk_build_column<kGroupAhead>(c, cand, pi, i, X, Y, zlo, zhi, wantPlain, [&](unsigned int e) {
            if (at >= first + kGroupStash && at < c.cap) row[row_entry_offset(at)] = e;
            ++at;
        });
    }
    if (valid && q == 0) {
        nbrCount[i] = total;
        if (total > c.cap && c.overflowMax) atomicMax(c.overflowMax, total);
    }
    return cell_id(c0.x, c0.y, c0.z, c.g);
}


// ------------------------------------------------------------------------------------------------------------
// Compact-brick LDS stage (r03; tolerance arithmetic only; the north star's "LDS-staged neighbour cells").
// A block of kBrickThreads owns the particles of a brick of 4 x 4 cell columns x 4 cells along z.  z is the fastest cell axis, so
// the brick's part of a column is ONE contiguous particle run and the neighbourhood of the brick is 6 x 6 runs covering
// [z0 - 1, z0 + 4]: they are staged with coalesced loads as 16-byte LDS records -- position+mass, and the one per-neighbour field
// of the sweep -- fluid runs first, then the boundary runs.  Rows hold 16-bit LDS slots (bit 15: boundary), 8 per 16-byte chunk,
// in the ordinary row storage; a pair costs two ds_read_b128 instead of two divergent global gathers (measured -15..-21 % per
// sweep against the quad walk at 10 M particles, profiles/r03_ubench_brick.txt; nothing under strict arithmetic, whose exact
// chains bind first, so the strict path never uses it).  A brick whose stage would not fit kBrickSlots is processed in 2 or 4
// slices along z (same rule in the builder and in every sweep: it depends on the cell tables only).
constexpr int kBrickEdge = 4;
constexpr int kBrickThreads = 512;
constexpr int kBrickSlots = 2304;           // staged records per slice: 36 KB per 16-byte field
constexpr int kBrickRuns = 36;              // (kBrickEdge + 2)^2 halo columns
struct BrickTables {
    int runStart[2 * kBrickRuns];           // [0,36): fluid runs, [36,72): boundary runs (index into the fluid / boundary array)
    int runLen[2 * kBrickRuns];
    int runBase[2 * kBrickRuns];            // LDS slot of the run's first record
    int ownStart[16], ownLen[16], ownEnd[16];   // the brick's own z-runs (fluid), inclusive prefix of their lengths
    int staged, stagedFluid, own;
    int x0, y0, z0;                             // first cell of the brick
    int pad[2];
};
static_assert(sizeof(BrickTables) % 16 == 0, "tables are copied as 16-byte words");
struct BrickGeom { int x0, y0, z0; bool any; };

__device__ __forceinline__ BrickGeom brick_geom(const SweepCtx& c, int b)
{
    const int nbz = (c.g.gz + kBrickEdge - 1) / kBrickEdge, nby = (c.g.gy + kBrickEdge - 1) / kBrickEdge, nbx = (c.g.gx + kBrickEdge - 1) / kBrickEdge;
    BrickGeom G;
    G.any = b < nbx * nby * nbz;
    G.z0 = (b % nbz) * kBrickEdge; G.y0 = ((b / nbz) % nby) * kBrickEdge; G.x0 = (b / (nbz * nby)) * kBrickEdge;
    return G;
}
inline int brick_count(const GridDesc& g)
{
    return ((g.gx + kBrickEdge - 1) / kBrickEdge) * ((g.gy + kBrickEdge - 1) / kBrickEdge) * ((g.gz + kBrickEdge - 1) / kBrickEdge);
}
// inclusive scan over the 64 lanes of a wave
__device__ __forceinline__ int wave_inclusive_scan(int v)
{
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { const int t = __shfl_up(v, off, 64); if (lane >= off) v += t; }
    return v;
}
// Tables of one slice [zs0, zs1) of the brick at (x0, y0): waves 0 and 1 of the block fill the fluid / boundary halo runs, wave 2
// the own runs.  Ends with a block barrier.  (Every thread of the block must call it.)
__device__ __forceinline__ void brick_slice_tables(const SweepCtx& c, BrickTables& T, int x0, int y0, int zs0, int zs1)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (wave < 2) {
        int start = 0, len = 0;
        if (lane < kBrickRuns) {
            const int X = x0 - 1 + lane / 6, Y = y0 - 1 + lane % 6;
            const int zlo = max(zs0 - 1, 0), zhi = min(zs1, c.g.gz - 1);
            if (X >= 0 && X < c.g.gx && Y >= 0 && Y < c.g.gy && zlo <= zhi) {
                const int* cs = wave == 0 ? c.csF : c.csB;
                const int base = (X * c.g.gy + Y) * c.g.gz;
                start = cs[base + zlo]; len = cs[base + zhi + 1] - start;
            }
        }
        const int end = wave_inclusive_scan(len);
        if (lane < kBrickRuns) { T.runStart[wave * kBrickRuns + lane] = start; T.runLen[wave * kBrickRuns + lane] = len; T.runBase[wave * kBrickRuns + lane] = end - len; }
        if (lane == 63) { if (wave == 0) T.stagedFluid = end; else T.staged = end; }      // (boundary total for now)
    } else if (wave == 2) {
        int start = 0, len = 0;
        if (lane < 16) {
            const int X = x0 + lane / 4, Y = y0 + lane % 4;
            const int z1 = min(zs1, c.g.gz);
            if (X < c.g.gx && Y < c.g.gy && zs0 < z1) {
                const int base = (X * c.g.gy + Y) * c.g.gz;
                start = c.csF[base + zs0]; len = c.csF[base + z1] - start;
            }
        }
        const int end = wave_inclusive_scan(len);
        if (lane < 16) { T.ownStart[lane] = start; T.ownLen[lane] = len; T.ownEnd[lane] = end; }
        if (lane == 63) T.own = end;
    }
    __syncthreads();
    if (threadIdx.x < kBrickRuns) T.runBase[kBrickRuns + threadIdx.x] += T.stagedFluid;     // boundary records follow the fluid ones
    if (threadIdx.x == 64) T.staged += T.stagedFluid;
    __syncthreads();
}
// number of z slices the brick is processed in: the smallest of 1, 2, 4 whose slices all fit the stage (0: not even single cells do)
__device__ __forceinline__ int brick_parts(const SweepCtx& c, BrickTables& T, const BrickGeom& G)
{
#pragma unroll 1
    for (int parts = 1; parts <= kBrickEdge; parts *= 2) {
        const int h = kBrickEdge / parts;
        bool fits = true;
#pragma unroll 1
        for (int sl = 0; sl < parts; ++sl) {
            brick_slice_tables(c, T, G.x0, G.y0, G.z0 + sl * h, G.z0 + (sl + 1) * h);
            fits = fits && T.staged <= kBrickSlots;            // (block-uniform: read after the barrier)
            __syncthreads();
        }
        if (fits) return parts;
    }
    return 0;
}
// own particle `p` of the current slice -> global index (p < T.own)
__device__ __forceinline__ int brick_own_index(const BrickTables& T, int p)
{
    int col = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) col += (p >= T.ownEnd[k]) ? 1 : 0;
    return T.ownStart[col] + (p - (T.ownEnd[col] - T.ownLen[col]));
}
// stage the halo runs: positions (+mass) always, and the sweep's per-neighbour field when `lf` is given
template <class Field, class Stage>
__device__ __forceinline__ void brick_stage(const SweepCtx& c, const BrickTables& T, float4* lp, Field* lf, Stage&& stage)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int r = wave; r < 2 * kBrickRuns; r += kBrickThreads / 64) {
        const int len = T.runLen[r];
        if (len <= 0) continue;
        const bool isB = r >= kBrickRuns;
        const int u0 = T.runStart[r] + (isB ? c.bOff : 0), b0 = T.runBase[r];
        for (int t = lane; t < len; t += 64) {
            lp[b0 + t] = c.posm[u0 + t];
            if (lf) lf[b0 + t] = stage(isB, u0 + t);
        }
    }
    __syncthreads();
}
// copies the tables k_brick_list left for brick `b` of the list into this block's LDS (ends with a block barrier)
__device__ __forceinline__ void brick_load_tables(const SweepCtx& c, BrickTables& T, int b)
{
    const uint4* src = reinterpret_cast<const uint4*>(c.brickTab + b);
    uint4* dst = reinterpret_cast<uint4*>(&T);
    if (threadIdx.x < sizeof(BrickTables) / 16) dst[threadIdx.x] = src[threadIdx.x];
    __syncthreads();
}
// The slices of brick `b` of the list, one after the other: `perSlice(T)` runs with the slice's tables in LDS (every thread of the
// block calls it; it must end with a block barrier).  Blocks stride over the list, so any launch size covers it.
template <class PerSlice>
__device__ __forceinline__ void brick_for_each_slice(const SweepCtx& c, BrickTables& T, PerSlice&& perSlice)
{
    const int count = *c.brickCount;
#pragma unroll 1
    for (int b = logical_block(); b < count; b += (int)gridDim.x) {
        brick_load_tables(c, T, b);
        if (T.staged <= kBrickSlots) { perSlice(T); continue; }
        BrickGeom G; G.x0 = T.x0; G.y0 = T.y0; G.z0 = T.z0; G.any = true;
        __syncthreads();
        const int parts = brick_parts(c, T, G);
        if (parts == 0) { if (threadIdx.x == 0 && c.brickFault) *c.brickFault = 1; continue; }
        const int h = kBrickEdge / parts;
#pragma unroll 1
        for (int sl = 0; sl < parts; ++sl) {
            brick_slice_tables(c, T, G.x0, G.y0, G.z0 + sl * h, G.z0 + (sl + 1) * h);
            perSlice(T);
        }
    }
}
constexpr unsigned int kBrickBoundaryBit = 0x8000u;
// The record an op stages per neighbour is its Field, unless the op names a more compact BrickField (with brick_pack /
// brick_unpack): the stage must stay at 16 bytes per record for two blocks to share a CU.
template <class Op, class = void> struct BrickFieldOf { using type = typename Op::Field; };
template <class Op> struct BrickFieldOf<Op, std::void_t<typename Op::BrickField>> { using type = typename Op::BrickField; };
template <class Op> using brick_field_t = typename BrickFieldOf<Op>::type;
template <class Op> __device__ __forceinline__ auto brick_pack_impl(const Op& op, const typename Op::Field& f, int) -> decltype(op.brick_pack(f)) { return op.brick_pack(f); }
template <class Op> __device__ __forceinline__ typename Op::Field brick_pack_impl(const Op&, const typename Op::Field& f, long) { return f; }
template <class Op> __device__ __forceinline__ brick_field_t<Op> brick_pack(const Op& op, const typename Op::Field& f) { return brick_pack_impl(op, f, 0); }
template <class Op> __device__ __forceinline__ auto brick_unpack_impl(const Op& op, const brick_field_t<Op>& t, int) -> decltype(op.brick_unpack(t)) { return op.brick_unpack(t); }
template <class Op> __device__ __forceinline__ typename Op::Field brick_unpack_impl(const Op&, const typename Op::Field& t, long) { return t; }
template <class Op> __device__ __forceinline__ typename Op::Field brick_unpack(const Op& op, const brick_field_t<Op>& t) { return brick_unpack_impl(op, t, 0); }

// the row walk of one own particle from the stage (tolerance arithmetic; lane-per-particle)
template <bool WANT_BOUNDARY, class Op, class Body>
__device__ __forceinline__ void sweep_brick(const Op& op, const SweepCtx& c, const float4* lp, const typename Op::Field* lfRaw, const int i,
                                            const bool valid, const float3 pi, Body& body)
{
    if (!valid) return;
    const int cnt = c.nbrCount[i];
    if (cnt > 2 * c.cap) {                 // row overflow: this particle walks the cells over global memory (the host enlarges the rows)
        walk_cells<WANT_BOUNDARY>(c, pi, [&](int j, bool isB, float3 d, float r2, float mj) {
            body.pair_tol(op.stage(isB, j + (isB ? c.bOff : 0)), isB, d, r2, mj);
        });
        return;
    }
    // (the stage holds brick_field_t<Op> records; the Op's operator() hands the pointer through typed as its Field)
    const brick_field_t<Op>* lf = reinterpret_cast<const brick_field_t<Op>*>(lfRaw);
    const uint4* row = reinterpret_cast<const uint4*>(c.nbr + row_base_offset(i, c.cap));
    const int chunks = (cnt + 7) >> 3;
    uint4 nxt = chunks > 0 ? row[0] : make_uint4(0u, 0u, 0u, 0u);
    for (int k = 0; k < chunks; ++k) {
        const uint4 cur = nxt;
        if (k + 1 < chunks) nxt = row[(size_t)(k + 1) * 64u];
        const unsigned int w[4] = {cur.x, cur.y, cur.z, cur.w};
#pragma unroll
        for (int h = 0; h < 2; ++h) {             // 4 entries in flight (8 measured no faster and spills more)
            float4 pj[4]; brick_field_t<Op> fr[4]; bool isB[4], use[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const unsigned int word = w[h * 2 + (u >> 1)];
                const unsigned int e = (u & 1) ? (word >> 16) : (word & 0xffffu);
                use[u] = k * 8 + h * 4 + u < cnt;
                const unsigned int slot = use[u] ? (e & 0x7fffu) : 0u;          // (slots past the row's end hold stale bits)
                isB[u] = (e & kBrickBoundaryBit) != 0u;
                pj[u] = lp[slot]; fr[u] = lf[slot];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float3 d = sub3(pi, v3(pj[u].x, pj[u].y, pj[u].z));
                const float r2 = dot3(d, d);
                const bool on = use[u] && (WANT_BOUNDARY || !isB[u]);
                body.pair_tol(brick_unpack(op, fr[u]), isB[u], d, r2, on ? pj[u].w : 0.0f);       // a dropped entry enters with mass 0
            }
        }
    }
}

// 16-bit row entries are collected 8 at a time and leave as 16-byte chunks
__device__ __forceinline__ void brick_put_entry(unsigned int* row, int cnt, unsigned int e, uint4& pend, int capEntries)
{
    const int w = cnt & 7;
    const unsigned int sh = (w & 1) ? 16u : 0u, keep = (w & 1) ? 0x0000ffffu : 0xffff0000u;
    unsigned int* word = (w >> 1) == 0 ? &pend.x : ((w >> 1) == 1 ? &pend.y : ((w >> 1) == 2 ? &pend.z : &pend.w));
    *word = (*word & keep) | (e << sh);
    if (w == 7 && cnt < capEntries) *reinterpret_cast<uint4*>(row + (size_t)(cnt >> 3) * 256u) = pend;
}
// rows of the current slice's own particles: candidates are read from the stage (fluid of the 9 columns, then boundary: the order
// of a particle's sum is free under the tolerance contract).  Every thread of the block calls it (after brick_stage).
__device__ __forceinline__ void brick_build_rows(const SweepCtx& c, const BrickTables& T, const float4* lp, int x0, int y0,
                                                 unsigned int* nbr, int* nbrCount)
{
    const int capEntries = 2 * c.cap;
    for (int p = threadIdx.x; p < T.own; p += kBrickThreads) {
        const int i = brick_own_index(T, p);
        const float4 self = c.posm[i];
        const float3 pi = v3(self.x, self.y, self.z);
        const int3 c0 = cell_of(pi, c.g);
        unsigned int* row = nbr + row_base_offset(i, c.cap);
        const int zlo = max(c0.z - 1, 0), zhi = min(c0.z + 1, c.g.gz - 1);
        int cnt = 0; uint4 pend = make_uint4(0u, 0u, 0u, 0u);
#pragma unroll 1
        for (int kind = 0; kind < 2; ++kind) {
            const int* cs = kind == 0 ? c.csF : c.csB;
            for (int dx = -1; dx <= 1; ++dx) {
                const int X = c0.x + dx;
                if (X < 0 || X >= c.g.gx) continue;
                for (int dy = -1; dy <= 1; ++dy) {
                    const int Y = c0.y + dy;
                    if (Y < 0 || Y >= c.g.gy) continue;
                    const int base = (X * c.g.gy + Y) * c.g.gz;
                    const int e = cs[base + zhi + 1];
                    int j = cs[base + zlo];
                    const int r = kind * kBrickRuns + (X - (x0 - 1)) * 6 + (Y - (y0 - 1));
                    const int shift = T.runBase[r] - T.runStart[r];
                    for (; j + 4 <= e; j += 4) {                      // 4 candidates in flight: one LDS round trip per 4 tests
                        float4 pj[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) pj[u] = lp[j + u + shift];
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const float3 d = sub3(pi, v3(pj[u].x, pj[u].y, pj[u].z));
                            const float r2 = dot3(d, d);
                            if (r2 > c.buildCut || (kind == 0 && j + u == i)) continue;
                            brick_put_entry(row, cnt, (unsigned int)(j + u + shift) | (kind ? kBrickBoundaryBit : 0u), pend, capEntries);
                            ++cnt;
                        }
                    }
                    for (; j < e; ++j) {
                        const float4 pj = lp[j + shift];
                        const float3 d = sub3(pi, v3(pj.x, pj.y, pj.z));
                        const float r2 = dot3(d, d);
                        if (r2 > c.buildCut || (kind == 0 && j == i)) continue;
                        brick_put_entry(row, cnt, (unsigned int)(j + shift) | (kind ? kBrickBoundaryBit : 0u), pend, capEntries);
                        ++cnt;
                    }
                }
            }
        }
        nbrCount[i] = cnt;
        if (cnt > capEntries && c.overflowMax) atomicMax(c.overflowMax, (cnt + 1) / 2);
        if ((cnt & 7) != 0 && cnt < capEntries) *reinterpret_cast<uint4*>(row + (size_t)(cnt >> 3) * 256u) = pend;
    }
}

__device__ __forceinline__ float3 ld3(const float3* __restrict__ p, int i) { return p[i]; }
__device__ __forceinline__ float3 xyz(const float4 v) { return v3(v.x, v.y, v.z); }

// exact, order-independent |error| accumulation: 2^-32 fixed point
__device__ __forceinline__ long long error_fixed(float e)
{
    float s = fabsf(e) * 4294967296.0f;
    if (!(s < 4.0e18f)) s = 4.0e18f;
    return (long long)s;
}

}  // namespace sphx
