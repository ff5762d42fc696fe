// runtime.hip — engine stream, error reporting, kernel constants, packed views, element-wise
// helpers and the optional per-kernel timer.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <mutex>

#include "engine.hpp"

namespace sphx {

// ------------------------------------------------------------------------------ stream / errors
static hipStream_t g_stream = nullptr;
static std::once_flag g_stream_once;
static thread_local std::string g_last_error;
static std::string g_last_error_global;

static hipStream_t g_external_stream = nullptr;
static bool g_use_external = false;
void use_external_stream(hipStream_t s) { g_external_stream = s; g_use_external = true; }

static hipStream_t g_scoped_stream = nullptr;
ScopedStream::ScopedStream(hipStream_t s) : previous(g_scoped_stream) { g_scoped_stream = s; }
ScopedStream::~ScopedStream() { g_scoped_stream = previous; }

hipStream_t stream()
{
    if (g_scoped_stream) return g_scoped_stream;
    if (g_use_external) return g_external_stream;
    std::call_once(g_stream_once, [] {
        if (hipStreamCreateWithFlags(&g_stream, hipStreamNonBlocking) != hipSuccess) g_stream = nullptr;
    });
    return g_stream;
}

void report_hip_error(hipError_t e, const char* file, int line)
{
    char buf[512];
    snprintf(buf, sizeof(buf), "HIP error at %s:%d: %s (%d)", file, line, hipGetErrorString(e), (int)e);
    g_last_error_global = buf;
    fprintf(stderr, "%s\n", buf);
}

const std::string& last_error_text() { return g_last_error_global; }
void set_error_text(const std::string& s) { g_last_error_global = s; }

// ------------------------------------------------------------------------------ tuning block
sphx_tuning default_tuning()
{
    sphx_tuning t;
    std::memset(&t, 0, sizeof(t));
    t.struct_size = (int)sizeof(sphx_tuning);
    t.quad_mask = t.duo_mask = t.quad_mask_tol = t.tol_strict_rate = -1;
    t.range_order = -1;
    t.dfsph_window = -1;
    t.pbd_skin = -1.0f;
    t.persist_controller = -1;
    t.slab_edge_stream = -1;
    return t;
}
static sphx_tuning g_tuning = default_tuning();
const sphx_tuning& tuning() { return g_tuning; }
void install_tuning(const sphx_tuning& t) { g_tuning = t; }
int g_lastRateVariant = kRateNone;

// ------------------------------------------------------------------------------ kernel constants
namespace {
float f_cube(float x) { return x * x * x; }
float bits_to_float(unsigned int b) { float f; std::memcpy(&f, &b, 4); return f; }

// largest non-negative float r2 with pred(r2) true, assuming pred is true at 0 and monotone
template <class Pred>
float largest_true(Pred pred)
{
    unsigned int lo = 0u, hi = 0x7f7fffffu;   // +0 .. FLT_MAX
    if (pred(bits_to_float(hi))) return bits_to_float(hi);
    while (hi - lo > 1u) {
        const unsigned int mid = lo + (hi - lo) / 2u;
        if (pred(bits_to_float(mid))) lo = mid; else hi = mid;
    }
    return bits_to_float(lo);
}
}  // namespace

KernelConsts make_kernel_consts(float R)
{
    KernelConsts k;
    k.R = R;
    k.wA = 0.25f / (kPi * R * R * R);
    k.viscDen = kPi * powf(R, 6);
    k.stK = kPi * f_cube(R) * f_cube(R) * f_cube(R);
    k.stC = 0.0156f * f_cube(R) * f_cube(R);
    // volatile keeps the host compiler from folding sqrt/div differently from run-time IEEE ops
    const float tW = largest_true([R](float r2) { volatile float r = sqrtf(r2); volatile float q = 2.0f * r / R; return !(q > 2.0f); });
    const float tR = largest_true([R](float r2) { volatile float r = sqrtf(r2); return r <= R; });
    k.tCut = tW > tR ? tW : tR;
    k.q2Free = (tW >= tR) ? 1 : 0;
    k.rcpR = 1.0f / R;
    k.fastQ = 0;
    k.fastDiv = 0;
    k.rcpViscDen = 1.0f / k.viscDen;
    k.fastVisc = 0;
    k.tol = 0;
    k.twoOverR = 2.0f / R;
    k.gradScale = 1.0f / (kPi * R * R * R * R * R);
    k.viscScale = 45.0f / k.viscDen;
    k.stScale = 136.0241f / k.stK;
    return k;
}

// ---- validation of the exact fast paths (sph_device.hpp) ------------------------------------------------
// every float x in {0} U [2^-47, lastBits] (q_of sends smaller positive x to the plain operator)
__global__ void k_check_q_division(KernelConsts k, unsigned int lastBits, unsigned int* mismatches)
{
    const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
    const unsigned int first = 0x28000000u;   // 2^-47
    unsigned int bad = 0;
    for (unsigned long long b = first + (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; b <= lastBits; b += stride) {
        const float x = __uint_as_float((unsigned int)b);
        bad += (__float_as_uint(div_by_radius<true>(x, k)) != __float_as_uint(x / k.R)) ? 1u : 0u;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) bad += (__float_as_uint(div_by_radius<true>(0.0f, k)) != __float_as_uint(0.0f / k.R)) ? 1u : 0u;
    if (bad) atomicAdd(mismatches, bad);
}
// every float x in {0} U [firstBits, lastBits]: the refined x * (1/den) against the plain quotient
__global__ void k_check_const_division(float den, float rcp, unsigned int firstBits, unsigned int lastBits, unsigned int* mismatches)
{
    const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
    unsigned int bad = 0;
    for (unsigned long long b = firstBits + (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; b <= lastBits; b += stride) {
        const float x = __uint_as_float((unsigned int)b);
        bad += (__float_as_uint(div_by_const_refined(x, den, rcp)) != __float_as_uint(x / den)) ? 1u : 0u;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) bad += (__float_as_uint(div_by_const_refined(0.0f, den, rcp)) != __float_as_uint(0.0f / den)) ? 1u : 0u;
    if (bad) atomicAdd(mismatches, bad);
}
__global__ void k_check_sqrt(unsigned int firstBits, unsigned int lastBits, unsigned int* mismatches)
{
    const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
    unsigned int bad = 0;
    for (unsigned long long b = firstBits + (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; b <= lastBits; b += stride) {
        const float x = __uint_as_float((unsigned int)b);
        if (x != 0.0f && x < 1.2621774483536189e-29f) continue;   // below 2^-96 the sweeps use sqrtf
        bad += (__float_as_uint(sqrt_sel<true>(x)) != __float_as_uint(sqrtf(x))) ? 1u : 0u;
    }
    if (bad) atomicAdd(mismatches, bad);
}
__device__ __forceinline__ unsigned int mix32(unsigned int v)
{
    v ^= v >> 16; v *= 0x7feb352du; v ^= v >> 15; v *= 0x846ca68bu; v ^= v >> 16;
    return v;
}
// pseudo-random numerators (zero, tiny, ordinary, both signs) over denominators spanning the
// accepted range; compares div3_exact with three IEEE divisions
__global__ void k_check_div3(KernelConsts k, float denLo, float denHi, unsigned long long samples, unsigned int* mismatches)
{
    const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
    unsigned int bad = 0;
    const float lgLo = log2f(denLo), lgHi = log2f(denHi);
    for (unsigned long long t = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; t < samples; t += stride) {
        const unsigned int h0 = mix32((unsigned int)t * 4u + 1u), h1 = mix32((unsigned int)t * 4u + 2u);
        const unsigned int h2 = mix32((unsigned int)t * 4u + 3u), h3 = mix32((unsigned int)t * 4u + 4u);
        // denominator: random mantissa, exponent uniform in [lgLo, lgHi]
        const float den = exp2f(lgLo + (lgHi - lgLo) * ((h0 >> 8) * (1.0f / 16777216.0f))) * (1.0f + (h1 & 0x7fffffu) * (1.0f / 8388608.0f) * 0.5f);
        auto numer = [](unsigned int h) {
            const unsigned int kind = h & 7u;
            if (kind == 0u) return 0.0f;
            const int e = (kind == 1u) ? (-140 + (int)((h >> 3) % 45u)) : (-40 + (int)((h >> 3) % 47u));   // tiny or ordinary
            const float m = 1.0f + ((h >> 9) & 0x7fffffu) * (1.0f / 8388608.0f);
            const float v = ldexpf(m, e);
            return (h & 0x80000000u) ? -v : v;
        };
        const float3 n = make_float3(numer(h1), numer(h2), numer(h3));
        // numerators below 2^-101 take the plain operators in the sweeps (pair_needs_plain_ops)
        const float3 a = pair_needs_plain_ops(n, 1.0f) ? div3s(n, den) : div3_sel<true>(n, den), b = div3s(n, den);
        bad += (__float_as_uint(a.x) != __float_as_uint(b.x)) + (__float_as_uint(a.y) != __float_as_uint(b.y)) +
               (__float_as_uint(a.z) != __float_as_uint(b.z));
    }
    if (bad) atomicAdd(mismatches, bad);
}

// enables the fast paths of `k` that hold for its radius; runs a few ms of device work and one sync
void validate_fast_math(KernelConsts& k)
{
    if (tuning().no_fastmath) return;
    const float R = k.R;
    // denominators seen by div3_exact: PI*(q+EPS)*R^5 for q in [0, 2], and stK*x for x in [EPS, R]
    const float r5 = R * R * R * R * R;
    const float dLo = fminf(kPi * kEps * r5, k.stK * kEps), dHi = fmaxf(kPi * (2.0f + kEps) * r5 * 1.01f, k.stK * R);
    k.fastDiv = (dLo >= ldexpf(1.0f, -90) && dHi <= ldexpf(1.0f, 16) && R < 64.0f) ? 1 : 0;
    unsigned int* d_bad = nullptr;
    if (hipMalloc((void**)&d_bad, sizeof(unsigned int)) != hipSuccess) return;
    HIP_CALL(hipMemsetAsync(d_bad, 0, sizeof(unsigned int), stream()));
    unsigned int lastBits; const float top = 2.2f * R; std::memcpy(&lastBits, &top, 4);
    { KernelConsts kq = k; kq.fastQ = 1; k_check_q_division<<<8192, 256, 0, stream()>>>(kq, lastBits, d_bad); }
    unsigned int bad = 1;
    HIP_CALL(hipMemcpyAsync(&bad, d_bad, sizeof(bad), hipMemcpyDeviceToHost, stream()));
    HIP_CALL(hipStreamSynchronize(stream()));
    k.fastQ = (bad == 0) ? 1 : 0;
    // the viscosity laplacian's division by PI R^6 (numerators 45 (R - r): 0 or >= 45 ulp(R))
    HIP_CALL(hipMemsetAsync(d_bad, 0, sizeof(unsigned int), stream()));
    unsigned int lastV; const float topV = 46.0f * R; std::memcpy(&lastV, &topV, 4);
    k_check_const_division<<<8192, 256, 0, stream()>>>(k.viscDen, k.rcpViscDen, 0x2b800000u /* 2^-40 */, lastV, d_bad);
    bad = 1;
    HIP_CALL(hipMemcpyAsync(&bad, d_bad, sizeof(bad), hipMemcpyDeviceToHost, stream()));
    HIP_CALL(hipStreamSynchronize(stream()));
    k.fastVisc = (bad == 0 && 45.0f * (R - nextafterf(R, 0.0f)) >= ldexpf(1.0f, -40)) ? 1 : 0;
    (void)hipFree(d_bad);
}

// sphx_fastmath_selftest: mismatch counts of {q division (exhaustive), sqrt (exhaustive over all
// non-negative finite floats), div3 (samples)} against the plain IEEE operators
void fastmath_selftest(float R, unsigned long long samples, unsigned int out[3], int flags[2])
{
    KernelConsts k = make_kernel_consts(R);
    validate_fast_math(k);
    flags[0] = k.fastQ; flags[1] = k.fastDiv;
    unsigned int* d_bad = nullptr;
    HIP_CALL(hipMalloc((void**)&d_bad, 3 * sizeof(unsigned int)));
    HIP_CALL(hipMemsetAsync(d_bad, 0, 3 * sizeof(unsigned int), stream()));
    unsigned int lastBits; const float top = 2.2f * R; std::memcpy(&lastBits, &top, 4);
    { KernelConsts kq = k; kq.fastQ = 1; k_check_q_division<<<8192, 256, 0, stream()>>>(kq, lastBits, d_bad); }
    k_check_sqrt<<<16384, 256, 0, stream()>>>(0u, 0x7f7fffffu, d_bad + 1);
    const float r5 = R * R * R * R * R;
    KernelConsts kf = k; kf.fastDiv = 1;
    k_check_div3<<<8192, 256, 0, stream()>>>(kf, fminf(kPi * kEps * r5, k.stK * kEps), fmaxf(kPi * 2.1f * r5, k.stK * R), samples, d_bad + 2);
    // ... and over the denominators of the surface sweep's division by max(EPS, |colorGrad|): [2^-20, 2^16]
    k_check_div3<<<8192, 256, 0, stream()>>>(kf, ldexpf(1.0f, -20), ldexpf(1.0f, 16), samples / 4 + 1, d_bad + 2);
    HIP_CALL(hipMemcpyAsync(out, d_bad, 3 * sizeof(unsigned int), hipMemcpyDeviceToHost, stream()));
    HIP_CALL(hipStreamSynchronize(stream()));
    (void)hipFree(d_bad);
}

GridDesc make_grid_desc(int3 cs, float cellLength, int cellOffsetX)
{
    GridDesc g;
    g.gx = cs.x; g.gy = cs.y; g.gz = cs.z; g.C = cs.x * cs.y * cs.z;
    g.cellLength = cellLength;
    g.xOff = cellOffsetX;
    return g;
}

// ------------------------------------------------------------------------------ element-wise
__global__ void k_gather_float3(float3* __restrict__ dst, const float3* __restrict__ src, const int* __restrict__ perm, int n)
{
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q < n) dst[q] = src[perm[q]];
}
__global__ void k_gather_float(float* __restrict__ dst, const float* __restrict__ src, const int* __restrict__ perm, int n)
{
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q < n) dst[q] = src[perm[q]];
}
__global__ void k_gather_int(int* __restrict__ dst, const int* __restrict__ src, const int* __restrict__ perm, int n)
{
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q < n) dst[q] = src[perm[q]];
}
__global__ void k_fill_float(float* __restrict__ dst, float v, int n)
{
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q < n) dst[q] = v;
}
__global__ void k_iota(int* __restrict__ dst, int n)
{
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q < n) dst[q] = q;
}
__global__ void k_pack4(float4* __restrict__ dst, const float3* __restrict__ pos, const float* __restrict__ w, int n)
{
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q < n) { const float3 p = pos[q]; dst[q] = make_float4(p.x, p.y, p.z, w[q]); }
}
// (massUniform is preset to 1 by a memset; any particle whose mass differs from particle 0's clears it)
__global__ void k_pack_fluid(float4* __restrict__ posm, float4* __restrict__ vel4, float4* __restrict__ posf,
                             int* __restrict__ massUniform, const float3* __restrict__ pos, const float* __restrict__ mass,
                             const float3* __restrict__ vel, int n)
{
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= n) return;
    const float3 p = pos[q], v = vel[q];
    const float m = mass[q];
    posm[q] = make_float4(p.x, p.y, p.z, m);
    posf[q] = make_float4(p.x, p.y, p.z, 0.0f);
    vel4[q] = make_float4(v.x, v.y, v.z, 0.0f);
    if (m != mass[0]) *massUniform = 0;
}

__global__ void k_gather_float_if(float* __restrict__ dst, const float* __restrict__ src, const int* __restrict__ perm, int n, const int* __restrict__ flag)
{
    if (*flag == 0) return;
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q < n) dst[q] = src[perm[q]];
}
__global__ void k_copy_float_if(float* __restrict__ dst, const float* __restrict__ src, int n, const int* __restrict__ flag)
{
    if (*flag == 0) return;
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q < n) dst[q] = src[q];
}
void ew_gather_float_if(float* dst, const float* src, const int* perm, int n, const int* flag)
{
    if (n > 0) k_gather_float_if<<<blocks_for(n), 256, 0, stream()>>>(dst, src, perm, n, flag);
}
void ew_copy_float_if(float* dst, const float* src, int n, const int* flag)
{
    if (n > 0) k_copy_float_if<<<blocks_for(n), 256, 0, stream()>>>(dst, src, n, flag);
}
void ew_gather_float3(float3* dst, const float3* src, const int* perm, int n)
{
    if (n <= 0) return;
    ScopedKernel t("gather_float3");
    k_gather_float3<<<blocks_for(n), 256, 0, stream()>>>(dst, src, perm, n);
}
void ew_gather_float(float* dst, const float* src, const int* perm, int n)
{
    if (n <= 0) return;
    ScopedKernel t("gather_float");
    k_gather_float<<<blocks_for(n), 256, 0, stream()>>>(dst, src, perm, n);
}
void ew_gather_int(int* dst, const int* src, const int* perm, int n)
{
    if (n <= 0) return;
    ScopedKernel t("gather_int");
    k_gather_int<<<blocks_for(n), 256, 0, stream()>>>(dst, src, perm, n);
}
void ew_copy(void* dst, const void* src, size_t bytes)
{
    if (bytes) HIP_CALL(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, stream()));
}
void ew_fill_float(float* dst, float value, int n)
{
    if (n <= 0) return;
    k_fill_float<<<blocks_for(n), 256, 0, stream()>>>(dst, value, n);
}
void ew_iota(int* dst, int n)
{
    if (n <= 0) return;
    k_iota<<<blocks_for(n), 256, 0, stream()>>>(dst, n);
}

// ------------------------------------------------------------------------------ SweepCache
__global__ void k_pack_kick_rt(float4* __restrict__ posm, float4* __restrict__ vel4, float4* __restrict__ posf,
                               int* __restrict__ massUniform, const float3* __restrict__ pos, const float* __restrict__ mass,
                               float3* __restrict__ vel, float3 dv, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float3 p = pos[i];
    const float m = mass[i];
    posm[i] = make_float4(p.x, p.y, p.z, m);
    posf[i] = make_float4(p.x, p.y, p.z, 0.0f);
    if (m != mass[0]) *massUniform = 0;
    const float3 v = add3(vel[i], dv);
    vel[i] = v;
    vel4[i] = make_float4(v.x, v.y, v.z, 0.0f);
}
// ---- tile schedule: counting sort of the tiles by (y-chunk, x) --------------------------------------
// onlyIf (persistent rows): the particle order changes only when the rows are rebuilt (device flag); otherwise the schedule in place stays
__global__ void k_tile_bucket_count(const float4* __restrict__ posm, int n, GridDesc g, int chunkCells, int chunks,
                                    int* __restrict__ key, int* __restrict__ hist, int numTiles, const int* __restrict__ onlyIf)
{
    if (onlyIf && *onlyIf == 0) return;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= numTiles) return;
    const int3 c = cell_of(xyz4(posm[min(t * kTile, n - 1)]), g);
    const int X = min(max(c.x, 0), g.gx - 1);
    const int Y = min(max(c.y, 0), g.gy - 1);
    const int k = min(Y / chunkCells, chunks - 1) * g.gx + X;
    key[t] = k;
    atomicAdd(&hist[k], 1);
}
// single block: exclusive scan of `m` bucket counts in place (m is a few thousand at most)
__global__ void __launch_bounds__(256) k_tile_bucket_scan(int* __restrict__ hist, int m, const int* __restrict__ onlyIf)
{
    if (onlyIf && *onlyIf == 0) return;
    __shared__ int carry;
    __shared__ int part[256];
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int base = 0; base < m; base += 256) {
        const int idx = base + threadIdx.x;
        const int v = idx < m ? hist[idx] : 0;
        part[threadIdx.x] = v;
        __syncthreads();
        for (int off = 1; off < 256; off <<= 1) {
            const int add = threadIdx.x >= off ? part[threadIdx.x - off] : 0;
            __syncthreads();
            part[threadIdx.x] += add;
            __syncthreads();
        }
        const int c0 = carry;
        if (idx < m) hist[idx] = c0 + part[threadIdx.x] - v;
        __syncthreads();
        if (threadIdx.x == 255) carry = c0 + part[255];
        __syncthreads();
    }
}
__global__ void k_tile_bucket_place(const int* __restrict__ key, int* __restrict__ cursor, int* __restrict__ order, int numTiles,
                                    const int* __restrict__ onlyIf)
{
    if (onlyIf && *onlyIf == 0) return;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < numTiles) order[atomicAdd(&cursor[key[t]], 1)] = t;
}

void SweepCache::ensureTileOrder()
{
    const int numTiles = (n + kTile - 1) / kTile;
    if (orderValid && orderTiles != numTiles) orderValid = false;      // a schedule must be a permutation of the current tiles
    if (orderValid || (flags & kFlagLinearTiles) || n <= 0) return;
    // y-chunks: about 1.5 MB of (position + one 16-byte field) for three x-layers of a chunk
    const double layerBytes = (double)g.gy * g.gz * 7.0 * 32.0;
    // only worth its ~40 us when three x-layers clearly exceed an XCD's 4 MB L2 (measured: +5 % at
    // 10 M particles, -4 % at 1 M)
    if (3.0 * layerBytes < 12.0 * 1024 * 1024 && !tuning().force_tile_order) { orderValid = false; flags |= 0; return; }
    int chunks = (int)std::ceil(3.0 * layerBytes / (1.5 * 1024 * 1024));
    chunks = std::max(8, std::min(64, ((chunks + 7) / 8) * 8));
    const int chunkCells = std::max(1, (g.gy + chunks - 1) / chunks);
    const int buckets = chunks * g.gx;
    if (!tileBuckets || (int)tileBuckets->length() < buckets + 1) tileBuckets.reset(new DArray<int>((unsigned)buckets + 1u));
    ScopedKernel t("tile_schedule");
    // persistent rows: a schedule of THIS tile count that exists already is only replaced in steps that rebuild the rows (a replayed
    // step graph holds these launches in every step: 0.12 ms at 10.3 M particles, needed in every second step or so)
    const int* onlyIf = (persistRows && orderBuiltTiles == numTiles) ? persistFlags.addr(0) : nullptr;
    HIP_CALL(hipMemsetAsync(tileBuckets->addr(), 0, sizeof(int) * (buckets + 1), stream()));
    k_tile_bucket_count<<<blocks_for(numTiles), 256, 0, stream()>>>(fluid4(), n, g, chunkCells, chunks, tileKey.addr(), tileBuckets->addr(), numTiles, onlyIf);
    k_tile_bucket_scan<<<1, 256, 0, stream()>>>(tileBuckets->addr(), buckets, onlyIf);
    k_tile_bucket_place<<<blocks_for(numTiles), 256, 0, stream()>>>(tileKey.addr(), tileBuckets->addr(), tileOrder.addr(), numTiles, onlyIf);
    orderValid = true; orderAge = 0; orderTiles = numTiles; orderBuiltTiles = numTiles;
}

// Row construction, one wave per 64-particle tile.  STREAM: each wave first decides whether its tile
// can use LDS-streamed entries (fmt 2), records that in tileFmt, and stages candidates through LDS.
// Skin rows (PBD): `refresh` non-null makes the build CONDITIONAL on the device flag refresh[0] ("a particle has moved
// too far since the rows were built"): not raised -> every wave leaves at once; raised -> the rows are rebuilt for
// the current positions, which become the new reference positions.  refresh[1] is the flag the coming position
// updates will raise; it is cleared here.  No host round trip, so the schedule stays graph-replayable.
template <bool STREAM>
__global__ void __launch_bounds__(kWideBlock) k_build_list(SweepCtx c, unsigned int* nbr, int* nbrCount, int* tileFmt,
                                                           float4* posBuild, int* rowCell, const int* flagNow, int* flagNext, int* rebuilds)
{
    __shared__ float4 pos[STREAM ? kWideBlock / kTile : 1][STREAM ? kGroupSlots : 1];
    __shared__ unsigned int stage[(STREAM || SPHX_BUILD_REGSTAGE) ? 1 : kWideBlock / kTile][(STREAM || SPHX_BUILD_REGSTAGE) ? 1 : kRowStage * kTile];
    if (flagNext && blockIdx.x == 0 && threadIdx.x == 0) {
        *flagNext = 0;
        if (rebuilds && *flagNow != 0) *rebuilds += 1;      // diagnostics: conditional rebuilds since creation
    }
    if (flagNow && *flagNow == 0) return;      // launch-uniform
    const int tile = wave_tile(c);
    if (tile < 0) return;                  // whole wave past the end
    const int i = tile * kTile + (int)(threadIdx.x & 63);
    const int i0 = tile * kTile;
    bool streamed = false;
    if (STREAM) {
        streamed = wave_ranges(c, i0).ok;
        if ((threadIdx.x & 63) == 0) tileFmt[i >> 6] = streamed ? 2 : 0;
    }
    build_neighbor_rows(c, STREAM ? pos[threadIdx.x >> 6] : nullptr, streamed, nbr, nbrCount, i, i < c.n && in_range(c, i),
                        (STREAM || SPHX_BUILD_REGSTAGE) ? nullptr : stage[threadIdx.x >> 6]);
    if (posBuild && i < c.n) {
        const float4 p = c.posm[i];
        posBuild[i] = p;
        const int3 c0 = cell_of(xyz4(p), c.g);
        rowCell[i] = cell_id(c0.x, c0.y, c0.z, c.g);
    }
}



#ifndef SPHX_GROUP_WAVES
#define SPHX_GROUP_WAVES 8
#endif
constexpr int kGroupBuildMax = 81920;      // measured crossover of a WCSPH step: 70 k particles 0.138 against 0.149 ms, 96 k particles 0.155 against 0.156
// The builder of small scenes: 16 lanes per particle (build_neighbor_rows_group), the conditional-rebuild words as above.
__global__ void __launch_bounds__(kWideBlock, SPHX_GROUP_WAVES) k_build_list_group(SweepCtx c, unsigned int* nbr, int* nbrCount, float4* posBuild, int* rowCell,
                                                                 const int* flagNow, int* flagNext, int* rebuilds)
{
    if (flagNext && blockIdx.x == 0 && threadIdx.x == 0) {
        *flagNext = 0;
        if (rebuilds && *flagNow != 0) *rebuilds += 1;
    }
    if (flagNow && *flagNow == 0) return;      // launch-uniform
    __shared__ unsigned int stash[kGroupStash * kWideBlock];
    const int i = (int)((blockIdx.x * kWideBlock + threadIdx.x) / kBuildGroup);
    const bool valid = i < c.n && in_range(c, i);
    const int cell = build_neighbor_rows_group(c, c.posm, nbr, nbrCount, i, valid, stash);
    if (posBuild && valid && (threadIdx.x & (kBuildGroup - 1)) == 0) {
        posBuild[i] = c.posm[i];
        rowCell[i] = cell;
    }
}
// Skin rows between two Jacobi iterations (r06): the rows of the particles the last position update found in another cell than
// the one their row was built around (SkinWatch), rebuilt around the new cell from the positions of the last build of
// everything -- unless that update also raised `staleNow`: then the launch behind this one rebuilds everything anyway.
__global__ void __launch_bounds__(kWideBlock, SPHX_GROUP_WAVES) k_build_list_changed(SweepCtx c, unsigned int* nbr, int* nbrCount, const float4* posBuild,
                                                                   int* rowCell, const int* staleNow, const int* countNow, const int* list,
                                                                   int listCap, int* countNext, int* partials)
{
    __shared__ unsigned int stash[kGroupStash * kWideBlock];
    const int m = min(*countNow, listCap);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        *countNext = 0;
        if (partials && m > 0 && *staleNow == 0) *partials += 1;      // diagnostics
    }
    if (*staleNow != 0 || m <= 0) return;      // launch-uniform
    const int t = (int)((blockIdx.x * kWideBlock + threadIdx.x) / kBuildGroup);
    if (((int)(blockIdx.x * kWideBlock) / kBuildGroup) >= m) return;      // whole block past the list
    const bool valid = t < m;
    const int i = valid ? list[t] : 0;
    const int cell = build_neighbor_rows_group(c, posBuild, nbr, nbrCount, i, valid, stash);
    if (valid && (threadIdx.x & (kBuildGroup - 1)) == 0) rowCell[i] = cell;
}

// The non-empty bricks of this step and their whole-brick tables (one block per brick of the grid; once per step).
__global__ void __launch_bounds__(256) k_brick_list(SweepCtx c, BrickTables* tab, int* count, int capacity, int* fault)
{
    __shared__ __attribute__((aligned(16))) BrickTables T;
    __shared__ int slot;
    const BrickGeom G = brick_geom(c, (int)blockIdx.x);
    if (!G.any) return;
    brick_slice_tables(c, T, G.x0, G.y0, G.z0, G.z0 + kBrickEdge);
    if (T.own == 0) return;
    if (threadIdx.x == 0) { T.x0 = G.x0; T.y0 = G.y0; T.z0 = G.z0; slot = atomicAdd(count, 1); }
    __syncthreads();
    if (slot >= capacity) { if (threadIdx.x == 0) { *fault = 1; atomicSub(count, 1); } return; }      // (the host leaves brick mode)
    if (threadIdx.x < sizeof(BrickTables) / 16) reinterpret_cast<uint4*>(tab + slot)[threadIdx.x] = reinterpret_cast<const uint4*>(&T)[threadIdx.x];
}
// Row builder of the compact-brick path: candidates read from the staged positions, rows of 16-bit slots.
__global__ void __launch_bounds__(kBrickThreads, 4) k_build_brick(SweepCtx c, unsigned int* nbr, int* nbrCount)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char brickLds[];
    __shared__ __attribute__((aligned(16))) BrickTables T;
    float4* lp = reinterpret_cast<float4*>(brickLds);
    brick_for_each_slice(c, T, [&](BrickTables& tab) {
        brick_stage(c, tab, lp, (float*)nullptr, [](bool, int) { return 0.0f; });
        brick_build_rows(c, tab, lp, tab.x0, tab.y0, nbr, nbrCount);
        __syncthreads();
    });
}

SweepCache::SweepCache(int num)
    : n(num), posm(4u * (unsigned)num), pterm((unsigned)num), aux3((unsigned)num), vel4(4u * (unsigned)num),
      cg4(4u * (unsigned)num), posf(4u * (unsigned)num), massUniform(1u), nbrCount((unsigned)num),
      tileFmt((unsigned)(num / kTile + 2)), tileOrder((unsigned)(num / kTile + 2)), tileKey((unsigned)(num / kTile + 2)), capN(num),
      rowOverflow(4u), staleFlag(6u), persistFlags(4u)
{
    const sphx_tuning& T = tuning();
    capAuto = true; cap = 48;
    if (T.row_capacity >= 8 && T.row_capacity <= 1024) { cap = (T.row_capacity + kRowChunk - 1) / kRowChunk * kRowChunk; capAuto = false; }   // rows are stored in chunks of 4
    flags = T.engine_flags;
    if (T.range_order >= 0) rangeOrder = T.range_order != 0;
    if (T.range_order_min > 0) rangeOrderMin = T.range_order_min;
    if (T.quad_mask >= 0) quadMask = quadMaskSmall = T.quad_mask;  // experiments: which sweeps run quad-per-particle (at every size)
    if (T.duo_mask >= 0) { duoMask = T.duo_mask; duoMaskLarge = 0; quadMaskSmall = quadMask; }   // ... and which with two lanes per particle
    if (T.quad_mask_tol >= 0) quadMaskTol = quadMaskTolSmall = T.quad_mask_tol;       // ... quad walks under the tolerance arithmetic
    brickWanted = T.brick != 0;                                    // compact-brick LDS stage under the tolerance arithmetic
    if (brickWanted) {      //  the stage needs ~74 KB of dynamic LDS per block: parts with 64 KB (gfx942) cannot run it
        int dev = 0, ldsMax = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&ldsMax, hipDeviceAttributeMaxSharedMemoryPerBlock, dev) != hipSuccess ||
            (size_t)ldsMax < (size_t)(kBrickSlots + 1) * 2 * sizeof(float4) + 4096) {
            (void)hipGetLastError();
            fprintf(stderr, "sphx: sphx_tuning.brick ignored: this device offers %d bytes of LDS per block, the compact-brick stage needs %zu\n", ldsMax,
                    (size_t)(kBrickSlots + 1) * 2 * sizeof(float4) + 4096);
            brickWanted = false; brickFailed = true;
        }
    }
    if (T.brick_min > 0) brickMin = T.brick_min;
    if (T.tol_strict_rate >= 0) strictRateInTol = T.tol_strict_rate != 0;      // A/B measurements
}

void SweepCache::setup(int3 cellSize, float cellLength, float radius)
{
    if (radius != radiusKey) { k = make_kernel_consts(radius); validate_fast_math(k); radiusKey = radius; }
    g.xOff = cellOffsetX;
    if (cellLength != cellKey || cellSize.x != cellsKey.x || cellSize.y != cellsKey.y || cellSize.z != cellsKey.z) {
        g = make_grid_desc(cellSize, cellLength, cellOffsetX);
        cellKey = cellLength; cellsKey = cellSize;
    }
    if (persistWanted && !skinRows) {
        // the skin may not exceed the slack of the cell length: a pair within R + skin must lie in adjacent cells for the 27-cell
        // candidate walk of the builder to see it (the reference scene: cellLength = 1.01 R)
        const float slack = cellLength - radius;
        const float s = slack > 0.0f ? std::min(0.05f * radius, 0.95f * slack) : 0.0f;
        const bool on = tolerance && !isSlab && s > 0.0f && !(flags & (kFlagNoList | kFlagTiles | kFlagUnfused));
        if (on != persistRows || (on && s != skin)) { persistRows = on; skin = on ? s : 0.0f; listValid = false; ++generation; requestRebuild(); }
    }
}

void SweepCache::requestRebuild()
{
    const int one = 1;
    HIP_CALL(hipMemcpyAsync(persistFlags.addr(1), &one, sizeof(int), hipMemcpyHostToDevice, stream()));
    HIP_CALL(hipStreamSynchronize(stream()));      // (`one` lives on this stack frame)
}

void SweepCache::packFluid(const SPHParticles& fluids)
{
    if (fluidValid) return;
    n = (int)fluids.size();
    ScopedKernel t("pack_fluid");
    if (n > 0) {
        HIP_CALL(hipMemsetAsync(massUniform.addr(), 1, 1, stream()));   // low byte 1 -> flag value 1
        k_pack_fluid<<<blocks_for(n), 256, 0, stream()>>>(fluid4w(), vel4w(), posfw(), massUniform.addr(), fluids.getPosPtr(),
                                                          fluids.getMassPtr(), fluids.getVelPtr(), n);
    }
    fluidValid = true;
    listValid = false;
}

void SweepCache::packFluidKick(const SPHParticles& fluids, float3 dv)
{
    n = (int)fluids.size();
    ScopedKernel t("pack_kick");
    if (n > 0) {
        HIP_CALL(hipMemsetAsync(massUniform.addr(), 1, 1, stream()));
        k_pack_kick_rt<<<blocks_for(n), 256, 0, stream()>>>(fluid4w(), vel4w(), posfw(), massUniform.addr(), fluids.getPosPtr(),
                                                            fluids.getMassPtr(), fluids.getVelPtr(), dv, n);
    }
    fluidValid = true;
    listValid = false;
}

void SweepCache::reserveBoundary(int count)
{
    if (count <= nbCap) return;
    const unsigned int len = 4u * (unsigned)(capN + count);
    auto grow = [&](DArray<float>& a) {
        DArray<float> t(len);                                   // zero-filled
        ew_copy(t.addr(), a.addr(), sizeof(float) * 4u * (size_t)capN);
        a.swap(t);
    };
    grow(posm); grow(posf); grow(vel4); grow(cg4);
    posmAlt.reset(); posfAlt.reset();
    nbCap = count;
    boundaryValid = false;
    listValid = false;
    ++generation;      // the four arrays moved: a captured graph holds stale pointers
}

void SweepCache::ensureAltPositions()
{
    if (posmAlt && posmAlt->length() == posm.length() && posfAlt && posfAlt->length() == posf.length()) return;
    posmAlt.reset(new DArray<float>(posm.length()));
    posfAlt.reset(new DArray<float>(posf.length()));
    ew_copy(posmAlt->addr(), posm.addr(), sizeof(float) * (size_t)posm.length());      // (the boundary tail is what matters)
    ew_copy(posfAlt->addr(), posf.addr(), sizeof(float) * (size_t)posf.length());
    ++generation;
}

void SweepCache::packBoundary(const SPHParticles& boundaries)
{
    const int count = (int)boundaries.size();
    if (boundaryValid && boundaryKey == (const void*)boundaries.getPosPtr() && nb == count) return;
    reserveBoundary(count);
    nb = count;
    // PBD's second halves of posm / posf (ensureAltPositions) carry the boundary tail too.  They are repacked in place, not dropped:
    // this runs inside a stream capture when a boundary invalidation made the step re-capture, and a capture may contain neither
    // hipFree nor hipMalloc.  Halves of another length (reserveBoundary grew the arrays) are already gone.
    const bool alt = posmAlt && posfAlt && posmAlt->length() == posm.length() && posfAlt->length() == posf.length();
    if (!alt) { posmAlt.reset(); posfAlt.reset(); }
    ScopedKernel t("pack_boundary");
    if (count > 0) {
        // (x, y, z, mass) into the boundary tail of both position arrays (plain and one-gather view)
        k_pack4<<<blocks_for(count), 256, 0, stream()>>>(fluid4w() + capN, boundaries.getPosPtr(), boundaries.getMassPtr(), count);
        k_pack4<<<blocks_for(count), 256, 0, stream()>>>(posfw() + capN, boundaries.getPosPtr(), boundaries.getMassPtr(), count);
        if (alt) {
            k_pack4<<<blocks_for(count), 256, 0, stream()>>>(reinterpret_cast<float4*>(posmAlt->addr()) + capN, boundaries.getPosPtr(), boundaries.getMassPtr(), count);
            k_pack4<<<blocks_for(count), 256, 0, stream()>>>(reinterpret_cast<float4*>(posfAlt->addr()) + capN, boundaries.getPosPtr(), boundaries.getMassPtr(), count);
        }
    }
    boundaryKey = (const void*)boundaries.getPosPtr();
    boundaryValid = true;
    listValid = false;
}

SweepCtx SweepCache::ctx(const DArray<int>& csF, const DArray<int>& csB) const
{
    SweepCtx c;
    c.g = g; c.k = k;
    c.k.tol = tolerance ? 1 : 0;
    c.csF = csF.addr(); c.posm = fluid4();
    c.csB = csB.addr(); c.bposm = boundary4(); c.bOff = capN;
    const bool use = listValid && nbr && !(flags & kFlagNoList);
    if (use) { c.csF = listCsF; c.csB = listCsB; }   // rows (and tile tables) are tied to these tables
    c.nbr = use ? nbr->rows : nullptr;
    c.nbrCount = nbrCount.addr();
    c.cap = cap;
    c.tileFmt = (use && allowTiles && (flags & kFlagTiles)) ? tileFmt.addr() : nullptr;
    // (strict: the surface sweeps add TWO terms per entry to one accumulator, (a + t1) + t2, which the ordered one-term-per-lane
    // accumulation of the quad walk cannot reproduce: they stay lane-per-particle whatever the mask says)
    // (small scenes: every sweep that has the variant, see SweepCache::smallBelow)
    const bool small = n < smallBelow;
    c.quad = (use && !c.tileFmt && !(flags & kFlagNoQuad))
                 ? (tolerance ? (n >= 4000000 ? quadMaskTol : (small ? quadMaskTolSmall : (quadMaskTol & 7))) : ((small ? quadMaskSmall : quadMask) & ~kQuadSurfaceBit))
                 : 0;
    // two lanes per particle: from 4 M particles on the head and the viscosity+colour sweep gain 7-8 % (r03: 1.10 -> 1.02 ms and
    // 1.18 -> 1.09 ms at 10.3 M; below, where they are not bound by the L1, they lose 10-25 %: r02)
    c.duo = (use && !c.tileFmt && !(flags & kFlagNoQuad)) ? (n >= 4000000 ? (duoMask | duoMaskLarge) : duoMask) : 0;
    c.n = n;
    c.vel4 = vel4w();
    c.cg4 = cg4w();
    c.numTiles = (n + kTile - 1) / kTile;
    c.tileOrder = (orderValid && orderTiles == c.numTiles && !(flags & kFlagLinearTiles)) ? tileOrder.addr() : nullptr;
    c.tile0 = 0; c.lo = 0; c.hi = n;
    c.lo2 = c.hi2 = 0; c.tileSplit = 0x7fffffff; c.tile1 = 0;
    if (rangeLo >= 0) {                       // a contiguous sub-range: linear tiles from the first one it touches
        c.lo = std::min(rangeLo, n); c.hi = std::min(std::max(rangeHi, c.lo), n);
        // a big range that covers most of the tiles (the interior of a wide slab) keeps the schedule: the launch visits every tile
        // and the tiles outside leave at once (tile_outside).  Measured at 10.3 M particles: -1.2 % per step with 1 or 2 slabs,
        // +1.2 % with 8 (1.3 M particles per slab: its x-layers fit the L2 anyway), so smaller ranges walk their tiles linearly.
        const bool scheduled = c.tileOrder && rangeLo2 < 0 && rangeOrder && 2LL * (c.hi - c.lo) >= n && c.hi - c.lo >= rangeOrderMin;
        if (!scheduled) {
            c.tile0 = c.lo / kTile;
            c.numTiles = c.hi > c.lo ? (c.hi - 1) / kTile - c.tile0 + 1 : 0;
            c.tileOrder = nullptr;
        }
        if (rangeLo2 >= 0) {                  // a second range behind the first: its tiles follow in the same launch; a tile both ranges
            const int lo2 = std::min(std::max(rangeLo2, c.hi), n), hi2 = std::min(std::max(rangeHi2, lo2), n);   // touch is launched once
            if (hi2 > lo2) {
                const int firstB = std::max(lo2 / kTile, c.tile0 + c.numTiles), lastB = (hi2 - 1) / kTile;
                c.lo2 = lo2; c.hi2 = hi2;
                c.tileSplit = c.numTiles; c.tile1 = firstB - c.numTiles;
                c.numTiles += std::max(lastB - firstB + 1, 0);
            }
        }
    }
    c.posf = posfw();
    const bool skinNow = skinRows && skin > 0.0f && use;
    c.stale = skinNow ? staleFlag.addr(activeFlag) : nullptr;
    c.rowCell = (skinNow && rowCell) ? rowCell->addr() : nullptr;
    c.buildCut = k.tCut;
    if ((skinRows || persistRows) && skin > 0.0f) { const float rc = sqrtf(k.tCut) + skin; c.buildCut = rc * rc; }
    c.gate = gate;
    // (r04: from 4 M particles on the rate sweeps of a tolerance-mode step took the STRICT quad kernel -- cut for 8 waves per SIMD where
    // the tolerance walk needed 6: 0.787 vs 0.840 ms per launch at 10.3 M.  r05: with its fused multiply-adds written out the tolerance
    // walk needs 64 VGPRs, runs at 8 waves too and wins, 12.13 vs 12.32 ms per step; sphx_tuning.tol_strict_rate = 1 restores the r04
    // choice for measurements.  Never with persistent rows: their strict walk would re-derive the plain-operator predicate per pair.)
    c.plainBits = (tolerance && !(persistRows && skin > 0.0f) && !(skinRows && skin > 0.0f) && n >= 4000000 && strictRateInTol) ? 1 : 0;
    c.persist = (persistRows && skin > 0.0f && use && rowCell) ? 1 : 0;
    if (c.persist) { c.rowCell = rowCell->addr(); c.tileFmt = nullptr; }
    c.massUniform = allowPacked ? massUniform.addr() : nullptr;
    c.overflowMax = nullptr;
    c.brick = (use && listIsBrick) ? 1 : 0;
    c.brickFault = rowOverflow.addr(1);
    c.brickTab = brickTab ? reinterpret_cast<const BrickTables*>(brickTab->addr()) : nullptr;
    c.brickCount = rowOverflow.addr(2);
    c.brickBlocks = brickBlocks;
    return c;
}

// The compact-brick stage serves whole-domain systems under the tolerance arithmetic whose sweeps run on the binned positions
// (not PBD), from `brickMin` particles on (below, the quad walks are as fast: profiles/r03_ubench_brick.txt).
bool SweepCache::brickMode() const
{
    return brickWanted && !brickFailed && tolerance && !isSlab && allowTiles && !(flags & (kFlagTiles | kFlagNoList)) && rangeLo < 0 &&
           !(skinRows && skin > 0.0f) && !persistRows && n >= brickMin;
}

// Adaptive row capacity: called between steps (never inside a captured graph).  One 4-byte read every 8+ steps.
void SweepCache::tuneRowCapacity(int stepsSinceLastCall)
{
    if ((!capAuto && !listIsBrick) || !nbr) return;
    capCheckSteps += stepsSinceLastCall;
    // (the opt-in brick schedule reports a brick it could not stage through a device flag and SKIPS that brick meanwhile: look at
    // the flag at every call -- after every step() and every 16 steps of a stepN batch -- not every 8+ steps: ADVICE r03)
    if (capCheckSteps < 8 && !listIsBrick) return;
    capCheckSteps = 0;
    int words[3] = {0, 0, 0};
    HIP_CALL(hipMemcpyAsync(words, rowOverflow.addr(), 3 * sizeof(int), hipMemcpyDeviceToHost, stream()));
    HIP_CALL(hipStreamSynchronize(stream()));
    if (listIsBrick && words[2] > 0) {         // launch about as many blocks as there are bricks (they stride over the list anyway)
        const int want = std::min(brick_count(g), words[2] + words[2] / 8 + 8);
        if (want > brickBlocks || want < brickBlocks - brickBlocks / 4) { brickBlocks = want; ++generation; }
    }
    if (words[1] != 0 && !brickFailed) {       // a single cell's neighbourhood outgrew the brick stage: back to the global rows for good
        fprintf(stderr, "sphx: a brick's one-cell slice exceeded the LDS stage (density far beyond rest): the compact-brick path is switched off; "
                        "the particles of that brick kept their previous values in the sweeps since the last check\n");
        brickFailed = true; listValid = false; ++generation;
    }
    const int longest = words[0];
    if (!capAuto || longest <= cap) return;
    HIP_CALL(hipMemsetAsync(rowOverflow.addr(), 0, sizeof(int), stream()));
    // Bounded growth: one dense clump must not size the rows of ALL particles -- beyond kRowCapMax entries a
    // particle walks the cells directly (same bits, slower), which is what the fixed 96-entry rows of r02 did.
    constexpr int kRowCapMax = 192;
    const int want = std::min(kRowCapMax, (longest + 8 + kRowChunk - 1) / kRowChunk * kRowChunk);
    if (want <= cap) return;                   // already at the bound: nothing to reallocate, no graph to drop
    // reallocate HERE, between steps: the next step may be captured into a hipGraph, and a capture must not allocate.  The new
    // store is allocated BEFORE the old one is dropped; when that fails the old rows (and capacity) stay in service.
    const unsigned long long entries = (unsigned long long)((std::max(capN, n) + 63) / 64) * 64ull * (unsigned long long)want;
    std::unique_ptr<RowStore> bigger;
    try { bigger.reset(new RowStore(entries)); }
    catch (const DeviceAllocError&) {
        (void)hipGetLastError();
        capAuto = false;                       // no further attempts: overflowing particles keep walking the cells
        fprintf(stderr, "sphx: no memory for %d-entry neighbour rows; staying at %d entries per particle\n", want, cap);
        return;
    }
    cap = want;
    nbr = std::move(bigger);
    listValid = false;
    ++generation;
    if (persistRows) requestRebuild();
}

// dst = src when *flag != 0 (persistent rows: the cell table of the build is kept beside the live one)
__global__ void k_copy_int_if(int* __restrict__ dst, const int* __restrict__ src, int count, const int* __restrict__ flag)
{
    if (*flag == 0) return;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) dst[i] = src[i];
}

void SweepCache::ensureList(const DArray<int>& csF, const DArray<int>& csB)
{
    if (listValid || (flags & kFlagNoList) || n <= 0) return;
    if (persistRows && skin > 0.0f) {
        // Persistent rows: the same launches every step -- copy of the cell table, then the builder -- and every wave of them
        // leaves at once unless the grid pass of this step raised persistFlags[0] (SPHSystem::persistentSearch: some particle
        // moved more than 0.49 skin relative to the others since the build, or the host asked).  Graph-replayable.
        const unsigned long long entries = (unsigned long long)((std::max(capN, n) + 63) / 64) * 64ull * (unsigned long long)cap;
        if ((unsigned long long)capN + (unsigned long long)nbCap > (unsigned long long)kIndexMask) { persistRows = false; flags |= kFlagNoList; ++generation; return; }
        if (!nbr || nbr->entries < entries) { nbr.reset(); nbr.reset(new RowStore(entries)); ++generation; requestRebuild(); }
        if (!posBuild) { posBuild.reset(new DArray<float>(4u * (unsigned)capN)); rowCell.reset(new DArray<int>((unsigned)capN)); ++generation; requestRebuild(); }
        if (!csBuild || csBuild->length() != csF.length()) { csBuild.reset(new DArray<int>(csF.length())); ++generation; requestRebuild(); }
        ensureTileOrder();
        listValid = false;                         // ctx() must hand out the LIVE cell tables to the builder
        SweepCtx c = ctx(csF, csB);
        c.nbr = nullptr; c.overflowMax = rowOverflow.addr(); c.persist = 0; c.rowCell = nullptr;
        ScopedKernel t("build_neighbor_list");
        k_copy_int_if<<<blocks_for((int)csF.length()), 256, 0, stream()>>>(csBuild->addr(), csF.addr(), (int)csF.length(), persistFlags.addr(0));
        launchBuild(c, reinterpret_cast<float4*>(posBuild->addr()), persistFlags.addr(0), nullptr);
        listCsF = csBuild->addr(); listCsB = csB.addr();
        listIsBrick = false;
        listValid = true;
        return;
    }
    // sized for the CAPACITY, not the current count: sphx_set_count may raise n up to capN later
    // (slab drivers do so every step) and the rows of all ceil(n/64) tiles must fit
    const unsigned long long entries = (unsigned long long)((std::max(capN, n) + 63) / 64) * 64ull * (unsigned long long)cap;
    if ((unsigned long long)capN + (unsigned long long)nbCap > (unsigned long long)kIndexMask) { flags |= kFlagNoList; ++generation; return; }
    if (!nbr || nbr->entries < entries) { nbr.reset(); nbr.reset(new RowStore(entries)); ++generation; }
    ensureTileOrder();
    const int keepLo = rangeLo, keepHi = rangeHi, keepLo2 = rangeLo2, keepHi2 = rangeHi2;
    rangeLo = rangeHi = rangeLo2 = rangeHi2 = -1;                    // rows are always built for every particle
    SweepCtx c = ctx(csF, csB);
    rangeLo = keepLo; rangeHi = keepHi; rangeLo2 = keepLo2; rangeHi2 = keepHi2;
    c.nbr = nullptr; c.overflowMax = rowOverflow.addr();
    listCsF = csF.addr(); listCsB = csB.addr();
    ScopedKernel t("build_neighbor_list");
    const bool skinMode = skinRows && skin > 0.0f;
    if (skinMode) {          // remember where every particle is: the position updates measure against it
        if (!posBuild) { posBuild.reset(new DArray<float>(4u * (unsigned)capN)); rowCell.reset(new DArray<int>((unsigned)capN)); ++generation; }
        if (!changedList) { changedCap = std::max(1024, capN / 8); changedList.reset(new DArray<int>(2u * (unsigned)changedCap)); ++generation; }
        HIP_CALL(hipMemsetAsync(staleFlag.addr(), 0, 2 * sizeof(int), stream()));
        HIP_CALL(hipMemsetAsync(staleFlag.addr(3), 0, 2 * sizeof(int), stream()));
        activeFlag = 0;
    }
    listIsBrick = brickMode();
    if (listIsBrick) {
        // the bricks of this step: list + tables, then the rows.  Launches use `brickBlocks` blocks that stride over the list
        // (refined from the list length between steps); the table store holds min(all bricks, n / 32 + 4096) entries
        const int total = brick_count(g);
        const int capacity = std::min(total, n / 32 + 4096);
        if (!brickTab || (long long)brickTab->length() < (long long)capacity * (long long)(sizeof(BrickTables) / 4)) {
            brickTab.reset(new DArray<int>((unsigned)((long long)capacity * (long long)(sizeof(BrickTables) / 4))));
            ++generation;
        }
        if (brickBlocks <= 0) brickBlocks = std::min(total, 8192);
        HIP_CALL(hipMemsetAsync(rowOverflow.addr(2), 0, sizeof(int), stream()));
        k_brick_list<<<total, 256, 0, stream()>>>(c, reinterpret_cast<BrickTables*>(brickTab->addr()), rowOverflow.addr(2), capacity, rowOverflow.addr(1));
        c = ctx(csF, csB); c.nbr = nullptr; c.overflowMax = rowOverflow.addr();      // (now with the table pointers)
        c.brickTab = reinterpret_cast<const BrickTables*>(brickTab->addr()); c.brickCount = rowOverflow.addr(2); c.brickBlocks = brickBlocks;
        const size_t lds = (size_t)(kBrickSlots + 1) * sizeof(float4);
        k_build_brick<<<xcd_grid(brickBlocks * kBrickThreads, kBrickThreads), kBrickThreads, lds, stream()>>>(c, nbr->rows, nbrCount.addr());
    } else launchBuild(c, skinMode ? reinterpret_cast<float4*>(posBuild->addr()) : nullptr, nullptr, nullptr);
    listValid = true;
}

// Rows of the particles [rangeLo, rangeHi) only, for the positions as they are NOW (slab layer, PBD: the interior of a
// slab is rebuilt and swept while the halo of the edge layers' new positions is still in flight; the edges follow).
// Rows are per particle, so building them range by range gives the rows one build of everything would give.
void SweepCache::buildListForRange(const DArray<int>& csF, const DArray<int>& csB)
{
    if ((flags & kFlagNoList) || n <= 0) return;
    const unsigned long long entries = (unsigned long long)((std::max(capN, n) + 63) / 64) * 64ull * (unsigned long long)cap;
    if ((unsigned long long)capN + (unsigned long long)nbCap > (unsigned long long)kIndexMask) { flags |= kFlagNoList; ++generation; return; }
    if (!nbr || nbr->entries < entries) { nbr.reset(); nbr.reset(new RowStore(entries)); ++generation; }
    listValid = false;                         // ctx() must hand out the live cell tables
    SweepCtx c = ctx(csF, csB);                // keeps the launch range
    c.nbr = nullptr; c.overflowMax = rowOverflow.addr();
    listCsF = csF.addr(); listCsB = csB.addr();
    ScopedKernel t("build_neighbor_list");
    listIsBrick = false;
    if (c.numTiles > 0) launchBuild(c, nullptr, nullptr, nullptr);
    listValid = true;
}

void SweepCache::launchBuild(const SweepCtx& c, float4* posBuildOut, const int* flagNow, int* flagNext)
{
    unsigned int* rows = nbr->rows;
    // few particles: the walk's latency, not its throughput, is the builder's time -- 16 lanes per particle up to the measured
    // crossover with the lane-per-particle builder (kGroupBuildMax; profiles/r06_small_scene_builder.txt)
    if (tuning().group_build_max >= 0 && n <= (tuning().group_build_max > 0 ? tuning().group_build_max : kGroupBuildMax) && !(flags & kFlagTiles))
        k_build_list_group<<<blocks_for(n * kBuildGroup, kWideBlock), kWideBlock, 0, stream()>>>(c, rows, nbrCount.addr(), posBuildOut, rowCell ? rowCell->addr() : nullptr, flagNow, flagNext, staleFlag.addr(2));
    else if (allowTiles && (flags & kFlagTiles))
        k_build_list<true><<<xcd_grid(n, kWideBlock), kWideBlock, 0, stream()>>>(c, rows, nbrCount.addr(), tileFmt.addr(), posBuildOut, rowCell ? rowCell->addr() : nullptr, flagNow, flagNext, staleFlag.addr(2));
    else
        k_build_list<false><<<xcd_grid(n, kWideBlock), kWideBlock, 0, stream()>>>(c, rows, nbrCount.addr(), tileFmt.addr(), posBuildOut, rowCell ? rowCell->addr() : nullptr, flagNow, flagNext, staleFlag.addr(2));
}

SkinWatch SweepCache::skinWatch() const
{
    SkinWatch w;
    w.posBuild = reinterpret_cast<const float4*>(posBuild->addr()); w.rowCell = rowCell->addr();
    w.stale = staleFlag.addr(activeFlag); w.limit2 = staleLimit2();
    if (changedList && !tuning().pbd_no_partial) {
        w.changedCount = staleFlag.addr(3 + activeFlag); w.changedList = changedList->addr(activeFlag * changedCap); w.listCap = changedCap;
    }
    return w;
}

// Skin rows between two Jacobi iterations: the rows of particles that changed their cell are rebuilt, and all of them if (and only
// if) the last position update raised the flag
void SweepCache::rebuildIfStale(const DArray<int>& csF, const DArray<int>& csB)
{
    if (!(skinRows && skin > 0.0f) || !listValid || !nbr || !posBuild || n <= 0) return;
    const int keepLo = rangeLo, keepHi = rangeHi, keepLo2 = rangeLo2, keepHi2 = rangeHi2;
    rangeLo = rangeHi = rangeLo2 = rangeHi2 = -1;
    SweepCtx c = ctx(csF, csB);
    rangeLo = keepLo; rangeHi = keepHi; rangeLo2 = keepLo2; rangeHi2 = keepHi2;
    c.nbr = nullptr; c.stale = nullptr; c.overflowMax = rowOverflow.addr();
    ScopedKernel t("rebuild_rows_if_stale");
    if (changedList && !tuning().pbd_no_partial)
        k_build_list_changed<<<blocks_for(changedCap * kBuildGroup, kWideBlock), kWideBlock, 0, stream()>>>(
            c, nbr->rows, nbrCount.addr(), reinterpret_cast<const float4*>(posBuild->addr()), rowCell->addr(), staleFlag.addr(activeFlag),
            staleFlag.addr(3 + activeFlag), changedList->addr(activeFlag * changedCap), changedCap, staleFlag.addr(3 + (activeFlag ^ 1)), staleFlag.addr(5));
    launchBuild(c, reinterpret_cast<float4*>(posBuild->addr()), staleFlag.addr(activeFlag), staleFlag.addr(activeFlag ^ 1));
    activeFlag ^= 1;
}

// ------------------------------------------------------------------------------ KernelTimer
bool KernelTimer::enabled = false;
std::string KernelTimer::filter;
namespace {
struct TimedSpan { std::string name; hipEvent_t a, b; };
std::vector<TimedSpan> g_spans;
std::vector<size_t> g_open;      // indices of spans begun but not ended (scopes may nest)
}
void KernelTimer::begin(const char* name)
{
    TimedSpan s; s.name = name;
    HIP_CALL(hipEventCreate(&s.a)); HIP_CALL(hipEventCreate(&s.b));
    HIP_CALL(hipEventRecord(s.a, stream()));
    g_open.push_back(g_spans.size());
    g_spans.push_back(s);
}
void KernelTimer::end()
{
    HIP_CALL(hipEventRecord(g_spans[g_open.back()].b, stream()));
    g_open.pop_back();
}
void KernelTimer::collect(std::vector<std::string>& names, std::vector<float>& ms)
{
    HIP_CALL(hipStreamSynchronize(stream()));
    for (auto& s : g_spans) {
        float t = 0.0f;
        HIP_CALL(hipEventElapsedTime(&t, s.a, s.b));
        names.push_back(s.name); ms.push_back(t);
    }
}
void KernelTimer::reset()
{
    for (auto& s : g_spans) { HIP_CALL(hipEventDestroy(s.a)); HIP_CALL(hipEventDestroy(s.b)); }
    g_spans.clear();
    g_open.clear();
}

}  // namespace sphx
