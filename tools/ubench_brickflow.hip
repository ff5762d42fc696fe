// ubench_brickflow.hip — micro-benchmark of the latency-hidden LDS-brick sweep.
//
// Same scene, rows and tolerance pair arithmetic as tools/ubench_sweep.hip, three data paths:
//   Q4     the engine's hot path: quad-per-particle walk over 32-bit global rows, 16-byte global gathers       (baseline)
//   BRICK  r03's compact-brick kernel: one 512-thread block per 4x4x4-cell brick, stage -> barrier -> lane-per-particle walk
//   FLOW   the new form: persistent blocks (one per CU) stride over a balanced, XCD-contiguous share of the brick list;
//          NL loader waves fill an NBUF-deep ring of stages with global_load_lds_dwordx4 (LDS DMA, no VGPRs, no ds_write)
//          one or more bricks ahead; the consumer waves walk rows quad-per-particle (ds_read_b128 per neighbour record)
//          and are NOT tied to a block barrier: work items are groups of 16 particles dealt round-robin over the consumer
//          waves across brick borders, a wave waits for a stage through an LDS counter (ready) and hands it back through
//          another (left), and it prefetches the next group's own records and row blocks from global memory while it
//          computes the current one.
// Layouts of FLOW (what an engine builder would write):
//   descs[k]        {runFirst, numRuns, staged, groupBase}; groups of a brick = ceil(own / 16); slot 0 of a stage = dummy record
//   runs[]          {start, len, base}: halo run -> stage slots [base, base + len)
//   ownRec[G*16+q]  {index | -1, slot | count << 16}
//   grpBlocks[G]    row blocks of group G (a block = 8 entries per particle = one u32 per lane)
//   rows[((G*2 + b/4)*64 + lane)*4 + b%4]   block b of lane: lo/hi 16 bits = entries (2b)*4+g and (2b+1)*4+g of particle q = lane/4, g = lane%4
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fopenmp tools/ubench_brickflow.hip -o tools/ubench_brickflow
//   ./ubench_brickflow [nx=190] [reps=10] [variants: letters, default all]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

constexpr float kEps = 1e-6f;
constexpr float kPi = 3.14159265358979323846f;
constexpr int kCap = 64;          // row capacity (entries)
constexpr int kCapB = kCap / 8;   // row blocks per group

struct Consts { float twoOverR, gradScale; };

__device__ __forceinline__ int logical_block() { return (int)(blockIdx.x & 7u) * (int)(gridDim.x >> 3) + (int)(blockIdx.x >> 3); }
static unsigned int xcd_grid(int n, int block) { const int nb = n > 0 ? (n - 1) / block + 1 : 1; return (unsigned int)(((nb + 7) / 8) * 8); }

// tolerance pair term (v_rsq / v_rcp, written-out FMAs): m_j * dot(v_i - v_j, gradW(x_i - x_j))
__device__ __forceinline__ float pair_tol(const Consts& c, float px, float py, float pz, float vx, float vy, float vz, float4 pj, float4 vj)
{
    const float dx = px - pj.x, dy = py - pj.y, dz = pz - pj.z;
    const float r2 = __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
    const float r = r2 * __builtin_amdgcn_rsqf(fmaxf(r2, 1e-30f));
    const float q = r * c.twoOverR;
    const float poly = (q > 1.0f) ? __builtin_fmaf(__builtin_fmaf(-3.0f, q, 12.0f), q, -12.0f) : __builtin_fmaf(9.0f, q, -12.0f) * q;
    const float s = poly * c.gradScale * __builtin_amdgcn_rcpf(q + kEps);
    const float dv = __builtin_fmaf(vz - vj.z, dz, __builtin_fmaf(vy - vj.y, dy, (vx - vj.x) * dx));
    return pj.w * s * dv;
}

__device__ __forceinline__ float4 gather16(const float4* __restrict__ base, unsigned int off)
{
    return *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(base) + off);
}
template <int CTRL> __device__ __forceinline__ float dppf(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}

// ---- Q4: quad walk over global rows (the engine's walk_row_quad under the tolerance contract: one partial sum per lane) ----
template <bool TWO>
__global__ void __launch_bounds__(256) k_q4(Consts c, const float4* __restrict__ posm, const float4* __restrict__ vel4,
                                            const unsigned int* __restrict__ rows, const int* __restrict__ tileSteps,
                                            float* __restrict__ out, int n, int numTilesQ, int capSteps)
{
    constexpr int U = 4;
    const int tile = logical_block() * 4 + (int)(threadIdx.x >> 6);
    if (tile >= numTilesQ) return;
    const int lane = threadIdx.x & 63;
    const int ip = tile * 16 + lane / 4;
    const int i = min(ip, n - 1);
    const float4 self = posm[i];
    const float4 sv = vel4[i];
    const unsigned int* row = rows + ((size_t)tile * capSteps) * 64u + (unsigned)lane;
    const int steps = tileSteps[tile];
    float e = 0.0f;
    for (int s = 0; s < steps; s += U) {
        unsigned int idx[U];
        float4 pj[U], vj[U];
#pragma unroll
        for (int u = 0; u < U; ++u) idx[u] = (s + u < steps) ? row[(size_t)(s + u) * 64u] : (unsigned)n;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            pj[u] = gather16(posm, idx[u] << 4);
            vj[u] = TWO ? gather16(vel4, idx[u] << 4) : make_float4(pj[u].w, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) e += pair_tol(c, self.x, self.y, self.z, sv.x, sv.y, sv.z, pj[u], vj[u]);
    }
    e += dppf<0xB1>(e);
    e += dppf<0x4E>(e);
    if (ip < n && (lane & 3) == 0) out[i] = e;
}

// ---- BRICK (r03): one block per brick, lane-per-particle, single stage ------------------------------------------------------
struct BrickDesc { int runFirst, numRuns, staged, own, ownFirst, rowBase, rounds; };
struct BrickRun { int start, len, base; };

template <int T, bool TWO>
__global__ void __launch_bounds__(T) k_brick(Consts c, const float4* __restrict__ posm, const float4* __restrict__ vel4,
                                             const BrickDesc* __restrict__ bricks, const BrickRun* __restrict__ runs,
                                             const int* __restrict__ ownIndex, const unsigned short* __restrict__ ownSlot,
                                             const uint4* __restrict__ rows, const unsigned char* __restrict__ waveChunks,
                                             float* __restrict__ out, int numBricks, int slots)
{
    extern __shared__ float4 lds[];
    float4* lpos = lds;
    float4* lvel = lds + slots;
    constexpr int kWaves = T / 64;
    const int blk = logical_block();
    if (blk >= numBricks) return;
    const BrickDesc B = bricks[blk];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int r = wave; r < B.numRuns; r += kWaves) {
        const BrickRun R = runs[B.runFirst + r];
        for (int t = lane; t < R.len; t += 64) {
            lpos[R.base + t] = posm[R.start + t];
            if (TWO) lvel[R.base + t] = vel4[R.start + t];
        }
    }
    if (threadIdx.x == 0) {
        lpos[B.staged] = make_float4(1.0e3f, 1.0e3f, 1.0e3f, 0.0f);
        if (TWO) lvel[B.staged] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    __syncthreads();
    int rowAt = B.rowBase;
    for (int rd = 0; rd < B.rounds; ++rd) {
        const int p = rd * T + (int)threadIdx.x;
        const bool has = p < B.own;
        const int i = has ? ownIndex[B.ownFirst + p] : -1;
        const int self = has ? (int)ownSlot[B.ownFirst + p] : B.staged;
        const float4 sp = lpos[self];
        const float4 sv = TWO ? lvel[self] : (has ? vel4[i] : make_float4(0.f, 0.f, 0.f, 0.f));
        const int chunksRound = waveChunks[(size_t)(blk * 8 + rd) * 17 + 16];
        const int chunks = waveChunks[(size_t)(blk * 8 + rd) * 17 + wave];
        const uint4* row = rows + (size_t)rowAt * T + threadIdx.x;
        float e = 0.0f;
        uint4 nxt = chunks > 0 ? row[0] : make_uint4(0, 0, 0, 0);
        for (int ch = 0; ch < chunks; ++ch) {
            const uint4 cur = nxt;
            if (ch + 1 < chunks) nxt = row[(size_t)(ch + 1) * T];
            const unsigned int w[4] = {cur.x, cur.y, cur.z, cur.w};
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                float4 pj[4], vj[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const unsigned int word = w[h * 2 + (u >> 1)];
                    const unsigned int slot = (u & 1) ? (word >> 16) : (word & 0xffffu);
                    pj[u] = lpos[slot];
                    vj[u] = TWO ? lvel[slot] : make_float4(pj[u].w, 0.f, 0.f, 0.f);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) e += pair_tol(c, sp.x, sp.y, sp.z, sv.x, sv.y, sv.z, pj[u], vj[u]);
            }
        }
        if (has) out[i] = e;
        rowAt += chunksRound;
    }
}

// ---- FLOW ---------------------------------------------------------------------------------------------------------------------
struct BDesc { int runFirst, numRuns, staged, groupBase; };
struct OwnRec { int index; unsigned int slotCnt; };

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;
__device__ __forceinline__ void glds16(const float4* src, float4* ldsDst)      // per-lane source, wave-uniform destination base
{
    __builtin_amdgcn_global_load_lds((gbl_ptr_t)src, (lds_ptr_t)ldsDst, 16, 0, 0);
}
__device__ __forceinline__ int lds_load_acquire(int* p) { return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void lds_add_release(int* p, int v) { __hip_atomic_fetch_add(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP); }

// T threads = W consumer waves + NL loader waves; NBUF stages of F * slotsCap records; UB row blocks (2 pairs per lane each) per batch
template <int T, int NBUF, int NL, bool TWO, int UB, bool PROF = false>
__global__ void __launch_bounds__(T) k_flow(Consts c, const float4* __restrict__ posm, const float4* __restrict__ vel4,
                                            const BDesc* __restrict__ descs, const BrickRun* __restrict__ runs,
                                            const OwnRec* __restrict__ ownRec, const unsigned char* __restrict__ grpBlocks,
                                            const unsigned int* __restrict__ rows, const int2* __restrict__ blockRanges,
                                            float* __restrict__ out, int slotsCap, int outDump, int* __restrict__ fault, int mode, unsigned long long* __restrict__ prof = nullptr)
{
    extern __shared__ float4 lds[];
    __shared__ int readyCnt[NBUF], leftW[NBUF][16];      // leftW[s][w]: how many bricks of ring slot s consumer wave w has left behind
    constexpr int W = T / 64 - NL;
    constexpr int F = TWO ? 2 : 1;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    if (threadIdx.x < NBUF) {
        readyCnt[threadIdx.x] = 0;
        for (int w = 0; w < 16; ++w) leftW[threadIdx.x][w] = 0;
        lds[(size_t)threadIdx.x * F * slotsCap] = make_float4(1.0e3f, 1.0e3f, 1.0e3f, 0.0f);          // slot 0: the dummy record
        if (TWO) lds[(size_t)threadIdx.x * F * slotsCap + slotsCap] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    __syncthreads();
    const int2 R = blockRanges[logical_block()];
    const int kFirst = R.x, kLast = R.y;
    if (kFirst >= kLast) return;

    if (wave >= W) {
        // ---- loader wave j: stages brick k into ring slot (k - kFirst) % NBUF as soon as every consumer has left the slot's last brick
        const int j = wave - W;
        // Pieces of 64 consecutive stage slots (slot 0 is the dummy record: the runs cover slots 1 .. staged without gaps).  The run
        // table of a brick is expanded into a slot -> source map in LDS (one ds_write per run), so that a piece is one ds_read + two
        // DMA instructions with every lane busy (run by run a piece has 45 of 64 lanes, and an LDS-DMA instruction costs its ~70
        // issue cycles whatever its lanes).  Pipeline: the table of brick k + 2 is loaded behind the fills of brick k, the map of
        // brick k + 1 is built while those fills are in flight (their source addresses left the map when they were issued).
        int* srcMap = reinterpret_cast<int*>(lds + (size_t)NBUF * F * slotsCap) + (size_t)j * slotsCap;
        auto loadTable = [&](const BDesc& D, bool on) {
            BrickRun t; t.start = 0; t.len = 0; t.base = 0;
            if (on && lane < D.numRuns) t = runs[D.runFirst + lane];
            return t;
        };
        auto buildMap = [&](const BDesc& D, const BrickRun& t) {
#pragma unroll 1
            for (int r = 0; r < D.numRuns; ++r) {
                const int st = __builtin_amdgcn_readlane(t.start, r), ln = __builtin_amdgcn_readlane(t.len, r), bs = __builtin_amdgcn_readlane(t.base, r);
                // straight-line for runs of up to 128 records (a loop here becomes a maze of exec-mask blocks: 260 cycles per run)
                if (lane < ln) srcMap[bs + lane] = st + lane;
                if (lane + 64 < ln) srcMap[bs + lane + 64] = st + lane + 64;
                if (__builtin_expect(ln > 128, 0)) for (int u = lane + 128; u < ln; u += 64) srcMap[bs + u] = st + u;
            }
        };
        __builtin_amdgcn_s_setprio(3);        // the loader is the youngest wave of its SIMD: without this it issues in the consumers' gaps
        unsigned long long lWait = 0, lIssue = 0, lBuild = 0, lDrain = 0, lLast = PROF ? __builtin_amdgcn_s_memtime() : 0ull;
        auto lstamp = [&](unsigned long long& acc) { if (PROF) { const unsigned long long now = __builtin_amdgcn_s_memtime(); acc += now - lLast; lLast = now; } };
        BDesc D1 = descs[kFirst], D2 = descs[min(kFirst + 1, kLast)];
        BrickRun my1 = loadTable(D1, true), my2 = loadTable(D2, kFirst + 1 < kLast);
        __builtin_amdgcn_s_waitcnt(0x0F70);
        buildMap(D1, my1);
#pragma unroll 1
        for (int k = kFirst; k < kLast; ++k) {
            const int kr = k - kFirst, s = kr % NBUF, round = kr / NBUF;
            if (round > 0 && !(mode & 4)) { int spins = 0; while (!__all(lane >= W || lds_load_acquire(&leftW[s][lane < W ? lane : 0]) >= round)) { __builtin_amdgcn_s_sleep(2); if (++spins > (1 << 20)) { if (lane == 0) atomicAdd(fault, 1); break; } } }
            lstamp(lWait);
            float4* bufPos = lds + (size_t)s * F * slotsCap;
            float4* bufVel = bufPos + slotsCap;
            const int staged = (mode & 1) ? 0 : D1.staged;
#pragma unroll 1
            for (int p0 = 1 + 64 * j; p0 <= staged; p0 += 64 * NL * 4) {
                int src[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) src[u] = srcMap[min(p0 + 64 * NL * u + lane, slotsCap - 1)];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int pu = p0 + 64 * NL * u;
                    if (pu + lane <= staged) {
                        glds16(posm + src[u], bufPos + pu);
                        if (TWO) glds16(vel4 + src[u], bufVel + pu);
                    }
                }
            }
            lstamp(lIssue);
            const BDesc D3 = descs[min(k + 2, kLast)];
            const BrickRun my3 = loadTable(D3, k + 2 < kLast);
            if (k + 1 < kLast) buildMap(D2, my2);
            lstamp(lBuild);
            // (the builtin, not inline asm: the compiler's own waitcnt bookkeeping must see that the table has arrived, or it drains
            // the DMA queue in front of every readlane of it -- one run in flight at a time, 3x slower fills)
            __builtin_amdgcn_s_waitcnt(0x0F70);          // vmcnt(0): this brick's fills have landed, the table of brick k + 2 is here
            asm volatile("" ::: "memory");
            if (lane == 0) lds_add_release(&readyCnt[s], 1);
            lstamp(lDrain);
            D1 = D2; D2 = D3; my1 = my2; my2 = my3;
        }
        if (PROF && lane == 0 && j == 0) {
            atomicAdd(&prof[8], lWait); atomicAdd(&prof[9], lIssue); atomicAdd(&prof[10], lBuild); atomicAdd(&prof[11], lDrain); atomicAdd(&prof[12], (unsigned long long)(kLast - kFirst));
        }
        return;
    }

    // ---- consumer wave: groups G = first + wave, + W, ... of the block's bricks
    if (mode & 4) return;
    const int q = lane >> 2;
    const int Gend = descs[kLast].groupBase;
    int G = descs[kFirst].groupBase + wave;
    int k = kFirst;
    int gbNext = descs[kFirst + 1].groupBase;
    OwnRec own; own.index = -1; own.slotCnt = 0u;
    int nb = 0;
    unsigned int rw[kCapB];
#pragma unroll
    for (int b = 0; b < kCapB; ++b) rw[b] = 0u;
    // two stages ahead: the block count of a group is loaded one iteration before its rows are (the row loads are guarded by
    // it; fetched together, the wave would sit out the count's latency in the middle of every iteration)
    auto prefetch = [&](int Gp, int nbOfGp) {
        own = ownRec[(size_t)Gp * 16 + q];
        nb = __builtin_amdgcn_readfirstlane(nbOfGp);
        // rows: [G][half][lane] uint4 = blocks 4 * half .. 4 * half + 3 of this lane: one 16-byte load covers a lattice group's 4 blocks
        const uint4* rp = reinterpret_cast<const uint4*>(rows) + (size_t)Gp * 128 + lane;
        const uint4 a = rp[0];
        uint4 bq = make_uint4(0u, 0u, 0u, 0u);
        if (nb > 4) bq = rp[64];
        rw[0] = a.x; rw[1] = a.y; rw[2] = a.z; rw[3] = a.w; rw[4] = bq.x; rw[5] = bq.y; rw[6] = bq.z; rw[7] = bq.w;
    };
    auto leave = [&](int kk) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        // (a per-wave count, not a sum over waves: a wave without a group in a brick passes it before others have finished the
        // slot's previous brick, and a sum cannot tell "15 waves left brick k" from "some left k + NBUF already")
        if (lane == 0) __hip_atomic_store(&leftW[(kk - kFirst) % NBUF][wave], (kk - kFirst) / NBUF + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    };
    int nbAhead = G < Gend ? (int)grpBlocks[G] : 0;
    if (G < Gend) prefetch(G, nbAhead);
    nbAhead = G + W < Gend ? (int)grpBlocks[G + W] : 0;
    unsigned long long tPoll = 0, tComp = 0, tVm = 0, tTop = 0, tLast = PROF ? __builtin_amdgcn_s_memtime() : 0ull; int nIter = 0;
    auto stamp = [&](unsigned long long& acc) { if (PROF) { const unsigned long long now = __builtin_amdgcn_s_memtime(); acc += now - tLast; tLast = now; } };
    int pendIdx = -1; float pendE = 0.0f;          // the previous group's result: stored BEFORE the next prefetch is issued, so that
                                                   // the wait for the prefetched words is not a wait for the store as well
#pragma unroll 1
    while (G < Gend) {
        const OwnRec cOwn = own;
        const int cnb = nb;
        unsigned int cr[kCapB];
#pragma unroll
        for (int b = 0; b < kCapB; ++b) cr[b] = rw[b];
        const int Gn = G + W;
        out[((lane & 3) == 0 && pendIdx >= 0) ? pendIdx : outDump + lane] = pendE;      // (no branch around the store: the compiler can then count it)
        if (Gn < Gend) prefetch(Gn, nbAhead);
        const int nbAhead2 = Gn + W < Gend ? (int)grpBlocks[Gn + W] : 0;
        stamp(tTop);
        while (G >= gbNext) { leave(k); ++k; gbNext = descs[k + 1].groupBase; }
        const int kr = k - kFirst, s = kr % NBUF, round = kr / NBUF;
        { int spins = 0; while (lds_load_acquire(&readyCnt[s]) < NL * (round + 1)) { __builtin_amdgcn_s_sleep(1); if (++spins > (1 << 20)) { if (lane == 0) atomicAdd(fault, 1); break; } } }
        stamp(tPoll);
        const float4* bufPos = lds + (size_t)s * F * slotsCap;
        const float4* bufVel = bufPos + slotsCap;
        const unsigned int ownSlot = cOwn.slotCnt & 0xffffu;
        const float4 sp = bufPos[ownSlot];
        const float4 sv = TWO ? bufVel[ownSlot] : (cOwn.index >= 0 ? vel4[cOwn.index] : make_float4(0.f, 0.f, 0.f, 0.f));
        float e = 0.0f;
        // the first FB blocks as ONE batch (2 * FB pairs per lane in flight: a group of the lattice has 4 blocks; blocks past the
        // group's count hold the dummy slot), the rest in batches of UB
        constexpr int FB = 4;
        if (!(mode & 2)) {
            float4 pj[2 * FB], vj[2 * FB];
#pragma unroll
            for (int u = 0; u < 2 * FB; ++u) {
                const unsigned int word = cr[u >> 1];
                const unsigned int slot = (u & 1) ? (word >> 16) : (word & 0xffffu);
                pj[u] = bufPos[slot];
                vj[u] = TWO ? bufVel[slot] : make_float4(pj[u].w, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int u = 0; u < 2 * FB; ++u) e += pair_tol(c, sp.x, sp.y, sp.z, sv.x, sv.y, sv.z, pj[u], vj[u]);
        }
#pragma unroll
        for (int b = FB; b < kCapB; b += UB) {
            if (b < cnb && !(mode & 2)) {
                float4 pj[2 * UB], vj[2 * UB];
#pragma unroll
                for (int u = 0; u < 2 * UB; ++u) {
                    const unsigned int word = cr[b + (u >> 1)];
                    const unsigned int slot = (u & 1) ? (word >> 16) : (word & 0xffffu);
                    pj[u] = bufPos[slot];
                    vj[u] = TWO ? bufVel[slot] : make_float4(pj[u].w, 0.f, 0.f, 0.f);
                }
#pragma unroll
                for (int u = 0; u < 2 * UB; ++u) e += pair_tol(c, sp.x, sp.y, sp.z, sv.x, sv.y, sv.z, pj[u], vj[u]);
            }
        }
        e += dppf<0xB1>(e);
        e += dppf<0x4E>(e);
        pendIdx = cOwn.index; pendE = e;
        if (PROF) { asm volatile("" :: "v"(e)); stamp(tComp); __builtin_amdgcn_s_waitcnt(0x0F70); stamp(tVm); ++nIter; }
        nbAhead = nbAhead2;
        G = Gn;
    }
    if ((lane & 3) == 0 && pendIdx >= 0) out[pendIdx] = pendE;
    for (; k < kLast; ++k) leave(k);
    if (PROF && lane == 0) {
        atomicAdd(&prof[0], tTop); atomicAdd(&prof[1], tPoll); atomicAdd(&prof[2], tComp); atomicAdd(&prof[3], tVm); atomicAdd(&prof[4], (unsigned long long)nIter);
    }
}


// ---- SELF: the consumers stage the next brick themselves --------------------------------------------------------------------
// No loader wave and no LDS DMA (an LDS-DMA instruction costs ~70 issue cycles per KiB and a loader wave drains its queue once per
// brick; an ordinary 16-byte load costs the texture path 16 cycles per KiB).  The first min(WC, groups) groups of brick k each carry
// a share of brick k + 1's staging items (an item = up to 64 consecutive records of one halo run): loads issued at the top of the
// iteration, next to the prefetch of the next group's rows, land behind the group's arithmetic; the 16-byte ds_writes happen at
// the top of the wave's NEXT iteration (by then every wave has long left the brick that used the stage before).
// Two stages, stage of brick k = k & 1; `stagedCnt[s]` counts items written into stage s (cumulative over the block's bricks),
// `leftW[s][w]` = how many bricks of stage s wave w has left behind.
// Everything a wave needs about a group comes with the group's header, loaded two iterations ahead:
//   grpHdr[G] = {row blocks | items << 8, first item, brick, items of the block's bricks of this parity up to and incl. this brick}
struct GrpHdr { int nbItems, itemFirst, brick, cum; };
template <int T, bool TWO, int MAXI, bool PROF = false, int MINW = 1>
__global__ void __launch_bounds__(T, MINW) k_self(Consts c, const float4* __restrict__ posm, const float4* __restrict__ vel4,
                                            const BDesc* __restrict__ descs, const int4* __restrict__ items, const int4* __restrict__ firstItems,
                                            const OwnRec* __restrict__ ownRec, const GrpHdr* __restrict__ grpHdr,
                                            const unsigned int* __restrict__ rows, const int2* __restrict__ blockRanges,
                                            float* __restrict__ out, int slotsCap, int outDump, int* __restrict__ fault, int mode, unsigned long long* __restrict__ prof = nullptr)
{
    extern __shared__ float4 lds[];
    __shared__ int stagedCnt[2], leftW[2][16];
    constexpr int W = T / 64, F = TWO ? 2 : 1;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    if (threadIdx.x < 2) {
        stagedCnt[threadIdx.x] = 0;
        for (int w = 0; w < 16; ++w) leftW[threadIdx.x][w] = 0;
        lds[(size_t)threadIdx.x * F * slotsCap] = make_float4(1.0e3f, 1.0e3f, 1.0e3f, 0.0f);
        if (TWO) lds[(size_t)threadIdx.x * F * slotsCap + slotsCap] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    __syncthreads();
    const int2 R = blockRanges[logical_block()];
    const int kFirst = R.x, kLast = R.y;
    if (kFirst >= kLast) return;
    auto spinUntil = [&](auto&& cond) { if (mode & 8) return; int spins = 0; while (!cond()) { __builtin_amdgcn_s_sleep(1); if (++spins > (1 << 20)) { if (lane == 0) atomicAdd(fault, 1); break; } } };
    const BDesc D0 = descs[kFirst];
    // cumulative item counts in the headers run over ALL bricks of a parity: what lay before this block's first brick is taken off
    const int Gfirst = D0.groupBase, Gend = descs[kLast].groupBase;
    int cumBase[2];
    {
        const GrpHdr h0 = grpHdr[Gfirst];                                       // brick kFirst: cum includes its own items
        cumBase[kFirst & 1] = h0.cum - D0.numRuns;
        const int k1 = kFirst + 1;                                              // the first brick of the other parity
        cumBase[k1 & 1] = k1 < kLast ? grpHdr[descs[k1].groupBase].cum - descs[k1].numRuns : 0;
    }
    {   // the block's first brick: every wave stages its share (its items in run order: firstItems)
        float4* dstPos = lds + (size_t)(kFirst & 1) * F * slotsCap;
        int done = 0;
        for (int i = wave; i < D0.numRuns; i += W) {
            const int4 it = firstItems[D0.runFirst + i];
            if (lane < it.y) {
                dstPos[it.z + lane] = posm[it.x + lane];
                if (TWO) dstPos[slotsCap + it.z + lane] = vel4[it.x + lane];
            }
            ++done;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (lane == 0 && done) lds_add_release(&stagedCnt[kFirst & 1], done);
    }
    const int q = lane >> 2;
    int G = Gfirst + wave;
    // software pipeline: header two groups ahead; own records, row blocks and item records one group ahead
    GrpHdr hNext, hNext2; hNext.nbItems = 0; hNext.itemFirst = 0; hNext.brick = kFirst; hNext.cum = 0; hNext2 = hNext;
    OwnRec own; own.index = -1; own.slotCnt = 0u;
    unsigned int rw[kCapB];
#pragma unroll
    for (int b = 0; b < kCapB; ++b) rw[b] = 0u;
    int4 itemsNext = make_int4(0, 0, 0, 0);
    auto prefetch = [&](int Gp, const GrpHdr& h) {
        own = ownRec[(size_t)Gp * 16 + q];
        const int nbp = __builtin_amdgcn_readfirstlane(h.nbItems & 0xff), cnt = __builtin_amdgcn_readfirstlane(h.nbItems >> 8);
        const uint4* rp = reinterpret_cast<const uint4*>(rows) + (size_t)Gp * 128 + lane;
        const uint4 a = rp[0];
        uint4 bq = make_uint4(0u, 0u, 0u, 0u);
        if (nbp > 4) bq = rp[64];
        rw[0] = a.x; rw[1] = a.y; rw[2] = a.z; rw[3] = a.w; rw[4] = bq.x; rw[5] = bq.y; rw[6] = bq.z; rw[7] = bq.w;
        itemsNext = make_int4(0, 0, 0, 0);
        if (lane < cnt) itemsNext = items[__builtin_amdgcn_readfirstlane(h.itemFirst) + lane];
    };
    if (G < Gend) { hNext = grpHdr[G]; prefetch(G, hNext); }
    if (G + W < Gend) hNext2 = grpHdr[G + W];
    unsigned long long tTop = 0, tPoll = 0, tComp = 0, tStage = 0, tLast = PROF ? __builtin_amdgcn_s_memtime() : 0ull; int nIter = 0, nDuty = 0;
    auto stamp = [&](unsigned long long& acc) { if (PROF) { const unsigned long long now = __builtin_amdgcn_s_memtime(); acc += now - tLast; tLast = now; } };
    int pendIdx = -1; float pendE = 0.0f;
    int kCur = kFirst;                         // the brick this wave has last announced itself in
    // staging state carried into the next iteration
    int pCount = 0, pBrick = kFirst; int4 pItems = make_int4(0, 0, 0, 0);
    float4 stP[MAXI], stV[MAXI];
#pragma unroll
    for (int m = 0; m < MAXI; ++m) { stP[m] = make_float4(0.f, 0.f, 0.f, 0.f); stV[m] = stP[m]; }
    auto bricksOfParityBefore = [&](int kk, int par) {      // bricks j in [kFirst, kk) with (j & 1) == par
        const int first = kFirst + (((kFirst & 1) == par) ? 0 : 1);
        return kk > first ? (kk - first + 1) / 2 : 0;
    };
    auto flushStage = [&]() {
        if (pCount > 0) {
            // destination: the stage of brick pBrick + 1, used by brick pBrick - 1 before
            const int sd = (pBrick + 1) & 1;
            float4* dstPos = lds + (size_t)sd * F * slotsCap;
            float4* dstVel = dstPos + slotsCap;
            const int need = bricksOfParityBefore(pBrick, sd);
            if (need > 0) spinUntil([&] { return __all(lane >= W || lds_load_acquire(&leftW[sd][lane < W ? lane : 0]) >= need); });
#pragma unroll
            for (int m = 0; m < MAXI; ++m) {
                const int ln = __builtin_amdgcn_readlane(pItems.y, m), bs = __builtin_amdgcn_readlane(pItems.z, m);
                if (lane < ln) {
                    dstPos[bs + lane] = stP[m];
                    if (TWO) dstVel[bs + lane] = stV[m];
                }
            }
            for (int m = MAXI; m < pCount; ++m) {          // (bricks with few carrier groups: the rest of the share, unpipelined)
                const int st = __builtin_amdgcn_readlane(pItems.x, m), ln = __builtin_amdgcn_readlane(pItems.y, m), bs = __builtin_amdgcn_readlane(pItems.z, m);
                if (lane < ln) {
                    dstPos[bs + lane] = posm[st + lane];
                    if (TWO) dstVel[bs + lane] = vel4[st + lane];
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (lane == 0) lds_add_release(&stagedCnt[sd], pCount);
            pCount = 0;
        }
    };
    auto announce = [&](int kk) {               // this wave now works in brick kk: every earlier brick is behind it
        if (kk != kCur) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (lane < 2) __hip_atomic_store(&leftW[lane][wave], bricksOfParityBefore(kk, lane), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            kCur = kk;
        }
    };
#pragma unroll 1
    while (G < Gend) {
        flushStage();
        stamp(tStage);
        const GrpHdr h = hNext;
        const OwnRec cOwn = own;
        unsigned int cr[kCapB];
#pragma unroll
        for (int b = 0; b < kCapB; ++b) cr[b] = rw[b];
        const int4 cItems = itemsNext;
        const int cnb = __builtin_amdgcn_readfirstlane(h.nbItems & 0xff), cCount = __builtin_amdgcn_readfirstlane(h.nbItems >> 8);
        const int k = __builtin_amdgcn_readfirstlane(h.brick), cum = __builtin_amdgcn_readfirstlane(h.cum);
        const int Gn = G + W;
        out[((lane & 3) == 0 && pendIdx >= 0) ? pendIdx : outDump + lane] = pendE;
        hNext = hNext2;
        if (Gn < Gend) prefetch(Gn, hNext);
        if (Gn + W < Gend) hNext2 = grpHdr[Gn + W];
        announce(k);
        const int s = k & 1;
        // this group's share of the next brick's stage: loads now, writes at the top of the next iteration
#pragma unroll
        for (int m = 0; m < MAXI; ++m) {
            const int st = __builtin_amdgcn_readlane(cItems.x, m), ln = __builtin_amdgcn_readlane(cItems.y, m);
            if (lane < ln && !(mode & 1)) {
                stP[m] = posm[st + lane];
                if (TWO) stV[m] = vel4[st + lane];
            }
        }
        pCount = cCount; pBrick = k; pItems = cItems;
        stamp(tTop);
        spinUntil([&] { return lds_load_acquire(&stagedCnt[s]) >= cum - cumBase[s]; });
        stamp(tPoll);
        const float4* bufPos = lds + (size_t)s * F * slotsCap;
        const float4* bufVel = bufPos + slotsCap;
        const unsigned int ownSlot = cOwn.slotCnt & 0xffffu;
        const float4 sp = bufPos[ownSlot];
        const float4 sv = TWO ? bufVel[ownSlot] : (cOwn.index >= 0 ? vel4[cOwn.index] : make_float4(0.f, 0.f, 0.f, 0.f));
        float e = 0.0f;
        constexpr int FB = 4;
        if (!(mode & 2)) {
            float4 pj[2 * FB], vj[2 * FB];
#pragma unroll
            for (int u = 0; u < 2 * FB; ++u) {
                const unsigned int word = cr[u >> 1];
                const unsigned int slot = (u & 1) ? (word >> 16) : (word & 0xffffu);
                pj[u] = bufPos[slot];
                vj[u] = TWO ? bufVel[slot] : make_float4(pj[u].w, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int u = 0; u < 2 * FB; ++u) e += pair_tol(c, sp.x, sp.y, sp.z, sv.x, sv.y, sv.z, pj[u], vj[u]);
        }
#pragma unroll
        for (int b = FB; b < kCapB; b += 2) {
            if (b < cnb && !(mode & 2)) {
                float4 pj[4], vj[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const unsigned int word = cr[b + (u >> 1)];
                    const unsigned int slot = (u & 1) ? (word >> 16) : (word & 0xffffu);
                    pj[u] = bufPos[slot];
                    vj[u] = TWO ? bufVel[slot] : make_float4(pj[u].w, 0.f, 0.f, 0.f);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) e += pair_tol(c, sp.x, sp.y, sp.z, sv.x, sv.y, sv.z, pj[u], vj[u]);
            }
        }
        e += dppf<0xB1>(e);
        e += dppf<0x4E>(e);
        pendIdx = cOwn.index; pendE = e;
        if (PROF) { asm volatile("" :: "v"(e)); stamp(tComp); ++nIter; nDuty += cCount > 0 ? 1 : 0; }
        G = Gn;
    }
    flushStage();
    out[((lane & 3) == 0 && pendIdx >= 0) ? pendIdx : outDump + lane] = pendE;
    announce(kLast);
    if (PROF && lane == 0) { atomicAdd(&prof[0], tTop); atomicAdd(&prof[1], tPoll); atomicAdd(&prof[2], tComp); atomicAdd(&prof[3], tStage); atomicAdd(&prof[4], (unsigned long long)nIter); atomicAdd(&prof[5], (unsigned long long)nDuty); }
}


// ---- PARK: r03's block-per-brick walk made persistent, the NEXT brick's stage parked in registers while this one is walked ------
// One stage per block (two 512-thread blocks per CU), one barrier pair per brick.  While brick k is walked, every thread holds the
// records it will write into the stage for brick k + 1 (NS slots per thread, two fields: 8 NS registers), its own index / slot / row
// chunks for brick k + 1, and the source indices of brick k + 2's slots.  Nothing a thread needs at the start of a brick is still
// in flight: no second LDS stage, no loader wave, no flags.
template <int T, bool TWO, int NS, int NR, int mode = 0, int MINW = 4>
__global__ void __launch_bounds__(T, MINW) k_park(Consts c, const float4* __restrict__ posm, const float4* __restrict__ vel4,
                                            const BrickDesc* __restrict__ bricks, const int* __restrict__ stageBase, const int* __restrict__ srcIdx,
                                            const int* __restrict__ ownIndex, const unsigned short* __restrict__ ownSlot, const float4* __restrict__ ownVel,
                                            const uint4* __restrict__ rows, const int* __restrict__ waveChunks,
                                            const int2* __restrict__ blockRanges, float* __restrict__ out, int slots)
{
    extern __shared__ float4 lds[];
    float4* lpos = lds;
    float4* lvel = lds + slots;
    const int2 Rg = blockRanges[logical_block()];
    const int kFirst = Rg.x, kLast = Rg.y;
    if (kFirst >= kLast) return;
    const int t = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const float4 far = make_float4(1.0e3f, 1.0e3f, 1.0e3f, 0.0f), zero = make_float4(0.f, 0.f, 0.f, 0.f);
    // per-thread state of a brick: own particle (round 0), first NR row chunks, the wave's chunk count
    struct Own { int i, self, chunks, chunksRound; float4 sv; uint4 r[NR]; };
    auto loadOwn = [&](const BrickDesc& B, int blk, Own& o) {
        const bool has = t < B.own;
        o.i = has ? ownIndex[B.ownFirst + t] : -1;
        o.self = has ? (int)ownSlot[B.ownFirst + t] : B.staged;
        o.chunks = waveChunks[(size_t)(blk * 8) * 17 + wave];
        o.chunksRound = waveChunks[(size_t)(blk * 8) * 17 + 16];
        o.sv = (!TWO && has) ? ownVel[B.ownFirst + t] : make_float4(0.f, 0.f, 0.f, 0.f);      // (one-field form only: the engine's own fields sit in the stage)
        const uint4* row = rows + (size_t)B.rowBase * T + t;
#pragma unroll
        for (int u = 0; u < NR; ++u) o.r[u] = row[(size_t)u * T];          // (the rows array is padded: chunks past this brick's belong to the next)
    };
    BrickDesc B = bricks[kFirst];
    {   // the block's first brick goes straight into the stage
        const int sb = stageBase[kFirst];
        for (int s = t; s < B.staged; s += T) {
            const int src = srcIdx[sb + s];
            lpos[s] = posm[src];
            if (TWO) lvel[s] = vel4[src];
        }
        if (t == 0) { lpos[B.staged] = far; if (TWO) lvel[B.staged] = zero; }
    }
    Own cur; loadOwn(B, kFirst, cur);
    int S[NS];                                                      // source indices of the NEXT brick's slots t, t + T, ...
    {
        const BrickDesc Bn = bricks[min(kFirst + 1, kLast)];      // (bricks[] has one entry past the end)
        const int sb = stageBase[min(kFirst + 1, kLast)];
#pragma unroll
        for (int j = 0; j < NS; ++j) { const int s = t + T * j; S[j] = (kFirst + 1 < kLast && s < Bn.staged) ? srcIdx[sb + s] : -1; }
    }
    __syncthreads();
#pragma unroll 1
    for (int k = kFirst; k < kLast; ++k) {
        const bool more = k + 1 < kLast;
        const BrickDesc Bn = bricks[min(k + 1, kLast)];
        // 1. the next brick's records (parked until the walk is over) ...
        float4 P[NS], V[NS];
#pragma unroll
        for (int j = 0; j < NS; ++j) {
            P[j] = zero; V[j] = zero;
            if (S[j] >= 0) { P[j] = posm[S[j]]; if (TWO) V[j] = vel4[S[j]]; }
        }
        // ... its own / row words, and the source indices of the brick after it
        Own nxt; nxt.i = -1; nxt.self = 0; nxt.chunks = 0; nxt.chunksRound = 0; nxt.sv = zero;
#pragma unroll
        for (int u = 0; u < NR; ++u) nxt.r[u] = make_uint4(0u, 0u, 0u, 0u);
        if (more) loadOwn(Bn, k + 1, nxt);
        int Sn[NS];
        {
            const BrickDesc Bnn = bricks[min(k + 2, kLast)];
            const int sb = stageBase[min(k + 2, kLast)];
#pragma unroll
            for (int j = 0; j < NS; ++j) { const int s = t + T * j; Sn[j] = (k + 2 < kLast && s < Bnn.staged) ? srcIdx[sb + s] : -1; }
        }
        // 2. walk this brick.  Round 0 (every brick; its inputs are in registers: no memory instruction in the loop) ...
        auto pairs8 = [&](const uint4 curw, const float4 sp, const float4 sv, float& e) {
            const unsigned int w[4] = {curw.x, curw.y, curw.z, curw.w};
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                float4 pj[4], vj[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const unsigned int word = w[h * 2 + (u >> 1)];
                    unsigned int slot = (u & 1) ? (word >> 16) : (word & 0xffffu);
                    if (mode == 1) slot = (slot & 0x7c0u) + (t & 63);                 // ablation: conflict-free slots (one record per lane, consecutive)
                    if (mode == 3) { pj[u] = make_float4(sp.x + slot * 1e-3f, sp.y, sp.z, sp.w); vj[u] = sv; continue; }      // ablation: no LDS reads
                    pj[u] = lpos[slot];
                    vj[u] = TWO ? lvel[slot] : make_float4(pj[u].w, 0.f, 0.f, 0.f);
                }
                if (mode == 2) { for (int u = 0; u < 4; ++u) e += pj[u].x + vj[u].y; continue; }               // ablation: no pair arithmetic
#pragma unroll
                for (int u = 0; u < 4; ++u) e += pair_tol(c, sp.x, sp.y, sp.z, sv.x, sv.y, sv.z, pj[u], vj[u]);
            }
        };
        {
            const float4 sp = lpos[cur.self];
            const float4 sv = TWO ? lvel[cur.self] : cur.sv;
            float e = 0.0f;
#pragma unroll
            for (int ch = 0; ch < NR; ++ch) if (ch < cur.chunks) pairs8(cur.r[ch], sp, sv, e);
            if (__builtin_expect(cur.chunks > NR, 0)) {
                const uint4* row = rows + (size_t)B.rowBase * T + t;
#pragma unroll 1
                for (int ch = NR; ch < cur.chunks; ++ch) pairs8(row[(size_t)ch * T], sp, sv, e);
            }
            if (cur.i >= 0) out[cur.i] = e;
        }
        // ... further rounds (bricks with more than T own particles): unpipelined
        if (__builtin_expect(B.rounds > 1, 0)) {
            int rowAt = B.rowBase + cur.chunksRound;
#pragma unroll 1
            for (int rd = 1; rd < B.rounds; ++rd) {
                const int p = rd * T + t;
                const bool has = p < B.own;
                const int i = has ? ownIndex[B.ownFirst + p] : -1;
                const int self = has ? (int)ownSlot[B.ownFirst + p] : B.staged;
                const int chunks = waveChunks[(size_t)(k * 8 + rd) * 17 + wave], chunksRound = waveChunks[(size_t)(k * 8 + rd) * 17 + 16];
                const float4 sp = lpos[self];
                const float4 sv = TWO ? lvel[self] : ((!TWO && has) ? ownVel[B.ownFirst + p] : zero);
                const uint4* row = rows + (size_t)rowAt * T + t;
                float e = 0.0f;
#pragma unroll 1
                for (int ch = 0; ch < chunks; ++ch) pairs8(row[(size_t)ch * T], sp, sv, e);
                if (i >= 0) out[i] = e;
                rowAt += chunksRound;
            }
        }
        // 3. everyone is done reading: the parked records become the stage
        __syncthreads();
        if (more) {
#pragma unroll
            for (int j = 0; j < NS; ++j) if (S[j] >= 0) { lpos[t + T * j] = P[j]; if (TWO) lvel[t + T * j] = V[j]; }
            if (Bn.staged > NS * T) {                               // (a stage beyond NS slots per thread: the rest unpipelined)
                const int sb = stageBase[k + 1];
                for (int s = t + NS * T; s < Bn.staged; s += T) { const int src = srcIdx[sb + s]; lpos[s] = posm[src]; if (TWO) lvel[s] = vel4[src]; }
            }
            if (t == 0) { lpos[Bn.staged] = far; if (TWO) lvel[Bn.staged] = zero; }
        }
        __syncthreads();
        B = Bn; cur = nxt;
#pragma unroll
        for (int j = 0; j < NS; ++j) S[j] = Sn[j];
    }
}

// ---- host --------------------------------------------------------------------------------------------------------------------
int main(int argc, char** argv)
{
    const int nx = argc > 1 ? atoi(argv[1]) : 190;
    const int reps = argc > 2 ? atoi(argv[2]) : 10;
    const char* which = argc > 3 ? argv[3] : "qbf";
    const float spacing = 0.02f, R = 0.04f, cellLength = 1.01f * R, scale = nx / 24.0f;
    const int ny = 3 * nx / 2, nz = nx;
    const int n = nx * ny * nz;
    const int gx = (int)ceilf(scale / cellLength), gy = gx, gz = gx, C = gx * gy * gz;
    printf("scene: %d x %d x %d = %d particles, grid %d^3, R = %g\n", nx, ny, nz, n, gx, R);

    std::vector<float4> P0(n);
    std::vector<int> cell(n);
    unsigned int rng = 12345u;
    auto jitter = [&]() { rng = rng * 1664525u + 1013904223u; return ((rng >> 8) * (1.0f / 16777216.0f) - 0.5f) * 0.08f * spacing; };
    {
        int qq = 0;
        for (int iy = 0; iy < ny; ++iy) for (int ix = 0; ix < nx; ++ix) for (int iz = 0; iz < nz; ++iz, ++qq) {
            P0[qq] = make_float4(0.27f * scale + spacing * ix + jitter(), 0.10f * scale + spacing * iy + jitter(),
                                 0.27f * scale + spacing * iz + jitter(), 76.596750762082e-6f);
            const int cx = (int)(P0[qq].x / cellLength), cy = (int)(P0[qq].y / cellLength), cz = (int)(P0[qq].z / cellLength);
            cell[qq] = (cx * gy + cy) * gz + cz;
        }
    }
    std::vector<int> cs(C + 2, 0), order(n);
    for (int qq = 0; qq < n; ++qq) cs[cell[qq] + 1]++;
    for (int kk = 0; kk < C + 1; ++kk) cs[kk + 1] += cs[kk];
    {
        std::vector<int> cur(cs.begin(), cs.begin() + C + 1);
        for (int qq = 0; qq < n; ++qq) order[cur[cell[qq]]++] = qq;
    }
    std::vector<float4> posm(n + 1), vel4(n + 1);
    std::vector<int> scell(n);
    for (int qq = 0; qq < n; ++qq) {
        posm[qq] = P0[order[qq]]; scell[qq] = cell[order[qq]];
        rng = rng * 1664525u + 1013904223u;
        vel4[qq] = make_float4(jitter() * 50.f, -0.04f + jitter() * 50.f, jitter() * 50.f, 0.0f);
    }
    posm[n] = make_float4(1.0e3f, 1.0e3f, 1.0e3f, 0.0f);
    vel4[n] = make_float4(0.f, 0.f, 0.f, 0.f);

    const float tCut = R * R;
    std::vector<int> cnt(n, 0);
    std::vector<std::vector<int>> nbr(n);
    long long pairs = 0; int maxCnt = 0;
#pragma omp parallel for reduction(+ : pairs) reduction(max : maxCnt) schedule(dynamic, 4096)
    for (int i = 0; i < n; ++i) {
        const int c0 = scell[i], cz = c0 % gz, cy = (c0 / gz) % gy, cx = c0 / (gz * gy);
        std::vector<int>& my = nbr[i];
        for (int dx = -1; dx <= 1; ++dx) for (int dy = -1; dy <= 1; ++dy) for (int dz = -1; dz <= 1; ++dz) {
            const int X = cx + dx, Y = cy + dy, Z = cz + dz;
            if (X < 0 || X >= gx || Y < 0 || Y >= gy || Z < 0 || Z >= gz) continue;
            const int cc = (X * gy + Y) * gz + Z;
            for (int j = cs[cc]; j < cs[cc + 1]; ++j) {
                if (j == i) continue;
                const float ddx = posm[i].x - posm[j].x, ddy = posm[i].y - posm[j].y, ddz = posm[i].z - posm[j].z;
                if (ddx * ddx + ddy * ddy + ddz * ddz <= tCut) my.push_back(j);
            }
        }
        pairs += (long long)my.size();
        maxCnt = std::max(maxCnt, (int)my.size());
    }
    if (maxCnt > kCap) { printf("row capacity %d exceeded (%d)\n", kCap, maxCnt); return 1; }
    for (int i = 0; i < n; ++i) cnt[i] = (int)nbr[i].size();
    printf("pairs: %lld (%.1f per particle), max %d\n", pairs, (double)pairs / n, maxCnt);

    Consts c; c.twoOverR = 2.0f / R; c.gradScale = 1.0f / (kPi * R * R * R * R * R);

    float4 *dPos, *dVel; float* dOut;
    CK(hipMalloc(&dPos, sizeof(float4) * (n + 1))); CK(hipMalloc(&dVel, sizeof(float4) * (n + 1))); CK(hipMalloc(&dOut, sizeof(float) * (n + 64)));
    CK(hipMemcpy(dPos, posm.data(), sizeof(float4) * (n + 1), hipMemcpyHostToDevice));
    CK(hipMemcpy(dVel, vel4.data(), sizeof(float4) * (n + 1), hipMemcpyHostToDevice));
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    int numCUs = 256;
    { hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0)); numCUs = prop.multiProcessorCount; printf("device: %s, %d CUs\n", prop.name, numCUs); }

    std::vector<float> ref[2], got(n);
    auto run = [&](const char* name, int two, auto&& launch) {
        CK(hipMemsetAsync(dOut, 0, sizeof(float) * n, st));
        for (int w = 0; w < 2; ++w) launch();
        CK(hipEventRecord(e0, st));
        for (int r = 0; r < reps; ++r) launch();
        CK(hipEventRecord(e1, st));
        CK(hipEventSynchronize(e1));
        CK(hipGetLastError());
        float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
        ms /= reps;
        CK(hipMemcpy(got.data(), dOut, sizeof(float) * n, hipMemcpyDeviceToHost));
        std::vector<float>& rf = ref[two];
        char verdict[128] = "reference";
        if (rf.empty()) rf = got;
        else {
            double scaleV = 0, maxAbs = 0; long long bad = 0;
            for (int i = 0; i < n; ++i) scaleV = std::max(scaleV, (double)fabsf(rf[i]));
            for (int i = 0; i < n; ++i) {
                const double d = fabs((double)got[i] - rf[i]);
                maxAbs = std::max(maxAbs, d);
                if (!(d <= 1e-5 * scaleV)) ++bad;
            }
            snprintf(verdict, sizeof(verdict), "max |diff| / max|ref| = %.2e, %lld beyond 1e-5", maxAbs / std::max(scaleV, 1e-30), bad);
        }
        printf("%-46s %8.3f ms   %7.1f Gpair/s   alg %6.1f GB/s   [%s]\n", name, ms, pairs / ms * 1e-6, 44.0 * n / ms * 1e-6, verdict);
        fflush(stdout);
    };

    // ---- Q4 rows ----
    unsigned int* dRowsQ = nullptr; int* dStepsQ = nullptr; int numTilesQ = (n + 15) / 16; const int capSteps = kCap / 4;
    if (strchr(which, 'q')) {
        std::vector<unsigned int> rq((size_t)numTilesQ * capSteps * 64, (unsigned)n);
        std::vector<int> steps(numTilesQ, 0);
        for (int i = 0; i < n; ++i) {
            const int tile = i / 16, p = i % 16, m = cnt[i];
            steps[tile] = std::max(steps[tile], (m + 3) / 4);
            for (int t = 0; t < m; ++t) rq[(((size_t)tile * capSteps + t / 4) * 64 + p * 4 + (t % 4))] = (unsigned)nbr[i][t];
        }
        CK(hipMalloc(&dRowsQ, sizeof(unsigned int) * rq.size())); CK(hipMalloc(&dStepsQ, sizeof(int) * numTilesQ));
        CK(hipMemcpy(dRowsQ, rq.data(), sizeof(unsigned int) * rq.size(), hipMemcpyHostToDevice));
        CK(hipMemcpy(dStepsQ, steps.data(), sizeof(int) * numTilesQ, hipMemcpyHostToDevice));
        const unsigned gridQ = xcd_grid(numTilesQ * 64, 256);
        run("Q4 global quad walk tol 2f", 1, [&] { hipLaunchKernelGGL((k_q4<true>), dim3(gridQ), dim3(256), 0, st, c, dPos, dVel, dRowsQ, dStepsQ, dOut, n, numTilesQ, capSteps); });
        run("Q4 global quad walk tol 1f", 0, [&] { hipLaunchKernelGGL((k_q4<false>), dim3(gridQ), dim3(256), 0, st, c, dPos, dVel, dRowsQ, dStepsQ, dOut, n, numTilesQ, capSteps); });
    }

    // ---- bricks (4x4x4 cells) ----
    const int BX = 4, BY = 4, BZ = 4;
    const int nbx = (gx + BX - 1) / BX, nby = (gy + BY - 1) / BY, nbz = (gz + BZ - 1) / BZ;
    struct HostBrick { std::vector<int> own; std::vector<BrickRun> runs; int staged; };
    std::vector<HostBrick> HB;
    for (int bxi = 0; bxi < nbx; ++bxi) for (int byi = 0; byi < nby; ++byi) for (int bzi = 0; bzi < nbz; ++bzi) {
        const int x0 = bxi * BX, y0 = byi * BY, z0 = bzi * BZ;
        const int x1 = std::min(x0 + BX, gx), y1 = std::min(y0 + BY, gy), z1 = std::min(z0 + BZ, gz);
        HostBrick B; B.staged = 0;
        for (int X = x0; X < x1; ++X) for (int Y = y0; Y < y1; ++Y)
            for (int j = cs[(X * gy + Y) * gz + z0]; j < cs[(X * gy + Y) * gz + z1]; ++j) B.own.push_back(j);
        if (B.own.empty()) continue;
        const int zlo = std::max(z0 - 1, 0), zhi = std::min(z1, gz - 1);
        for (int X = std::max(x0 - 1, 0); X <= std::min(x1, gx - 1); ++X) for (int Y = std::max(y0 - 1, 0); Y <= std::min(y1, gy - 1); ++Y) {
            const int a = cs[(X * gy + Y) * gz + zlo], b = cs[(X * gy + Y) * gz + zhi + 1];
            if (b == a) continue;
            B.runs.push_back({a, b - a, B.staged});
            B.staged += b - a;
        }
        HB.push_back(std::move(B));
    }
    const int numBricks = (int)HB.size();
    int maxStaged = 0, maxOwn = 0, maxRuns = 0; long long stagedSum = 0;
    for (auto& B : HB) { maxStaged = std::max(maxStaged, B.staged); maxOwn = std::max(maxOwn, (int)B.own.size()); maxRuns = std::max(maxRuns, (int)B.runs.size()); stagedSum += B.staged; }
    printf("bricks 4x4x4: %d, own avg %.0f max %d, staged avg %.0f max %d (x%.2f), runs max %d\n", numBricks, (double)n / numBricks, maxOwn,
           (double)stagedSum / numBricks, maxStaged, (double)stagedSum / n, maxRuns);
    if (maxRuns > 64 || maxStaged + 2 > 65535) { printf("brick tables do not fit\n"); return 1; }
    std::vector<int> slotOf(n, -1);

    // ---- BRICK (r03 kernel), T = 512 ----
    if (strchr(which, 'b')) {
        const int T = 512;
        std::vector<BrickDesc> descs; std::vector<BrickRun> bruns; std::vector<int> ownIdx; std::vector<unsigned short> ownSlot;
        std::vector<unsigned char> wch((size_t)numBricks * 8 * 17, 0); std::vector<uint4> rowsFlat; long long rowChunkRows = 0;
        for (int blk = 0; blk < numBricks; ++blk) {
            const HostBrick& B = HB[blk];
            BrickDesc D; D.runFirst = (int)bruns.size(); D.numRuns = (int)B.runs.size(); D.staged = B.staged; D.own = (int)B.own.size();
            D.ownFirst = (int)ownIdx.size(); D.rounds = (D.own + T - 1) / T; D.rowBase = (int)rowChunkRows;
            if (D.rounds > 8) { printf("rounds\n"); return 1; }
            for (auto& r : B.runs) { bruns.push_back(r); for (int j = 0; j < r.len; ++j) slotOf[r.start + j] = r.base + j; }
            for (int j : B.own) { ownIdx.push_back(j); ownSlot.push_back((unsigned short)slotOf[j]); }
            for (int rd = 0; rd < D.rounds; ++rd) {
                int roundMax = 0;
                for (int t = 0; t < T; ++t) {
                    const int p = rd * T + t;
                    const int ch = p < D.own ? (cnt[B.own[p]] + 7) / 8 : 0;
                    unsigned char& wv = wch[(size_t)(blk * 8 + rd) * 17 + t / 64];
                    wv = (unsigned char)std::max<int>(wv, ch);
                    roundMax = std::max(roundMax, ch);
                }
                wch[(size_t)(blk * 8 + rd) * 17 + 16] = (unsigned char)roundMax;
                const size_t at = rowsFlat.size();
                rowsFlat.resize(at + (size_t)roundMax * T, make_uint4(0, 0, 0, 0));
                for (int t = 0; t < T; ++t) {
                    const int p = rd * T + t;
                    const int i = p < D.own ? B.own[p] : -1;
                    const int m = i >= 0 ? cnt[i] : 0;
                    for (int kk = 0; kk < roundMax * 8; ++kk) {
                        int slot = D.staged;
                        if (kk < m) slot = slotOf[nbr[i][kk]];
                        reinterpret_cast<unsigned short*>(&rowsFlat[at + (size_t)(kk / 8) * T + t])[kk % 8] = (unsigned short)slot;
                    }
                }
                rowChunkRows += roundMax;
            }
            for (auto& r : B.runs) for (int j = 0; j < r.len; ++j) slotOf[r.start + j] = -1;
            descs.push_back(D);
        }
        const int slots = maxStaged + 1;
        BrickDesc* dDesc; BrickRun* dRuns; int* dOwnIdx; unsigned short* dOwnSlot; uint4* dRowsB; unsigned char* dWch;
        CK(hipMalloc(&dDesc, sizeof(BrickDesc) * descs.size())); CK(hipMalloc(&dRuns, sizeof(BrickRun) * bruns.size()));
        CK(hipMalloc(&dOwnIdx, sizeof(int) * ownIdx.size())); CK(hipMalloc(&dOwnSlot, sizeof(unsigned short) * ownSlot.size()));
        CK(hipMalloc(&dRowsB, sizeof(uint4) * std::max<size_t>(rowsFlat.size(), 1))); CK(hipMalloc(&dWch, wch.size()));
        CK(hipMemcpy(dDesc, descs.data(), sizeof(BrickDesc) * descs.size(), hipMemcpyHostToDevice));
        CK(hipMemcpy(dRuns, bruns.data(), sizeof(BrickRun) * bruns.size(), hipMemcpyHostToDevice));
        CK(hipMemcpy(dOwnIdx, ownIdx.data(), sizeof(int) * ownIdx.size(), hipMemcpyHostToDevice));
        CK(hipMemcpy(dOwnSlot, ownSlot.data(), sizeof(unsigned short) * ownSlot.size(), hipMemcpyHostToDevice));
        CK(hipMemcpy(dRowsB, rowsFlat.data(), sizeof(uint4) * rowsFlat.size(), hipMemcpyHostToDevice));
        CK(hipMemcpy(dWch, wch.data(), wch.size(), hipMemcpyHostToDevice));
        const unsigned grid = (unsigned)(((numBricks + 7) / 8) * 8);
        for (int two = 1; two >= 0; --two) {
            const size_t ldsBytes = (size_t)slots * 16 * (two ? 2 : 1);
            char nm[96]; snprintf(nm, sizeof(nm), "BRICK r03 T=512 tol %s (%zu KB)", two ? "2f" : "1f", ldsBytes / 1024);
            if (two) { CK(hipFuncSetAttribute((const void*)k_brick<512, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsBytes));
                run(nm, two, [&] { hipLaunchKernelGGL((k_brick<512, true>), dim3(grid), dim3(512), ldsBytes, st, c, dPos, dVel, dDesc, dRuns, dOwnIdx, dOwnSlot, dRowsB, dWch, dOut, numBricks, slots); }); }
            else { CK(hipFuncSetAttribute((const void*)k_brick<512, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsBytes));
                run(nm, two, [&] { hipLaunchKernelGGL((k_brick<512, false>), dim3(grid), dim3(512), ldsBytes, st, c, dPos, dVel, dDesc, dRuns, dOwnIdx, dOwnSlot, dRowsB, dWch, dOut, numBricks, slots); }); }
        }

        {   // ---- PARK ----
            std::vector<int> stageBaseH(numBricks + 1, 0), srcIdxH;
            for (int blk = 0; blk < numBricks; ++blk) {
                stageBaseH[blk] = (int)srcIdxH.size();
                for (auto& r : HB[blk].runs) for (int j = 0; j < r.len; ++j) srcIdxH.push_back(r.start + j);
            }
            stageBaseH[numBricks] = (int)srcIdxH.size();
            srcIdxH.resize(srcIdxH.size() + 8192, 0);
            descs.push_back(BrickDesc{0, 0, 0, 0, 0, (int)rowChunkRows, 0});
            int *dStageBase, *dSrcIdx; int2* dRangesP; BrickDesc* dDescP; uint4* dRowsP; int* dWchP; float4* dOwnVel;
            { std::vector<float4> ov(ownIdx.size() + 1024); for (size_t q2 = 0; q2 < ownIdx.size(); ++q2) ov[q2] = vel4[ownIdx[q2]];
              CK(hipMalloc(&dOwnVel, sizeof(float4) * ov.size())); CK(hipMemcpy(dOwnVel, ov.data(), sizeof(float4) * ov.size(), hipMemcpyHostToDevice)); }
            std::vector<uint4> rowsPad(rowsFlat); rowsPad.resize(rowsPad.size() + (size_t)8 * T, make_uint4(0, 0, 0, 0));
            std::vector<int> wchPad(wch.begin(), wch.end()); wchPad.resize(wchPad.size() + 8 * 17, 0);      // (int, not byte: hipcc converts a byte right behind its load, i.e. waits for it)
            CK(hipMalloc(&dStageBase, 4 * stageBaseH.size())); CK(hipMalloc(&dSrcIdx, 4 * srcIdxH.size())); CK(hipMalloc(&dDescP, sizeof(BrickDesc) * descs.size()));
            CK(hipMalloc(&dRowsP, sizeof(uint4) * rowsPad.size())); CK(hipMalloc(&dWchP, 4 * wchPad.size()));
            CK(hipMemcpy(dStageBase, stageBaseH.data(), 4 * stageBaseH.size(), hipMemcpyHostToDevice));
            CK(hipMemcpy(dSrcIdx, srcIdxH.data(), 4 * srcIdxH.size(), hipMemcpyHostToDevice));
            CK(hipMemcpy(dDescP, descs.data(), sizeof(BrickDesc) * descs.size(), hipMemcpyHostToDevice));
            CK(hipMemcpy(dRowsP, rowsPad.data(), sizeof(uint4) * rowsPad.size(), hipMemcpyHostToDevice));
            CK(hipMemcpy(dWchP, wchPad.data(), 4 * wchPad.size(), hipMemcpyHostToDevice));
            auto ranges = [&](int nblk) {
                std::vector<int2> rg(nblk); long long tot = 0; for (auto& Bk : HB) tot += (long long)Bk.own.size();
                int kk = 0; long long acc = 0;
                for (int b = 0; b < nblk; ++b) {
                    const long long target = tot * (b + 1) / nblk; const int first = kk;
                    while (kk < numBricks && acc + (long long)HB[kk].own.size() <= target) { acc += (long long)HB[kk].own.size(); ++kk; }
                    if (b == nblk - 1) kk = numBricks;
                    rg[b] = make_int2(first, kk);
                }
                CK(hipMalloc(&dRangesP, sizeof(int2) * nblk));
                CK(hipMemcpy(dRangesP, rg.data(), sizeof(int2) * nblk, hipMemcpyHostToDevice));
            };
#define PARK(TW, NSS, NRR, BPC, MODE)                                                                                                 \
    do {                                                                                                                        \
        const size_t ldsBytes = (size_t)slots * 16 * ((TW) ? 2 : 1);                                                            \
        const int nblk = numCUs * (BPC);                                                                                        \
        ranges(nblk);                                                                                                           \
        char nm[96]; snprintf(nm, sizeof(nm), "PARK T=512 NS=%d NR=%d x%d/CU tol %s%s (%zu KB)", NSS, NRR, BPC, (TW) ? "2f" : "1f", MODE == 0 ? "" : (MODE == 1 ? " CONFLICT-FREE" : (MODE == 2 ? " NO-ARITH" : " NO-LDS")), ldsBytes / 1024); \
        CK(hipFuncSetAttribute((const void*)k_park<512, TW, NSS, NRR, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsBytes)); \
        run(nm, (TW) ? 1 : 0, [&] { hipLaunchKernelGGL((k_park<512, TW, NSS, NRR, MODE>), dim3(nblk), dim3(512), ldsBytes, st, c, dPos, dVel, dDescP, dStageBase, dSrcIdx, dOwnIdx, dOwnSlot, dOwnVel, dRowsP, dWchP, dRangesP, dOut, slots); }); \
        CK(hipFree(dRangesP));                                                                                                  \
    } while (0)
            PARK(true, 4, 2, 2, 0);
            PARK(true, 4, 2, 2, 1);
            PARK(true, 4, 2, 2, 2);
            PARK(true, 4, 2, 2, 3);
            PARK(false, 4, 4, 2, 0);
            PARK(false, 4, 4, 2, 1);
            PARK(false, 4, 4, 2, 2);
            PARK(false, 4, 4, 2, 3);
#undef PARK
            CK(hipFree(dStageBase)); CK(hipFree(dSrcIdx)); CK(hipFree(dDescP)); CK(hipFree(dRowsP)); CK(hipFree(dWchP));
        }
        CK(hipFree(dDesc)); CK(hipFree(dRuns)); CK(hipFree(dOwnIdx)); CK(hipFree(dOwnSlot)); CK(hipFree(dRowsB)); CK(hipFree(dWch));
    }

    // ---- FLOW ----
    if (strchr(which, 'f')) {
        std::vector<BDesc> descs(numBricks + 1); std::vector<BrickRun> bruns; std::vector<OwnRec> ownRec; std::vector<unsigned char> grpBlocks;
        std::vector<unsigned int> rows;
        int G = 0; double blockSum = 0;
        for (int blk = 0; blk < numBricks; ++blk) {
            const HostBrick& B = HB[blk];
            BDesc D; D.runFirst = (int)bruns.size(); D.numRuns = (int)B.runs.size(); D.staged = B.staged; D.groupBase = G;
            for (auto& r : B.runs) { bruns.push_back({r.start, r.len, r.base + 1}); for (int j = 0; j < r.len; ++j) slotOf[r.start + j] = r.base + 1 + j; }   // slot 0 = dummy
            const int own = (int)B.own.size(), groups = (own + 15) / 16;
            ownRec.resize((size_t)(G + groups) * 16); grpBlocks.resize(G + groups); rows.resize((size_t)(G + groups) * kCapB * 64, 0u);
            for (int g = 0; g < groups; ++g) {
                int nbk = 0;
                for (int qq = 0; qq < 16; ++qq) {
                    const int p = g * 16 + qq;
                    OwnRec& o = ownRec[(size_t)(G + g) * 16 + qq];
                    if (p >= own) { o.index = -1; o.slotCnt = 0u; continue; }
                    const int i = B.own[p], m = cnt[i];
                    o.index = i; o.slotCnt = (unsigned)slotOf[i] | ((unsigned)m << 16);
                    nbk = std::max(nbk, (m + 7) / 8);
                    for (int t = 0; t < m; ++t) {
                        const int b = t / 8, h = (t % 8) / 4, g4 = t % 4;
                        unsigned int& wv = rows[(((size_t)(G + g) * 2 + b / 4) * 64 + qq * 4 + g4) * 4 + (b % 4)];
                        wv |= (unsigned)slotOf[nbr[i][t]] << (h ? 16 : 0);
                    }
                }
                grpBlocks[G + g] = (unsigned char)nbk;
                blockSum += nbk;
            }
            for (auto& r : B.runs) for (int j = 0; j < r.len; ++j) slotOf[r.start + j] = -1;
            descs[blk] = D;
            G += groups;
        }
        descs[numBricks] = BDesc{(int)bruns.size(), 0, 0, G};
        const int Gtot = G;
        printf("FLOW: %d groups, row blocks avg %.2f per group (padding x%.2f of pairs)\n", Gtot, blockSum / Gtot, blockSum * 16 * 8 / pairs);
        const int slotsCap = maxStaged + 1;
        BDesc* dDesc; BrickRun* dRuns; OwnRec* dOwn; unsigned char* dGB; unsigned int* dRows; int2* dRanges;
        CK(hipMalloc(&dDesc, sizeof(BDesc) * descs.size())); CK(hipMalloc(&dRuns, sizeof(BrickRun) * bruns.size()));
        CK(hipMalloc(&dOwn, sizeof(OwnRec) * ownRec.size())); CK(hipMalloc(&dGB, grpBlocks.size())); CK(hipMalloc(&dRows, 4 * rows.size()));
        CK(hipMemcpy(dDesc, descs.data(), sizeof(BDesc) * descs.size(), hipMemcpyHostToDevice));
        CK(hipMemcpy(dRuns, bruns.data(), sizeof(BrickRun) * bruns.size(), hipMemcpyHostToDevice));
        CK(hipMemcpy(dOwn, ownRec.data(), sizeof(OwnRec) * ownRec.size(), hipMemcpyHostToDevice));
        CK(hipMemcpy(dGB, grpBlocks.data(), grpBlocks.size(), hipMemcpyHostToDevice));
        CK(hipMemcpy(dRows, rows.data(), 4 * rows.size(), hipMemcpyHostToDevice));
        int* dFault; CK(hipMalloc(&dFault, 4)); CK(hipMemset(dFault, 0, 4));
        const int maxBlocks = numCUs * 2;
        CK(hipMalloc(&dRanges, sizeof(int2) * maxBlocks));
        auto setRanges = [&](int nblk) {
            std::vector<int2> rg(nblk);
            int kk = 0;
            for (int b = 0; b < nblk; ++b) {
                const long long target = (long long)Gtot * (b + 1) / nblk;
                const int first = kk;
                while (kk < numBricks && descs[kk + 1].groupBase <= target) ++kk;
                if (b == nblk - 1) kk = numBricks;
                rg[b] = make_int2(first, kk);
            }
            CK(hipMemcpy(dRanges, rg.data(), sizeof(int2) * nblk, hipMemcpyHostToDevice));
        };
#define FLOW(TT, NB, NLD, TW, UBB, BPC, MODE)                                                                                         \
    do {                                                                                                                        \
        const size_t ldsBytes = (size_t)(NB) * ((TW) ? 2 : 1) * slotsCap * 16 + (size_t)(NLD) * slotsCap * 4;                   \
        const int nblk = numCUs * (BPC);                                                                                        \
        char nm[96]; snprintf(nm, sizeof(nm), "FLOW T=%d NBUF=%d NL=%d UB=%d x%d/CU %s%s (%zu KB)", TT, NB, NLD, UBB, BPC, (TW) ? "2f" : "1f", MODE == 0 ? "" : (MODE == 1 ? " NO-DMA" : (MODE == 2 ? " NO-COMPUTE" : (MODE == 3 ? " NEITHER" : " LOADER-ALONE"))), ldsBytes / 1024); \
        if (ldsBytes + 64 > 160 * 1024 / (BPC)) { printf("%s: does not fit\n", nm); break; }                                    \
        setRanges(nblk);                                                                                                        \
        CK(hipFuncSetAttribute((const void*)k_flow<TT, NB, NLD, TW, UBB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsBytes)); \
        run(nm, (TW) ? 1 : 0, [&] { hipLaunchKernelGGL((k_flow<TT, NB, NLD, TW, UBB>), dim3(nblk), dim3(TT), ldsBytes, st, c, dPos, dVel, dDesc, dRuns, dOwn, dGB, dRows, dRanges, dOut, slotsCap, n, dFault, MODE); }); \
        { int hf = 0; CK(hipMemcpy(&hf, dFault, 4, hipMemcpyDeviceToHost)); if (hf) { printf("   !! %d spin limits hit (results invalid)\n", hf); CK(hipMemset(dFault, 0, 4)); } } \
    } while (0)
        FLOW(1024, 2, 1, true, 2, 1, 0);
        FLOW(1024, 2, 2, true, 2, 1, 0);
        FLOW(1024, 2, 1, false, 2, 1, 0);
        FLOW(512, 2, 1, false, 2, 2, 0);
#undef FLOW

        {   // ---- SELF: staging items (runs cut into pieces of up to 64 records), dealt to the first WC groups of the brick before ----
            struct SelfSet { int WC; BDesc* dSd; int4 *dItems, *dFirst; GrpHdr* dHdr; };
            auto buildSelf = [&](int WC) {
                std::vector<BDesc> sd(numBricks + 1); std::vector<int4> runItems;              // items in run order (a block's first brick)
                for (int blk = 0; blk < numBricks; ++blk) {
                    sd[blk] = descs[blk]; sd[blk].runFirst = (int)runItems.size();
                    for (int r = 0; r < descs[blk].numRuns; ++r) {
                        const BrickRun& Rn = bruns[descs[blk].runFirst + r];
                        for (int o = 0; o < Rn.len; o += 64) runItems.push_back(make_int4(Rn.start + o, std::min(64, Rn.len - o), Rn.base + o, 0));
                    }
                    sd[blk].numRuns = (int)runItems.size() - sd[blk].runFirst;
                }
                sd[numBricks] = BDesc{(int)runItems.size(), 0, 0, Gtot};
                std::vector<GrpHdr> hdr(Gtot); std::vector<int4> dealt; dealt.reserve(runItems.size());
                long long cum[2] = {0, 0};
                for (int blk = 0; blk < numBricks; ++blk) {
                    const int g0 = sd[blk].groupBase, groups = sd[blk + 1].groupBase - g0;
                    cum[blk & 1] += sd[blk].numRuns;
                    const int C = std::min(WC, groups), nNext = blk + 1 < numBricks ? sd[blk + 1].numRuns : 0;
                    for (int g = 0; g < groups; ++g) {
                        GrpHdr& H = hdr[g0 + g];
                        int cnt = 0; const int first = (int)dealt.size();
                        if (g < C) for (int i = g; i < nNext; i += C) { dealt.push_back(runItems[sd[blk + 1].runFirst + i]); ++cnt; }
                        if (cnt > 63) { printf("too many items for one group\n"); exit(1); }
                        H.nbItems = (int)grpBlocks[g0 + g] | (cnt << 8); H.itemFirst = first; H.brick = blk; H.cum = (int)cum[blk & 1];
                    }
                }
                dealt.resize(dealt.size() + 64, make_int4(0, 0, 0, 0));
                SelfSet S; S.WC = WC;
                CK(hipMalloc(&S.dSd, sizeof(BDesc) * sd.size())); CK(hipMalloc(&S.dItems, sizeof(int4) * dealt.size())); CK(hipMalloc(&S.dFirst, sizeof(int4) * runItems.size())); CK(hipMalloc(&S.dHdr, sizeof(GrpHdr) * hdr.size()));
                CK(hipMemcpy(S.dSd, sd.data(), sizeof(BDesc) * sd.size(), hipMemcpyHostToDevice));
                CK(hipMemcpy(S.dItems, dealt.data(), sizeof(int4) * dealt.size(), hipMemcpyHostToDevice));
                CK(hipMemcpy(S.dFirst, runItems.data(), sizeof(int4) * runItems.size(), hipMemcpyHostToDevice));
                CK(hipMemcpy(S.dHdr, hdr.data(), sizeof(GrpHdr) * hdr.size(), hipMemcpyHostToDevice));
                printf("SELF WC=%d: %.1f staging items per brick\n", WC, (double)runItems.size() / numBricks);
                return S;
            };
            SelfSet sets16 = buildSelf(16), sets12 = buildSelf(12), sets8 = buildSelf(8);
            auto pick = [&](int TT) -> SelfSet& { return TT == 1024 ? sets16 : (TT == 768 ? sets12 : sets8); };
#define SELF(TT, TW, MI, BPC, MODE)                                                                                              \
    do {                                                                                                                        \
        const size_t ldsBytes = (size_t)2 * ((TW) ? 2 : 1) * slotsCap * 16;                                                     \
        const int nblk = numCUs * (BPC);                                                                                        \
        char nm[96]; snprintf(nm, sizeof(nm), "SELF T=%d MAXI=%d x%d/CU %s%s (%zu KB)", TT, MI, BPC, (TW) ? "2f" : "1f", MODE == 0 ? "" : (MODE == 1 ? " NO-STAGE-LOADS" : " NO-COMPUTE"), ldsBytes / 1024); \
        if (ldsBytes + 256 > 160 * 1024 / (BPC)) { printf("%s: does not fit\n", nm); break; }                                   \
        setRanges(nblk);                                                                                                        \
        CK(hipFuncSetAttribute((const void*)k_self<TT, TW, MI>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsBytes));    \
        { SelfSet& SS = pick(TT); run(nm, (TW) ? 1 : 0, [&] { hipLaunchKernelGGL((k_self<TT, TW, MI>), dim3(nblk), dim3(TT), ldsBytes, st, c, dPos, dVel, SS.dSd, SS.dItems, SS.dFirst, dOwn, SS.dHdr, dRows, dRanges, dOut, slotsCap, n, dFault, MODE); }); } \
        { int hf = 0; CK(hipMemcpy(&hf, dFault, 4, hipMemcpyDeviceToHost)); if (hf) { printf("   !! %d spin limits hit (results invalid)\n", hf); CK(hipMemset(dFault, 0, 4)); } } \
    } while (0)
            SELF(1024, true, 3, 1, 0);
            SELF(1024, true, 3, 1, 1);
            SELF(1024, true, 3, 1, 2);
            SELF(1024, true, 2, 1, 0);
            SELF(768, true, 3, 1, 0);
            SELF(512, true, 4, 1, 0);
            SELF(1024, false, 3, 1, 0);
            SELF(1024, false, 3, 1, 2);
            SELF(512, false, 3, 2, 0);
            SELF(512, false, 4, 2, 0);
            {
                const size_t ldsBytes = (size_t)2 * slotsCap * 16;
                setRanges(numCUs * 2);
                CK(hipFuncSetAttribute((const void*)k_self<1024, false, 2, false, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsBytes));
                CK(hipFuncSetAttribute((const void*)k_self<768, false, 3, false, 6>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsBytes));
                CK(hipFuncSetAttribute((const void*)k_self<768, false, 2, false, 6>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsBytes));
                run("SELF T=1024 MAXI=2 x2/CU 1f (8 waves/SIMD)", 0, [&] { hipLaunchKernelGGL((k_self<1024, false, 2, false, 8>), dim3(numCUs * 2), dim3(1024), ldsBytes, st, c, dPos, dVel, sets16.dSd, sets16.dItems, sets16.dFirst, dOwn, sets16.dHdr, dRows, dRanges, dOut, slotsCap, n, dFault, 0, nullptr); });
                run("SELF T=768 MAXI=3 x2/CU 1f (6 waves/SIMD)", 0, [&] { hipLaunchKernelGGL((k_self<768, false, 3, false, 6>), dim3(numCUs * 2), dim3(768), ldsBytes, st, c, dPos, dVel, sets12.dSd, sets12.dItems, sets12.dFirst, dOwn, sets12.dHdr, dRows, dRanges, dOut, slotsCap, n, dFault, 0, nullptr); });
                run("SELF T=768 MAXI=2 x2/CU 1f (6 waves/SIMD)", 0, [&] { hipLaunchKernelGGL((k_self<768, false, 2, false, 6>), dim3(numCUs * 2), dim3(768), ldsBytes, st, c, dPos, dVel, sets12.dSd, sets12.dItems, sets12.dFirst, dOwn, sets12.dHdr, dRows, dRanges, dOut, slotsCap, n, dFault, 0, nullptr); });
                { int hf = 0; CK(hipMemcpy(&hf, dFault, 4, hipMemcpyDeviceToHost)); if (hf) { printf("   !! %d spin limits hit (results invalid)\n", hf); CK(hipMemset(dFault, 0, 4)); } }
            }
#undef SELF
            {
                unsigned long long* dProf; CK(hipMalloc(&dProf, 16 * 8));
                const size_t lb2 = (size_t)2 * 2 * slotsCap * 16;
                setRanges(numCUs);
                CK(hipFuncSetAttribute((const void*)k_self<1024, true, 3, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lb2));
                for (int md : {0, 1, 2, 8, 9, 10}) {
                    CK(hipMemset(dProf, 0, 128));
                    hipLaunchKernelGGL((k_self<1024, true, 3, true>), dim3(numCUs), dim3(1024), lb2, st, c, dPos, dVel, sets16.dSd, sets16.dItems, sets16.dFirst, dOwn, sets16.dHdr, dRows, dRanges, dOut, slotsCap, n, dFault, md, dProf);
                    CK(hipDeviceSynchronize());
                    unsigned long long h[16]; CK(hipMemcpy(h, dProf, 128, hipMemcpyDeviceToHost));
                    const double it = std::max(1.0, (double)h[4]);
                    printf("PROF SELF T=1024 2f mode %2d  cycles per group-iteration: top %.0f  ready-wait %.0f  comp %.0f  stage(left-wait + vm + writes) %.0f   (%.0f%% of iterations carry items)\n", md,
                           h[0] / it, h[1] / it, h[2] / it, h[3] / it, 100.0 * h[5] / it);
                }
                CK(hipMemset(dFault, 0, 4));
            }
        }
        {   // where a consumer wave's time goes (s_memtime stamps; 100 MHz ticks): top = store + prefetch issue, poll = cursor + ready wait,
            // comp = LDS reads + arithmetic, vm = wait for the prefetched words
            unsigned long long* dProf; CK(hipMalloc(&dProf, 16 * 8));
            auto prof = [&](const char* nm, auto&& launch) {
                CK(hipMemset(dProf, 0, 128)); launch(); CK(hipDeviceSynchronize());
                unsigned long long h[16]; CK(hipMemcpy(h, dProf, 128, hipMemcpyDeviceToHost));
                const double it = std::max(1.0, (double)h[4]), bk = std::max(1.0, (double)h[12]);
                printf("PROF %-26s cycles per group-iteration: top %.0f  poll %.0f  comp %.0f  vm %.0f | loader per brick: wait %.0f  issue %.0f  table+map %.0f  drain %.0f\n", nm,
                       h[0] / it, h[1] / it, h[2] / it, h[3] / it, h[8] / bk, h[9] / bk, h[10] / bk, h[11] / bk);
            };
            const size_t lb2 = (size_t)2 * 2 * slotsCap * 16 + (size_t)2 * slotsCap * 4, lb1 = (size_t)2 * 1 * slotsCap * 16 + (size_t)2 * slotsCap * 4;
            setRanges(numCUs);
            CK(hipFuncSetAttribute((const void*)k_flow<1024, 2, 1, true, 2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lb2));
            CK(hipFuncSetAttribute((const void*)k_flow<1024, 2, 2, true, 2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lb2));
            CK(hipFuncSetAttribute((const void*)k_flow<1024, 2, 1, false, 2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lb1));
            for (int md : {0, 1, 2, 4}) {
                char nm[64]; snprintf(nm, sizeof(nm), "T=1024 NL=1 2f mode %d", md);
                prof(nm, [&] { hipLaunchKernelGGL((k_flow<1024, 2, 1, true, 2, true>), dim3(numCUs), dim3(1024), lb2, st, c, dPos, dVel, dDesc, dRuns, dOwn, dGB, dRows, dRanges, dOut, slotsCap, n, dFault, md, dProf); });
            }
            prof("T=1024 NL=2 2f mode 0", [&] { hipLaunchKernelGGL((k_flow<1024, 2, 2, true, 2, true>), dim3(numCUs), dim3(1024), lb2, st, c, dPos, dVel, dDesc, dRuns, dOwn, dGB, dRows, dRanges, dOut, slotsCap, n, dFault, 0, dProf); });
            prof("T=1024 NL=1 1f mode 0", [&] { hipLaunchKernelGGL((k_flow<1024, 2, 1, false, 2, true>), dim3(numCUs), dim3(1024), lb1, st, c, dPos, dVel, dDesc, dRuns, dOwn, dGB, dRows, dRanges, dOut, slotsCap, n, dFault, 0, dProf); });
            prof("T=1024 NL=1 1f mode 1", [&] { hipLaunchKernelGGL((k_flow<1024, 2, 1, false, 2, true>), dim3(numCUs), dim3(1024), lb1, st, c, dPos, dVel, dDesc, dRuns, dOwn, dGB, dRows, dRanges, dOut, slotsCap, n, dFault, 1, dProf); });
        }
    }
    return 0;
}
