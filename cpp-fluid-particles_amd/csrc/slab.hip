// slab.hip — the native multi-GPU layer (include/sphx_slab.h): SPHSystem::step() over x-slabs, one process per
// GPU, halo exchange by RCCL point-to-point messages, or all slabs in one process (loopback) for testing.
//
// The decomposition follows from the reference's cell id (x slowest, CUDAFunctions.cuh:64-70) and cell length >=
// support radius (main.cpp:56-57): one ghost cell column per side suffices (two for PBD, whose sweeps run on
// positions that moved after binning, PBDSolver.cu:225-258).  See sphx_slab.h for the invariants that make the
// result bit-identical to the single-device run.
#include <dlfcn.h>

#include <algorithm>
#include <chrono>
#include <climits>
#include <cmath>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include <rccl/rccl.h>          // types and enums only: the library is opened at run time (sphx works without it)

#include "capi_internal.hpp"
#include "engine.hpp"
#include "sphx_slab.h"

using namespace sphx;

namespace {

// ================================================================================ RCCL, opened lazily
struct RcclApi {
    void* lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;          // optional: what the bench line reports about the communicator
    ncclResult_t (*CommUserRank)(const ncclComm_t, int*) = nullptr;

    bool load(std::string& why)
    {
        if (lib) return true;
        // SPHX_RCCL_LIBRARY: a specific RCCL build (or the test suite's stand-in, tests/mock_rccl.cpp)
        const char* chosen = std::getenv("SPHX_RCCL_LIBRARY");
        if (chosen && *chosen) lib = dlopen(chosen, RTLD_NOW | RTLD_LOCAL);
        else
            for (const char* name : {"librccl.so.1", "librccl.so"}) {
                lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
                if (lib) break;
            }
        if (!lib) { why = std::string("cannot open librccl: ") + dlerror(); return false; }
        auto sym = [&](const char* n) { return dlsym(lib, n); };
        GetUniqueId = (decltype(GetUniqueId))sym("ncclGetUniqueId");
        CommInitRank = (decltype(CommInitRank))sym("ncclCommInitRank");
        CommDestroy = (decltype(CommDestroy))sym("ncclCommDestroy");
        Send = (decltype(Send))sym("ncclSend");
        Recv = (decltype(Recv))sym("ncclRecv");
        AllReduce = (decltype(AllReduce))sym("ncclAllReduce");
        GroupStart = (decltype(GroupStart))sym("ncclGroupStart");
        GroupEnd = (decltype(GroupEnd))sym("ncclGroupEnd");
        GetErrorString = (decltype(GetErrorString))sym("ncclGetErrorString");
        CommCount = (decltype(CommCount))sym("ncclCommCount");
        CommUserRank = (decltype(CommUserRank))sym("ncclCommUserRank");
        if (!GetUniqueId || !CommInitRank || !CommDestroy || !Send || !Recv || !AllReduce || !GroupStart || !GroupEnd) {
            why = "librccl lacks an expected symbol";
            return false;
        }
        return true;
    }
};
RcclApi g_rccl;

struct SlabError { std::string text; };
[[noreturn]] void die(const std::string& s) { throw SlabError{s}; }
void hip_ok(hipError_t e, const char* what) { if (e != hipSuccess) die(std::string(what) + ": " + hipGetErrorString(e)); }
void nccl_ok(ncclResult_t r, const char* what)
{
    if (r != ncclSuccess) die(std::string(what) + ": " + (g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "rccl error"));
}

// ================================================================================ transports
struct Msg { int from, to; void* buf; size_t bytes; };     // global slab ranks; device buffers

struct Transport {
    virtual ~Transport() {}
    // Executes the posted messages ordered after everything enqueued so far on the engine stream.  With
    // async = true the transfer may run beside later engine work; wait() orders later engine work after it.
    virtual void exchange(std::vector<Msg>& sends, std::vector<Msg>& recvs, bool async) = 0;
    virtual void wait() = 0;
    virtual long long allreduce_sum(long long localSum) = 0;   // blocking; every process calls it
    // the same for a device word, ordered on the engine stream like an exchange and WITHOUT a host wait: *dOut = sum over the
    // processes of *dIn once the engine stream gets there (the failure word of a step travels with its size messages)
    virtual void allreduce_device(const long long* dIn, long long* dOut) = 0;
    // what the bench line reports: 0 loopback / 1 RCCL, ranks and own rank of the communicator as the LIBRARY reports them
    // (-1: the library has no ncclCommCount), payload bytes posted so far, exchanges (grouped send/recv rounds) and all-reduces
    virtual void describe(int& kind, int& ranks, int& rank) const { kind = 0; ranks = 1; rank = 0; }
    long long bytesSent = 0, bytesReceived = 0, exchanges = 0, allreduces = 0;
    void account(const std::vector<Msg>& sends, const std::vector<Msg>& recvs)
    {
        for (const Msg& m : sends) bytesSent += (long long)m.bytes;
        for (const Msg& m : recvs) bytesReceived += (long long)m.bytes;
        ++exchanges;
    }
};

// all slabs in this process: a message is one device-to-device copy on the engine stream
struct LoopbackTransport final : Transport {
    void exchange(std::vector<Msg>& sends, std::vector<Msg>& recvs, bool) override
    {
        account(sends, recvs);
        std::vector<char> used(sends.size(), 0);
        for (const Msg& r : recvs) {
            size_t k = 0;
            while (k < sends.size() && (used[k] || sends[k].from != r.from || sends[k].to != r.to)) ++k;
            if (k == sends.size()) die("loopback: a receive has no matching send");
            if (sends[k].bytes != r.bytes) die("loopback: message size mismatch between neighbours");
            used[k] = 1;
            if (r.bytes) hip_ok(hipMemcpyAsync(r.buf, sends[k].buf, r.bytes, hipMemcpyDeviceToDevice, sphx::stream()), "loopback copy");
        }
        for (size_t k = 0; k < sends.size(); ++k) if (!used[k]) die("loopback: a send has no matching receive");
        sends.clear(); recvs.clear();
    }
    void wait() override {}
    long long allreduce_sum(long long v) override { return v; }
    void allreduce_device(const long long* dIn, long long* dOut) override
    {
        hip_ok(hipMemcpyAsync(dOut, dIn, sizeof(long long), hipMemcpyDeviceToDevice, sphx::stream()), "loopback all-reduce");
    }
};

// one process per GPU: grouped ncclSend / ncclRecv on a communication stream of its own.  A process may drive several
// CONSECUTIVE slabs (slabsPerProcess, the same number everywhere): slab r lives in process r / slabsPerProcess, and a
// message between two slabs of one process is an RCCL send to self.  RCCL matches the sends and receives of a pair of
// processes in the order they were posted, so both sides post theirs sorted by (from, to).
struct RcclTransport final : Transport {
    int slabsPerProcess = 1;
    ncclComm_t comm = nullptr;
    hipStream_t commStream = nullptr;
    hipEvent_t ready = nullptr, done = nullptr;
    bool pending = false, pendingAsync = false;
    long long* dScalar = nullptr;       // 2 x int64 on the device for the all-reduce
    long long* hScalar = nullptr;       // pinned

    RcclTransport(int rank, int world, const char* id128, int slabsPerProcess_) : slabsPerProcess(slabsPerProcess_)
    {
        std::string why;
        if (!g_rccl.load(why)) die(why);
        ncclUniqueId id;
        static_assert(sizeof(id) == 128, "ncclUniqueId is 128 bytes");
        std::memcpy(&id, id128, 128);
        nccl_ok(g_rccl.CommInitRank(&comm, world, id, rank), "ncclCommInitRank");
        try {       // a constructor that throws runs no destructor: give back what exists before passing the error on
            // the transfers run beside the interior sweep of the same stage, which fills every CU: the communication stream gets
            // the highest priority so that RCCL's copy kernels are dispatched as soon as wave slots free up instead of behind
            // the sweep's remaining workgroups (sphx_tuning.slab_comm_priority: 1 a default-priority stream, 2 the least priority; for measurements)
            int least = 0, greatest = 0;
            hip_ok(hipDeviceGetStreamPriorityRange(&least, &greatest), "stream priority range");
            const int pr = sphx::tuning().slab_comm_priority;      // 0 highest (default), 1 a flags-created stream: default priority, 2 the least the device has
            if (pr == 1) hip_ok(hipStreamCreateWithFlags(&commStream, hipStreamNonBlocking), "comm stream");
            else hip_ok(hipStreamCreateWithPriority(&commStream, hipStreamNonBlocking, pr == 2 ? least : greatest), "comm stream");
            hip_ok(hipEventCreateWithFlags(&ready, hipEventDisableTiming), "event");
            hip_ok(hipEventCreateWithFlags(&done, hipEventDisableTiming), "event");
            hip_ok(hipMalloc((void**)&dScalar, 2 * sizeof(long long)), "scalar buffer");
            hip_ok(hipHostMalloc((void**)&hScalar, 2 * sizeof(long long), hipHostMallocDefault), "pinned scalar");
        } catch (...) {
            release();
            throw;
        }
    }
    ~RcclTransport() override { release(); }
    void release()
    {
        if (commStream) (void)hipStreamSynchronize(commStream);
        if (comm) (void)g_rccl.CommDestroy(comm);
        if (dScalar) (void)hipFree(dScalar);
        if (hScalar) (void)hipHostFree(hScalar);
        if (ready) (void)hipEventDestroy(ready);
        if (done) (void)hipEventDestroy(done);
        if (commStream) (void)hipStreamDestroy(commStream);
        comm = nullptr; dScalar = nullptr; hScalar = nullptr; ready = done = nullptr; commStream = nullptr;
    }
    void describe(int& kind, int& ranks, int& rank) const override
    {
        kind = 1; ranks = -1; rank = -1;
        if (g_rccl.CommCount && g_rccl.CommCount(comm, &ranks) != ncclSuccess) ranks = -1;
        if (g_rccl.CommUserRank && g_rccl.CommUserRank(comm, &rank) != ncclSuccess) rank = -1;
    }
    void exchange(std::vector<Msg>& sends, std::vector<Msg>& recvs, bool async) override
    {
        account(sends, recvs);
        if (pending) wait();
        hip_ok(hipEventRecord(ready, sphx::stream()), "event record");
        hip_ok(hipStreamWaitEvent(commStream, ready, 0), "stream wait");
        if (slabsPerProcess > 1) {
            auto byPair = [](const Msg& a, const Msg& b) { return a.from != b.from ? a.from < b.from : a.to < b.to; };
            std::stable_sort(sends.begin(), sends.end(), byPair);
            std::stable_sort(recvs.begin(), recvs.end(), byPair);
        }
        nccl_ok(g_rccl.GroupStart(), "ncclGroupStart");
        for (const Msg& m : sends) if (m.bytes) nccl_ok(g_rccl.Send(m.buf, m.bytes, ncclInt8, m.to / slabsPerProcess, comm, commStream), "ncclSend");
        for (const Msg& m : recvs) if (m.bytes) nccl_ok(g_rccl.Recv(m.buf, m.bytes, ncclInt8, m.from / slabsPerProcess, comm, commStream), "ncclRecv");
        nccl_ok(g_rccl.GroupEnd(), "ncclGroupEnd");
        hip_ok(hipEventRecord(done, commStream), "event record");
        pending = true; pendingAsync = async;
        sends.clear(); recvs.clear();
        if (!async) wait();
    }
    void wait() override
    {
        if (!pending) return;
        // fault injection for the tests ("skipwait"): the engine stream is NOT ordered after an overlapped halo transfer.
        // With a transport that really completes late (tests/mock_rccl.cpp, deferred mode) results must then be wrong;
        // tests/test_gpu_slab.py asserts that, which proves the several-ranks tests can see a missing wait().
        // Compiled only into the TEST build of the library (-DSPHX_TEST_HOOKS, tests/libsphx_hooks.so); the product has no such hook.
#ifdef SPHX_TEST_HOOKS
        static const bool skip = [] { const char* f = std::getenv("SPHX_SLAB_FAULT"); return f && std::strcmp(f, "skipwait") == 0; }();
        if (skip && pendingAsync) { pending = false; return; }
#endif
        hip_ok(hipStreamWaitEvent(sphx::stream(), done, 0), "stream wait");
        pending = false;
    }
    long long allreduce_sum(long long v) override
    {
        ++allreduces;
        wait();
        hScalar[0] = v;
        hip_ok(hipMemcpyAsync(dScalar, hScalar, sizeof(long long), hipMemcpyHostToDevice, sphx::stream()), "scalar upload");
        hip_ok(hipEventRecord(ready, sphx::stream()), "event record");
        hip_ok(hipStreamWaitEvent(commStream, ready, 0), "stream wait");
        nccl_ok(g_rccl.AllReduce(dScalar, dScalar + 1, 1, ncclInt64, ncclSum, comm, commStream), "ncclAllReduce");
        hip_ok(hipMemcpyAsync(hScalar + 1, dScalar + 1, sizeof(long long), hipMemcpyDeviceToHost, commStream), "scalar download");
        hip_ok(hipStreamSynchronize(commStream), "comm sync");
        return hScalar[1];
    }
    void allreduce_device(const long long* dIn, long long* dOut) override
    {
        ++allreduces;
        wait();
        hip_ok(hipEventRecord(ready, sphx::stream()), "event record");
        hip_ok(hipStreamWaitEvent(commStream, ready, 0), "stream wait");
        nccl_ok(g_rccl.AllReduce(dIn, dOut, 1, ncclInt64, ncclSum, comm, commStream), "ncclAllReduce");
        hip_ok(hipEventRecord(done, commStream), "event record");
        pending = true; pendingAsync = false;
        wait();                                   // the engine stream continues behind the reduction; the host does not wait here
    }
};

// ================================================================================ device helpers
// Classification of the owned particles after last step's advect: global cell column by the engine's own expression (true fp32
// division, truncation: cell_of), which neighbour needs a copy, whether the particle stays in this slab's ghost range, and a sanity flag.
// r05: the three stable compactions (to the left neighbour / to the right neighbour / kept) no longer go through per-particle flag
// and scan arrays (24 bytes written, 48 scanned, 24 read back per particle): a counting pass leaves three counts per BLOCK of 256
// particles, the scan runs over those, and the packing pass recomputes the flags and ranks them inside the block with ballots.
struct SlabClass {
    float cellLength; int x0, x1, g, hasLeft, hasRight, loOK, hiOK;
    __device__ __forceinline__ int column(const float3 p) const { return (int)(p.x / cellLength); }
    __device__ __forceinline__ int left(int col) const { return (hasLeft && col <= x0 + g - 1) ? 1 : 0; }
    __device__ __forceinline__ int right(int col) const { return (hasRight && col >= x1 - g) ? 1 : 0; }
    // still inside this slab's ghost range?  (after a cut moved, a former owner may hold particles two columns out:
    // they travel to the neighbour like every migrant and are dropped here)
    __device__ __forceinline__ int kept(int col) const { return (col >= x0 - g && col <= x1 + g - 1) ? 1 : 0; }
    // travelled farther than ONE exchange reaches (columns [loOK, hiOK] are fine: sphx_slab_group::reach): the step then takes the
    // hop-by-hop path (sphx_slab_group::forwardFarFlyers)
    __device__ __forceinline__ bool crossed(int col) const { return col < loOK || col > hiOK; }
};
constexpr int kSlabBlock = 256;
// exclusive rank of this thread among the threads of its block with flag set, and the block's total (all threads call it; the
// three calls of a kernel use the table rows 0, 1, 2, so no call waits for the readers of the one before)
__device__ __forceinline__ int block_rank_256(int flag, int row, int* total)
{
    __shared__ int waveSums[3][kSlabBlock / 64];
    static_assert(kSlabBlock == 256, "four waves");
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned long long votes = __ballot(flag != 0);
    const int inWave = __popcll(votes & ((1ull << lane) - 1ull));
    if (lane == 0) waveSums[row][wave] = __popcll(votes);
    __syncthreads();
    int base = 0;
    for (int w = 0; w < wave; ++w) base += waveSums[row][w];
    *total = waveSums[row][0] + waveSums[row][1] + waveSums[row][2] + waveSums[row][3];
    return base + inWave;
}
__global__ void __launch_bounds__(kSlabBlock) k_slab_count(const float3* __restrict__ pos, int m, SlabClass c, int* __restrict__ blockL,
                                                           int* __restrict__ blockR, int* __restrict__ blockK, int* __restrict__ violation)
{
    const int k = blockIdx.x * kSlabBlock + threadIdx.x;
    int fl = 0, fr = 0, fk = 0;
    if (k < m) {
        const int col = c.column(pos[k]);
        fl = c.left(col); fr = c.right(col); fk = c.kept(col);
        if (c.crossed(col)) *violation = 1;
    }
    int tl, tr, tk;
    (void)block_rank_256(fl, 0, &tl); (void)block_rank_256(fr, 1, &tr); (void)block_rank_256(fk, 2, &tk);
    if (threadIdx.x == 0) { blockL[blockIdx.x] = tl; blockR[blockIdx.x] = tr; blockK[blockIdx.x] = tk; }
}

// the size words of a step before the classification fills in the rest: everything zero, {0, owned, width} for both neighbours
__global__ void k_slab_prepare(long long* __restrict__ counts, int* __restrict__ violation, long long owned, long long width)
{
    const int t = threadIdx.x;
    if (t < 13) counts[t] = (t == 1 || t == 4) ? owned : ((t == 2 || t == 5) ? width : 0);
    if (t == 13) *violation = 0;
}

// the failure word of a slab for this step, formed on the device once the neighbours' size messages have arrived: capacity = 1 << 20
// (the encoding the host reports from), a particle beyond the reach of one exchange = 1 << 40 (not a failure: every rank then takes
// the hop-by-hop path of this step together); added to the process's word, which is all-reduced on the stream
__global__ void k_slab_verdict(const long long* __restrict__ counts, const int* __restrict__ violation, long long capacity, int hasLeft,
                               int hasRight, long long inject, unsigned long long* __restrict__ processBad)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    long long bad = inject;
    if (*violation != 0) bad += 1LL << 40;
    const long long rl = hasLeft ? counts[6] : 0, rr = hasRight ? counts[9] : 0;
    if (counts[12] + rl + rr > capacity) bad += 1LL << 20;
    if (bad) atomicAdd(processBad, (unsigned long long)bad);
}

// payload row of a particle: pos(3) vel(3) id(1, bit pattern) extras(E).  `own` receives the owned particles still in
// this slab's range, sendL / sendR the copies for the neighbours: three stable compactions (block offsets = exclusive scans of the
// block counts of k_slab_count; inside a block the rank among the flagged threads)
__global__ void __launch_bounds__(kSlabBlock) k_slab_pack(const float3* __restrict__ pos, const float3* __restrict__ vel, const int* __restrict__ ids,
                                                          const float* __restrict__ extra, int E, int m, SlabClass c, const int* __restrict__ blockL,
                                                          const int* __restrict__ blockR, const int* __restrict__ blockK, float* __restrict__ own,
                                                          float* __restrict__ sendL, float* __restrict__ sendR, long long* __restrict__ counts)
{
    const int k = blockIdx.x * kSlabBlock + threadIdx.x;
    const int W = 7 + E;
    float row[10];
    int fl = 0, fr = 0, fk = 0;
    if (k < m) {
        const float3 p = pos[k], v = vel[k];
        const int col = c.column(p);
        fl = c.left(col); fr = c.right(col); fk = c.kept(col);
        row[0] = p.x; row[1] = p.y; row[2] = p.z; row[3] = v.x; row[4] = v.y; row[5] = v.z; row[6] = __int_as_float(ids[k]);
        for (int e = 0; e < E; ++e) row[7 + e] = extra[(size_t)k * E + e];
    }
    int tl, tr, tk;
    const int rl = blockL[blockIdx.x] + block_rank_256(fl, 0, &tl), rr = blockR[blockIdx.x] + block_rank_256(fr, 1, &tr),
              rk = blockK[blockIdx.x] + block_rank_256(fk, 2, &tk);
    if (fk) { float* d = own + (size_t)rk * W; for (int t = 0; t < W; ++t) d[t] = row[t]; }
    if (fl) { float* d = sendL + (size_t)rl * W; for (int t = 0; t < W; ++t) d[t] = row[t]; }
    if (fr) { float* d = sendR + (size_t)rr * W; for (int t = 0; t < W; ++t) d[t] = row[t]; }
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) {
        counts[0] = blockL[blockIdx.x] + tl; counts[3] = blockR[blockIdx.x] + tr; counts[12] = blockK[blockIdx.x] + tk;
    }
}

// new pre-sort arrays = [from left | previously owned | from right]: ascending in last step's global order, which
// keeps the stable cell sort identical to the single-device one
__global__ void k_slab_unpack(float3* __restrict__ pos, float3* __restrict__ vel, int* __restrict__ ids, float* __restrict__ extra,
                              int E, const float* __restrict__ fromL, int nl, const float* __restrict__ own, int m,
                              const float* __restrict__ fromR, int nr)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nl + m + nr) return;
    const int W = 7 + E;
    const float* row = t < nl ? fromL + (size_t)t * W : (t < nl + m ? own + (size_t)(t - nl) * W : fromR + (size_t)(t - nl - m) * W);
    pos[t] = make_float3(row[0], row[1], row[2]);
    vel[t] = make_float3(row[3], row[4], row[5]);
    ids[t] = __float_as_int(row[6]);
    for (int e = 0; e < E; ++e) extra[(size_t)t * E + e] = row[7 + e];
}

// ---- rows that arrived from a neighbour in a hop-by-hop step: which stay here (owned or ghost), which travel on --------------------
// dir 0: the rows came from the LEFT neighbour (they travel right), dir 1: from the right.  Stable, like k_slab_count / k_slab_pack.
__global__ void __launch_bounds__(kSlabBlock) k_rows_count(const float* __restrict__ rows, int n, int W, SlabClass c, int dir,
                                                           int* __restrict__ blockK, int* __restrict__ blockF)
{
    const int k = blockIdx.x * kSlabBlock + threadIdx.x;
    int fk = 0, ff = 0;
    if (k < n) {
        const int col = c.column(make_float3(rows[(size_t)k * W], 0.0f, 0.0f));
        fk = c.kept(col); ff = dir == 0 ? c.right(col) : c.left(col);
    }
    int tk, tf;
    (void)block_rank_256(fk, 0, &tk); (void)block_rank_256(ff, 1, &tf);
    if (threadIdx.x == 0) { blockK[blockIdx.x] = tk; blockF[blockIdx.x] = tf; }
}
__global__ void __launch_bounds__(kSlabBlock) k_rows_pack(const float* __restrict__ rows, int n, int W, SlabClass c, int dir,
                                                          const int* __restrict__ blockK, const int* __restrict__ blockF,
                                                          float* __restrict__ keep, float* __restrict__ fwd, long long* __restrict__ outCounts)
{
    const int k = blockIdx.x * kSlabBlock + threadIdx.x;
    int fk = 0, ff = 0;
    if (k < n) {
        const int col = c.column(make_float3(rows[(size_t)k * W], 0.0f, 0.0f));
        fk = c.kept(col); ff = dir == 0 ? c.right(col) : c.left(col);
    }
    int tk, tf;
    const int rk = blockK[blockIdx.x] + block_rank_256(fk, 0, &tk), rf = blockF[blockIdx.x] + block_rank_256(ff, 1, &tf);
    if (k < n) {
        const float* src = rows + (size_t)k * W;
        if (fk) { float* d = keep + (size_t)rk * W; for (int t = 0; t < W; ++t) d[t] = src[t]; }
        if (ff) { float* d = fwd + (size_t)rf * W; for (int t = 0; t < W; ++t) d[t] = src[t]; }
    }
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) { outCounts[0] = blockK[blockIdx.x] + tk; outCounts[1] = blockF[blockIdx.x] + tf; }
}

__global__ void k_slab_pick6(const int* __restrict__ cellStart, int i0, int i1, int i2, int i3, int i4, int i5, int* __restrict__ out)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        out[0] = cellStart[i0]; out[1] = cellStart[i1]; out[2] = cellStart[i2]; out[3] = cellStart[i3]; out[4] = cellStart[i4];
        out[5] = cellStart[i5];
    }
}

template <class T>
struct DevBuf {
    T* p = nullptr; size_t count = 0;
    void alloc(size_t n) { count = n ? n : 1; hip_ok(hipMalloc((void**)&p, sizeof(T) * count), "slab scratch"); hip_ok(hipMemsetAsync(p, 0, sizeof(T) * count, sphx::stream()), "memset"); }
    ~DevBuf() { if (p) (void)hipFree(p); }
};

// ================================================================================ one slab
struct Slab {
    int rank = 0, world = 1;
    int x0 = 0, x1 = 0, ghost = 1, cellsPerColumn = 0, gx = 0;
    // what the neighbours reported with last step's size message: their owned particle counts and widths (cut re-balancing)
    long long ownedLeft = -1, ownedRight = -1; int widthLeft = 0, widthRight = 0;
    float cellLength = 0.0f;
    int solver = SPHX_WCSPH;
    sphx_system* sys = nullptr;
    int capacity = 0, extraFloats = 0;
    int o0 = 0, o1 = 0;              // owned particles [o0, o1) of the engine arrays
    int layer[5] = {0, 0, 0, 0, 0};  // [0,l0) left ghosts, [l0,l1) first owned layers, [l2,l3) last owned layers, [l3,l4) right ghosts
    int held = 0;
    bool hasLeft = false, hasRight = false;
    long stepsDone = 0;
    // engine arrays
    float3 *pos = nullptr, *vel = nullptr; int* ids = nullptr; float* extra = nullptr; float* density = nullptr; int* cellStart = nullptr;
    // scratch
    DevBuf<int> blockL, blockR, blockK, blockSums, violation, layerOut;      // per block of 256 owned particles: rows for the left / right neighbour / kept
    DevBuf<float> own, sendL, sendR, recvL, recvR;
    DevBuf<float> farBuf;                     // hop-by-hop steps: rows in transit and the kept rows of the later hops (allocated by the first such step)
    DevBuf<long long> farCounts;              // [0,1] kept / forwarded of the rows from the left, [2,3] from the right, [4] to left [5] to right [6] from left [7] from right
    long long* hFar = nullptr;                // pinned copy
    // size messages, 3 x int64 each: {payload particles, owned particles, width in columns}
    //   [0..2] to left  [3..5] to right  [6..8] from left  [9..11] from right   [12] owned particles kept here
    DevBuf<long long> counts;
    long long* hCounts = nullptr; int* hInts = nullptr;   // pinned: 13 size words; 6 layer offsets
    hipEvent_t layersReady = nullptr;                      // the layer offsets of this step have arrived in hInts
    long long sentOwned = 0; int sentWidth = 0;            // what this slab reported with its last size message

    ~Slab()
    {
        if (sys) sphx_destroy(sys);
        if (hCounts) (void)hipHostFree(hCounts);
        if (hInts) (void)hipHostFree(hInts);
        if (hFar) (void)hipHostFree(hFar);
        if (layersReady) (void)hipEventDestroy(layersReady);
    }
    int width() const { return 7 + extraFloats; }
    void* field(int f, size_t* bytesPerParticle) const
    {
        void* p = nullptr; size_t total = 0;
        if (sphx_locate(sys, f, &p, &total)) die("slab: the engine lacks a field of this solver");
        *bytesPerParticle = total / (size_t)std::max(capacity, 1);
        return p;
    }
};

}  // namespace

// ================================================================================ the group
#ifdef SPHX_TEST_HOOKS
// test build only: SPHX_SLAB_TRACE=1 prints where the host side of every rank is (the last line of a rank that hangs says where)
static void slab_trace(int rank, int step, const char* what, long long arg = 0)
{
    static const bool on = std::getenv("SPHX_SLAB_TRACE") != nullptr;
    if (on) { std::fprintf(stderr, "[slab rank %d step %d] %s %lld\n", rank, step, what, arg); std::fflush(stderr); }
}
#define SLAB_TRACE(what, arg) slab_trace(slabs[0]->rank, slabs[0]->stepsDone, what, (long long)(arg))
#else
#define SLAB_TRACE(what, arg) ((void)0)
#endif
struct sphx_slab_group {
    std::vector<std::unique_ptr<Slab>> slabs;
    std::unique_ptr<Transport> transport;
    sphx_params global{};
    int world = 1, flags = 0;
    bool surface = false, adaptive = false;
    long long nGlobal = 0;
    int lastDiv = 0, lastDen = 0;
    double waitSeconds = 0.0;
    std::vector<Msg> sends, recvs;
    bool failed = false;        // a step threw: posted messages were dropped, slabs may be half-updated -> only destroy is allowed
    DevBuf<long long> dBad;                               // [0] failure word of this process's slabs in the current step, [1] summed over all processes
    long long* hBad = nullptr;                            // pinned copy of both
    hipStream_t edgeStream = nullptr;                     // DFSPH / WCSPH edge layers + their halo, beside the interior (sphx_tuning.slab_edge_stream = 0: off)
    hipEvent_t forkEvent = nullptr, joinEvent = nullptr;

    ~sphx_slab_group()
    {
        if (edgeStream) { (void)hipStreamSynchronize(edgeStream); (void)hipStreamDestroy(edgeStream); }
        if (hBad) (void)hipHostFree(hBad);
        if (forkEvent) (void)hipEventDestroy(forkEvent);
        if (joinEvent) (void)hipEventDestroy(joinEvent);
    }
    void createEdgeStream()
    {
        if (sphx::tuning().slab_edge_stream == 0) return;
        int least = 0, greatest = 0;
        hip_ok(hipDeviceGetStreamPriorityRange(&least, &greatest), "stream priority range");
        // A DEFAULT-priority stream (r04).  Until r04 this was the highest priority the device has; with 8 processes sharing the one
        // test GPU (tests/test_gpu_slab.py, 8 ranks, transfers completing late) a rank then now and then computed different bits or died
        // of HSA_STATUS_ERROR_ILLEGAL_INSTRUCTION in its first step -- 4 of 30 runs, in the r03 sources as well; 0 of 24 with this
        // stream at default priority, 0 of 12 without it (profiles/r04_slab_edge_stream_priority.txt).  The edge kernels are enqueued
        // before the interior of their stage, so they start first anyway.
        // r06: the fault is NAMED (profiles/r06_slab_edge_stream.txt: it needs the TEST stand-in's deferred completion -- pinned staging
        // copies and a stream host callback -- beside a highest-priority queue of the same process; 30 of 30 runs clean with the
        // immediate stand-in and with the installed librccl at that priority, 4 of 15 wrong with the deferred stand-in) and the switch
        // is gone from the product: only the test build of the library can still ask for the old stream, to keep the bisection reproducible.
        bool highest = false;
#ifdef SPHX_TEST_HOOKS
        highest = std::getenv("SPHX_SLAB_EDGE_HIGHEST") != nullptr;
#endif
        if (highest) hip_ok(hipStreamCreateWithPriority(&edgeStream, hipStreamNonBlocking, greatest), "edge stream");
        else hip_ok(hipStreamCreateWithFlags(&edgeStream, hipStreamNonBlocking), "edge stream");
        hip_ok(hipEventCreateWithFlags(&forkEvent, hipEventDisableTiming), "event");
        hip_ok(hipEventCreateWithFlags(&joinEvent, hipEventDisableTiming), "event");
    }

    bool overlap() const { return (flags & SPHX_SLAB_NO_OVERLAP) == 0; }

    void sync(const char* what)
    {
        const auto t0 = std::chrono::steady_clock::now();
        transport->wait();
        hip_ok(hipStreamSynchronize(sphx::stream()), what);
        waitSeconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    }

    // ---- cut re-balancing ---------------------------------------------------------------------------------------------
    // Every `rebalanceEvery` steps a cut plane moves by one column towards the lighter neighbour when the two owned
    // counts differ by more than `rebalanceTol`.  Both neighbours evaluate the same rule on the same numbers (the owned
    // counts and widths exchanged with last step's size messages), so they agree without another message.  Ownership
    // follows by itself: the particles of the column that changed hands are migrants of the ordinary exchange.
    int rebalanceEvery = 16;
    float rebalanceTol = 0.05f;
    static int cut_shift(long long ownedA, long long ownedB, int widthA, int widthB, int ghost, float tol)
    {
        const int minShrinkable = ghost + 4;     // stays >= ghost + 2 even if its other cut shrinks it too
        // A column changes hands only when that REDUCES the difference: the heavier slab's average column must weigh less than
        // the difference itself (r03: at 10.3 M particles over 8 slabs a column is 8 % of a slab; with the 5 % dead band alone
        // the cuts ping-ponged and the worst slab drifted from +5 % to +9.5 % of the mean).
        const double diff = (double)ownedA - (double)ownedB;
        if (diff > (double)ownedB * tol && diff > (double)ownedA / (double)std::max(widthA, 1) && widthA >= minShrinkable) return -1;   // left slab hands a column over
        if (-diff > (double)ownedA * tol && -diff > (double)ownedB / (double)std::max(widthB, 1) && widthB >= minShrinkable) return +1;
        return 0;
    }
    // How far an owned particle may travel in one step.  A migrant goes to the ADJACENT slab, and the exchange is complete as long as
    // (a) the column it lands in is owned by that neighbour and (b) is not one of the neighbour's FAR edge columns -- those are ghost
    // columns of the slab beyond, which takes its ghosts from what the neighbour owned BEFORE this step.  Everything nearer is
    // handled by the one exchange there is: the pre-sort order [from left | kept | from right] is the global order of the last
    // step for any displacement, the sender keeps what stays in its own ghost range, the receiver sorts what it is handed.
    // So the reach is the neighbour's width minus its ghost width (minus one column for a far cut that moves in this very step),
    // not one column (r02-r05): 20+ columns at 8 slabs of the 10.3 M scene, i.e. |v| dt of most of a slab.  The end slabs own the
    // rest of the domain on their outer side: no limit there.  Widths are the ones exchanged with the last size messages (the
    // planned ones before the first step); [x0b, x1b) is this slab's range before this step's own cut moves.
    static void reach(const Slab& s, int x0b, int x1b, int slackL, int slackR, int& loOK, int& hiOK)
    {
        loOK = INT_MIN / 2; hiOK = INT_MAX / 2;
        if (s.solver == SPHX_PBD) {
            // PBD is different: its sweeps run on positions that moved INSIDE the step over the cell table of the step's start
            // (PBDSolver.cu:139-141), so a slab's neighbourhoods are complete only while a particle stays within
            // (ghost - 1) = one column of the column it was binned in.  The travel of a step is the solver's own result (constraint
            // projection), not known before the exchange: a farther move is found by the NEXT exchange and ends the run, as in r02-r05.
            if (s.hasLeft) loOK = s.x0 - 1 - slackL;
            if (s.hasRight) hiOK = s.x1 + slackR;
            return;
        }
        if (s.hasLeft) {
            const int oneColumn = s.x0 - 1 - slackL;                                  // (the r02 rule: always safe)
            loOK = s.rank >= 2 ? std::min(oneColumn, x0b - s.widthLeft + s.ghost + 1) : (s.widthLeft > 0 ? INT_MIN / 2 : oneColumn);
            if (s.widthLeft <= 0) loOK = oneColumn;
        }
        if (s.hasRight) {
            const int oneColumn = s.x1 + slackR;
            hiOK = s.rank + 2 < s.world ? std::max(oneColumn, x1b + s.widthRight - s.ghost - 2) : (s.widthRight > 0 ? INT_MAX / 2 : oneColumn);
            if (s.widthRight <= 0) hiOK = oneColumn;
        }
    }
    void rebalance(Slab& s, int& slackL, int& slackR)
    {
        slackL = slackR = 0;
        if (rebalanceEvery <= 0 || s.stepsDone == 0 || (s.stepsDone % rebalanceEvery) != 0) return;
        if (s.hasLeft && s.ownedLeft >= 0) {
            const int d = cut_shift(s.ownedLeft, s.sentOwned, s.widthLeft, s.sentWidth, s.ghost, rebalanceTol);
            s.x0 += d; slackL = d < 0 ? -d : d;
        }
        if (s.hasRight && s.ownedRight >= 0) {
            const int d = cut_shift(s.sentOwned, s.ownedRight, s.sentWidth, s.widthRight, s.ghost, rebalanceTol);
            s.x1 += d; slackR = d < 0 ? -d : d;
        }
    }

    // ---- particle exchange (migrants and ghost copies alike) ------------------------------------------------
    void exchangeParticles()
    {
        hipStream_t st = sphx::stream();
        for (auto& sp : slabs) {
            Slab& s = *sp;
            int slackL = 0, slackR = 0;
            const int x0b = s.x0, x1b = s.x1;
            rebalance(s, slackL, slackR);
            int loOK = 0, hiOK = 0;
            reach(s, x0b, x1b, slackL, slackR, loOK, hiOK);
            const int m = s.o1 - s.o0;
            // size words that do not come from the device: owned count and width, for both neighbours
            s.sentOwned = m; s.sentWidth = s.x1 - s.x0;
            k_slab_prepare<<<1, 64, 0, st>>>(s.counts.p, s.violation.p, (long long)m, (long long)(s.x1 - s.x0));
            if (m > 0) {
                const SlabClass cls{s.cellLength, s.x0, s.x1, s.ghost, s.hasLeft ? 1 : 0, s.hasRight ? 1 : 0, loOK, hiOK};
                const int blocks = (m - 1) / kSlabBlock + 1;
                k_slab_count<<<blocks, kSlabBlock, 0, st>>>(s.pos + s.o0, m, cls, s.blockL.p, s.blockR.p, s.blockK.p, s.violation.p);
                device_exclusive_scan3(s.blockL.p, s.blockR.p, s.blockK.p, blocks, s.blockSums.p);      // over the BLOCK counts: a few thousand words
                k_slab_pack<<<blocks, kSlabBlock, 0, st>>>(s.pos + s.o0, s.vel + s.o0, s.ids + s.o0,
                                                           s.extraFloats ? s.extra + (size_t)s.o0 * s.extraFloats : nullptr, s.extraFloats, m, cls,
                                                           s.blockL.p, s.blockR.p, s.blockK.p, s.own.p, s.sendL.p, s.sendR.p, s.counts.p);
            }
            // size messages travel first (24 bytes per neighbour)
            if (s.hasLeft) { sends.push_back({s.rank, s.rank - 1, s.counts.p + 0, 24}); recvs.push_back({s.rank - 1, s.rank, s.counts.p + 6, 24}); }
            if (s.hasRight) { sends.push_back({s.rank, s.rank + 1, s.counts.p + 3, 24}); recvs.push_back({s.rank + 1, s.rank, s.counts.p + 9, 24}); }
        }
        SLAB_TRACE("size exchange: posting", sends.size());
        transport->exchange(sends, recvs, false);
        SLAB_TRACE("size exchange: posted", 0);
        // Rank-local failures are only known now (the violation flag, and the capacity check needs the neighbours' sizes), but the
        // neighbours are about to post receives for THIS rank's payload: a rank that simply returned an error here would leave them
        // waiting in ncclRecv.  So every process contributes its failure word to one all-reduce and all of them leave the step
        // together.  r05: the word is formed on the device behind the size messages and reduced ON THE STREAM, and it
        // comes back with the sizes: one host wait per step instead of two.
        hip_ok(hipMemsetAsync(dBad.p, 0, 2 * sizeof(long long), st), "memset");
        for (auto& sp : slabs) {
            Slab& s = *sp;
            long long inject = 0;
#ifdef SPHX_TEST_HOOKS
            if (const char* f = std::getenv("SPHX_SLAB_FAULT")) {      // fault injection, test build only: "capacity:<rank>:<step>"
                int r = -1, at = -1;
                if (std::sscanf(f, "capacity:%d:%d", &r, &at) == 2 && s.rank == r && s.stepsDone == at) inject = 1LL << 20;
            }
#endif
            k_slab_verdict<<<1, 64, 0, st>>>(s.counts.p, s.violation.p, (long long)s.capacity, s.hasLeft ? 1 : 0, s.hasRight ? 1 : 0, inject,
                                             reinterpret_cast<unsigned long long*>(dBad.p));
        }
        // (the RCCL transport reduces even when this process drives every slab -- a one-rank ncclAllReduce: the runs over the installed
        // librccl on the one-GPU box then exercise the same calls, events and stream order a node would)
        int kind = 0, ranks = 0, rk = 0;
        transport->describe(kind, ranks, rk);
        if (world > (int)slabs.size() || kind == 1) transport->allreduce_device(dBad.p, dBad.p + 1);
        else hip_ok(hipMemcpyAsync(dBad.p + 1, dBad.p, sizeof(long long), hipMemcpyDeviceToDevice, st), "failure word");
        for (auto& sp : slabs) {
            Slab& s = *sp;
            hip_ok(hipMemcpyAsync(s.hCounts, s.counts.p, 13 * sizeof(long long), hipMemcpyDeviceToHost, st), "counts");
        }
        hip_ok(hipMemcpyAsync(hBad, dBad.p, 2 * sizeof(long long), hipMemcpyDeviceToHost, st), "failure word");
        sync("particle exchange (sizes)");
        SLAB_TRACE("size exchange: synchronised", 0);
        const long long bad = hBad[0], anyBad = hBad[1];
        SLAB_TRACE("failure word reduced", anyBad);
        const bool farStep = (anyBad >> 40) != 0;            // somewhere a particle flew past the reach of one exchange
        if (farStep && global.solver == SPHX_PBD)
            die(std::string("slab: PBD: a particle moved more than one cell column within a step -- the sweeps of that step ran on positions outside "
                            "the two ghost columns (") + ((bad >> 40) ? "this process" : "another rank") + ")");
        if ((anyBad >> 20) & ((1LL << 20) - 1)) {
            const char* here = ((bad >> 20) & ((1LL << 20) - 1)) ? "this process" : "another rank";
            die(std::string("slab: capacity exceeded (particles piled up in one slab; ") + here + ")");
        }
        for (auto& sp : slabs) {
            Slab& s = *sp;
            const size_t rowBytes = sizeof(float) * (size_t)s.width();
            const long long sl = s.hCounts[0], sr = s.hCounts[3], rl = s.hasLeft ? s.hCounts[6] : 0, rr = s.hasRight ? s.hCounts[9] : 0;
            if (s.hasLeft) { s.ownedLeft = s.hCounts[7]; s.widthLeft = (int)s.hCounts[8]; }
            if (s.hasRight) { s.ownedRight = s.hCounts[10]; s.widthRight = (int)s.hCounts[11]; }
            if (s.hasLeft) { sends.push_back({s.rank, s.rank - 1, s.sendL.p, (size_t)sl * rowBytes}); recvs.push_back({s.rank - 1, s.rank, s.recvL.p, (size_t)rl * rowBytes}); }
            if (s.hasRight) { sends.push_back({s.rank, s.rank + 1, s.sendR.p, (size_t)sr * rowBytes}); recvs.push_back({s.rank + 1, s.rank, s.recvR.p, (size_t)rr * rowBytes}); }
        }
        transport->exchange(sends, recvs, false);
        std::vector<int> nlv(slabs.size()), nrv(slabs.size());
        for (size_t i = 0; i < slabs.size(); ++i) {
            nlv[i] = slabs[i]->hasLeft ? (int)slabs[i]->hCounts[6] : 0;
            nrv[i] = slabs[i]->hasRight ? (int)slabs[i]->hCounts[9] : 0;
        }
        if (farStep) forwardFarFlyers(nlv, nrv);
        for (size_t i = 0; i < slabs.size(); ++i) {
            Slab& s = *slabs[i];
            const int m = (int)s.hCounts[12], nl = nlv[i], nr = nrv[i];   // m: kept
            const int n = nl + m + nr;
            // DFSPH / WCSPH: the rows [from left | kept | from right] ARE the pre-sort order of this step; the SEARCH stage sorts them
            // straight into the engine's arrays (no unpack pass, no copy back, the warm stiffness arrives sorted).  PBD unpacks.
            if (s.solver != SPHX_PBD) s.sys->system->setCellWindow(s.x0 - s.ghost - 1, s.x1 + s.ghost + 1);      // held columns + an empty one per side
            if (s.solver != SPHX_PBD)
                s.sys->system->setStagedInput(SPHSystem::StagedRows{{s.recvL.p, s.own.p, s.recvR.p}, {nl, m, nr}, s.extraFloats, s.extraFloats ? s.extra : nullptr});
            else if (n > 0)
                k_slab_unpack<<<blocks_for(n), 256, 0, st>>>(s.pos, s.vel, s.ids, s.extra, s.extraFloats, s.recvL.p, nl, s.own.p, m, s.recvR.p, nr);
            s.sys->system->getFluids()->setActiveCount((unsigned)n);
            s.held = n;
        }
    }

    // ---- hop by hop: a step in which some particle flew past the reach of one exchange (r06) -----------------------------------------
    // The first exchange has delivered every owned particle that left its slab to the ADJACENT slab, whatever column it landed in.
    // Now the rows a slab received are looked at with the same two questions the owned particles were asked: does this slab hold the
    // column (as owner or as ghost) -> keep a copy; does the neighbour on the far side need it (owner or ghost) -> send it on.  Rows
    // keep their direction, every hop is one grouped send/recv with both neighbours, and the hops end when no rank has sent anything
    // on (one all-reduce per hop).  Order: rows that arrive from the left in a LATER hop come from slabs farther left, i.e. from lower
    // global indices of the last step: the pre-sort order is [kept of hop J | ... | kept of hop 1 | own | kept of hop 1 | ... | hop J],
    // each part in its sender's order -- the global order of the last step again, so the stable cell sort still reproduces the
    // single-device permutation (/root/reference/src/SPHSystem.cu:114-127 re-bins any displacement).
    // Cost: two small kernels per side and hop, one host wait and one blocking all-reduce per hop -- in steps that need it only
    // (a particle beyond the reach of one exchange: |v| dt of most of a slab).
    void forwardFarFlyers(std::vector<int>& nlv, std::vector<int>& nrv)
    {
        hipStream_t st = sphx::stream();
        struct Chunk { const float* p; long long rows; };
        const size_t S = slabs.size();
        std::vector<std::vector<Chunk>> keptL(S), keptR(S);
        std::vector<const float*> curL(S), curR(S);       // the rows that arrived in the last hop
        std::vector<long long> curNl(S), curNr(S), farUsed(S, 0);
        for (size_t i = 0; i < S; ++i) {
            Slab& s = *slabs[i];
            if (!s.farBuf.p) s.farBuf.alloc((size_t)s.capacity * (size_t)s.width());
            curL[i] = s.recvL.p; curR[i] = s.recvR.p; curNl[i] = nlv[i]; curNr[i] = nrv[i];
        }
        for (int hop = 1; hop <= world + 1; ++hop) {
            if (hop > world) die("slab: rows still in transit after as many hops as there are slabs");
            // classify what arrived: kept rows and rows to send on.  Hop 1 keeps into the (now idle) send buffers of the first
            // exchange, later hops into the transit buffer; forwarded rows always go into the transit buffer.
            std::vector<float*> fwdL(S, nullptr), fwdR(S, nullptr);
            for (size_t i = 0; i < S; ++i) {
                Slab& s = *slabs[i];
                const int W = s.width();
                const SlabClass cls{s.cellLength, s.x0, s.x1, s.ghost, s.hasLeft ? 1 : 0, s.hasRight ? 1 : 0, 0, 0};
                hip_ok(hipMemsetAsync(s.farCounts.p, 0, 8 * sizeof(long long), st), "memset");
                auto take = [&](long long rows) {
                    if ((farUsed[i] + rows) * W > (long long)s.farBuf.count) die("slab: too many rows in transit in one step (hop-by-hop buffer)");
                    float* p = s.farBuf.p + (size_t)farUsed[i] * W; farUsed[i] += rows; return p;
                };
                float* keepLp = hop == 1 ? s.sendL.p : take(curNl[i]);
                float* keepRp = hop == 1 ? s.sendR.p : take(curNr[i]);
                fwdR[i] = take(curNl[i]); fwdL[i] = take(curNr[i]);          // rows from the left travel right and the other way round
                if (curNl[i] > 0) {
                    const int blocks = (int)((curNl[i] - 1) / kSlabBlock + 1);
                    k_rows_count<<<blocks, kSlabBlock, 0, st>>>(curL[i], (int)curNl[i], W, cls, 0, s.blockK.p, s.blockR.p);
                    device_exclusive_scan3(s.blockL.p, s.blockR.p, s.blockK.p, blocks, s.blockSums.p);
                    k_rows_pack<<<blocks, kSlabBlock, 0, st>>>(curL[i], (int)curNl[i], W, cls, 0, s.blockK.p, s.blockR.p, keepLp, fwdR[i], s.farCounts.p + 0);
                }
                if (curNr[i] > 0) {
                    const int blocks = (int)((curNr[i] - 1) / kSlabBlock + 1);
                    k_rows_count<<<blocks, kSlabBlock, 0, st>>>(curR[i], (int)curNr[i], W, cls, 1, s.blockK.p, s.blockL.p);
                    device_exclusive_scan3(s.blockL.p, s.blockR.p, s.blockK.p, blocks, s.blockSums.p);
                    k_rows_pack<<<blocks, kSlabBlock, 0, st>>>(curR[i], (int)curNr[i], W, cls, 1, s.blockK.p, s.blockL.p, keepRp, fwdL[i], s.farCounts.p + 2);
                }
                keptL[i].push_back({keepLp, 0}); keptR[i].push_back({keepRp, 0});
                // what goes on: [4] to the left = forwarded rows that came from the right, [5] to the right = those that came from the left
                hip_ok(hipMemcpyAsync(s.farCounts.p + 4, s.farCounts.p + 3, sizeof(long long), hipMemcpyDeviceToDevice, st), "count");
                hip_ok(hipMemcpyAsync(s.farCounts.p + 5, s.farCounts.p + 1, sizeof(long long), hipMemcpyDeviceToDevice, st), "count");
                if (s.hasLeft) { sends.push_back({s.rank, s.rank - 1, s.farCounts.p + 4, 8}); recvs.push_back({s.rank - 1, s.rank, s.farCounts.p + 6, 8}); }
                if (s.hasRight) { sends.push_back({s.rank, s.rank + 1, s.farCounts.p + 5, 8}); recvs.push_back({s.rank + 1, s.rank, s.farCounts.p + 7, 8}); }
            }
            transport->exchange(sends, recvs, false);
            for (auto& sp : slabs) hip_ok(hipMemcpyAsync(sp->hFar, sp->farCounts.p, 8 * sizeof(long long), hipMemcpyDeviceToHost, st), "far counts");
            sync("particle exchange (hop by hop)");
            long long inTransit = 0;
            for (size_t i = 0; i < S; ++i) {
                Slab& s = *slabs[i];
                keptL[i].back().rows = s.hFar[0]; keptR[i].back().rows = s.hFar[2];
                if (!s.hasLeft) s.hFar[4] = s.hFar[6] = 0;
                if (!s.hasRight) s.hFar[5] = s.hFar[7] = 0;
                inTransit += s.hFar[4] + s.hFar[5];
            }
            int kind = 0, ranks = 0, rk = 0;
            transport->describe(kind, ranks, rk);
            if (world > (int)slabs.size() || kind == 1) inTransit = transport->allreduce_sum(inTransit);
            SLAB_TRACE("hop: rows sent on", inTransit);
            if (inTransit == 0) break;
            for (size_t i = 0; i < S; ++i) {
                Slab& s = *slabs[i];
                const int W = s.width();
                const size_t rowBytes = sizeof(float) * (size_t)W;
                auto take = [&](long long rows) {
                    if ((farUsed[i] + rows) * W > (long long)s.farBuf.count) die("slab: too many rows in transit in one step (hop-by-hop buffer)");
                    float* p = s.farBuf.p + (size_t)farUsed[i] * W; farUsed[i] += rows; return p;
                };
                float* inL = take(s.hFar[6]); float* inR = take(s.hFar[7]);
                if (s.hasLeft) { sends.push_back({s.rank, s.rank - 1, fwdL[i], (size_t)s.hFar[4] * rowBytes}); recvs.push_back({s.rank - 1, s.rank, inL, (size_t)s.hFar[6] * rowBytes}); }
                if (s.hasRight) { sends.push_back({s.rank, s.rank + 1, fwdR[i], (size_t)s.hFar[5] * rowBytes}); recvs.push_back({s.rank + 1, s.rank, inR, (size_t)s.hFar[7] * rowBytes}); }
                curL[i] = inL; curR[i] = inR; curNl[i] = s.hFar[6]; curNr[i] = s.hFar[7];
            }
            transport->exchange(sends, recvs, false);
        }
        // the kept rows of all hops, in the global order of the last step, back into the receive buffers of the first exchange
        for (size_t i = 0; i < S; ++i) {
            Slab& s = *slabs[i];
            const size_t rowBytes = sizeof(float) * (size_t)s.width();
            long long nl = 0, nr = 0, total = (long long)s.hCounts[12];
            for (auto& c : keptL[i]) total += c.rows;
            for (auto& c : keptR[i]) total += c.rows;
            if (total > s.capacity) die("slab: capacity exceeded (particles piled up in one slab; this process)");
            for (size_t h = keptL[i].size(); h-- > 0;) {           // farthest hop first
                const Chunk& c = keptL[i][h];
                if (c.rows) hip_ok(hipMemcpyAsync(s.recvL.p + (size_t)nl * s.width(), c.p, (size_t)c.rows * rowBytes, hipMemcpyDeviceToDevice, st), "kept rows");
                nl += c.rows;
            }
            for (const Chunk& c : keptR[i]) {
                if (c.rows) hip_ok(hipMemcpyAsync(s.recvR.p + (size_t)nr * s.width(), c.p, (size_t)c.rows * rowBytes, hipMemcpyDeviceToDevice, st), "kept rows");
                nr += c.rows;
            }
            nlv[i] = (int)nl; nrv[i] = (int)nr;
        }
    }

    // after the local sort: where the ghost and edge layers are (cell starts at six column boundaries of the global grid).
    // The read-back is enqueued by the engine's after-sort hook, i.e. BEFORE the packing and the row build of the SEARCH stage:
    // the host waits for that copy only and issues the first sweep stage while the rows are still being built.
    static void readLayers(Slab& s)
    {
        hipStream_t st = sphx::stream();
        const int L = s.cellsPerColumn, w = s.ghost;
        auto at = [&](int column) { return std::min(std::max(column, 0), s.gx) * L; };
        k_slab_pick6<<<1, 64, 0, st>>>(s.cellStart, at(s.x0 - w), at(s.x0), at(s.x0 + w), at(s.x1 - w), at(s.x1), at(s.x1 + w), s.layerOut.p);
        hip_ok(hipMemcpyAsync(s.hInts, s.layerOut.p, 6 * sizeof(int), hipMemcpyDeviceToHost, st), "layers");
        hip_ok(hipEventRecord(s.layersReady, st), "event record");
    }
    void updateLayers()
    {
        const auto t0 = std::chrono::steady_clock::now();
        SLAB_TRACE("waiting for the layer offsets", 0);
        for (auto& sp : slabs) hip_ok(hipEventSynchronize(sp->layersReady), "layer offsets");
        SLAB_TRACE("layer offsets here", 0);
        waitSeconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        for (auto& sp : slabs) {
            Slab& s = *sp;
            if (s.hInts[0] != 0 || s.hInts[5] != s.held) die("slab: a held particle lies outside the slab's ghost range");
            for (int k = 0; k < 5; ++k) s.layer[k] = s.hInts[k + 1];
            s.o0 = s.layer[0]; s.o1 = s.layer[3];
        }
    }

    // Halo refresh of one or more per-particle fields: the edge layers are contiguous ranges of the sorted arrays, so
    // a message is a plain slice of the engine's own array (no pack kernels, no staging copies); the sizes are known
    // on both sides from their own cell tables (my right ghosts ARE the neighbour's first owned layers).
    void postHalo(const std::vector<int>& fields, bool async)
    {
        if (fields.empty()) return;
        for (int fieldId : fields)
            for (auto& sp : slabs) {
                Slab& s = *sp;
                size_t bpp = 0;
                char* base = static_cast<char*>(s.field(fieldId, &bpp));
                const int* l = s.layer;
                if (s.hasLeft) {
                    sends.push_back({s.rank, s.rank - 1, base + (size_t)l[0] * bpp, (size_t)(l[1] - l[0]) * bpp});
                    recvs.push_back({s.rank - 1, s.rank, base, (size_t)l[0] * bpp});
                }
                if (s.hasRight) {
                    sends.push_back({s.rank, s.rank + 1, base + (size_t)l[2] * bpp, (size_t)(l[3] - l[2]) * bpp});
                    recvs.push_back({s.rank + 1, s.rank, base + (size_t)l[3] * bpp, (size_t)(l[4] - l[3]) * bpp});
                }
            }
        transport->exchange(sends, recvs, async);
    }

    void runAll(int phase)
    {
        SLAB_TRACE("stage (all held particles)", phase);
        transport->wait();
        for (auto& sp : slabs) sp->sys->system->phase(phase);
    }

    // The layers of a DFSPH / WCSPH stage.  An edge layer exists where a neighbour reads it (it goes first, its halo is posted while the
    // interior is swept); on a side WITHOUT a neighbour the first owned columns are ordinary interior particles.  (Until r05 both edge
    // layers of every slab were launched apart whether anyone waited for them or not: two launches, a fork and a join per stage for a
    // slab on its own -- +1.1 % per step at 10.3 M, profiles/r05_slab_one_slab_probe.txt.)
    struct StageParts { int e0, e1, e20, e21, in0, in1; };
    static StageParts stageParts(const Slab& s)
    {
        const int* l = s.layer;
        StageParts p{-1, -1, -1, -1, s.hasLeft ? l[1] : l[0], s.hasRight ? l[2] : l[3]};
        if (s.hasLeft) { p.e0 = l[0]; p.e1 = l[1]; if (s.hasRight) { p.e20 = l[2]; p.e21 = l[3]; } }
        else if (s.hasRight) { p.e0 = l[2]; p.e1 = l[3]; }
        return p;
    }

    // A sweep stage of DFSPH / WCSPH: only owned particles are swept (ghost values arrive by halo).  With overlap the
    // edge layers go first, the halo of the stage's output starts, the interior follows.
    // reduce: the stage accumulates the exact |error| total of the owned particles (adaptive DFSPH).
    void sweepStage(int phase, const std::vector<int>& halo, bool reduce = false)
    {
        SLAB_TRACE("stage", phase);
        transport->wait();                      // the edges read ghost values written by the previous stage's halo
        const bool sweepGhosts = !overlap() && (flags & SPHX_SLAB_SWEEP_GHOSTS);
        bool anyEdge = false;
        for (auto& sp : slabs) anyEdge = anyEdge || sp->hasLeft || sp->hasRight;
        if (!overlap() || !anyEdge) {           // (no neighbour anywhere: one launch per slab, nothing to post)
            for (auto& sp : slabs) {
                Slab& s = *sp;
                if (sweepGhosts) s.sys->system->phaseEx(phase, -1, -1, reduce, s.o0, s.o1, false);
                else s.sys->system->phaseEx(phase, s.o0, s.o1, reduce, s.o0, s.o1, false);
            }
            postHalo(halo, false);
            return;
        }
        if (edgeStream && global.solver != SPHX_PBD) {
            // DFSPH and (r04) WCSPH: the edge layers and the interior of a stage are independent (both read the previous stage's output
            // -- neighbour velocities through the vel4 mirror of the stage before --, each writes its own particles), so the edges -- one small launch -- and the halo they feed run on a stream of their own BESIDE the
            // interior instead of in front of it.  fork: the edge stream starts where the engine stream stands (halo of the previous
            // stage arrived, previous interior done); join: the engine stream's next work waits for the edges and, with the loopback
            // transport, for the halo copies behind them.  An error stage zeroes its accumulators before the fork; both parts add.
            hipStream_t main = sphx::stream();
            if (reduce) for (auto& sp : slabs) sp->sys->system->resetErrorTotal();
            hip_ok(hipEventRecord(forkEvent, main), "event record");
            hip_ok(hipStreamWaitEvent(edgeStream, forkEvent, 0), "stream wait");
            {
                ScopedStream onEdges(edgeStream);
                for (auto& sp : slabs) {
                    Slab& s = *sp;
                    const StageParts p = stageParts(s);
                    if (p.e0 >= 0) s.sys->system->phaseEx(phase, p.e0, p.e1, reduce, s.o0, s.o1, true, p.e20, p.e21);
                }
                postHalo(halo, true);
                hip_ok(hipEventRecord(joinEvent, edgeStream), "event record");
            }
            for (auto& sp : slabs) {
                Slab& s = *sp;
                const StageParts p = stageParts(s);
                s.sys->system->phaseEx(phase, p.in0, p.in1, reduce, s.o0, s.o1, true);
            }
            hip_ok(hipStreamWaitEvent(main, joinEvent, 0), "stream wait");
            return;
        }
        for (auto& sp : slabs) {
            Slab& s = *sp;
            const StageParts p = stageParts(s);
            if (p.e0 >= 0) s.sys->system->phaseEx(phase, p.e0, p.e1, reduce, s.o0, s.o1, false, p.e20, p.e21);       // both edge layers in one launch
        }
        postHalo(halo, true);
        for (auto& sp : slabs) {
            Slab& s = *sp;
            const StageParts p = stageParts(s);
            s.sys->system->phaseEx(phase, p.in0, p.in1, reduce, s.o0, s.o1, p.e0 >= 0);
        }
    }

    // global |error| total of the last error stage as the fp32 value DFSPHSolver compares with its threshold
    float globalError()
    {
        long long local = 0;
        const auto t0 = std::chrono::steady_clock::now();
        for (auto& sp : slabs) local += sp->sys->system->errorTotalFixed();
        const long long total = transport->allreduce_sum(local);
        waitSeconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        return (float)((double)total * (1.0 / 4294967296.0));
    }

    void stepDfsph()
    {
        const int v = global.dfsph_fixed_div, d = global.dfsph_fixed_den;
        exchangeParticles();
        runAll(SPHX_PH_SEARCH);
        updateLayers();
        sweepStage(SPHX_PH_HEAD, {SPHX_F_KAPPA, SPHX_F_POSF});
        int itDiv = 0, itDen = 0;
        // fixed counts with at least one divergence correction: the gravity kick rides in the LAST correction's store (edges and interior
        // each kick their own particles; ghost velocities arrive kicked with that stage's halo), as in the whole-domain step
        const bool kickFused = !adaptive && v >= 1 && !sphx::tuning().no_kick_fusion && (overlap() || !(flags & SPHX_SLAB_SWEEP_GHOSTS));
        if (!adaptive) {
            const float3 G = make_float3(global.gravity[0], global.gravity[1], global.gravity[2]);
            for (; itDiv < v; ++itDiv) {
                const bool last = kickFused && itDiv + 1 == v;
                if (last) for (auto& sp : slabs) sp->sys->dfsph->setKickInCorrect(true, global.dt, G);
                sweepStage(SPHX_PH_DIV_CORRECT, {SPHX_F_VEL4});
                if (last) for (auto& sp : slabs) sp->sys->dfsph->setKickInCorrect(false);
                // (fixed counts: the error sweep behind the LAST correction -- and its halo -- has no reader, DFSPHSolver::step)
                if (itDiv + 1 < v) sweepStage(SPHX_PH_DIV_ERROR, {SPHX_F_KAPPA, SPHX_F_POSF});
            }
        } else {       // DFSPHSolver.cu:347-361
            const float limit = global.dfsph_divergence_thr * (float)nGlobal * global.rho0;
            float total = 3.4028235e38f;
            while ((itDiv < 1 || total > limit) && itDiv < global.dfsph_max_iter) {
                sweepStage(SPHX_PH_DIV_CORRECT, {SPHX_F_VEL4});
                sweepStage(SPHX_PH_DIV_ERROR, {SPHX_F_KAPPA, SPHX_F_POSF}, true);
                total = globalError();
                ++itDiv;
            }
        }
        if (!kickFused) runAll(SPHX_PH_FORCE);
        sweepStage(SPHX_PH_VISC_COLOR, surface ? std::vector<int>{SPHX_F_CG4} : std::vector<int>{});
        if (surface) {
            // (the fused sweep takes the neighbour's warm-start stiffness from posf.w, the record it gathers anyway: VISC_COLOR left it
            // there for the owned particles, the ghosts' -- known locally since the exchange -- are written here)
            for (auto& sp : slabs) { sp->sys->dfsph->packWarmIntoPosf(0, sp->o0); sp->sys->dfsph->packWarmIntoPosf(sp->o1, sp->held); }
            sweepStage(SPHX_PH_SURFACE_WARM, {SPHX_F_VEL4});      // one row walk for both, one halo instead of two
        } else {
            sweepStage(SPHX_PH_SURFACE, {SPHX_F_VEL4});
            sweepStage(SPHX_PH_WARM_CORRECT, {SPHX_F_VEL4});
        }
        sweepStage(SPHX_PH_DEN_ERROR_SET, {SPHX_F_KAPPA, SPHX_F_POSF});
        if (!adaptive) {
            for (; itDen < d; ++itDen) {
                sweepStage(SPHX_PH_DEN_CORRECT, {SPHX_F_VEL4});
                sweepStage(SPHX_PH_DEN_ERROR_ACC, itDen + 1 < d ? std::vector<int>{SPHX_F_KAPPA, SPHX_F_POSF} : std::vector<int>{});
            }
        } else {       // DFSPHSolver.cu:187-208
            const float limit = global.dfsph_density_thr * (float)nGlobal * global.rho0;
            float total = 3.4028235e38f;
            while ((itDen < 2 || total > limit) && itDen < global.dfsph_max_iter) {
                sweepStage(SPHX_PH_DEN_CORRECT, {SPHX_F_VEL4});
                ++itDen;
                const bool needTotal = itDen >= 2;
                sweepStage(SPHX_PH_DEN_ERROR_ACC, {SPHX_F_KAPPA, SPHX_F_POSF}, needTotal);
                if (needTotal) total = globalError();
            }
        }
        lastDiv = itDiv; lastDen = itDen;
        for (auto& sp : slabs) if (sp->sys->dfsph) sp->sys->dfsph->noteIterations(itDiv, itDen);
        runAll(SPHX_PH_ADVECT);
    }

    void stepWcsph()
    {
        exchangeParticles();
        runAll(SPHX_PH_W_SEARCH);
        updateLayers();
        // W_PROPS writes the colour gradient (read by W_SURFACE) and the pressure term (read by W_PRESSURE)
        sweepStage(SPHX_PH_W_PROPS, surface ? std::vector<int>{SPHX_F_CG4, SPHX_F_PTERM, SPHX_F_POSF} : std::vector<int>{SPHX_F_PTERM, SPHX_F_POSF});
        if (surface) {
            sweepStage(SPHX_PH_W_SURFACE_PRESSURE, {});
        } else {
            sweepStage(SPHX_PH_W_SURFACE, {});
            sweepStage(SPHX_PH_W_PRESSURE, {});
        }
        runAll(SPHX_PH_ADVECT);
    }

    // PBDSolver::step (PBDSolver.cu:34-79): two ghost columns, every stage on all held particles, halo after each
    // stage that writes a neighbour-read field.  The first call only sorts and records positions (PBDSolver.cu:45-49).
    // The owned particles of a slab as three disjoint ranges: left edge layer, interior, right edge layer (a PBD slab may be
    // narrower than its two edge layers together: the right edge then starts where the left one ends).
    struct Parts { int eL0, eL1, in0, in1, eR0, eR1; };
    static Parts parts(const Slab& s)
    {
        const int* l = s.layer;
        Parts p;
        p.eL0 = l[0]; p.eL1 = std::min(std::max(l[1], l[0]), l[3]);
        p.eR0 = std::max(std::min(l[2], l[3]), p.eL1); p.eR1 = l[3];
        p.in0 = p.eL1; p.in1 = p.eR0;
        return p;
    }
    void pbdInterior(int phase) { for (auto& sp : slabs) { const Parts p = parts(*sp); sp->sys->system->phaseEx(phase, p.in0, p.in1, false, 0, 0, false); } }
    void pbdEdges(int phase)
    {
        for (auto& sp : slabs) {
            const Parts p = parts(*sp);
            sp->sys->system->phaseEx(phase, p.eL0, p.eL1, false, 0, 0, false);
            sp->sys->system->phaseEx(phase, p.eR0, p.eR1, false, 0, 0, false);
        }
    }

    // PBD with halo traffic hidden behind the interior (r03).  Every stage sweeps its INTERIOR first -- interior particles
    // read owned particles only, so the halo posted by the previous stage may still be in flight -- then waits, sweeps the two
    // edge layers (which read ghosts) and posts the halo of its own output at once.  Jacobi semantics are kept: every delta-p
    // is computed (interior, then edges) before any position moves (PBDSolver.cu:225-258), the rows of a range
    // are rebuilt by the first stage that meets moved positions (P_LAMBDA, P_XSPH), ghosts are never swept.
    void stepPbdOverlapped()
    {
        exchangeParticles();
        runAll(SPHX_PH_P_SEARCH);
        updateLayers();
        if (slabs[0]->stepsDone == 0) return;
        for (int it = 0; it < global.pbd_iters; ++it) {
            pbdInterior(SPHX_PH_P_LAMBDA);               // (the position halo of the previous iteration is in flight)
            transport->wait();
            pbdEdges(SPHX_PH_P_LAMBDA);
            postHalo({SPHX_F_LAMBDA, SPHX_F_POSF}, true);
            pbdInterior(SPHX_PH_P_DELTA_SWEEP);
            transport->wait();
            pbdEdges(SPHX_PH_P_DELTA_SWEEP);
            pbdEdges(SPHX_PH_P_APPLY);                   // all delta-p of this slab are computed: positions may move now
            postHalo({SPHX_F_POS4}, true);
            pbdInterior(SPHX_PH_P_APPLY);
        }
        for (auto& sp : slabs) sp->sys->system->phaseEx(SPHX_PH_P_VELOCITY, sp->o0, sp->o1, false, 0, 0, false);
        postHalo({SPHX_F_VEL4}, true);                   // (orders the engine stream after the position halo first)
        pbdInterior(SPHX_PH_P_XSPH);
        transport->wait();
        pbdEdges(SPHX_PH_P_XSPH);
        if (surface) {
            postHalo({SPHX_F_CG4}, true);
            pbdInterior(SPHX_PH_P_SURFACE);
            transport->wait();
            pbdEdges(SPHX_PH_P_SURFACE);
        } else {
            // every XSPH sum has been formed: the new velocities become the live ones (owned particles)
            for (auto& sp : slabs) sp->sys->system->phaseEx(SPHX_PH_P_SURFACE, sp->o0, sp->o1, false, 0, 0, false);
        }
        runAll(SPHX_PH_P_TAIL);
    }

    void stepPbd()
    {
        if (overlap()) { stepPbdOverlapped(); return; }
        exchangeParticles();
        runAll(SPHX_PH_P_SEARCH);
        updateLayers();
        Slab& any = *slabs[0];
        const bool first = any.stepsDone == 0;
        if (first) return;
        for (int it = 0; it < global.pbd_iters; ++it) {
            runAll(SPHX_PH_P_LAMBDA); postHalo({SPHX_F_LAMBDA, SPHX_F_POSF}, false);
            runAll(SPHX_PH_P_DELTA); postHalo({SPHX_F_POS4}, false);
        }
        runAll(SPHX_PH_P_VELOCITY); postHalo({SPHX_F_VEL4}, false);
        runAll(SPHX_PH_P_XSPH);
        if (surface) postHalo({SPHX_F_CG4}, false);
        runAll(SPHX_PH_P_SURFACE);
        runAll(SPHX_PH_P_TAIL);
    }

    void step()
    {
        if (global.solver == SPHX_DFSPH) stepDfsph();
        else if (global.solver == SPHX_WCSPH) stepWcsph();
        else stepPbd();
        for (auto& sp : slabs) sp->stepsDone++;
    }
};

namespace {

int slab_fail(int code, const std::string& msg) { return sphx_fail(code, msg); }

template <class F>
int slab_guarded(const char* where, F&& body)
{
    try {
        return body();
    } catch (const SlabError& e) {
        return slab_fail(SPHX_ERR_STATE, std::string(where) + ": " + e.text);
    } catch (const char* msg) {
        return slab_fail(SPHX_ERR_STATE, std::string(where) + ": " + msg);
    } catch (const sphx::DeviceAllocError& e) {
        return slab_fail(SPHX_ERR_HIP, std::string(where) + ": " + e.what());
    } catch (const std::exception& e) {
        return slab_fail(SPHX_ERR_STATE, std::string(where) + ": " + e.what());
    } catch (...) {
        return slab_fail(SPHX_ERR_STATE, std::string(where) + ": unknown exception");
    }
}

// cut planes x_0 = 0 < x_1 < ... < x_world = gx balancing the particle counts; every slab at least minWidth columns
// (ghost width + 1: a particle that moves one column must be deliverable by its last owner to every rank that
// needs it, owner or ghost holder, with neighbour messages only)
std::vector<int> choose_cuts(const std::vector<int>& column, int gx, int world, int minWidth)
{
    std::vector<long long> cdf((size_t)gx, 0);
    for (int c : column) cdf[(size_t)std::min(std::max(c, 0), gx - 1)]++;
    for (int x = 1; x < gx; ++x) cdf[x] += cdf[x - 1];
    const long long total = cdf[gx - 1];
    std::vector<int> cuts{0};
    for (int r = 1; r < world; ++r) {
        const double target = (double)total * r / world;
        // the column in which the running count crosses the target goes to whichever side leaves the smaller error
        const int xc = (int)(std::lower_bound(cdf.begin(), cdf.end(), target, [](long long a, double t) { return (double)a < t; }) - cdf.begin());
        const double below = xc > 0 ? (double)cdf[(size_t)xc - 1] : 0.0, above = xc < gx ? (double)cdf[(size_t)xc] : (double)total;
        int x = (target - below < above - target) ? xc : xc + 1;
        x = std::max(x, cuts.back() + minWidth);
        x = std::min(x, gx - minWidth * (world - r));
        cuts.push_back(x);
    }
    cuts.push_back(gx);
    for (size_t k = 0; k + 1 < cuts.size(); ++k)
        if (cuts[k + 1] - cuts[k] < minWidth) die("domain too narrow for this many slabs");
    return cuts;
}

}  // namespace

extern "C" {

int sphx_slab_rccl_unique_id(char id128[128])
{
    if (!id128) return slab_fail(SPHX_ERR_INVALID, "sphx_slab_rccl_unique_id: null buffer");
    return slab_guarded("sphx_slab_rccl_unique_id", [&] {
        std::string why;
        if (!g_rccl.load(why)) die(why);
        ncclUniqueId id;
        nccl_ok(g_rccl.GetUniqueId(&id), "ncclGetUniqueId");
        std::memcpy(id128, &id, 128);
        return (int)SPHX_OK;
    });
}

// The decomposition of a scene: cut planes and the particle capacity every slab's engine is created with.
// capacity: twice the most particles any slab HOLDS at the start (owned + ghost columns, plus two columns: a cut may
// move towards it on either side) — the fluid piles up against a wall while the cuts follow it with a delay — but never
// more than the whole scene plus its ghost copies.  ~560 B of HBM per slot.  Pure host arithmetic on the global scene,
// so every rank arrives at the same numbers.
struct SlabPlan { std::vector<int> cuts; std::vector<long long> owned; long long capacity = 0; };
static SlabPlan plan_slabs(const sphx_params& P, const float* fluid_xyz, int n_fluid, int world)
{
    const int gx = P.cells[0];
    const int ghost = P.solver == SPHX_PBD ? 2 : 1;
    // columns by the engine's expression on the host (IEEE division, truncation)
    auto column_of = [&](float x) { volatile float q = x / P.cell_length; return (int)q; };
    std::vector<int> col((size_t)n_fluid);
    for (int i = 0; i < n_fluid; ++i) col[i] = column_of(fluid_xyz[3 * (size_t)i]);
    SlabPlan plan;
    plan.cuts = choose_cuts(col, gx, world, ghost + 1);
    std::vector<long long> perColumn((size_t)std::max(gx, 1), 0);
    for (int c : col) if (c >= 0 && c < gx) perColumn[c]++;
    plan.owned.assign((size_t)world, 0);
    long long most = 0;
    for (int r = 0; r < world; ++r) {
        long long heldHere = 0;
        for (int x = std::max(plan.cuts[r] - ghost, 0); x < std::min(plan.cuts[r + 1] + ghost, gx); ++x) heldHere += perColumn[x];
        for (int x = std::max(plan.cuts[r], 0); x < std::min(plan.cuts[r + 1], gx); ++x) plan.owned[r] += perColumn[x];
        most = std::max(most, heldHere);
    }
    const long long densestColumn = *std::max_element(perColumn.begin(), perColumn.end());
    most += 2 * densestColumn;
    plan.capacity = std::min<long long>(std::min<long long>(2 * most, (long long)n_fluid + 4LL * ghost * densestColumn) + 4096, 2000000000LL);
    return plan;
}

int sphx_slab_create(const sphx_params* params, const float* fluid_xyz, const float* fluid_vel, int n_fluid,
                     const float* boundary_xyz, int n_boundary, int world, int first_rank, int local_ranks,
                     const char* rccl_id128, int flags, sphx_slab_group** out)
{
    if (!params || !out || n_fluid < 0 || n_boundary < 0 || (n_fluid && !fluid_xyz) || (n_boundary && !boundary_xyz) || world < 1 ||
        first_rank < 0 || local_ranks < 1 || first_rank + local_ranks > world)
        return slab_fail(SPHX_ERR_INVALID, "sphx_slab_create: bad argument");
    if (params->reserved[3] < 0 || params->reserved[3] > 2)
        return slab_fail(SPHX_ERR_INVALID, "sphx_slab_create: reserved[3] (arithmetic) must be 0 (strict), 1 (tolerance) or 2 (tolerance with persistent rows)");
    if (!rccl_id128 && (first_rank != 0 || local_ranks != world))
        return slab_fail(SPHX_ERR_INVALID, "sphx_slab_create: without an RCCL token all slabs must be local (loopback)");
    if (rccl_id128 && (world % local_ranks != 0 || first_rank % local_ranks != 0))
        return slab_fail(SPHX_ERR_INVALID, "sphx_slab_create: with RCCL every process drives the same number of consecutive slabs");
    return slab_guarded("sphx_slab_create", [&] {
        *out = nullptr;
        std::unique_ptr<sphx_slab_group> G(new sphx_slab_group());
        const sphx_params& P = *params;
        G->global = P; G->world = world; G->flags = flags; G->nGlobal = n_fluid;
        G->surface = P.surface_tension > EPSILON || P.air_pressure > EPSILON;
        G->adaptive = P.solver == SPHX_DFSPH && (P.dfsph_fixed_div < 0 || P.dfsph_fixed_den < 0);
        const int gx = P.cells[0], gy = P.cells[1], gz = P.cells[2];
        const int ghost = P.solver == SPHX_PBD ? 2 : 1;
        // transport first: it binds this process to its device-side communicator
        if (rccl_id128) G->transport.reset(new RcclTransport(first_rank / local_ranks, world / local_ranks, rccl_id128, local_ranks));
        else G->transport.reset(new LoopbackTransport());
        // (r05: WCSPH too -- sweepStage has taken this path for it since r04, but the stream was only ever created for DFSPH)
        if (P.solver != SPHX_PBD && G->overlap()) G->createEdgeStream();
        G->dBad.alloc(2);
        hip_ok(hipHostMalloc((void**)&G->hBad, 2 * sizeof(long long), hipHostMallocDefault), "pinned failure word");
        G->hBad[0] = G->hBad[1] = 0;

        const SlabPlan plan = plan_slabs(P, fluid_xyz, n_fluid, world);
        const std::vector<int>& cuts = plan.cuts;
        auto column_of = [&](float x) { volatile float q = x / P.cell_length; return (int)q; };

        for (int r = first_rank; r < first_rank + local_ranks; ++r) {
            std::unique_ptr<Slab> S(new Slab());
            Slab& s = *S;
            s.rank = r; s.world = world; s.x0 = cuts[r]; s.x1 = cuts[r + 1]; s.ghost = ghost;
            s.widthLeft = r > 0 ? cuts[r] - cuts[r - 1] : 0; s.widthRight = r + 1 < world ? cuts[r + 2] - cuts[r + 1] : 0;      // (the planned widths: the reach of the first exchange)
            s.cellsPerColumn = gy * gz; s.gx = gx; s.cellLength = P.cell_length;
            s.solver = P.solver; s.hasLeft = r > 0; s.hasRight = r + 1 < world;
            s.extraFloats = P.solver == SPHX_DFSPH ? 1 : (P.solver == SPHX_PBD ? 3 : 0);
            s.capacity = (int)plan.capacity;
            // The slab's engine works on the WHOLE grid (cell tables are a few tens of MB even at 10 M particles) and
            // holds the whole boundary set, whose masses it computes like any system (SPHSystem.cu:69-71): only the
            // particles it is handed are local.  Cut planes are then just two numbers of this driver and may move.
            sphx_params Pl = P;
            Pl.reserved[1] = 0; Pl.reserved[2] = 1;          // a slab system: no initial fluid sort, stage-wise stepping
            // arithmetic contract of the slabs' sweeps: strict (0) or tolerance (1) as asked.  The rows of a slab are slices of the
            // single-device rows, so a tolerance run equals the single-device tolerance engine bit for bit as long as both pick the
            // same kernel variants (they depend on the particle count per device from 4 M particles on: SweepCache::ctx)
            Pl.reserved[3] = P.reserved[3] >= 1 ? 1 : 0;
            {
                std::vector<float> zeros((size_t)3 * s.capacity, 0.0f);
                const int rc = sphx_create_impl(&Pl, zeros.data(), s.capacity, boundary_xyz, n_boundary, 0, &s.sys);
                if (rc) die(std::string("slab engine: ") + sphx_last_error());
            }
            // engine arrays
            const auto f = s.sys->system->getFluids();
            s.pos = f->getPosPtr(); s.vel = f->getVelPtr(); s.ids = f->getIdPtr(); s.density = f->getDensityPtr();
            s.cellStart = s.sys->system->getCellStartFluid().addr();
            if (P.solver == SPHX_DFSPH) s.extra = s.sys->dfsph->getWarmStiffness().addr();
            if (P.solver == SPHX_PBD) s.extra = reinterpret_cast<float*>(s.sys->pbd->getPosLast().addr());
            // scratch
            const size_t cap = (size_t)s.capacity, W = (size_t)s.width();
            const size_t blocks = cap / kSlabBlock + 2;
            s.blockL.alloc(blocks); s.blockR.alloc(blocks); s.blockK.alloc(blocks); s.blockSums.alloc(3 * (blocks / 2048 + 2));
            s.violation.alloc(1); s.layerOut.alloc(8); s.counts.alloc(13); s.farCounts.alloc(8);
            hip_ok(hipHostMalloc((void**)&s.hFar, 8 * sizeof(long long), hipHostMallocDefault), "pinned far counts");
            s.own.alloc(cap * W); s.sendL.alloc(cap * W); s.sendR.alloc(cap * W); s.recvL.alloc(cap * W); s.recvR.alloc(cap * W);
            hip_ok(hipHostMalloc((void**)&s.hCounts, 13 * sizeof(long long), hipHostMallocDefault), "pinned counts");
            hip_ok(hipHostMalloc((void**)&s.hInts, 8 * sizeof(int), hipHostMallocDefault), "pinned ints");
            hip_ok(hipEventCreateWithFlags(&s.layersReady, hipEventDisableTiming), "event");
            { Slab* self = S.get(); s.sys->system->setAfterSortHook([self] { sphx_slab_group::readLayers(*self); }); }
            // initial distribution: this slab's particles in generation order, ids = global generation index
            std::vector<float> p0, v0; std::vector<int> id0;
            for (int i = 0; i < n_fluid; ++i) {
                const int ci = column_of(fluid_xyz[3 * (size_t)i]);
                if (ci >= s.x0 && ci < s.x1) {
                    p0.insert(p0.end(), {fluid_xyz[3 * (size_t)i], fluid_xyz[3 * (size_t)i + 1], fluid_xyz[3 * (size_t)i + 2]});
                    if (fluid_vel) v0.insert(v0.end(), {fluid_vel[3 * (size_t)i], fluid_vel[3 * (size_t)i + 1], fluid_vel[3 * (size_t)i + 2]});
                    id0.push_back(i);
                }
            }
            const int m = (int)id0.size();
            if (m > s.capacity) die("slab: capacity too small for the initial distribution");
            hipStream_t st = sphx::stream();
            if (m > 0) {
                hip_ok(hipMemcpyAsync(s.pos, p0.data(), sizeof(float) * p0.size(), hipMemcpyHostToDevice, st), "upload");
                if (fluid_vel) hip_ok(hipMemcpyAsync(s.vel, v0.data(), sizeof(float) * v0.size(), hipMemcpyHostToDevice, st), "upload");
                else hip_ok(hipMemsetAsync(s.vel, 0, sizeof(float3) * (size_t)m, st), "memset");
                hip_ok(hipMemcpyAsync(s.ids, id0.data(), sizeof(int) * id0.size(), hipMemcpyHostToDevice, st), "upload");
                if (P.solver == SPHX_DFSPH) hip_ok(hipMemsetAsync(s.extra, 0, sizeof(float) * (size_t)m, st), "memset");
                if (P.solver == SPHX_PBD) {
                    // PBDSolver.h:56-60: the first step records the positions.  PBD derives velocities from
                    // displacements, so initial velocities enter as last positions moved back by dt * vel.
                    if (fluid_vel) for (size_t t = 0; t < p0.size(); ++t) v0[t] = p0[t] - P.dt * v0[t];
                    hip_ok(hipMemcpyAsync(s.extra, fluid_vel ? v0.data() : p0.data(), sizeof(float) * p0.size(), hipMemcpyHostToDevice, st), "upload");
                }
            }
            hip_ok(hipStreamSynchronize(st), "upload sync");
            if (P.solver == SPHX_PBD) s.sys->pbd->markPosLastInitialized();
            s.o0 = 0; s.o1 = m; s.held = m;
            G->slabs.push_back(std::move(S));
        }
        *out = G.release();
        return (int)SPHX_OK;
    });
}

int sphx_slab_destroy(sphx_slab_group* g)
{
    if (!g) return SPHX_OK;
    (void)hipStreamSynchronize(sphx::stream());
    delete g;
    return SPHX_OK;
}

int sphx_slab_step(sphx_slab_group* g, int n, float* ms_total)
{
    if (!g || n < 0) return slab_fail(SPHX_ERR_INVALID, "sphx_slab_step: bad argument");
    if (g->failed) return slab_fail(SPHX_ERR_STATE, "sphx_slab_step: an earlier step of this group failed; destroy it (sphx_slab_destroy) and create a new one");
    return slab_guarded("sphx_slab_step", [&] {
        struct Guard {           // any exception leaves the group unusable: posted messages are dropped, never replayed
            sphx_slab_group* g; bool ok = false;
            ~Guard() { if (!ok) { g->failed = true; g->sends.clear(); g->recvs.clear(); } }
        } guard{g};
        const auto t0 = std::chrono::steady_clock::now();
        for (int k = 0; k < n; ++k) g->step();
        g->transport->wait();
        hip_ok(hipStreamSynchronize(sphx::stream()), "step sync");
        if (ms_total) *ms_total = (float)(std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() * 1e3);
        guard.ok = true;
        return (int)SPHX_OK;
    });
}

int sphx_slab_info(const sphx_slab_group* g, int index, int* x0, int* x1, int* owned, int* held)
{
    if (!g || index < 0 || index >= (int)g->slabs.size()) return slab_fail(SPHX_ERR_INVALID, "sphx_slab_info: bad argument");
    const Slab& s = *g->slabs[index];
    if (x0) *x0 = s.x0;
    if (x1) *x1 = s.x1;
    if (owned) *owned = s.o1 - s.o0;
    if (held) *held = s.held;
    return SPHX_OK;
}

int sphx_slab_gather(sphx_slab_group* g, int index, int capacity, int* ids, float* pos, float* vel, float* density, int* count)
{
    if (!g || index < 0 || index >= (int)g->slabs.size() || !count) return slab_fail(SPHX_ERR_INVALID, "sphx_slab_gather: bad argument");
    return slab_guarded("sphx_slab_gather", [&] {
        Slab& s = *g->slabs[index];
        const int m = s.o1 - s.o0;
        *count = m;
        if (m > capacity) die("host arrays too small");
        g->transport->wait();
        hipStream_t st = sphx::stream();
        if (m > 0) {
            if (ids) hip_ok(hipMemcpyAsync(ids, s.ids + s.o0, sizeof(int) * (size_t)m, hipMemcpyDeviceToHost, st), "gather");
            if (pos) hip_ok(hipMemcpyAsync(pos, s.pos + s.o0, sizeof(float3) * (size_t)m, hipMemcpyDeviceToHost, st), "gather");
            if (vel) hip_ok(hipMemcpyAsync(vel, s.vel + s.o0, sizeof(float3) * (size_t)m, hipMemcpyDeviceToHost, st), "gather");
            if (density) hip_ok(hipMemcpyAsync(density, s.density + s.o0, sizeof(float) * (size_t)m, hipMemcpyDeviceToHost, st), "gather");
        }
        hip_ok(hipStreamSynchronize(st), "gather sync");
        return (int)SPHX_OK;
    });
}

int sphx_slab_plan_cuts(const sphx_params* params, const float* fluid_xyz, int n_fluid, int world, int* cuts, long long* counts)
{
    if (!params || (n_fluid && !fluid_xyz) || n_fluid < 0 || world < 1 || !cuts) return slab_fail(SPHX_ERR_INVALID, "sphx_slab_plan_cuts: bad argument");
    return slab_guarded("sphx_slab_plan_cuts", [&] {
        const SlabPlan plan = plan_slabs(*params, fluid_xyz, n_fluid, world);
        for (int r = 0; r <= world; ++r) cuts[r] = plan.cuts[r];
        if (counts) for (int r = 0; r < world; ++r) counts[r] = plan.owned[r];
        return (int)SPHX_OK;
    });
}

int sphx_slab_plan_capacity(const sphx_params* params, const float* fluid_xyz, int n_fluid, int world, long long* capacity)
{
    if (!params || (n_fluid && !fluid_xyz) || n_fluid < 0 || world < 1 || !capacity) return slab_fail(SPHX_ERR_INVALID, "sphx_slab_plan_capacity: bad argument");
    return slab_guarded("sphx_slab_plan_capacity", [&] {
        *capacity = plan_slabs(*params, fluid_xyz, n_fluid, world).capacity;
        return (int)SPHX_OK;
    });
}

int sphx_slab_cut_rule(long long owned_left, long long owned_right, int width_left, int width_right, int ghost, float tolerance)
{
    return sphx_slab_group::cut_shift(owned_left, owned_right, width_left, width_right, ghost, tolerance);
}

int sphx_slab_set_rebalance(sphx_slab_group* g, int every_steps, float tolerance)
{
    if (!g || tolerance < 0.0f) return slab_fail(SPHX_ERR_INVALID, "sphx_slab_set_rebalance: bad argument");
    g->rebalanceEvery = every_steps;
    g->rebalanceTol = tolerance;
    return SPHX_OK;
}

int sphx_slab_iters(const sphx_slab_group* g, int* div, int* den)
{
    if (!g) return slab_fail(SPHX_ERR_INVALID, "sphx_slab_iters: null group");
    if (div) *div = g->lastDiv;
    if (den) *den = g->lastDen;
    return SPHX_OK;
}

int sphx_slab_system(const sphx_slab_group* g, int index, sphx_system** sys)
{
    if (!g || !sys || index < 0 || index >= (int)g->slabs.size()) return slab_fail(SPHX_ERR_INVALID, "sphx_slab_system: bad argument");
    *sys = g->slabs[index]->sys;
    return SPHX_OK;
}

int sphx_slab_comm_info(const sphx_slab_group* g, int* transport_kind, int* comm_ranks, int* comm_rank, long long counters4[4])
{
    if (!g || !g->transport) return slab_fail(SPHX_ERR_INVALID, "sphx_slab_comm_info: bad argument");
    int kind = 0, ranks = 0, rank = 0;
    g->transport->describe(kind, ranks, rank);
    if (transport_kind) *transport_kind = kind;
    if (comm_ranks) *comm_ranks = ranks;
    if (comm_rank) *comm_rank = rank;
    if (counters4) { counters4[0] = g->transport->bytesSent; counters4[1] = g->transport->bytesReceived; counters4[2] = g->transport->exchanges; counters4[3] = g->transport->allreduces; }
    return SPHX_OK;
}

int sphx_slab_wait_seconds(const sphx_slab_group* g, double* seconds)
{
    if (!g || !seconds) return slab_fail(SPHX_ERR_INVALID, "sphx_slab_wait_seconds: bad argument");
    *seconds = g->waitSeconds;
    return SPHX_OK;
}

}  // extern "C"
