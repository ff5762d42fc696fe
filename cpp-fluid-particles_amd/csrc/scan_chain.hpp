// scan_chain.hpp — single-launch prefix sums over tiles (decoupled look-back, Merrill & Garland 2016) for gfx950.
//
// The grid pass of a step (SPHSystem::neighborSearch, src/SPHSystem.cu:114-127: thrust::exclusive_scan over the cell counts) used
// three launches per scan (tiles, tile totals, add offsets) plus a flag pass for the ranks of the out-of-grid bucket; small scenes
// are bound by launches, not by bytes.  Here a tile publishes its total, looks back over the tiles before it until it meets one
// whose inclusive prefix is known, and publishes its own: one launch, one read and one write of the data.
// Where it runs (measured, profiles/r06_scan_chain.txt): a look-back round costs ~3 us on this device, so the cell table goes
// through it only while all its tiles are resident at once (<= 128 tiles; 0.19 ms against 0.06 for the three passes at 3,677), and
// the out-of-grid ranks always (their launch leaves at once in a step without such particles: one launch instead of four).
//
// What makes this safe on a device whose eight L2s are not coherent with each other:
//   * a tile's state is ONE 64-bit word (generation << 34 | flag << 32 | value) moved with agent-scope atomics only -- no payload
//     behind a flag, so no fence pairs;
//   * tiles are numbered by an atomic ticket in the order their blocks START, so every tile a block waits for belongs to a block
//     that is already running (no dependence on the dispatch order of blockIdx);
//   * nothing is reset between launches: the generation (a device word the last block to finish advances, together with the
//     ticket) tags the states of a launch, so a replayed hipGraph needs no memset node in front of the scan;
//   * the spin is bounded: a tile that never sees its predecessors raises the fault word (pinned host memory, read by the host at
//     its next call) and leaves with what it has instead of hanging the device.
#pragma once

#include <hip/hip_runtime.h>
#include <stdexcept>
#ifdef SPHX_TEST_HOOKS
#include <cstdlib>
#endif

namespace sphx {

struct ScanChain {
    unsigned long long* state;   // one word per tile and channel
    unsigned int* ctl;           // [0] tickets handed out, [1] tiles finished, [2] generation of the launch in flight
    int* fault;                  // host-visible; non-zero once a look-back gave up
    int stride;                  // tiles per channel (state of channel c starts at c * stride)
#ifdef SPHX_TEST_HOOKS
    int dropTile;                // test build (tests/libsphx_hooks.so): this tile never publishes -- the tiles behind it must give up and report
#endif
};

constexpr unsigned int kChainSpinLimit = 1u << 20;      // a look-back normally ends within a few reads; this is seconds

// host-side owner of a chain's scratch: `tiles` states for each of `channels`, the control words, the pinned fault word
struct ChainScratch {
    ChainScratch(int tiles, int channels) : stride(tiles > 0 ? tiles : 1)
    {
        const size_t words = (size_t)stride * (size_t)(channels > 0 ? channels : 1);
        if (hipMalloc((void**)&state, sizeof(unsigned long long) * words) != hipSuccess ||
            hipMalloc((void**)&ctl, 4 * sizeof(unsigned int)) != hipSuccess ||
            hipHostMalloc((void**)&fault, sizeof(int), hipHostMallocDefault) != hipSuccess)
            throw std::runtime_error("scan chain: scratch allocation failed");
        *fault = 0;
        (void)hipMemset(state, 0, sizeof(unsigned long long) * words);
        (void)hipMemset(ctl, 0, 4 * sizeof(unsigned int));
    }
    ~ChainScratch()
    {
        if (state) (void)hipFree(state);
        if (ctl) (void)hipFree(ctl);
        if (fault) (void)hipHostFree(fault);
    }
    ChainScratch(const ChainScratch&) = delete;
    ChainScratch& operator=(const ChainScratch&) = delete;
#ifdef SPHX_TEST_HOOKS
    ScanChain chain() const
    {
        const char* drop = std::getenv("SPHX_CHAIN_DROP_TILE");
        return ScanChain{state, ctl, fault, stride, drop ? std::atoi(drop) : -1};
    }
#else
    ScanChain chain() const { return ScanChain{state, ctl, fault, stride}; }
#endif
    bool faulted() const { return fault && *fault != 0; }
    unsigned long long* state = nullptr;
    unsigned int* ctl = nullptr;
    int* fault = nullptr;
    int stride;
};

__device__ __forceinline__ unsigned long long chain_load(const unsigned long long* p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void chain_store(unsigned long long* p, unsigned long long v)
{
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// every thread of the block calls this first: the block's tile number and the launch's generation
__device__ __forceinline__ int chain_enter(const ScanChain& c, unsigned int& gen)
{
    __shared__ unsigned int entry[2];
    if (threadIdx.x == 0) {
        entry[1] = __hip_atomic_load(c.ctl + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        entry[0] = atomicAdd(c.ctl + 0, 1u);
    }
    __syncthreads();
    gen = entry[1];
    return (int)entry[0];
}

// every thread of the block calls this last (tiles = blocks of the launch): the last block to finish re-arms the chain
__device__ __forceinline__ void chain_leave(const ScanChain& c, unsigned int gen, int tiles)
{
    if (threadIdx.x != 0) return;
    const unsigned int done = atomicAdd(c.ctl + 1, 1u);
    if ((int)done == tiles - 1) {
        __hip_atomic_store(c.ctl + 0, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(c.ctl + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(c.ctl + 2, gen + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// All 64 lanes of ONE wave call this with the tile's total: returns the sum of the totals of all tiles before it (every lane).
__device__ __forceinline__ int chain_exclusive(const ScanChain& c, int channel, unsigned int gen, int tile, int total)
{
    unsigned long long* state = c.state + (size_t)channel * (size_t)c.stride;
    const int lane = (int)(threadIdx.x & 63);
    const unsigned long long tag = (unsigned long long)(gen & 0x3fffffffu) << 34;
#ifdef SPHX_TEST_HOOKS
    const bool publish = lane == 0 && tile != c.dropTile;
#else
    const bool publish = lane == 0;
#endif
    if (tile == 0) {
        if (publish) chain_store(state, tag | (2ull << 32) | (unsigned int)total);
        return 0;
    }
    if (publish) chain_store(state + tile, tag | (1ull << 32) | (unsigned int)total);
    int before = 0;
    unsigned int spins = 0;
    for (int look = tile - 1;;) {
        const int idx = look - lane;                                     // lane 0 reads the nearest tile
        const unsigned long long st = idx >= 0 ? chain_load(state + idx) : (tag | (2ull << 32));
        const unsigned int flag = (st >> 34) == (tag >> 34) ? (unsigned int)(st >> 32) & 3u : 0u;
        const unsigned long long known = __ballot(flag != 0), closed = __ballot(flag == 2u);
        const int stop = closed ? __builtin_ctzll(closed) : 63;          // nearest tile whose inclusive prefix is there
        const unsigned long long need = stop >= 63 ? ~0ull : ((2ull << stop) - 1ull);
        if ((known & need) != need) {
            if (++spins > kChainSpinLimit) { if (lane == 0) *c.fault = 1; break; }
            __builtin_amdgcn_s_sleep(1);
            continue;
        }
        int v = lane <= stop ? (int)(unsigned int)st : 0;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
        before += v;
        if (closed) break;
        look -= 64;
    }
    if (publish) chain_store(state + tile, tag | (2ull << 32) | (unsigned int)(before + total));
    return before;
}

}  // namespace sphx
