/*
 * mock_rccl.cpp — TEST INFRASTRUCTURE: a stand-in for the nine RCCL entry points csrc/slab.hip binds with dlsym,
 * so that the slab layer's RCCL transport (one process per slab, every neighbour remote) can run with 2+ ranks
 * on a box that has ONE GPU.  The real RCCL refuses two ranks on one device ("Duplicate GPU detected").
 *
 * It keeps RCCL's contract where the slab protocol could violate it and only differs in the mechanics:
 *   - ncclSend/ncclRecv between one pair of ranks match in issue order, and the two sides must name the SAME byte
 *     count (a mismatch, which the real library answers with a hang or silent corruption, fails loudly here);
 *   - the operations of one ncclGroupStart/End make progress together (no ordering between different peers);
 *   - an operation is ordered after the work enqueued on its stream before it.
 * Mechanics: messages are staged through a POSIX shared-memory segment named by the unique id (host copies).
 * Two modes:
 *   immediate (default)       everything is complete when ncclGroupEnd / ncclAllReduce returns.
 *   deferred  (SPHX_MOCK_RCCL_DEFER_US=<microseconds>)   like the real library, ncclGroupEnd only ENQUEUES: on the
 *       caller's stream go a spin kernel of that many microseconds, the device-to-host copies of the sends into pinned
 *       staging, ONE host callback that moves all of the group's messages through the mailboxes, and the
 *       host-to-device copies of the receives.  ncclGroupEnd returns before any byte has moved; data lands in the
 *       receive buffers (and is read from the send buffers) late and only in stream order.  A consumer that forgets to
 *       order its kernels after the transfer reads stale ghosts, a producer that overwrites a send buffer early
 *       ships wrong data -- both show up as wrong results (tests/test_gpu_slab.py has the negative test).
 *
 * Loaded through SPHX_RCCL_LIBRARY=<this .so> (tests/test_gpu_slab.py); never part of the product.
 */
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <vector>

namespace {

constexpr int kMaxRanks = 8;
constexpr size_t kChunk = 1u << 20;            // bytes of one mailbox
// a protocol deadlock becomes an error, not a hung box (SPHX_MOCK_RCCL_TIMEOUT_S: diagnostics)
const double kTimeoutSeconds = [] { const char* e = std::getenv("SPHX_MOCK_RCCL_TIMEOUT_S"); return e ? std::atof(e) : 60.0; }();

struct Box {                                   // one direction of one pair: single producer, single consumer
    std::atomic<uint64_t> written, consumed;   // chunk counters
    uint64_t messageBytes;                     // total size of the message the current chunk belongs to
    uint64_t chunkBytes;
    char data[kChunk];
};

struct Shared {
    std::atomic<int> ready;                    // set by the creator when the segment is initialised
    std::atomic<int> barrierCount, barrierSense;
    int world;
    long long reduceSlot[kMaxRanks];
    Box box[kMaxRanks][kMaxRanks];             // [from][to]
};

struct Comm {
    Shared* sh = nullptr;
    int rank = 0, world = 0;
    int localSense = 0;
    char name[64];
};

struct Op { bool send; char* dev; size_t bytes, done; int peer; Comm* comm; hipStream_t stream; bool headerSeen; char* host; };   // host: pinned staging (deferred mode)

thread_local int t_depth = 0;
thread_local std::vector<Op> t_ops;
std::atomic<int> g_ids{0};

double now() { timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }

int fail(const char* what)
{
    std::fprintf(stderr, "mock_rccl: %s\n", what);
    std::fflush(stderr);
    return 5;   // ncclInvalidUsage
}

bool barrier(Comm* c)
{
    Shared* s = c->sh;
    c->localSense ^= 1;
    if (s->barrierCount.fetch_add(1) + 1 == c->world) { s->barrierCount.store(0); s->barrierSense.store(c->localSense); return true; }
    const double t0 = now();
    while (s->barrierSense.load() != c->localSense) {
        if (now() - t0 > kTimeoutSeconds) return false;
        usleep(50);
    }
    return true;
}

// one attempt to move the next chunk of `op`; returns false when it cannot make progress right now
int progress(Op& op, std::vector<char>& stage, bool& moved)
{
    moved = false;
    Comm* c = op.comm;
    if (op.send) {
        Box& b = c->sh->box[c->rank][op.peer];
        if (b.written.load() != b.consumed.load()) return 0;             // the consumer still holds the previous chunk
        const size_t n = op.bytes - op.done < kChunk ? op.bytes - op.done : kChunk;
        if (op.host) std::memcpy(b.data, op.host + op.done, n);           // deferred mode: we are inside a stream callback, no HIP calls
        else if (hipMemcpy(b.data, op.dev + op.done, n, hipMemcpyDeviceToHost) != hipSuccess) return fail("device-to-host copy failed");
        b.messageBytes = op.bytes; b.chunkBytes = n;
        b.written.fetch_add(1);
        op.done += n; moved = true;
    } else {
        Box& b = c->sh->box[op.peer][c->rank];
        if (b.written.load() == b.consumed.load()) return 0;             // nothing there yet
        if (b.messageBytes != op.bytes) {
            std::fprintf(stderr, "mock_rccl: rank %d expects %zu bytes from rank %d, which sends %llu\n", c->rank, op.bytes, op.peer,
                         (unsigned long long)b.messageBytes);
            return fail("send/recv size mismatch");
        }
        const size_t n = b.chunkBytes;
        if (op.done + n > op.bytes) return fail("chunk overruns the receive buffer");
        if (op.host) std::memcpy(op.host + op.done, b.data, n);
        else if (hipMemcpy(op.dev + op.done, b.data, n, hipMemcpyHostToDevice) != hipSuccess) return fail("host-to-device copy failed");
        b.consumed.fetch_add(1);
        op.done += n; moved = true;
    }
    (void)stage;
    return 0;
}

int move_messages(std::vector<Op>& ops)
{
    std::vector<char> stage;
    double lastMove = now();
    for (;;) {
        bool all = true, any = false;
        for (size_t k = 0; k < ops.size(); ++k) {
            Op& o = ops[k];
            if (o.done == o.bytes) continue;
            all = false;
            // same pair, same direction: strictly in issue order
            bool blocked = false;
            for (size_t j = 0; j < k; ++j)
                if (ops[j].done != ops[j].bytes && ops[j].send == o.send && ops[j].peer == o.peer) { blocked = true; break; }
            if (blocked) continue;
            bool moved = false;
            const int r = progress(o, stage, moved);
            if (r) return r;
            any = any || moved;
        }
        if (all) return 0;
        if (any) lastMove = now();
        else {
            if (now() - lastMove > kTimeoutSeconds) {
                for (const Op& o : ops)
                    if (o.done != o.bytes)
                        std::fprintf(stderr, "mock_rccl: rank %d stuck in %s of %zu bytes %s rank %d (%zu done)\n", o.comm->rank,
                                     o.send ? "send" : "recv", o.bytes, o.send ? "to" : "from", o.peer, o.done);
                return fail("no progress: unmatched send/recv");
            }
            usleep(20);
        }
    }
}

// ---- deferred mode ------------------------------------------------------------------------------------------------
__global__ void k_mock_delay(long long ticks)
{
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
}
struct Deferred { std::vector<Op> ops; hipEvent_t done; std::vector<char*> pinned; int status; int sleepUs; };
std::vector<Deferred*> g_inflight;         // groups whose stream work may still be running (single host thread per process)
std::atomic<int> g_deferredError{0};

void deferred_callback(void* p)
{
    Deferred* d = (Deferred*)p;
    if (d->sleepUs > 0) usleep((useconds_t)d->sleepUs);
    d->status = move_messages(d->ops);
    if (d->status) g_deferredError.store(d->status);
}
void reap(bool all)
{
    for (size_t k = 0; k < g_inflight.size();) {
        Deferred* d = g_inflight[k];
        if (all) (void)hipEventSynchronize(d->done);
        if (all || hipEventQuery(d->done) == hipSuccess) {
            for (char* h : d->pinned) (void)hipHostFree(h);
            (void)hipEventDestroy(d->done);
            delete d;
            g_inflight.erase(g_inflight.begin() + (long)k);
        } else ++k;
    }
}
int defer_us()
{
    static const int us = [] { const char* e = std::getenv("SPHX_MOCK_RCCL_DEFER_US"); return e ? std::atoi(e) : 0; }();
    return us;
}

int run_group()
{
    if (g_deferredError.load()) return fail("an earlier deferred group failed");
    if (defer_us() <= 0) {
        std::vector<Op> ops;
        ops.swap(t_ops);
        for (const Op& o : ops) if (hipStreamSynchronize(o.stream) != hipSuccess) return fail("stream synchronise failed");
        return move_messages(ops);
    }
    reap(false);
    Deferred* d = new Deferred;
    d->ops.swap(t_ops);
    d->status = 0;
    if (d->ops.empty()) { delete d; return 0; }
    hipStream_t st = d->ops[0].stream;
    for (const Op& o : d->ops) if (o.stream != st) { delete d; return fail("deferred mode expects one stream per group"); }
    // the delay: a spin kernel on the stream (SPHX_MOCK_RCCL_SPIN_KERNEL=1, the r03 form) or -- default -- a sleep at the start of the
    // stream-ordered host callback below.  Either way the transfers complete that long after ncclGroupEnd returned, in stream order.
    // (r04: with 8 processes on the one test GPU a process now and then died of HSA_STATUS_ERROR_ILLEGAL_INSTRUCTION in its first step in
    // deferred mode only; the long-running one-thread spin kernel being pre-empted was the one thing the immediate mode does not have.)
    static const bool spinKernel = [] { const char* e = std::getenv("SPHX_MOCK_RCCL_SPIN_KERNEL"); return e && std::atoi(e) != 0; }();
    if (spinKernel) hipLaunchKernelGGL(k_mock_delay, dim3(1), dim3(1), 0, st, (long long)defer_us() * 100);     // wall_clock64: 100 MHz
    d->sleepUs = spinKernel ? 0 : defer_us();
    for (Op& o : d->ops) {
        if (hipHostMalloc((void**)&o.host, o.bytes, hipHostMallocDefault) != hipSuccess) return fail("pinned staging allocation failed");
        d->pinned.push_back(o.host);
        if (o.send && hipMemcpyAsync(o.host, o.dev, o.bytes, hipMemcpyDeviceToHost, st) != hipSuccess) return fail("async device-to-host copy failed");
    }
    if (hipLaunchHostFunc(st, deferred_callback, d) != hipSuccess) return fail("hipLaunchHostFunc failed");
    for (Op& o : d->ops)
        if (!o.send && hipMemcpyAsync(o.dev, o.host, o.bytes, hipMemcpyHostToDevice, st) != hipSuccess) return fail("async host-to-device copy failed");
    if (hipEventCreateWithFlags(&d->done, hipEventDisableTiming) != hipSuccess || hipEventRecord(d->done, st) != hipSuccess) return fail("event failed");
    g_inflight.push_back(d);
    return 0;
}

int post(bool send, void* buf, size_t count, int type, int peer, Comm* c, hipStream_t stream)
{
    if (!c || peer < 0 || peer >= c->world) return fail("bad peer");     // (peer == own rank: a send to self, as RCCL allows inside a group)
    const size_t width = (type == 0 || type == 1) ? 1 : (type == 2 || type == 3 || type == 7) ? 4 : 8;
    if (count == 0) return fail("zero-byte message (RCCL would hang on an unmatched empty message)");
    t_ops.push_back(Op{send, (char*)buf, count * width, 0, peer, c, stream, false, nullptr});
    if (t_depth == 0) return run_group();
    return 0;
}

}  // namespace

extern "C" {

struct ncclUniqueIdMock { char internal[128]; };

int ncclGetUniqueId(ncclUniqueIdMock* id)
{
    std::memset(id, 0, sizeof(*id));
    std::snprintf(id->internal, sizeof(id->internal), "/sphx_mock_rccl_%d_%d", (int)getpid(), g_ids.fetch_add(1));
    shm_unlink(id->internal);
    const int fd = shm_open(id->internal, O_CREAT | O_EXCL | O_RDWR, 0600);
    if (fd < 0) return fail("shm_open(create) failed");
    if (ftruncate(fd, sizeof(Shared)) != 0) { close(fd); return fail("ftruncate failed"); }
    void* p = mmap(nullptr, sizeof(Shared), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) return fail("mmap failed");
    Shared* s = (Shared*)p;                    // a fresh segment is zero-filled: every counter starts at 0
    s->ready.store(1);
    munmap(p, sizeof(Shared));
    return 0;
}

int ncclCommInitRank(Comm** out, int world, ncclUniqueIdMock id, int rank)
{
    if (world < 1 || world > kMaxRanks || rank < 0 || rank >= world) return fail("bad communicator geometry");
    const int fd = shm_open(id.internal, O_RDWR, 0600);
    if (fd < 0) return fail("shm_open failed (unknown unique id)");
    void* p = mmap(nullptr, sizeof(Shared), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) return fail("mmap failed");
    Comm* c = new Comm;
    c->sh = (Shared*)p; c->rank = rank; c->world = world;
    std::snprintf(c->name, sizeof(c->name), "%s", id.internal);
    if (rank == 0) c->sh->world = world;
    if (!barrier(c)) { delete c; return fail("not every rank joined the communicator"); }
    *out = c;
    return 0;
}

int ncclCommDestroy(Comm* c)
{
    if (!c) return 0;
    reap(true);
    (void)barrier(c);
    if (c->rank == 0) shm_unlink(c->name);
    munmap(c->sh, sizeof(Shared));
    delete c;
    return 0;
}

int ncclGroupStart() { ++t_depth; return 0; }
int ncclGroupEnd()
{
    if (t_depth <= 0) return fail("ncclGroupEnd without ncclGroupStart");
    if (--t_depth == 0) return run_group();
    return 0;
}

int ncclSend(const void* buf, size_t count, int type, int peer, Comm* c, hipStream_t stream) { return post(true, (void*)buf, count, type, peer, c, stream); }
int ncclRecv(void* buf, size_t count, int type, int peer, Comm* c, hipStream_t stream) { return post(false, buf, count, type, peer, c, stream); }

int ncclAllReduce(const void* in, void* out, size_t count, int type, int op, Comm* c, hipStream_t stream)
{
    if (!c || count != 1 || type != 4 || op != 0) return fail("the mock only reduces one int64 with ncclSum");
    if (g_deferredError.load()) return fail("an earlier deferred group failed");
    if (hipStreamSynchronize(stream) != hipSuccess) return fail("stream synchronise failed");
    long long v = 0;
    if (hipMemcpy(&v, in, sizeof(v), hipMemcpyDeviceToHost) != hipSuccess) return fail("device-to-host copy failed");
    c->sh->reduceSlot[c->rank] = v;
    if (!barrier(c)) return fail("all-reduce: a rank is missing");
    long long sum = 0;
    for (int r = 0; r < c->world; ++r) sum += c->sh->reduceSlot[r];
    if (!barrier(c)) return fail("all-reduce: a rank is missing");
    if (hipMemcpy(out, &sum, sizeof(sum), hipMemcpyHostToDevice) != hipSuccess) return fail("host-to-device copy failed");
    return 0;
}

int ncclCommCount(const Comm* c, int* count) { if (!c || !count) return fail("ncclCommCount: bad argument"); *count = c->world; return 0; }
int ncclCommUserRank(const Comm* c, int* rank) { if (!c || !rank) return fail("ncclCommUserRank: bad argument"); *rank = c->rank; return 0; }
const char* ncclGetErrorString(int r) { return r == 0 ? "no error" : "mock_rccl error (see stderr)"; }

}  // extern "C"
