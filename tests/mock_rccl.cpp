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
 * Mechanics: messages are staged through a POSIX shared-memory segment named by the unique id (host copies), and
 * everything is complete when ncclGroupEnd / ncclAllReduce returns.  Timing means nothing here; order and sizes do.
 *
 * Loaded through SPHX_RCCL_LIBRARY=<this .so> (tests/test_gpu_slab.py); never part of the product.
 */
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdint>
#include <cstdio>
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
constexpr double kTimeoutSeconds = 60.0;       // a protocol deadlock becomes an error, not a hung box

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

struct Op { bool send; char* dev; size_t bytes, done; int peer; Comm* comm; hipStream_t stream; bool headerSeen; };

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
        if (hipMemcpy(b.data, op.dev + op.done, n, hipMemcpyDeviceToHost) != hipSuccess) return fail("device-to-host copy failed");
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
        if (hipMemcpy(op.dev + op.done, b.data, n, hipMemcpyHostToDevice) != hipSuccess) return fail("host-to-device copy failed");
        b.consumed.fetch_add(1);
        op.done += n; moved = true;
    }
    (void)stage;
    return 0;
}

int run_group()
{
    std::vector<Op> ops;
    ops.swap(t_ops);
    for (const Op& o : ops) if (hipStreamSynchronize(o.stream) != hipSuccess) return fail("stream synchronise failed");
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

int post(bool send, void* buf, size_t count, int type, int peer, Comm* c, hipStream_t stream)
{
    if (!c || peer < 0 || peer >= c->world || peer == c->rank) return fail("bad peer");
    const size_t width = (type == 0 || type == 1) ? 1 : (type == 2 || type == 3 || type == 7) ? 4 : 8;
    if (count == 0) return fail("zero-byte message (RCCL would hang on an unmatched empty message)");
    t_ops.push_back(Op{send, (char*)buf, count * width, 0, peer, c, stream, false});
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

const char* ncclGetErrorString(int r) { return r == 0 ? "no error" : "mock_rccl error (see stderr)"; }

}  // extern "C"
