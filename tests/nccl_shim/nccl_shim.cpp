// nccl_shim.cpp -- TEST INFRASTRUCTURE, not part of the product: a host-staged stand-in for the eleven RCCL entry points
// csrc/team.h resolves with dlsym (rccl_api()), so that the one-process-per-rank transport of team.h -- communicator
// bootstrap, ncclAllGather of the records, grouped ncclSend / ncclRecv of the halo, IPC handle exchange -- can run with
// SEVERAL rank processes on ONE GPU.  RCCL itself refuses two ranks on one device; the box the tests run on has one.
// Selected with MI355CG_RCCL_LIB=<this library> (tests/test_gpu_team_ranks.py); nothing under iterative_solvers_amd/ knows it.
//
// Ranks meet in a POSIX shared-memory segment named by the "unique id".  Every call is stream-synchronous: it waits for the
// stream, moves the bytes through the segment with the host, and returns when this rank's part is done.  That is a stricter
// schedule than RCCL's (which only enqueues), so anything that deadlocks here would be a real ordering bug; the converse does
// not hold, which is why bench.py still times the order-safe schedule first on real hardware.  Every wait is bounded.
//   NCCL_SHIM_HOST=1   buffers are host memory (memcpy instead of hipMemcpy, no stream): lets the CPU test suite run the shim.
//   NCCL_SHIM_TIMEOUT_MS (60000)
#include <hip/hip_runtime.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

extern "C" {
typedef enum { ncclSuccess = 0, ncclUnhandledCudaError = 1, ncclSystemError = 2, ncclInternalError = 3, ncclInvalidArgument = 4, ncclInvalidUsage = 5 } ncclResult_t;
typedef enum { ncclInt8 = 0, ncclUint8 = 1, ncclInt32 = 2, ncclUint32 = 3, ncclInt64 = 4, ncclUint64 = 5, ncclFloat16 = 6, ncclFloat32 = 7, ncclFloat64 = 8 } ncclDataType_t;
typedef struct { char internal[128]; } ncclUniqueId;
struct ShimComm;
typedef ShimComm* ncclComm_t;
}

namespace {

constexpr size_t kCollMax = 8192;            // bytes per rank of one collective
constexpr size_t kP2pMax = 256 * 1024;       // bytes of one point-to-point message
constexpr int kMaxWorld = 16;

struct Mailbox {                              // one per ordered pair (src, dst): a single message in flight
    std::atomic<uint64_t> written, read;
    uint64_t bytes;
    char data[kP2pMax];
};
struct Segment {
    std::atomic<int> attached;
    std::atomic<int> bar_count;
    std::atomic<int> bar_gen;
    int world;
    char coll[kMaxWorld][kCollMax];
    // Mailbox box[world * world] follows
};
size_t segment_bytes(int world) { return sizeof(Segment) + sizeof(Mailbox) * (size_t)world * world; }

bool host_mode() { const char* v = getenv("NCCL_SHIM_HOST"); return v && *v == '1'; }
double timeout_s() { const char* v = getenv("NCCL_SHIM_TIMEOUT_MS"); return (v && *v ? atof(v) : 60000.0) * 1e-3; }
size_t type_size(ncclDataType_t t) {
    switch (t) { case ncclInt8: case ncclUint8: return 1; case ncclFloat16: return 2; case ncclInt32: case ncclUint32: case ncclFloat32: return 4; default: return 8; }
}
struct Deadline {
    std::chrono::steady_clock::time_point end = std::chrono::steady_clock::now() + std::chrono::microseconds((long long)(timeout_s() * 1e6));
    bool expired() const { return std::chrono::steady_clock::now() > end; }
};
void relax(unsigned& spins) { if (++spins > 64) { std::this_thread::yield(); spins = 0; } }

struct Op { bool send; void* buf; size_t bytes; int peer; hipStream_t stream; ncclComm_t comm; bool done; };
thread_local int g_group_depth = 0;
thread_local std::vector<Op> g_ops;

}  // namespace

struct ShimComm {
    Segment* seg = nullptr;
    size_t bytes = 0;
    int world = 0, rank = 0;
    std::string name;
    Mailbox* box(int src, int dst) { return reinterpret_cast<Mailbox*>(reinterpret_cast<char*>(seg) + sizeof(Segment)) + (size_t)src * world + dst; }
};

namespace {

ncclResult_t sync_stream(hipStream_t st) {
    if (host_mode()) return ncclSuccess;
    return hipStreamSynchronize(st) == hipSuccess ? ncclSuccess : ncclUnhandledCudaError;
}
ncclResult_t to_host(void* dst, const void* src, size_t n) {
    if (host_mode()) { std::memcpy(dst, src, n); return ncclSuccess; }
    return hipMemcpy(dst, src, n, hipMemcpyDeviceToHost) == hipSuccess ? ncclSuccess : ncclUnhandledCudaError;
}
ncclResult_t to_device(void* dst, const void* src, size_t n) {
    if (host_mode()) { std::memcpy(dst, src, n); return ncclSuccess; }
    return hipMemcpy(dst, src, n, hipMemcpyHostToDevice) == hipSuccess ? ncclSuccess : ncclUnhandledCudaError;
}
ncclResult_t barrier(ShimComm* c) {
    Segment* s = c->seg;
    const int gen = s->bar_gen.load(std::memory_order_acquire);
    if (s->bar_count.fetch_add(1, std::memory_order_acq_rel) == c->world - 1) {
        s->bar_count.store(0, std::memory_order_relaxed);
        s->bar_gen.store(gen + 1, std::memory_order_release);
        return ncclSuccess;
    }
    Deadline d; unsigned spins = 0;
    while (s->bar_gen.load(std::memory_order_acquire) == gen) { if (d.expired()) return ncclSystemError; relax(spins); }
    return ncclSuccess;
}
// the queued point-to-point operations of this thread, progressed together: a send completes when the mailbox of its pair is
// free, a receive when the message at the head of its pair's mailbox has arrived; messages of one pair keep their order
ncclResult_t run_ops() {
    for (auto& o : g_ops) if (ncclResult_t r = sync_stream(o.stream)) return r;
    Deadline d; unsigned spins = 0;
    size_t left = g_ops.size();
    while (left) {
        bool progress = false;
        std::vector<int> blocked_send, blocked_recv;        // peers whose earlier operation of the same direction is still pending
        for (auto& o : g_ops) {
            if (o.done) continue;
            ShimComm* c = o.comm;
            auto& blocked = o.send ? blocked_send : blocked_recv;
            bool is_blocked = false;
            for (int b : blocked) if (b == o.peer) is_blocked = true;
            if (is_blocked) continue;
            Mailbox* m = o.send ? c->box(c->rank, o.peer) : c->box(o.peer, c->rank);
            const uint64_t w = m->written.load(std::memory_order_acquire), r = m->read.load(std::memory_order_acquire);
            if (o.send && w == r) {
                if (ncclResult_t e = to_host(m->data, o.buf, o.bytes)) return e;
                m->bytes = o.bytes;
                m->written.store(w + 1, std::memory_order_release);
                o.done = true; --left; progress = true;
            } else if (!o.send && w > r) {
                if (m->bytes != o.bytes) { fprintf(stderr, "nccl_shim: rank %d expected %zu bytes from %d, the message has %llu\n", c->rank, o.bytes, o.peer, (unsigned long long)m->bytes); return ncclInvalidArgument; }
                if (ncclResult_t e = to_device(o.buf, m->data, o.bytes)) return e;
                m->read.store(r + 1, std::memory_order_release);
                o.done = true; --left; progress = true;
            } else blocked.push_back(o.peer);
        }
        if (!progress) { if (d.expired()) return ncclSystemError; relax(spins); }
    }
    g_ops.clear();
    return ncclSuccess;
}

}  // namespace

extern "C" {

const char* ncclGetErrorString(ncclResult_t r) {
    switch (r) {
        case ncclSuccess: return "no error";
        case ncclUnhandledCudaError: return "shim: HIP call failed";
        case ncclSystemError: return "shim: timed out waiting for a peer";
        case ncclInvalidArgument: return "shim: invalid argument (message size mismatch / too large)";
        default: return "shim: error";
    }
}

ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
    if (!id) return ncclInvalidArgument;
    static std::atomic<int> counter{0};
    std::memset(id, 0, sizeof *id);
    const auto now = std::chrono::steady_clock::now().time_since_epoch().count();
    std::snprintf(id->internal, sizeof id->internal, "/mi355cg_shim_%d_%d_%llx", (int)getpid(), counter.fetch_add(1), (unsigned long long)now);
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t* out, int world, ncclUniqueId id, int rank) {
    if (!out || world < 1 || world > kMaxWorld || rank < 0 || rank >= world || id.internal[0] != '/') return ncclInvalidArgument;
    ShimComm* c = new ShimComm();
    c->world = world; c->rank = rank; c->name = id.internal; c->bytes = segment_bytes(world);
    const int fd = shm_open(c->name.c_str(), O_CREAT | O_RDWR, 0600);
    if (fd < 0 || ftruncate(fd, (off_t)c->bytes) != 0) { if (fd >= 0) close(fd); delete c; return ncclSystemError; }
    void* p = mmap(nullptr, c->bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);     // a fresh segment reads as zeros: every counter starts at 0
    close(fd);
    if (p == MAP_FAILED) { delete c; return ncclSystemError; }
    c->seg = static_cast<Segment*>(p);
    c->seg->world = world;
    c->seg->attached.fetch_add(1, std::memory_order_acq_rel);
    Deadline d; unsigned spins = 0;
    while (c->seg->attached.load(std::memory_order_acquire) < world) { if (d.expired()) { munmap(p, c->bytes); shm_unlink(c->name.c_str()); delete c; return ncclSystemError; } relax(spins); }
    if (ncclResult_t r = barrier(c)) { delete c; return r; }
    if (rank == 0) shm_unlink(c->name.c_str());          // everybody has it mapped: the name can go
    *out = c;
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t c) {
    if (!c) return ncclSuccess;
    if (c->seg) munmap(c->seg, c->bytes);
    delete c;
    return ncclSuccess;
}

ncclResult_t ncclCommCount(const ncclComm_t c, int* count) {
    if (!c || !count) return ncclInvalidArgument;
    *count = c->world;
    return ncclSuccess;
}

ncclResult_t ncclAllGather(const void* send, void* recv, size_t count, ncclDataType_t type, ncclComm_t c, hipStream_t stream) {
    if (!c || !send || !recv) return ncclInvalidArgument;
    const size_t bytes = count * type_size(type);
    if (bytes > kCollMax) return ncclInvalidArgument;
    if (ncclResult_t r = sync_stream(stream)) return r;
    if (ncclResult_t r = to_host(c->seg->coll[c->rank], send, bytes)) return r;
    if (ncclResult_t r = barrier(c)) return r;
    for (int j = 0; j < c->world; ++j) if (ncclResult_t r = to_device(static_cast<char*>(recv) + bytes * j, c->seg->coll[j], bytes)) return r;
    return barrier(c);                                   // nobody refills its slot before everybody has read it
}

ncclResult_t ncclBroadcast(const void* send, void* recv, size_t count, ncclDataType_t type, int root, ncclComm_t c, hipStream_t stream) {
    if (!c || !recv || root < 0 || root >= c->world) return ncclInvalidArgument;
    const size_t bytes = count * type_size(type);
    if (bytes > kCollMax) return ncclInvalidArgument;
    if (ncclResult_t r = sync_stream(stream)) return r;
    if (c->rank == root) if (ncclResult_t r = to_host(c->seg->coll[root], send, bytes)) return r;
    if (ncclResult_t r = barrier(c)) return r;
    if (ncclResult_t r = to_device(recv, c->seg->coll[root], bytes)) return r;
    return barrier(c);
}

ncclResult_t ncclGroupStart() { ++g_group_depth; return ncclSuccess; }
ncclResult_t ncclGroupEnd() {
    if (g_group_depth <= 0) return ncclInvalidUsage;
    if (--g_group_depth > 0) return ncclSuccess;
    const ncclResult_t r = run_ops();
    if (r != ncclSuccess) g_ops.clear();
    return r;
}
static ncclResult_t p2p(bool send, void* buf, size_t count, ncclDataType_t type, int peer, ncclComm_t c, hipStream_t stream) {
    if (!c || !buf || peer < 0 || peer >= c->world || peer == c->rank) return ncclInvalidArgument;
    const size_t bytes = count * type_size(type);
    if (bytes > kP2pMax) return ncclInvalidArgument;
    g_ops.push_back(Op{send, buf, bytes, peer, stream, c, false});
    if (g_group_depth > 0) return ncclSuccess;
    const ncclResult_t r = run_ops();
    if (r != ncclSuccess) g_ops.clear();
    return r;
}
ncclResult_t ncclSend(const void* buf, size_t count, ncclDataType_t type, int peer, ncclComm_t c, hipStream_t stream) { return p2p(true, const_cast<void*>(buf), count, type, peer, c, stream); }
ncclResult_t ncclRecv(void* buf, size_t count, ncclDataType_t type, int peer, ncclComm_t c, hipStream_t stream) { return p2p(false, buf, count, type, peer, c, stream); }

}  // extern "C"
