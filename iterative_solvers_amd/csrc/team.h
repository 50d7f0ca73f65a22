// team.h -- the native multi-GPU CG loop (included at the end of mi355cg.hip: one translation unit).
//
// The reference is single-process and has no collectives; this is the scaling surface of the same kernels.
// A TEAM is a decomposition of the grid into `world` parts -- row slabs balanced by unknown count, or a 2-D split whose
// x-cuts fall on 128-column strip boundaries -- plus a transport for the two things that cross parts per iteration:
//   * every part's 16-double record of partial sums / maxes, all-gathered after each of the two launches of an iteration;
//   * the boundary rows / columns of the RESIDUAL, neighbour to neighbour, once per iteration.  (The direction never
//     crosses parts: the fused stencil launch recomputes it on its halo from the ghost copies of r and the old direction and
//     keeps the result in its ghost rows and ghost columns.)
// Transports:
//   RCCL   one process per GPU.  The library owns its communicators (mi355cg_team_unique_id -> ncclCommInitRank); RCCL is
//          resolved at run time from the librccl already in the process (torch's) or from the ROCm installation.
//          The records: ncclAllGather issued on the COMPUTE stream, between producer and consumer launch (nothing else
//          could run there: both launches of an iteration need all records).  The halo: one ncclSend/ncclRecv group per
//          iteration on a SECOND stream and a second communicator, ordered with the compute stream by two events, so it
//          travels while the update records are all-gathered.  The host never blocks inside an iteration.
//   LOCAL  one process drives all parts, on one or several GPUs (one host thread per part when they are on different GPUs):
//          records are written straight into every part's gathered buffer, halo segments are device-to-device copies.
//          This is what a single-process host (the reference's DirichletSolver is one) uses, and what lets one GPU rehearse
//          an 8-part run bit for bit.
// Per iteration and part (default: ONE launch per phase; MI355CG_TEAM_SPLIT=1 cuts each phase into interior + edge launches
// so that the halo travels beside the interior items instead):
//     compute stream                                                       comm stream
//     wait halo ; stencil -> record A (last block of the launch)
//     all-gather A
//     update -> record B ; pack columns ---------------------------------> halo exchange of r ; unpack columns
//     all-gather B                                                        |
// Sums travel as double-double pairs and are reduced in part order by every consumer, so every decomposition takes
// bit-identical steps (tests/test_gpu_team.py).
#include <dlfcn.h>
#include <atomic>
#include <condition_variable>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>
#include <rccl/rccl.h>        // types and constants only: every RCCL function is resolved with dlsym

namespace {

struct Box { int y_lo, y_hi, s_lo, s_hi; };                  // owned rows (inclusive), owned 128-column strips [s_lo, s_hi)
struct Seg { int id, src, dst, kind, y0, y1, x0, x1; };      // halo message: kind 0 = cells [x0, x1) of row y0, 1 = column x0 over rows y0..y1
inline long long seg_count(const Seg& s) { return s.kind == 0 ? s.x1 - s.x0 : s.y1 - s.y0 + 1; }

// x-cuts of one y-slab into px pieces balanced by unknown count, on strip boundaries
void cut_strips(const GridParams& gp, int ya, int yb, int px, std::vector<int>& cuts) {
    const int ns_all = strips_total(gp, 2);
    const int nb = std::max(0, std::min(yb, gp.half) - ya + 1), nu = std::max(0, yb - std::max(ya, gp.half + 1) + 1);
    std::vector<long long> pre(ns_all + 1, 0);
    for (int s = 0; s < ns_all; ++s) {
        const int c0 = s * kStripCols, c1 = c0 + kStripCols;
        const long long cu = std::max(0, std::min(c1, gp.n) - std::max(c0, 1));
        const long long cbn = std::max(0, std::min(c1, gp.n) - std::max(c0, gp.half + 1));
        pre[s + 1] = pre[s] + nu * cu + nb * cbn;
    }
    cuts.assign(px + 1, 0);
    cuts[px] = ns_all;
    for (int k = 1; k < px; ++k) {
        const long long target = (pre[ns_all] * k + px / 2) / px;
        int s = 0;
        while (s < ns_all && pre[s] < target) ++s;
        if (s > 0 && target - pre[s - 1] < pre[s] - target) --s;      // nearer strip boundary
        cuts[k] = s;
    }
    for (int k = 1; k < px; ++k) cuts[k] = std::max(cuts[k], cuts[k - 1] + 1);
    for (int k = px - 1; k >= 1; --k) cuts[k] = std::min(cuts[k], cuts[k + 1] - 1);
}

// decomp 0: `world` row slabs.  decomp 1: py x px blocks, px = 2 when world is even (and the grid has >= 2 strips):
// y-cuts where every slab holds the same number of unknowns, then every slab cut in x where ITS unknowns halve
// (for 2 x 2 the lower, L-shaped slab is cut near x = 0.7 N, the upper one at x = N / 2).
int decompose(const GridParams& gp, int world, int decomp, std::vector<Box>& out) {
    const int ns_all = strips_total(gp, 2);
    int px = (decomp == 1 && world % 2 == 0 && ns_all >= 2) ? 2 : 1;
    const int py = world / px;
    if (py < 1 || py > gp.n - 1) return fail(MI355CG_ERR_INVALID, "cannot cut a %d-interval grid into %d parts", gp.n, world);
    out.clear();
    for (int ky = 0; ky < py; ++ky) {
        int ya = 0, yb = 0;
        if (int rc = mi355cg_slab_rows(gp.n, py, ky, &ya, &yb)) return rc;
        std::vector<int> cuts;
        cut_strips(gp, ya, yb, px, cuts);
        for (int kx = 0; kx < px; ++kx) out.push_back(Box{ya, yb, cuts[kx], cuts[kx + 1]});
    }
    return MI355CG_OK;
}

// Every message of one halo exchange, in an order all parts agree on.
std::vector<Seg> halo_segments(const GridParams& gp, const std::vector<Box>& bx) {
    std::vector<Seg> segs;
    const int Pu = (int)round_up(gp.n + 1, 32), cb = gp.half & ~31, s0b = first_bottom_strip(gp, 2);
    auto add = [&](int src, int dst, int kind, int y0, int y1, int x0, int x1) { segs.push_back(Seg{(int)segs.size(), src, dst, kind, y0, y1, x0, x1}); };
    for (int a = 0; a < (int)bx.size(); ++a) for (int b = 0; b < (int)bx.size(); ++b) {
        if (a == b) continue;
        const Box &A = bx[a], &B = bx[b];
        if (A.y_hi + 1 == B.y_lo) {                              // A below B: one row each way over the common strips
            const int sa = std::max(A.s_lo, B.s_lo), sb = std::min(A.s_hi, B.s_hi);
            if (sa < sb) {
                auto cols = [&](int y, int* x0, int* x1) { *x0 = std::max(sa * kStripCols, y <= gp.half ? cb : 0); *x1 = std::min(sb * kStripCols, Pu); };
                int x0, x1;
                cols(A.y_hi, &x0, &x1); if (x0 < x1) add(a, b, 0, A.y_hi, A.y_hi, x0, x1);
                cols(B.y_lo, &x0, &x1); if (x0 < x1) add(b, a, 0, B.y_lo, B.y_lo, x0, x1);
            }
        }
        if (A.s_hi == B.s_lo) {                                  // A left of B: one column each way over the common rows
            int ya = std::max(A.y_lo, B.y_lo);
            const int yb = std::min(A.y_hi, B.y_hi);
            if (A.s_hi <= s0b) ya = std::max(ya, gp.half + 1);   // in bottom-block rows the cut lies in the removed quadrant
            if (ya <= yb) {
                add(a, b, 1, ya, yb, A.s_hi * kStripCols - 1, A.s_hi * kStripCols);
                add(b, a, 1, ya, yb, A.s_hi * kStripCols, A.s_hi * kStripCols + 1);
            }
        }
    }
    return segs;
}

// ---- RCCL, resolved at run time -------------------------------------------------------------------------------------
struct RcclApi {
    void* lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
RcclApi* rccl_api() {
    static RcclApi api;
    static bool tried = false;
    if (tried) return api.lib ? &api : nullptr;
    tried = true;
    // the copy already in the process first (PyTorch-ROCm bundles its own), then the ROCm installation's
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void* h = dlopen(names[0], RTLD_NOW | RTLD_NOLOAD);
    for (int i = 0; !h && i < 3; ++i) h = dlopen(names[i], RTLD_NOW | RTLD_LOCAL);
    if (!h) return nullptr;
#define MI355CG_SYM(field, name) api.field = reinterpret_cast<decltype(api.field)>(dlsym(h, name)); if (!api.field) return nullptr
    MI355CG_SYM(GetUniqueId, "ncclGetUniqueId"); MI355CG_SYM(CommInitRank, "ncclCommInitRank"); MI355CG_SYM(CommDestroy, "ncclCommDestroy");
    MI355CG_SYM(AllGather, "ncclAllGather"); MI355CG_SYM(Send, "ncclSend"); MI355CG_SYM(Recv, "ncclRecv"); MI355CG_SYM(Broadcast, "ncclBroadcast");
    MI355CG_SYM(GroupStart, "ncclGroupStart"); MI355CG_SYM(GroupEnd, "ncclGroupEnd"); MI355CG_SYM(GetErrorString, "ncclGetErrorString");
#undef MI355CG_SYM
    api.lib = h;
    return &api;
}
#define NCCLCK(expr)                                                                                          \
    do {                                                                                                      \
        ncclResult_t r_ = (expr);                                                                             \
        if (r_ != ncclSuccess)                                                                                \
            return fail(MI355CG_ERR_HIP, "%s failed: %s (%s:%d)", #expr, rccl_api()->GetErrorString(r_), __FILE__, __LINE__); \
    } while (0)

// ---- small kernels of the team loop ------------------------------------------------------------------------------
constexpr int kMaxLocalParts = kMaxRecDst;
constexpr int kRecStop = kRecStopWord;
struct TeamRecArgs {
    const double* part; int n, stride;
    int nsum, lo_off, max_first, nmax;            // as RecordArgs
    const int* stop_req;                          // pinned host word (update records only), may be null
    double* dst[kMaxLocalParts]; int ndst;        // this part's slot in every local part's gathered buffer
};
__global__ __launch_bounds__(kBlock) void k_team_record(const TeamRecArgs a) {
    __shared__ double lds[2 * kWaves];
    __shared__ double rec[kRecHeader];
    if (threadIdx.x < kRecHeader) rec[threadIdx.x] = 0.0;
    __syncthreads();
    for (int f = 0; f < a.nsum; ++f) {
        const dd t = reduce_parts_dd(a.part + f * a.stride, a.part + (f + a.lo_off) * a.stride, a.n, 1, lds);
        if (threadIdx.x == 0) { rec[f] = t.hi; rec[f + a.lo_off] = t.lo; }
    }
    for (int f = a.max_first; f < a.max_first + a.nmax; ++f) {
        const double t = reduce_parts<true>(a.part + f * a.stride, a.n, 1, lds);
        if (threadIdx.x == 0) rec[f] = t;
    }
    if (threadIdx.x == 0 && a.stop_req) rec[kRecStop] = *(const volatile int*)a.stop_req ? 1.0 : 0.0;
    __syncthreads();
    for (int i = threadIdx.x; i < a.ndst * kRecHeader; i += kBlock) a.dst[i / kRecHeader][i % kRecHeader] = rec[i % kRecHeader];
}
// max over ranks of the stop word -> summary (after k_check wrote the rest of it)
__global__ void k_team_stop(const double* gathered, int nranks, CgState* summary) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        double m = 0.0;
        for (int i = 0; i < nranks; ++i) m = fmax(m, gathered[i * kRecHeader + kRecStop]);
        summary->pad_ = m > 0.0 ? 1 : 0;
    }
}
// column messages: gather a vector's columns into the send buffer / scatter the receive buffer into ghost columns
constexpr int kMaxColSegs = 8;
struct ColSeg { int x, y0, n; long long off; };
struct ColArgs { Geom g; double* v; double* buf; ColSeg s[kMaxColSegs]; int ns, scatter; };
__global__ __launch_bounds__(kBlock) void k_cols(const ColArgs a) {
    for (int k = 0; k < a.ns; ++k) {
        ColSeg s = a.s[0];
#pragma unroll
        for (int j = 1; j < kMaxColSegs; ++j) if (j == k) s = a.s[j];
        for (int i = blockIdx.x * kBlock + threadIdx.x; i < s.n; i += gridDim.x * kBlock) {
            double* cell = a.v + (row_off(a.g, s.y0 + i) - a.g.base0 + s.x);
            if (a.scatter) *cell = a.buf[s.off + i]; else a.buf[s.off + i] = *cell;
        }
    }
}

struct TeamPart {
    mi355cg_ctx* c = nullptr;
    int rank = 0;
    hipStream_t comm = nullptr;
    double *gA = nullptr, *gB = nullptr;                  // gathered records [world][kRecHeader]
    double *send_cols = nullptr, *recv_cols = nullptr;    // packed column messages
    double **dstA = nullptr, **dstB = nullptr;            // device arrays: where this part's records go (one slot per local part)
    std::vector<Seg> sends, recvs;                        // ordered by (peer, id)
    std::vector<long long> send_off, recv_off;            // column messages: offset in send_cols / recv_cols
    ColArgs pack{}, unpack{};
    bool split = false;                                   // interior / edge launches (the part has neighbours)
    hipEvent_t ev_recA = nullptr, ev_gA = nullptr, ev_redge = nullptr, ev_recB = nullptr, ev_gB = nullptr, ev_halo = nullptr;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> comm_pairs;     // profiling: event pairs on the comm stream
};

}  // namespace

struct mi355cg_team_s {
    GridParams gp;
    int world = 1, decomp = 0;
    std::vector<Box> boxes;
    std::vector<Seg> segs;
    std::vector<TeamPart> parts;            // the parts this process drives (all of them: LOCAL; one: RCCL)
    bool rccl = false;
    ncclComm_t comm = nullptr;              // records: ncclAllGather issued on the COMPUTE stream (no cross-stream hop on the critical path)
    ncclComm_t comm_halo = nullptr;         // halo messages on the comm stream; its own communicator, so the two streams never serialise on one
    hipStream_t hub = nullptr;              // LOCAL: joins the parts' record events
    hipEvent_t ev_hub = nullptr;
    int* stop_h = nullptr;                  // pinned: this process's stop request, read by k_team_record
    // Interior / edge launches per phase (the halo travels while the interior items run) or ONE launch per phase (the halo
    // travels while the update records are all-gathered).  See team_solve; MI355CG_TEAM_SPLIT=1 selects the former.
    bool split_phases = false;
    // RCCL: the halo group on the COMPUTE stream, between the update launch and the all-gather of its records (no second
    // stream, no events), instead of on the comm stream beside them.  MI355CG_TEAM_HALO_INLINE.
    bool halo_inline = false;
    bool profiling = false;
    double prof_kernel_ms = 0, prof_comm_ms = 0, prof_wall_ms = 0;     // per iteration, last profiled solve
    int hub_device = 0;
};

namespace {

void team_free(mi355cg_team_s* t) {
    if (!t) return;
    for (auto& p : t->parts) {
        if (p.c) hipSetDevice(p.c->device);
        if (p.c && p.c->stream) hipStreamSynchronize(p.c->stream);
        if (p.comm) { hipStreamSynchronize(p.comm); }
    }
    if (t->comm_halo && rccl_api()) rccl_api()->CommDestroy(t->comm_halo);
    if (t->comm && rccl_api()) rccl_api()->CommDestroy(t->comm);
    for (auto& p : t->parts) {
        if (p.c) hipSetDevice(p.c->device);
        for (void* q : {(void*)p.gA, (void*)p.gB, (void*)p.send_cols, (void*)p.recv_cols, (void*)p.dstA, (void*)p.dstB}) if (q) hipFree(q);
        for (hipEvent_t e : {p.ev_recA, p.ev_gA, p.ev_redge, p.ev_recB, p.ev_gB, p.ev_halo}) if (e) hipEventDestroy(e);
        if (p.comm) hipStreamDestroy(p.comm);
        if (p.c) mi355cg_destroy(p.c);
    }
    if (t->hub) { hipSetDevice(t->hub_device); hipStreamDestroy(t->hub); }
    if (t->ev_hub) hipEventDestroy(t->ev_hub);
    if (t->stop_h) hipHostFree(t->stop_h);
    delete t;
}

// Per-part resources and halo lists once the contexts exist.
int team_finish_setup(mi355cg_team_s* t) {
    t->segs = halo_segments(t->gp, t->boxes);
    // Events that only order streams of ONE device need no system-scope fence (a default event writes back and invalidates the
    // caches when it fires: ~10 us on the compute stream between two launches).  Parts on several GPUs read each other's rows
    // after these events: those keep the default.
    bool one_device = true;
    for (auto& p : t->parts) if (p.c->device != t->parts[0].c->device) one_device = false;
    const unsigned ev_flags = hipEventDisableTiming | (one_device && env_int("MI355CG_TEAM_EVENT_FENCE", 0) == 0 ? hipEventDisableSystemFence : 0u);
    for (auto& p : t->parts) {
        mi355cg_ctx* c = p.c;
        HIPCK(hipSetDevice(c->device));
        HIPCK(hipStreamCreateWithFlags(&p.comm, hipStreamNonBlocking));
        if (int rc = alloc_vec(&p.gA, (long long)t->world * kRecHeader)) return rc;
        if (int rc = alloc_vec(&p.gB, (long long)t->world * kRecHeader)) return rc;
        for (hipEvent_t* e : {&p.ev_recA, &p.ev_gA, &p.ev_redge, &p.ev_recB, &p.ev_gB, &p.ev_halo}) HIPCK(hipEventCreateWithFlags(e, ev_flags));
        for (auto& s : t->segs) { if (s.src == p.rank) p.sends.push_back(s); if (s.dst == p.rank) p.recvs.push_back(s); }
        auto by_peer = [](bool send) { return [send](const Seg& a, const Seg& b) { const int pa = send ? a.dst : a.src, pb = send ? b.dst : b.src; return pa != pb ? pa < pb : a.id < b.id; }; };
        std::sort(p.sends.begin(), p.sends.end(), by_peer(true));
        std::sort(p.recvs.begin(), p.recvs.end(), by_peer(false));
        p.split = !p.sends.empty() || !p.recvs.empty();
        long long so = 0, ro = 0;
        p.pack = ColArgs{}; p.unpack = ColArgs{};
        p.pack.g = c->g; p.unpack.g = c->g; p.unpack.scatter = 1;
        for (auto& s : p.sends) { p.send_off.push_back(so); if (s.kind == 1) { if (p.pack.ns >= kMaxColSegs) return fail(MI355CG_ERR_INVALID, "too many column messages"); p.pack.s[p.pack.ns++] = ColSeg{s.x0, s.y0, (int)seg_count(s), so}; so += seg_count(s); } }
        for (auto& s : p.recvs) { p.recv_off.push_back(ro); if (s.kind == 1) { if (p.unpack.ns >= kMaxColSegs) return fail(MI355CG_ERR_INVALID, "too many column messages"); p.unpack.s[p.unpack.ns++] = ColSeg{s.x0, s.y0, (int)seg_count(s), ro}; ro += seg_count(s); } }
        if (int rc = alloc_vec(&p.send_cols, std::max<long long>(so, 1))) return rc;
        if (int rc = alloc_vec(&p.recv_cols, std::max<long long>(ro, 1))) return rc;
        p.pack.buf = p.send_cols; p.unpack.buf = p.recv_cols;
        HIPCK(hipDeviceSynchronize());
    }
    for (auto& p : t->parts) {                                     // every part's record goes to every local part's gathered buffer
        HIPCK(hipSetDevice(p.c->device));
        std::vector<double*> da, db;
        for (auto& q : t->parts) { da.push_back(q.gA + (size_t)p.rank * kRecHeader); db.push_back(q.gB + (size_t)p.rank * kRecHeader); }
        HIPCK(hipMalloc((void**)&p.dstA, sizeof(double*) * da.size()));
        HIPCK(hipMalloc((void**)&p.dstB, sizeof(double*) * db.size()));
        HIPCK(hipMemcpy(p.dstA, da.data(), sizeof(double*) * da.size(), hipMemcpyHostToDevice));
        HIPCK(hipMemcpy(p.dstB, db.data(), sizeof(double*) * db.size(), hipMemcpyHostToDevice));
    }
    if (!t->rccl) {
        t->hub_device = t->parts[0].c->device;
        HIPCK(hipSetDevice(t->hub_device));
        HIPCK(hipStreamCreateWithFlags(&t->hub, hipStreamNonBlocking));
        HIPCK(hipEventCreateWithFlags(&t->ev_hub, ev_flags));
        // parts on different GPUs of one process reach each other's memory directly (xGMI peer access)
        for (auto& a : t->parts) for (auto& b : t->parts) if (a.c->device != b.c->device) {
            int can = 0;
            HIPCK(hipDeviceCanAccessPeer(&can, a.c->device, b.c->device));
            if (!can) return fail(MI355CG_ERR_HIP, "device %d cannot access device %d: a LOCAL team needs peer access", a.c->device, b.c->device);
            HIPCK(hipSetDevice(a.c->device));
            const hipError_t e = hipDeviceEnablePeerAccess(b.c->device, 0);
            if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) HIPCK(e);
            (void)hipGetLastError();
        }
    }
    HIPCK(hipHostMalloc((void**)&t->stop_h, sizeof(int)));
    *t->stop_h = 0;
    return MI355CG_OK;
}

double* seg_ptr(const mi355cg_ctx* c, double* v, const Seg& s) { return v + (row_off(c->g, s.y0) - c->g.base0 + s.x0); }

// all-gather of the records of phase `which` (0 = A, 1 = B): after it every part's compute stream may read its gathered buffer
int team_exchange_records(mi355cg_team_s* t, int which) {
    if (t->rccl) {
        // in-stream: the compute stream itself carries the all-gather between the record kernel and the consumer launch
        TeamPart& p = t->parts[0];
        double* g = which == 0 ? p.gA : p.gB;
        hipEvent_t e0 = nullptr, e1 = nullptr;
        if (t->profiling) { e0 = p.c->events.get(); if (e0) hipEventRecord(e0, p.c->stream); }
        if (t->world > 1 || env_int("MI355CG_FORCE_COLLECTIVES", 0))
            NCCLCK(rccl_api()->AllGather(g + (size_t)p.rank * kRecHeader, g, kRecHeader, ncclDouble, t->comm, p.c->stream));     // in place
        if (t->profiling && e0) { e1 = p.c->events.get(); if (e1) { hipEventRecord(e1, p.c->stream); p.comm_pairs.push_back({e0, e1}); } }
        return MI355CG_OK;
    }
    if (t->parts.size() == 1) return MI355CG_OK;                   // same stream wrote the record
    for (auto& p : t->parts) HIPCK(hipStreamWaitEvent(t->hub, which == 0 ? p.ev_recA : p.ev_recB, 0));
    HIPCK(hipEventRecord(t->ev_hub, t->hub));
    for (auto& p : t->parts) HIPCK(hipStreamWaitEvent(p.c->stream, t->ev_hub, 0));
    return MI355CG_OK;
}

int part_halo_in(mi355cg_team_s* t, TeamPart& p);
// boundary rows / columns of r to the neighbours; ev_halo of every part fires when its ghost cells are in place
int team_exchange_halo(mi355cg_team_s* t) {
    if (t->rccl) {
        TeamPart& p = t->parts[0];
        const hipStream_t hs = t->halo_inline ? p.c->stream : p.comm;
        if (!t->halo_inline) HIPCK(hipStreamWaitEvent(p.comm, p.ev_redge, 0));
        hipEvent_t e0 = nullptr, e1 = nullptr;
        if (t->profiling) { e0 = p.c->events.get(); if (e0) hipEventRecord(e0, hs); }
        if (!p.sends.empty() || !p.recvs.empty()) {
            NCCLCK(rccl_api()->GroupStart());
            for (size_t i = 0; i < p.sends.size(); ++i) {
                const Seg& s = p.sends[i];
                const double* src = s.kind == 0 ? seg_ptr(p.c, p.c->r, s) : p.send_cols + p.send_off[i];
                NCCLCK(rccl_api()->Send(src, (size_t)seg_count(s), ncclDouble, s.dst, t->comm_halo, hs));
            }
            for (size_t i = 0; i < p.recvs.size(); ++i) {
                const Seg& s = p.recvs[i];
                double* dst = s.kind == 0 ? seg_ptr(p.c, p.c->r, s) : p.recv_cols + p.recv_off[i];
                NCCLCK(rccl_api()->Recv(dst, (size_t)seg_count(s), ncclDouble, s.src, t->comm_halo, hs));
            }
            NCCLCK(rccl_api()->GroupEnd());
            if (p.unpack.ns) { ColArgs a = p.unpack; a.v = p.c->r; hipLaunchKernelGGL(k_cols, dim3(16), dim3(kBlock), 0, hs, a); }
        }
        if (t->profiling && e0) { e1 = p.c->events.get(); if (e1) { hipEventRecord(e1, hs); p.comm_pairs.push_back({e0, e1}); } }
        if (!t->halo_inline) HIPCK(hipEventRecord(p.ev_halo, p.comm));
        HIPCK(hipGetLastError());
        return MI355CG_OK;
    }
    for (auto& p : t->parts) { HIPCK(hipSetDevice(p.c->device)); if (int rc = part_halo_in(t, p)) return rc; }      // p = destination
    return MI355CG_OK;
}

// the record a producer launch writes itself when it ends a phase (cg_kernels.h: arrive_and_record)
RecSpec team_rec_spec(mi355cg_team_s* t, TeamPart& p, int which, int nslots) {
    RecSpec rs{};
    rs.enabled = 1; rs.nslots = nslots; rs.ticket = p.c->ticket; rs.stop_req = which == 1 ? t->stop_h : nullptr;
    rs.ndst = (int)t->parts.size(); rs.dst = which == 0 ? p.dstA : p.dstB;
    return rs;
}

// the same record from a separate one-block launch (after the initialisation pass)
void team_record(mi355cg_team_s* t, TeamPart& p, int which, int nslots) {
    mi355cg_ctx* c = p.c;
    TeamRecArgs a{};
    if (which == 0) { a.part = c->partA; a.stride = c->strideA; a.nsum = kNumSumsA; a.lo_off = FA_LO; a.max_first = 0; a.nmax = 0; }
    else { a.part = c->partB; a.stride = c->strideB; a.nsum = kNumSumsB; a.lo_off = FB_LO; a.max_first = FB_RMAX; a.nmax = 3; a.stop_req = t->stop_h; }
    a.n = nslots;
    a.ndst = 0;
    for (auto& q : t->parts) a.dst[a.ndst++] = (which == 0 ? q.gA : q.gB) + (size_t)p.rank * kRecHeader;
    hipLaunchKernelGGL(k_team_record, dim3(1), dim3(kBlock), 0, c->stream, a);
    if (!t->rccl) hipEventRecord(which == 0 ? p.ev_recA : p.ev_recB, c->stream);
}

PartSrc team_gsrc(const mi355cg_team_s* t, TeamPart& p, int which) { return PartSrc{which == 0 ? p.gA : p.gB, t->world, 1, kRecHeader}; }

// ---- one part's share of an iteration (called by the one driving thread, or by the part's own thread) --------------
// stencil phase: the launch that ends it writes the part's record (its last block)
int part_stencil_phase(mi355cg_team_s* t, TeamPart& p, const IterCfg& cfg) {
    mi355cg_ctx* c = p.c;
    hipEvent_t e0 = nullptr;
    if (p.split && !t->split_phases && !t->halo_inline) HIPCK(hipStreamWaitEvent(c->stream, p.ev_halo, 0));      // one launch: the halo has to be there first
    prof_begin(c, &e0);
    if (p.split && t->split_phases) {
        const RecSpec rs = team_rec_spec(t, p, 0, c->interior.grid + c->edge.grid);
        launch_iteration_stencil<double, 2>(c, cfg, c->r, c->p, Where{c->stream, &c->interior, 0}, team_gsrc(t, p, 1));
        prof_end(c, 0, e0);
        HIPCK(hipStreamWaitEvent(c->stream, p.ev_halo, 0));
        prof_begin(c, &e0);
        launch_iteration_stencil<double, 2>(c, cfg, c->r, c->p, Where{c->stream, &c->edge, c->interior.grid}, team_gsrc(t, p, 1), &rs);
    } else {
        const RecSpec rs = team_rec_spec(t, p, 0, c->whole.grid);
        launch_iteration_stencil<double, 2>(c, cfg, c->r, c->p, whole_part(c), team_gsrc(t, p, 1), &rs);
    }
    prof_end(c, 0, e0);
    c->cur = (c->cur + 1) % c->xsteps;
    if (!t->rccl) HIPCK(hipEventRecord(p.ev_recA, c->stream));
    return MI355CG_OK;
}
// update phase: edge items first, so the halo of r is on its way while the interior is updated
int part_update_phase(mi355cg_team_s* t, TeamPart& p, const IterCfg& cfg) {
    mi355cg_ctx* c = p.c;
    hipEvent_t e0 = nullptr;
    prof_begin(c, &e0);
    if (p.split && t->split_phases) {
        const RecSpec rs = team_rec_spec(t, p, 1, c->interior.grid + c->edge.grid);
        const bool has_int = c->interior.wl.nitems > 0;
        launch_iteration_update<double, 2>(c, cfg, c->x, c->r, c->p, c->u, Where{c->stream, &c->edge, c->interior.grid}, team_gsrc(t, p, 0), has_int ? nullptr : &rs);
        if (p.pack.ns) { ColArgs a = p.pack; a.v = c->r; hipLaunchKernelGGL(k_cols, dim3(16), dim3(kBlock), 0, c->stream, a); }
        prof_end(c, 1, e0);
        HIPCK(hipEventRecord(p.ev_redge, c->stream));
        prof_begin(c, &e0);
        launch_iteration_update<double, 2>(c, cfg, c->x, c->r, c->p, c->u, Where{c->stream, &c->interior, 0}, team_gsrc(t, p, 0), &rs);
    } else {
        const RecSpec rs = team_rec_spec(t, p, 1, c->whole.grid);
        launch_iteration_update<double, 2>(c, cfg, c->x, c->r, c->p, c->u, whole_part(c), team_gsrc(t, p, 0), &rs);
        if (p.pack.ns) { ColArgs a = p.pack; a.v = c->r; hipLaunchKernelGGL(k_cols, dim3(16), dim3(kBlock), 0, c->stream, a); }
        if (!t->halo_inline) HIPCK(hipEventRecord(p.ev_redge, c->stream));
    }
    prof_end(c, 1, e0);
    if (!t->rccl) HIPCK(hipEventRecord(p.ev_recB, c->stream));
    HIPCK(hipGetLastError());
    return MI355CG_OK;
}
// LOCAL: fetch this part's incoming halo segments (device-to-device copies on its comm stream); ev_halo fires when they are in place
int part_halo_in(mi355cg_team_s* t, TeamPart& p) {
    int last_src = -1;
    for (size_t i = 0; i < p.recvs.size(); ++i) {
        const Seg& s = p.recvs[i];
        TeamPart* q = nullptr;
        for (auto& o : t->parts) if (o.rank == s.src) q = &o;
        if (!q) return fail(MI355CG_ERR_STATE, "part %d is not in this process", s.src);
        if (s.src != last_src) { HIPCK(hipStreamWaitEvent(p.comm, q->ev_redge, 0)); last_src = s.src; }
        const double* src = nullptr;
        if (s.kind == 0) src = seg_ptr(q->c, q->c->r, s);
        else for (size_t j = 0; j < q->sends.size(); ++j) if (q->sends[j].id == s.id) src = q->send_cols + q->send_off[j];
        double* dst = s.kind == 0 ? seg_ptr(p.c, p.c->r, s) : p.recv_cols + p.recv_off[i];
        HIPCK(hipMemcpyAsync(dst, src, sizeof(double) * seg_count(s), hipMemcpyDefault, p.comm));
    }
    if (p.unpack.ns) { ColArgs a = p.unpack; a.v = p.c->r; hipLaunchKernelGGL(k_cols, dim3(16), dim3(kBlock), 0, p.comm, a); }
    HIPCK(hipEventRecord(p.ev_halo, p.comm));
    HIPCK(hipGetLastError());
    return MI355CG_OK;
}
// the decision of the last iteration, for the host (every part evaluates it: all parts hold the same records)
int part_poll_enqueue(mi355cg_team_s* t, TeamPart& p, const IterCfg& cfg) {
    launch_check(p.c, cfg, p.c->stream, team_gsrc(t, p, 1));
    hipLaunchKernelGGL(k_team_stop, dim3(1), dim3(64), 0, p.c->stream, p.gB, t->world, p.c->summary);
    return MI355CG_OK;
}

// ---- LOCAL transport with one host thread per part -------------------------------------------------------------------
// One thread issues ~20 runtime calls per part and iteration; with the parts on 8 different GPUs that is the host, not the GPUs,
// setting the pace.  Here every part has its own thread; the threads meet at two barriers per
// iteration, each right after the events the other parts are about to wait on have been recorded (an event has to be
// RECORDED before another thread may enqueue a wait on it).  Everything between the barriers is the part's own stream work.
struct TeamCrew {
    mi355cg_team_s* t = nullptr;
    IterCfg cfg{};
    int nthreads = 0;
    std::atomic<int> arrived{0}, generation{0};
    std::atomic<int> error{0};
    std::string error_text;
    std::mutex mu;                       // chunk hand-over (workers sleep between chunks: the host waits for the device there)
    std::condition_variable cv;
    int chunk_seq = 0, chunk_m = 0, done_count = 0;
    bool chunk_poll = false, quit = false;

    void note_error(int rc) { int zero = 0; if (error.compare_exchange_strong(zero, rc)) { std::lock_guard<std::mutex> g(mu); error_text = g_err; } }
    // spinning barrier for the two meeting points inside an iteration (microseconds apart); gives way when there are more
    // threads than cores; returns false once any thread has failed
    bool barrier() {
        const int gen = generation.load(std::memory_order_acquire);
        if (arrived.fetch_add(1, std::memory_order_acq_rel) == nthreads - 1) {
            arrived.store(0, std::memory_order_relaxed);
            generation.store(gen + 1, std::memory_order_release);
        } else {
            int spins = 0;
            while (generation.load(std::memory_order_acquire) == gen) {
                if (error.load(std::memory_order_relaxed)) return false;
                if (++spins > 256) { std::this_thread::yield(); spins = 0; }
            }
        }
        return error.load(std::memory_order_relaxed) == 0;
    }
    // the iterations of one chunk as seen by part i
    int run_chunk(int i, int m, bool poll) {
        TeamPart& p = t->parts[i];
        for (int k = 0; k < m; ++k) {
            if (int rc = part_stencil_phase(t, p, cfg)) return rc;
            if (!barrier()) return MI355CG_ERR_STATE;
            for (auto& q : t->parts) if (&q != &p) HIPCK(hipStreamWaitEvent(p.c->stream, q.ev_recA, 0));
            if (int rc = part_update_phase(t, p, cfg)) return rc;
            if (!barrier()) return MI355CG_ERR_STATE;
            if (int rc = part_halo_in(t, p)) return rc;
            for (auto& q : t->parts) if (&q != &p) HIPCK(hipStreamWaitEvent(p.c->stream, q.ev_recB, 0));
        }
        if (poll) if (int rc = part_poll_enqueue(t, p, cfg)) return rc;
        return MI355CG_OK;
    }
    void worker(int i) {
        if (hipSetDevice(t->parts[i].c->device) != hipSuccess) { fail(MI355CG_ERR_HIP, "hipSetDevice failed in a team thread"); note_error(MI355CG_ERR_HIP); }
        int seen = 0;
        for (;;) {
            int m; bool poll;
            {
                std::unique_lock<std::mutex> g(mu);
                cv.wait(g, [&] { return quit || chunk_seq != seen; });
                if (quit) return;
                seen = chunk_seq; m = chunk_m; poll = chunk_poll;
            }
            if (!error.load()) if (int rc = run_chunk(i, m, poll)) { if (rc != MI355CG_ERR_STATE || !error.load()) note_error(rc); }
            { std::lock_guard<std::mutex> g(mu); ++done_count; }
            cv.notify_all();
        }
    }
    // called by the solving thread (which is part 0's thread): run m iterations on every part, return when all are enqueued
    int chunk(int m, bool poll) {
        { std::lock_guard<std::mutex> g(mu); chunk_m = m; chunk_poll = poll; done_count = 0; ++chunk_seq; }
        cv.notify_all();
        if (hipSetDevice(t->parts[0].c->device) != hipSuccess) { fail(MI355CG_ERR_HIP, "hipSetDevice failed"); note_error(MI355CG_ERR_HIP); }
        if (!error.load()) if (int rc = run_chunk(0, m, poll)) { if (rc != MI355CG_ERR_STATE || !error.load()) note_error(rc); }
        { std::unique_lock<std::mutex> g(mu); cv.wait(g, [&] { return done_count == nthreads - 1; }); }
        if (const int rc = error.load()) { g_err = error_text; return rc; }
        return MI355CG_OK;
    }
};

int team_solve(mi355cg_team_s* t, const mi355cg_params* prm, mi355cg_iter_cb cb, void* user,
               const volatile int* stop_flag, mi355cg_results* out) {
    if (prm->rule != MI355CG_RULE_MSG_MAXNORM && prm->rule != MI355CG_RULE_REL_2NORM) return fail(MI355CG_ERR_INVALID, "unknown rule %d", prm->rule);
    if (prm->diagnostics) return fail(MI355CG_ERR_INVALID, "per-iteration diagnostics are not available on a team");
    const bool msg = prm->rule == MI355CG_RULE_MSG_MAXNORM;
    const IterCfg cfg = make_cfg(prm);             // has_u: u is read on every iteration here (the single-GPU path skips it where unobservable)
    const int W = t->world;
    const auto t0 = std::chrono::steady_clock::now();
    *t->stop_h = 0;
    t->split_phases = env_int("MI355CG_TEAM_SPLIT", 0) != 0;
    t->halo_inline = t->rccl && !t->split_phases && env_int("MI355CG_TEAM_HALO_INLINE", 0) != 0;

    // x = 0, r = b, z = 0; partial norms of r0; first record + halo of r0 = b
    for (auto& p : t->parts) {
        mi355cg_ctx* c = p.c;
        HIPCK(hipSetDevice(c->device));
        if (cfg.has_u) if (int rc = ensure_u_on_device(c)) return rc;
        c->events.reset(); c->ev_pairs[0].clear(); c->ev_pairs[1].clear(); p.comm_pairs.clear();
        c->profiling = t->profiling;
        // one pass over the owned range (mi355cg_solve does the same); the ghost cells of the first direction are zeroed too:
        // they still hold the neighbours' last direction of the previous solve
        HIPCK(hipMemsetAsync(c->p[0], 0, sizeof(double) * c->storage_len, c->stream));
        c->cur = 0;
        {
            FreshArgs<double> f{};
            f.begin = c->g.own_begin / 2; f.nvec = c->g.own_len / 2;
            f.b = c->b; f.x = c->x; f.r = c->r; f.p0 = c->p[0]; f.u = c->u;
            f.partB = c->partB; f.strideB = c->strideB; f.s_out = c->sB;
            if (cfg.has_u) hipLaunchKernelGGL((k_init_fresh<double, 2, true>), dim3(c->whole.grid), dim3(kBlock), 0, c->stream, f);
            else hipLaunchKernelGGL((k_init_fresh<double, 2, false>), dim3(c->whole.grid), dim3(kBlock), 0, c->stream, f);
        }
        if (p.pack.ns) { ColArgs a = p.pack; a.v = c->r; hipLaunchKernelGGL(k_cols, dim3(16), dim3(kBlock), 0, c->stream, a); }
        if (!t->halo_inline) HIPCK(hipEventRecord(p.ev_redge, c->stream));
        team_record(t, p, 1, c->whole.grid);
        HIPCK(hipGetLastError());
        c->solved = true;
    }
    if (int rc = team_exchange_halo(t)) return rc;
    if (int rc = team_exchange_records(t, 1)) return rc;

    TeamPart& lead = t->parts[0];
    auto poll_fetch = [&]() -> int {
        HIPCK(hipSetDevice(lead.c->device));
        HIPCK(hipMemcpyAsync(lead.c->summary_h, lead.c->summary, sizeof(CgState), hipMemcpyDeviceToHost, lead.c->stream));
        HIPCK(hipMemcpyAsync(lead.c->hist_h, lead.c->hist, sizeof(HistEntry) * kHist, hipMemcpyDeviceToHost, lead.c->stream));
        HIPCK(hipGetLastError());
        HIPCK(hipStreamSynchronize(lead.c->stream));
        return MI355CG_OK;
    };
    auto poll = [&]() -> int {
        for (auto& p : t->parts) { HIPCK(hipSetDevice(p.c->device)); if (int rc = part_poll_enqueue(t, p, cfg)) return rc; }
        return poll_fetch();
    };
    // LOCAL transport with several parts: one thread per part (MI355CG_TEAM_THREADS=0: the one-thread loop)
    std::unique_ptr<TeamCrew> crew;
    std::vector<std::thread> crew_threads;
    auto stop_crew = [&]() {
        if (!crew) return;
        { std::lock_guard<std::mutex> g(crew->mu); crew->quit = true; }
        crew->cv.notify_all();
        for (auto& th : crew_threads) th.join();
        crew_threads.clear(); crew.reset();
    };
    struct CrewGuard { std::function<void()> f; ~CrewGuard() { f(); } } crew_guard{stop_crew};      // every way out joins the threads
    // Default: threads when the parts live on different GPUs.  Parts sharing ONE GPU (the rehearsals of tests/ and tools/) gain
    // nothing -- their launches queue up on the same device whoever issues them (measured: 0.45 vs 0.42 ms per iteration with 4
    // parts of N = 4096, 0.77 vs 0.59 with 8) -- so they keep the one-thread loop.  MI355CG_TEAM_THREADS=0 | 1 overrides.
    bool several_devices = false;
    for (auto& p : t->parts) if (p.c->device != t->parts[0].c->device) several_devices = true;
    if (!t->rccl && t->parts.size() > 1 && env_int("MI355CG_TEAM_THREADS", several_devices ? 1 : 0) != 0) {
        crew.reset(new TeamCrew);
        crew->t = t; crew->cfg = cfg; crew->nthreads = (int)t->parts.size();
        for (int i = 1; i < crew->nthreads; ++i) crew_threads.emplace_back([&, i] { crew->worker(i); });
    }
    *lead.c->summary_h = CgState{};
    if (msg && cb) {                       // the state of iteration 0 is only fetched for its callback (msg_solver.cpp:75-77)
        if (int rc = poll()) return rc;
        cb(user, 0, DBL_MAX, lead.c->summary_h->rmax, cfg.has_u ? lead.c->summary_h->emax : DBL_MAX);
    }

    const int every = prm->callback_every;
    int sync_every = std::min(prm->sync_every > 0 ? prm->sync_every : (msg ? 100 : 200), kHist);
    int it_done = 0;
    bool interrupted = false, first_chunk = cb != nullptr || stop_flag != nullptr;
    if (!lead.c->ev_loop[0]) { HIPCK(hipEventCreate(&lead.c->ev_loop[0])); HIPCK(hipEventCreate(&lead.c->ev_loop[1])); }
    HIPCK(hipSetDevice(lead.c->device));
    HIPCK(hipEventRecord(lead.c->ev_loop[0], lead.c->stream));
    while (!lead.c->summary_h->done) {
        // A stop request has to reach every rank at the same iteration (a rank that left the loop alone would leave the
        // others waiting in a collective).  One process: act on it at once, like the reference's per-iteration check
        // (msg_solver.cpp:82-87).  Several: the flag travels with the update records of ONE more iteration and every rank
        // finds it in its summary at the poll that follows.
        const bool want_stop = stop_flag && *stop_flag;
        if (lead.c->summary_h->pad_ || (want_stop && (!t->rccl || t->world == 1))) { interrupted = true; break; }
        if (want_stop) *t->stop_h = 1;
        int m = std::min(sync_every, prm->max_iterations - it_done);
        if (msg && every > 0) m = std::min(m, every - it_done % every);
        if (first_chunk || want_stop) { m = 1; first_chunk = false; }
        if (m <= 0) m = 1;
        if (crew) {
            if (int rc = crew->chunk(m, true)) return rc;
            if (int rc = poll_fetch()) return rc;
        } else {
            for (int k = 0; k < m; ++k) {
                for (auto& p : t->parts) { HIPCK(hipSetDevice(p.c->device)); if (int rc = part_stencil_phase(t, p, cfg)) return rc; }
                if (int rc = team_exchange_records(t, 0)) return rc;
                for (auto& p : t->parts) { HIPCK(hipSetDevice(p.c->device)); if (int rc = part_update_phase(t, p, cfg)) return rc; }
                if (int rc = team_exchange_halo(t)) return rc;
                if (int rc = team_exchange_records(t, 1)) return rc;
            }
            if (int rc = poll()) return rc;
        }
        const int it_now = lead.c->summary_h->it;
        if (msg && cb) for (int it = it_done + 1; it <= it_now; ++it) {
            const bool stopped_here = lead.c->summary_h->done && lead.c->summary_h->reason != MI355CG_STOP_ITERATIONS && it == it_now;
            const HistEntry& h = lead.c->hist_h[it % kHist];
            if ((it == 1 || (every > 0 && it % every == 0)) && !stopped_here) cb(user, it, h.dmax, h.rmax, cfg.has_u ? h.emax : DBL_MAX);
        }
        it_done = it_now;
    }
    HIPCK(hipSetDevice(lead.c->device));
    HIPCK(hipEventRecord(lead.c->ev_loop[1], lead.c->stream));
    const CgState fin = *lead.c->summary_h;
    for (auto& p : t->parts) {
        mi355cg_ctx* c = p.c;
        HIPCK(hipSetDevice(c->device));
        if (!t->halo_inline) HIPCK(hipStreamWaitEvent(c->stream, p.ev_halo, 0));           // the last halo exchange writes this part's ghost cells
        c->cur = fin.it % c->xsteps;
        if (cfg.x2) launch_flush_x<double, 2>(c, c->whole, c->x, c->p, fin, c->stream);
        HIPCK(hipGetLastError());
    }
    for (auto& p : t->parts) { HIPCK(hipSetDevice(p.c->device)); HIPCK(hipStreamSynchronize(p.c->stream)); HIPCK(hipStreamSynchronize(p.comm)); }
    const double wall = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (t->profiling) {
        mi355cg_ctx* c = lead.c;
        double comm = 0;
        for (auto& pr : lead.comm_pairs) { float ms = 0; if (hipEventElapsedTime(&ms, pr.first, pr.second) == hipSuccess) comm += ms; }
        lead.comm_pairs.clear();
        prof_collect(c);
        const double its = std::max(1, fin.it);
        t->prof_kernel_ms = (c->kernel_ms[0] * c->kernel_launches[0] + c->kernel_ms[1] * c->kernel_launches[1]) / its;
        t->prof_comm_ms = comm / its;
        t->prof_wall_ms = 1e3 * wall / its;
        for (auto& p : t->parts) { p.c->profiling = false; p.c->events.reset(); p.c->ev_pairs[0].clear(); p.c->ev_pairs[1].clear(); }
    }
    mi355cg_results res{};
    res.iterations = fin.it;
    res.converged = interrupted ? 0 : fin.converged;
    res.stop_reason = interrupted ? MI355CG_STOP_INTERRUPTED : fin.reason;
    res.final_residual_norm = fin.rmax;
    res.final_precision = fin.it > 0 ? fin.dmax : DBL_MAX;
    res.final_error_norm = cfg.has_u ? fin.emax : DBL_MAX;
    res.r_norm2 = fin.rnorm2; res.initial_r_norm2 = fin.r0norm;
    res.solve_seconds = wall;
    { float ms = 0; if (hipEventElapsedTime(&ms, lead.c->ev_loop[0], lead.c->ev_loop[1]) == hipSuccess) res.loop_seconds = 1e-3 * ms; }
    if (msg && cb) cb(user, res.iterations, res.final_precision, res.final_residual_norm, res.final_error_norm);
    if (out) *out = res;
    return MI355CG_OK;
}

}  // namespace

extern "C" {

int mi355cg_decompose(int n, int world, int decomp, int rank, int* y_lo, int* y_hi, int* x_lo, int* x_hi) {
    GridParams gp;
    if (!grid_params_init(&gp, n, n, 0, 1, 0, 1)) return fail(MI355CG_ERR_INVALID, "grid %d rejected", n);
    if (world < 1 || rank < 0 || rank >= world) return fail(MI355CG_ERR_INVALID, "bad world/rank %d/%d", rank, world);
    if (decomp != MI355CG_DECOMP_ROWS && decomp != MI355CG_DECOMP_2D) return fail(MI355CG_ERR_INVALID, "unknown decomposition %d", decomp);
    std::vector<Box> bx;
    if (int rc = decompose(gp, world, decomp, bx)) return rc;
    const Box& b = bx[rank];
    if (y_lo) *y_lo = b.y_lo;
    if (y_hi) *y_hi = b.y_hi;
    if (x_lo) *x_lo = b.s_lo * kStripCols;
    if (x_hi) *x_hi = b.s_hi == strips_total(gp, 2) ? gp.n : b.s_hi * kStripCols;
    return MI355CG_OK;
}

int mi355cg_halo_plan(int n, int world, int decomp, int rank, int max_msgs, int* n_msgs, mi355cg_halo_msg* msgs) {
    GridParams gp;
    if (!grid_params_init(&gp, n, n, 0, 1, 0, 1)) return fail(MI355CG_ERR_INVALID, "grid %d rejected", n);
    if (world < 1 || rank < 0 || rank >= world || !n_msgs) return fail(MI355CG_ERR_INVALID, "bad argument");
    std::vector<Box> bx;
    if (int rc = decompose(gp, world, decomp, bx)) return rc;
    const std::vector<Seg> segs = halo_segments(gp, bx);
    int k = 0;
    for (auto& s : segs) {
        if (s.src != rank && s.dst != rank) continue;
        if (msgs && k < max_msgs) msgs[k] = mi355cg_halo_msg{s.id, s.src == rank ? s.dst : s.src, s.src == rank ? 1 : 0, s.kind, s.y0, s.y1, s.x0, s.x1, seg_count(s)};
        ++k;
    }
    *n_msgs = k;
    return MI355CG_OK;
}

// Pure host arithmetic, for tests: the launch plan (work items) a part of the given decomposition would get.
// which: 0 whole part, 1 interior items, 2 edge items.  panels: up to 8 rows of {y0, y1, s0, ns, ty, nchunks, item0, gc};
// cls: ncls followed by the class boundaries (ncls + 1 values) when ncls > 1.  Returns the number of panels in *np.
int mi355cg_debug_plan(int n, int world, int decomp, int rank, int which, int* np, int* panels, int* grid, int* nitems, int* cls) {
    GridParams gp;
    if (!grid_params_init(&gp, n, n, 0, 1, 0, 1)) return fail(MI355CG_ERR_INVALID, "grid %d rejected", n);
    if (world < 1 || rank < 0 || rank >= world || !np || !panels) return fail(MI355CG_ERR_INVALID, "bad argument");
    std::vector<Box> bx;
    if (int rc = decompose(gp, world, decomp, bx)) return rc;
    mi355cg_ctx c{};                                   // geometry and plans only: no device is touched
    c.gp = gp; c.dtype = MI355CG_F64; c.is_slab = world > 1;
    c.s_lo = bx[rank].s_lo; c.s_hi = bx[rank].s_hi;
    build_geom(&c, 2, bx[rank].y_lo, bx[rank].y_hi);
    build_plans(&c);
    const Plan& pl = which == 1 ? c.interior : which == 2 ? c.edge : c.whole;
    *np = pl.wl.np;
    for (int k = 0; k < pl.wl.np; ++k) {
        const Panel& P = pl.wl.p[k];
        const int row[8] = {P.y0, P.y1, P.s0, P.ns, P.ty, P.nchunks, P.item0, P.gc};
        std::memcpy(panels + 8 * k, row, sizeof row);
    }
    if (grid) *grid = pl.grid;
    if (nitems) *nitems = pl.wl.nitems;
    if (cls) { cls[0] = pl.wl.ncls; for (int k = 0; k <= kXcds; ++k) cls[1 + k] = pl.wl.ncls == kXcds ? pl.wl.cls0[k] : 0; }
    return MI355CG_OK;
}

static int team_create_common(int n, int m, double a, double b, double c_, double d, int world, int decomp, mi355cg_team_s** out_t) {
    if (!out_t) return fail(MI355CG_ERR_INVALID, "out is null");
    *out_t = nullptr;
    if (decomp != MI355CG_DECOMP_ROWS && decomp != MI355CG_DECOMP_2D) return fail(MI355CG_ERR_INVALID, "unknown decomposition %d", decomp);
    GridParams gp;
    if (!grid_params_init(&gp, n, m, a, b, c_, d))
        return fail(MI355CG_ERR_INVALID, "grid %dx%d rejected: the L-shaped index map is only consistent for n == m, even, >= 6", n, m);
    if (world < 1) return fail(MI355CG_ERR_INVALID, "world %d", world);
    mi355cg_team_s* t = new mi355cg_team_s();
    t->gp = gp; t->world = world; t->decomp = decomp;
    if (int rc = decompose(gp, world, decomp, t->boxes)) { delete t; return rc; }
    *out_t = t;
    return MI355CG_OK;
}

int mi355cg_team_create_local(int n, int m, double a, double b, double c_, double d, int world,
                              const int* devices, int ndevices, int decomp, mi355cg_team* out) {
    if (!out) return fail(MI355CG_ERR_INVALID, "out is null");
    *out = nullptr;
    if (world > kMaxLocalParts) return fail(MI355CG_ERR_INVALID, "a LOCAL team drives at most %d parts", kMaxLocalParts);
    mi355cg_team_s* t = nullptr;
    if (int rc = team_create_common(n, m, a, b, c_, d, world, decomp, &t)) return rc;
    const int ns_all = strips_total(t->gp, 2);
    for (int r = 0; r < world; ++r) {
        const Box& bx = t->boxes[r];
        TeamPart p; p.rank = r;
        const int dev = (devices && ndevices > 0) ? devices[r % ndevices] : 0;
        const bool whole = world == 1;
        const int rc = create_impl(n, m, a, b, c_, d, MI355CG_F64, dev, bx.y_lo, bx.y_hi, bx.s_lo, std::min(bx.s_hi, ns_all), !whole, &p.c);
        if (rc) { team_free(t); return rc; }
        t->parts.push_back(p);
    }
    if (int rc = team_finish_setup(t)) { team_free(t); return rc; }
    *out = t;
    return MI355CG_OK;
}

int mi355cg_team_unique_id(void* id128) {
    if (!id128) return fail(MI355CG_ERR_INVALID, "null argument");
    RcclApi* api = rccl_api();
    if (!api) return fail(MI355CG_ERR_HIP, "librccl could not be loaded: %s", dlerror() ? dlerror() : "no such library");
    static_assert(sizeof(ncclUniqueId) == 128, "mi355cg.h promises a 128-byte id");
    ncclUniqueId id;
    NCCLCK(api->GetUniqueId(&id));
    std::memcpy(id128, &id, sizeof id);
    return MI355CG_OK;
}

int mi355cg_team_create_rccl(int n, int m, double a, double b, double c_, double d, int world, int rank, int device,
                             const void* id128, int decomp, mi355cg_team* out) {
    if (!out) return fail(MI355CG_ERR_INVALID, "out is null");
    *out = nullptr;
    if (rank < 0 || rank >= world || !id128) return fail(MI355CG_ERR_INVALID, "bad rank %d of %d / null id", rank, world);
    RcclApi* api = rccl_api();
    if (!api) return fail(MI355CG_ERR_HIP, "librccl could not be loaded");
    mi355cg_team_s* t = nullptr;
    if (int rc = team_create_common(n, m, a, b, c_, d, world, decomp, &t)) return rc;
    t->rccl = true;
    const Box& bx = t->boxes[rank];
    TeamPart p; p.rank = rank;
    int rc = create_impl(n, m, a, b, c_, d, MI355CG_F64, device, bx.y_lo, bx.y_hi, bx.s_lo, bx.s_hi, world > 1, &p.c);
    if (rc) { team_free(t); return rc; }
    t->parts.push_back(p);
    if ((rc = team_finish_setup(t))) { team_free(t); return rc; }
    ncclUniqueId id;
    std::memcpy(&id, id128, sizeof id);
    if (hipSetDevice(device) != hipSuccess) { team_free(t); return fail(MI355CG_ERR_HIP, "hipSetDevice(%d) failed", device); }
    const ncclResult_t nr = api->CommInitRank(&t->comm, world, id, rank);
    if (nr != ncclSuccess) { t->comm = nullptr; team_free(t); return fail(MI355CG_ERR_HIP, "ncclCommInitRank failed: %s", api->GetErrorString(nr)); }
    if (world > 1) {
        // second communicator for the halo messages: its id comes from rank 0 through the first one
        ncclUniqueId id2;
        unsigned char* dbuf = nullptr;
        auto bail = [&](const char* what, const char* why) { if (dbuf) hipFree(dbuf); team_free(t); return fail(MI355CG_ERR_HIP, "%s failed: %s", what, why); };
        if (rank == 0) { const ncclResult_t r0 = api->GetUniqueId(&id2); if (r0 != ncclSuccess) return bail("ncclGetUniqueId", api->GetErrorString(r0)); }
        if (hipMalloc((void**)&dbuf, sizeof id2) != hipSuccess) return bail("hipMalloc", "id buffer");
        if (rank == 0 && hipMemcpy(dbuf, &id2, sizeof id2, hipMemcpyHostToDevice) != hipSuccess) return bail("hipMemcpy", "id upload");
        hipStream_t st = t->parts[0].comm;
        const ncclResult_t rb = api->Broadcast(dbuf, dbuf, sizeof id2, ncclUint8, 0, t->comm, st);
        if (rb != ncclSuccess) return bail("ncclBroadcast", api->GetErrorString(rb));
        if (hipStreamSynchronize(st) != hipSuccess || hipMemcpy(&id2, dbuf, sizeof id2, hipMemcpyDeviceToHost) != hipSuccess) return bail("hipMemcpy", "id download");
        hipFree(dbuf); dbuf = nullptr;
        const ncclResult_t n2 = api->CommInitRank(&t->comm_halo, world, id2, rank);
        if (n2 != ncclSuccess) { t->comm_halo = nullptr; return bail("ncclCommInitRank (halo communicator)", api->GetErrorString(n2)); }
    }
    *out = t;
    return MI355CG_OK;
}

void mi355cg_team_destroy(mi355cg_team t) { team_free(t); }

int mi355cg_team_solve(mi355cg_team t, const mi355cg_params* prm, mi355cg_iter_cb cb, void* user,
                       const volatile int* stop_flag, mi355cg_results* out) {
    if (!t || !prm) return fail(MI355CG_ERR_INVALID, "null argument");
    return team_solve(t, prm, cb, user, stop_flag, out);
}

int mi355cg_team_info(mi355cg_team t, int* world, int* nlocal, int* decomp, long long* size) {
    if (!t) return fail(MI355CG_ERR_INVALID, "null team");
    if (world) *world = t->world;
    if (nlocal) *nlocal = (int)t->parts.size();
    if (decomp) *decomp = t->decomp;
    if (size) *size = t->gp.size;
    return MI355CG_OK;
}

int mi355cg_team_part(mi355cg_team t, int local_index, mi355cg_handle* part, int* rank) {
    if (!t || local_index < 0 || local_index >= (int)t->parts.size()) return fail(MI355CG_ERR_INVALID, "bad part index");
    if (part) *part = t->parts[local_index].c;
    if (rank) *rank = t->parts[local_index].rank;
    return MI355CG_OK;
}

// which: 0 x, 1 recursive residual, 2 right-hand side, 3 exact solution.  Fills the entries of the caller's GLOBAL packed
// vector (length mi355cg_size) that this process's parts own; the others are left untouched.
int mi355cg_team_get_vector(mi355cg_team t, int which, double* global_packed) {
    if (!t || !global_packed) return fail(MI355CG_ERR_INVALID, "null argument");
    for (auto& p : t->parts) {
        mi355cg_ctx* c = p.c;
        HIPCK(hipSetDevice(c->device));
        std::vector<double> own(std::max<long long>(c->pk_len, 1));
        if (which == 2 || which == 3) { if (int rc = ensure_host_copies(c)) return rc; }
        if (which == 2) own = c->rhs_h;
        else if (which == 3) own = c->u_h;
        else if (which == 0 || which == 1) { if (int rc = download_packed<double>(c, which == 0 ? c->x : c->r, own.data())) return rc; }
        else return fail(MI355CG_ERR_INVALID, "vector %d (0 x, 1 r, 2 b, 3 u)", which);
        const PackGeom& pg = c->pg;
        long long i = 0;
        for (int k = 0; k < pg.nb_rows; ++k, i += pg.wb) std::memcpy(global_packed + packed_index(t->gp, pg.xb0, pg.yb0 + k), own.data() + i, sizeof(double) * pg.wb);
        for (int k = 0; k < pg.nu_rows; ++k, i += pg.wu) std::memcpy(global_packed + packed_index(t->gp, pg.xu0, pg.yu0 + k), own.data() + i, sizeof(double) * pg.wu);
    }
    return MI355CG_OK;
}

// out2[0] = sum of v, out2[1] = sum of v^2 over the cells of this process's parts (double-double inside, rounded once)
int mi355cg_team_checksum(mi355cg_team t, int which, double* out2) {
    if (!t || !out2) return fail(MI355CG_ERR_INVALID, "null argument");
    hdd s[2] = {{0, 0}, {0, 0}};
    for (auto& p : t->parts) if (int rc = ctx_checksum(p.c, which, s)) return rc;
    out2[0] = s[0].hi + s[0].lo; out2[1] = s[1].hi + s[1].lo;
    return MI355CG_OK;
}

// mi355cg_setup_on_device for every part this process drives
int mi355cg_team_setup_on_device(mi355cg_team t) {
    if (!t) return fail(MI355CG_ERR_INVALID, "null team");
    for (auto& p : t->parts) if (int rc = mi355cg_setup_on_device(p.c)) return rc;
    return MI355CG_OK;
}

int mi355cg_team_set_profiling(mi355cg_team t, int enable) {
    if (!t) return fail(MI355CG_ERR_INVALID, "null team");
    t->profiling = enable != 0;
    return MI355CG_OK;
}
// per iteration of the last profiled solve, on this process's first part: device time of its kernels, device time of the
// collectives / halo messages on the comm stream, wall time (what is left is waiting + the host driver)
int mi355cg_team_phase_times(mi355cg_team t, double* kernel_ms, double* comm_ms, double* wall_ms) {
    if (!t) return fail(MI355CG_ERR_INVALID, "null team");
    if (kernel_ms) *kernel_ms = t->prof_kernel_ms;
    if (comm_ms) *comm_ms = t->prof_comm_ms;
    if (wall_ms) *wall_ms = t->prof_wall_ms;
    return MI355CG_OK;
}

}  // extern "C"
