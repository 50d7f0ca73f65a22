// team.h -- the native multi-GPU CG loop (included at the end of mi355cg.hip: one translation unit).
//
// The reference is single-process and has no collectives; this is the scaling surface of the same kernels.
// A TEAM is a decomposition of the grid into `world` parts -- row slabs balanced by unknown count, or a 2-D split whose
// x-cuts fall on 128-column strip boundaries -- plus a transport for the two things that cross parts per iteration:
//   * every part's 16-double RECORD of partial sums / maxes / stop request, needed by every part after each of the two
//     launches of an iteration;
//   * the boundary rows / columns of the RESIDUAL, neighbour to neighbour, once per iteration.  (The direction never
//     crosses parts: the fused stencil launch recomputes it on its halo from the ghost copies of r and the old direction and
//     keeps the result in its ghost rows and ghost columns.)
// Who drives the parts:
//   RCCL   one process per GPU.  The library owns its communicator (mi355cg_team_unique_id -> ncclCommInitRank); RCCL is
//          resolved at run time from the librccl already in the process (torch's) or from the ROCm installation.  At team
//          creation the ranks exchange IPC handles through it and map each other's mailbox, residual vector and column buffer.
//   LOCAL  one process drives all parts, on one or several GPUs (one host thread per part when they are on different GPUs).
//          This is what a single-process host (the reference's DirichletSolver is one) uses, and what lets one GPU rehearse
//          an 8-part run bit for bit.
// How the records travel (mi355cg_team_s::rec_mode):
//   MAILBOX  every workgroup of a producer launch also stores its partials as flagged 64-bit words; a one-workgroup REDUCER launch on
//            a side stream, resident beside the producer, polls them as they arrive and stores the part's record (32 flagged words)
//            straight into the mailbox of every OTHER part (peer GPU memory over xGMI; IPC-mapped when the part is another process).
//            A consumer launch reduces its own part's partials itself, as on a single GPU, and polls its own mailbox only for the other
//            parts' records (bounded; WAIT_KERNEL): the hop costs the reducer's tail (2 - 3 us) + one store latency, and nothing sits
//            between the producer and the consumer launch on the stream.  Ranks that share one physical GPU (rehearsals) let their
//            STREAMS wait for an announcement word instead (hipStreamWaitValue64; WAIT_STREAM), because polling consumers of several
//            ranks could keep each other's producers off the CUs.
//   RCCL     ncclAllGather of the flagged words on the COMPUTE stream, between producer and consumer launch.
//   EVENTS   LOCAL teams whose parts share a GPU: direct stores + a hub stream joining the parts' events.
// How the halo travels (halo_mode):
//   PUSH         one small launch behind the update launch stores the boundary rows / packed columns into the neighbours' ghost
//                rows / receive buffers; when all its workgroups have released their stores, the last one stores the sequence number
//                into each neighbour's mailbox, and the neighbour's compute stream waits for that word (hipStreamWaitValue64)
//                before its next stencil launch.
//   RCCL_INLINE  one ncclSend/ncclRecv group per iteration on the compute stream (one communicator, one stream: order-safe).
//   RCCL_STREAM  the same group on a SECOND stream and communicator, ordered with the compute stream by two events.
//   LOCAL        device-to-device copies on the parts' comm streams (one process).
// Per iteration and part (default: ONE launch per phase; MI355CG_TEAM_SPLIT=1 cuts each phase into interior + edge launches
// so that the halo travels beside the interior items instead, = 2 the update phase only):
//     wait halo ; stencil (+ reducer -> record A) ............................ every part's update launch needs all records A
//     update (+ reducer -> record B) ; pack columns ; halo of r ............... every part's next stencil launch needs all records B
// F32_MIXED (mi355cg_team_set_dtype): the same loop on the parts' fp32 vectors inside fp64 refinement steps (team_solve_mixed).
// Sums travel as double-double pairs and are reduced in part order by every consumer, so every decomposition takes
// bit-identical steps (tests/test_gpu_team.py, tests/test_gpu_team_ranks.py).
#include <dlfcn.h>
#include <unistd.h>
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
    std::string lib_name;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
// The copy already in the process first (PyTorch-ROCm bundles its own), then the ROCm installation's.  MI355CG_RCCL_LIB names
// another library with the same eleven entry points instead (tests/nccl_shim: the host-staged stand-in that lets several rank
// processes share ONE GPU, which RCCL itself refuses).
RcclApi* rccl_api() {
    static RcclApi api;
    static std::once_flag once;
    std::call_once(once, [] {
        void* h = nullptr;
        const char* over = getenv("MI355CG_RCCL_LIB");
        if (over && *over) {
            h = dlopen(over, RTLD_NOW | RTLD_LOCAL);
            if (h) api.lib_name = over;
        } else {
            const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
            h = dlopen(names[0], RTLD_NOW | RTLD_NOLOAD);
            if (h) api.lib_name = names[0];
            for (int i = 0; !h && i < 3; ++i) { h = dlopen(names[i], RTLD_NOW | RTLD_LOCAL); if (h) api.lib_name = names[i]; }
        }
        if (!h) return;
        bool ok = true;
#define MI355CG_SYM(field, name) api.field = reinterpret_cast<decltype(api.field)>(dlsym(h, name)); if (!api.field) ok = false
        MI355CG_SYM(GetUniqueId, "ncclGetUniqueId"); MI355CG_SYM(CommInitRank, "ncclCommInitRank"); MI355CG_SYM(CommDestroy, "ncclCommDestroy");
        MI355CG_SYM(CommCount, "ncclCommCount");
        MI355CG_SYM(AllGather, "ncclAllGather"); MI355CG_SYM(Send, "ncclSend"); MI355CG_SYM(Recv, "ncclRecv"); MI355CG_SYM(Broadcast, "ncclBroadcast");
        MI355CG_SYM(GroupStart, "ncclGroupStart"); MI355CG_SYM(GroupEnd, "ncclGroupEnd"); MI355CG_SYM(GetErrorString, "ncclGetErrorString");
#undef MI355CG_SYM
        if (ok) api.lib = h;
    });
    return api.lib ? &api : nullptr;
}
#define NCCLCK(expr)                                                                                          \
    do {                                                                                                      \
        ncclResult_t r_ = (expr);                                                                             \
        if (r_ != ncclSuccess)                                                                                \
            return fail(MI355CG_ERR_HIP, "%s failed: %s (%s:%d)", #expr, rccl_api()->GetErrorString(r_), __FILE__, __LINE__); \
    } while (0)

// ---- mailboxes -------------------------------------------------------------------------------------------------------
// One per part, in the part's own device memory (uncached allocation: peers write it while its kernels poll it), 64-bit words:
//   rec[phase 0..1][slot 0..1][rank][kLLWords]   the parts' flagged records (cg_kernels.h); slot = iteration sequence number & 1
//   flag[phase 0..1][rank]                       sequence number of the last record rank published (what a STREAM-level wait watches)
//   halo[rank]                                   sequence number of the last halo rows / columns rank pushed into this part's ghost cells
struct MboxLayout {
    int W;
    size_t rec(int ph, int slot, int rank) const { return ((size_t)(ph * 2 + slot) * W + rank) * kLLWords; }
    size_t flag(int ph, int rank) const { return (size_t)4 * W * kLLWords + (size_t)ph * W + rank; }
    size_t halo(int rank) const { return (size_t)4 * W * kLLWords + (size_t)2 * W + rank; }
    size_t flags_begin() const { return (size_t)4 * W * kLLWords; }
    size_t words() const { return (size_t)4 * W * kLLWords + (size_t)3 * W; }
};
// the 32-bit stamp of sequence number q: never 0 (a fresh mailbox holds zeros)
inline unsigned stamp_of(u64 q) { return (unsigned)(q & 0x7fffffffu) | 0x80000000u; }

// ---- small kernels of the team loop ------------------------------------------------------------------------------
constexpr int kMaxLocalParts = kMaxRecDst;
struct TeamRecArgs {
    const double* part; int n, stride;
    int nsum, lo_off, max_first, nmax;            // as RecordArgs
    const int* stop_req;                          // pinned host word (update records only), may be null
    RecSpec rs;                                   // destinations, slot, stamp
};
// the record of the initialisation pass, from its plain partials (behind it on its stream: k_init_fresh is a flat kernel that
// stores nothing flagged)
__global__ __launch_bounds__(kBlock) void k_team_record(const TeamRecArgs a) {
    __shared__ double lds[2 * kWaves];
    __shared__ double rec[kRecWords];
    if (threadIdx.x < kRecWords) rec[threadIdx.x] = 0.0;
    __syncthreads();
    for (int f = 0; f < a.nsum; ++f) {
        const dd t = reduce_parts_dd(a.part + f * a.stride, a.part + (f + a.lo_off) * a.stride, a.n, 1, lds);
        if (threadIdx.x == 0) { rec[f] = t.hi; rec[f + a.lo_off] = t.lo; }
    }
    for (int f = a.max_first; f < a.max_first + a.nmax; ++f) {
        const double t = reduce_parts<true>(a.part + f * a.stride, a.n, 1, lds);
        if (threadIdx.x == 0) rec[f] = t;
    }
    if (threadIdx.x == 0 && a.stop_req) rec[kRecStopWord] = *(const volatile int*)a.stop_req ? 1.0 : 0.0;
    __syncthreads();
    publish_record(rec, a.rs);
}
// column messages <-> buffers: gather a vector's columns into the send buffer / scatter the receive buffer into ghost columns (a
// column is strided in storage; a row travels in place)
constexpr int kMaxColSegs = 8;
struct ColSeg { int x, y0, n, row; long long off; };     // row = 0: cells (x, y0 .. y0+n-1); row = 1: cells (x .. x+n-1, y0)
struct ColArgs { Geom g; double* v; double* buf; ColSeg s[kMaxColSegs]; int ns, scatter; };
__global__ __launch_bounds__(kBlock) void k_cols(const ColArgs a) {
    for (int k = 0; k < a.ns; ++k) {
        ColSeg s = a.s[0];
#pragma unroll
        for (int j = 1; j < kMaxColSegs; ++j) if (j == k) s = a.s[j];
        for (int i = blockIdx.x * kBlock + threadIdx.x; i < s.n; i += gridDim.x * kBlock) {
            double* cell = s.row ? a.v + (row_off(a.g, s.y0) - a.g.base0 + s.x + i) : a.v + (row_off(a.g, s.y0 + i) - a.g.base0 + s.x);
            if (a.scatter) *cell = a.buf[s.off + i]; else a.buf[s.off + i] = *cell;
        }
    }
}
// PUSH halo: this part's boundary rows straight into the neighbours' ghost rows and its packed columns into their receive
// buffers (their memory: a peer GPU's over xGMI, mapped through IPC when the neighbour is another process).
constexpr int kMaxPush = 12, kMaxPushPeers = 6;
struct PushSeg { const void* src; void* dst; int n; };           // n elements of the launch's element type
struct PushArgs {
    PushSeg s[kMaxPush]; int ns;
    unsigned* ticket;                 // arrival counter of this launch's workgroups, 0 between launches
    u64* flag[kMaxPushPeers]; int nflag; u64 flag_value;      // the neighbours' halo words: set by the LAST workgroup, after everybody's stores
};
// The announcement rides on the launch itself: every workgroup makes its stores visible at system scope (release fence), takes a
// ticket, and the last one stores the sequence number into the neighbours' halo words -- one hop less than a stream-ordered
// hipStreamWriteValue64 behind the launch (~2 us of blit launch on the path update -> neighbour's next stencil).
template <typename E>
__global__ __launch_bounds__(kBlock) void k_push(const PushArgs a) {
    for (int k = 0; k < a.ns; ++k) {
        PushSeg s = a.s[0];
#pragma unroll
        for (int j = 1; j < kMaxPush; ++j) if (j == k) s = a.s[j];
        for (int i = blockIdx.x * kBlock + threadIdx.x; i < s.n; i += gridDim.x * kBlock) ((E*)s.dst)[i] = ((const E*)s.src)[i];
    }
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) {
        const bool last = __hip_atomic_fetch_add(a.ticket, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1;
        if (last) {
            for (int f = 0; f < kMaxPushPeers; ++f) if (f < a.nflag) st_sys(a.flag[f], a.flag_value);
            __hip_atomic_store(a.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

enum { REC_EVENTS = 0, REC_RCCL = 1, REC_MAILBOX = 2 };                       // how a part's records reach the other parts
enum { WAIT_KERNEL = 0, WAIT_STREAM = 1 };                                     // REC_MAILBOX: who waits for them
enum { HALO_LOCAL = 0, HALO_RCCL_INLINE = 1, HALO_RCCL_STREAM = 2, HALO_PUSH = 3 };
const char* rec_name(int m) { return m == REC_EVENTS ? "events" : m == REC_RCCL ? "rccl" : "mailbox"; }
const char* wait_name(int m) { return m == WAIT_KERNEL ? "kernel" : "stream"; }
const char* halo_name(int m) { return m == HALO_LOCAL ? "copies" : m == HALO_RCCL_INLINE ? "rccl-inline" : m == HALO_RCCL_STREAM ? "rccl-stream" : "push"; }

struct TeamPart {
    mi355cg_ctx* c = nullptr;
    int rank = 0;
    hipStream_t comm = nullptr;
    // What other parts write into comes from a per-process pool that is never handed back to HIP (IpcPool): ONE slab of uncached
    // memory [mailbox | column receive buffer], and -- rank processes -- the residual vector (its ghost rows).  Nothing else is exported.
    void* slab = nullptr;
    u64* mbox = nullptr;                                  // this part's mailbox (MboxLayout)
    double* recv_cols = nullptr;                          // receive buffer of the packed column messages
    std::vector<u64*> peer_mbox;                          // [world]: part j's mailbox as this part's device addresses it (nullptr: not reachable)
    std::vector<double*> peer_cols;                       // [world]: part j's receive buffer, same
    std::vector<double*> peer_r;                          // [world]: part j's residual vector (row messages go straight into its ghost rows), same
    bool r_pooled = false;                                // the part's residual vector comes from the IPC pool (rank processes export it)
    std::vector<void*> ipc_opened;                        // mappings to close
    double* send_cols = nullptr;                          // packed column messages to send
    u64 **dst_self[2] = {nullptr, nullptr};               // device arrays for RecSpec::dst, per phase: own mailbox only (REC_RCCL) ...
    u64 **dst_all[2] = {nullptr, nullptr};                // ... or every OTHER reachable part's
    u64 **flag_all[2] = {nullptr, nullptr};               // RecSpec::flag, matching dst_all
    int ndst_all = 0;
    u64* pll[2] = {nullptr, nullptr};                     // this part's flagged partials per phase (what its reducer launches poll)
    hipStream_t side = nullptr;                           // the reducer launches of REC_MAILBOX run here, beside the producers
    int nslots[2] = {0, 0};                               // partial slots the last phase of each kind wrote (what the next consumer reduces)
    std::vector<Seg> sends, recvs;                        // ordered by (peer, id)
    std::vector<long long> send_off, recv_off;            // column messages: offset in send_cols / recv_cols
    std::vector<int> halo_from, halo_to;                  // distinct neighbour ranks
    ColArgs pack{}, unpack{};
    PushArgs push{}, push32{};                            // the fp64 residual's messages; the fp32 residual's (same cells, 4-byte elements, row slabs)
    unsigned* push_ticket = nullptr;
    bool split = false;                                   // interior / edge launches (the part has neighbours)
    hipEvent_t ev_recA = nullptr, ev_gA = nullptr, ev_redge = nullptr, ev_recB = nullptr, ev_gB = nullptr, ev_halo = nullptr;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> comm_pairs;     // profiling: event pairs on the comm stream
};

}  // namespace

struct mi355cg_team_s {
    GridParams gp;
    int world = 1, decomp = 0;
    std::vector<Box> boxes;
    std::vector<Geom> geoms;                // every part's storage geometry (a part addresses its neighbours' ghost rows with it)
    std::vector<Seg> segs;
    std::vector<TeamPart> parts;            // the parts this process drives (all of them: LOCAL; one: RCCL)
    bool rccl = false;
    ncclComm_t comm = nullptr;              // records (REC_RCCL) and the halo of HALO_RCCL_INLINE: everything on the COMPUTE stream
    ncclComm_t comm_halo = nullptr;         // HALO_RCCL_STREAM: the halo messages on the comm stream; its own communicator, so the two streams never serialise on one
    int rccl_nranks = 0;                    // what ncclCommCount says
    bool ipc_ok = false;                    // every rank could map every other rank's mailbox / residual / column buffer -- and reads ITS bytes through the mappings
    std::string ipc_note = "-";             // why not, if not
    bool shared_device = false;             // two ranks sit on the same physical GPU (rehearsals): kernels must not spin on each other
    hipStream_t hub = nullptr;              // LOCAL, REC_EVENTS: joins the parts' record events
    hipEvent_t ev_hub = nullptr;
    hipStream_t side = nullptr;             // releases stream-level waits when a solve is abandoned
    int* stop_h = nullptr;                  // pinned: this process's stop request, read by the update phase's record
    u64 seq = 0;                            // iteration sequence number: stamps and slots of the records (identical on every rank)
    int rec_mode = REC_EVENTS, wait_mode = WAIT_KERNEL, halo_mode = HALO_LOCAL;
    u64 budget_ticks = 0;                   // what a kernel may wait for a record (100 MHz ticks)
    double timeout_s = 30.0;
    int dtype = MI355CG_F64;                // MI355CG_F32_MIXED: solves are fp64 refinement around an fp32 inner CG (mi355cg_team_set_dtype)
    bool f32 = false;                       // the CG loop now running works on the parts' fp32 vectors (an inner solve of F32_MIXED)
    bool broken = false;                    // a solve was abandoned: the ranks' sequence numbers may differ, no further solves
    // Interior / edge launches per phase (the halo travels while the interior items run) or ONE launch per phase.
    int split_phases = 0;                   // 0: one launch per phase; 1: interior / edge launches in both phases; 2: in the update phase only
    bool profiling = false;
    double prof_kernel_ms = 0, prof_comm_ms = 0, prof_wall_ms = 0;     // per iteration, last profiled solve
    int hub_device = 0;
};

namespace {

// Memory that other processes map (IPC) comes from a pool that never returns anything to HIP: an owner that frees exported memory
// and allocates again has been seen to hand out a handle that a peer resolves to the OLD memory (and hipIpcGetMemHandle to refuse
// recycled hipMalloc memory) once a process has created and destroyed a few teams.  A slab keeps its handle for the life of the process.
struct IpcPool {
    struct Slab { void* ptr; size_t bytes; int device; bool uncached; bool in_use; };
    std::mutex mu;
    std::vector<Slab> slabs;
    // uncached: memory that is polled while peers write it (mailboxes, receive buffers); otherwise ordinary device memory (a vector)
    int acquire(int device, size_t bytes, bool uncached, void** out) {
        bytes = (bytes + 65535) / 65536 * 65536;
        std::lock_guard<std::mutex> g(mu);
        Slab* hit = nullptr;
        for (auto& s : slabs) if (!s.in_use && s.device == device && s.uncached == uncached && s.bytes >= bytes && s.bytes <= bytes + bytes / 4 && (!hit || s.bytes < hit->bytes)) hit = &s;
        if (!hit) {
            void* p = nullptr;
            if (uncached) HIPCK(hipExtMallocWithFlags(&p, bytes, hipDeviceMallocUncached)); else HIPCK(hipMalloc(&p, bytes));
            slabs.push_back(Slab{p, bytes, device, uncached, false});
            hit = &slabs.back();
        }
        hit->in_use = true;
        HIPCK(hipMemset(hit->ptr, 0, hit->bytes));
        HIPCK(hipDeviceSynchronize());
        *out = hit->ptr;
        return MI355CG_OK;
    }
    void release(void* p) { std::lock_guard<std::mutex> g(mu); for (auto& s : slabs) if (s.ptr == p) s.in_use = false; }
};
IpcPool& ipc_pool() { static IpcPool pool; return pool; }
inline size_t mbox_bytes(int world) { return (sizeof(u64) * MboxLayout{world}.words() + 255) / 256 * 256; }

int boot_all_gather(mi355cg_team_s* t, const void* mine, void* all, size_t bytes);
void team_free(mi355cg_team_s* t) {
    if (!t) return;
    if (t->broken) return;                 // streams may never drain (a peer is gone): leak rather than hang in a synchronise
    for (auto& p : t->parts) {
        if (p.c) hipSetDevice(p.c->device);
        if (p.c && p.c->stream) hipStreamSynchronize(p.c->stream);
        if (p.comm) { hipStreamSynchronize(p.comm); }
    }
    // IPC: every rank unmaps the other ranks' memory BEFORE any owner frees it (an owner that frees memory a peer still has mapped,
    // then allocates again, has been seen to hand out a handle that the peer resolves to the OLD memory)
    bool mapped = false;
    for (auto& p : t->parts) {
        if (p.c) hipSetDevice(p.c->device);
        for (void* q : p.ipc_opened) if (q) { hipIpcCloseMemHandle(q); mapped = true; }
        p.ipc_opened.clear();
    }
    (void)hipGetLastError();
    if (t->rccl && t->comm && t->world > 1 && (mapped || t->ipc_ok) && rccl_api()) { int mine = 1; std::vector<int> all(t->world); (void)boot_all_gather(t, &mine, all.data(), sizeof(int)); }
    if (t->comm_halo && rccl_api()) rccl_api()->CommDestroy(t->comm_halo);
    if (t->comm && rccl_api()) rccl_api()->CommDestroy(t->comm);
    for (auto& p : t->parts) {
        if (p.c) hipSetDevice(p.c->device);
        if (p.slab) ipc_pool().release(p.slab);
        if (p.c && p.c->rf == (float*)p.c->r) p.c->rf = nullptr;           // F32_MIXED: an alias of r
        if (p.r_pooled && p.c) { ipc_pool().release(p.c->r); p.c->r = nullptr; }
        if (p.side) { hipStreamSynchronize(p.side); hipStreamDestroy(p.side); }
        for (void* q : {(void*)p.send_cols, (void*)p.push_ticket, (void*)p.pll[0], (void*)p.pll[1], (void*)p.dst_self[0], (void*)p.dst_self[1], (void*)p.dst_all[0], (void*)p.dst_all[1],
                        (void*)p.flag_all[0], (void*)p.flag_all[1]}) if (q) hipFree(q);
        for (hipEvent_t e : {p.ev_recA, p.ev_gA, p.ev_redge, p.ev_recB, p.ev_gB, p.ev_halo}) if (e) hipEventDestroy(e);
        if (p.comm) hipStreamDestroy(p.comm);
        if (p.c) mi355cg_destroy(p.c);
    }
    if (t->hub) { hipSetDevice(t->hub_device); hipStreamDestroy(t->hub); }
    if (t->side) hipStreamDestroy(t->side);
    if (t->ev_hub) hipEventDestroy(t->ev_hub);
    if (t->stop_h) hipHostFree(t->stop_h);
    delete t;
}

// The ordered message lists of one part and the layout of its buffers: every rank can compute them for every part.
// Send / receive buffers: the column messages packed back to back (rows travel in place).
struct PartLists { std::vector<Seg> sends, recvs; std::vector<long long> send_off, recv_off; long long send_len = 0, recv_len = 0; };
PartLists part_lists(const std::vector<Seg>& segs, int rank) {
    PartLists L;
    for (auto& s : segs) { if (s.src == rank) L.sends.push_back(s); if (s.dst == rank) L.recvs.push_back(s); }
    auto by_peer = [](bool send) { return [send](const Seg& a, const Seg& b) { const int pa = send ? a.dst : a.src, pb = send ? b.dst : b.src; return pa != pb ? pa < pb : a.id < b.id; }; };
    std::sort(L.sends.begin(), L.sends.end(), by_peer(true));
    std::sort(L.recvs.begin(), L.recvs.end(), by_peer(false));
    for (auto& s : L.sends) { L.send_off.push_back(L.send_len); if (s.kind == 1) L.send_len += seg_count(s); }
    for (auto& s : L.recvs) { L.recv_off.push_back(L.recv_len); if (s.kind == 1) L.recv_len += seg_count(s); }
    return L;
}
Geom part_geom(const GridParams& gp, const Box& bx) {
    mi355cg_ctx c{};                                   // geometry only: no device is touched
    c.gp = gp; c.dtype = MI355CG_F64; c.is_slab = true;
    c.s_lo = bx.s_lo; c.s_hi = std::min(bx.s_hi, strips_total(gp, 2));
    build_geom(&c, 2, bx.y_lo, bx.y_hi);
    return c.g;
}
// The residual vector the halo is about, by element size: the fp64 one, or the fp32 one of an inner solve of F32_MIXED -- which lives
// in the SAME memory (a part's rf is its r reinterpreted: half of it is used), so every neighbour's mapping of r serves both.
inline int halo_esz(const mi355cg_team_s* t) { return t->f32 ? 4 : 8; }
void* seg_bytes_g(const Geom& g, void* v, const Seg& s, int esz) { return (char*)v + (size_t)esz * (size_t)(row_off(g, s.y0) - g.base0 + s.x0); }
void* seg_bytes(const mi355cg_ctx* c, void* v, const Seg& s, int esz) { return seg_bytes_g(c->g, v, s, esz); }

int upload_ptrs(u64*** dev, const std::vector<u64*>& v) {
    HIPCK(hipMalloc((void**)dev, sizeof(u64*) * std::max<size_t>(v.size(), 1)));
    if (!v.empty()) HIPCK(hipMemcpy(*dev, v.data(), sizeof(u64*) * v.size(), hipMemcpyHostToDevice));
    return MI355CG_OK;
}

// Per-part resources and halo lists once the contexts exist.  (The mailboxes of parts in other processes are mapped later:
// team_map_peers.)
int team_finish_setup(mi355cg_team_s* t) {
    t->segs = halo_segments(t->gp, t->boxes);
    t->geoms.clear();
    for (auto& b : t->boxes) t->geoms.push_back(part_geom(t->gp, b));
    const MboxLayout ml{t->world};
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
        HIPCK(hipStreamCreateWithFlags(&p.side, hipStreamNonBlocking));
        HIPCK(hipMalloc((void**)&p.push_ticket, sizeof(unsigned)));
        HIPCK(hipMemset(p.push_ticket, 0, sizeof(unsigned)));
        if (c->strideA > kMaxPartSlots || c->strideB > kMaxPartSlots) return fail(MI355CG_ERR_INVALID, "a phase writes %d partial slots, the reducer takes %d", std::max(c->strideA, c->strideB), kMaxPartSlots);
        for (int ph = 0; ph < 2; ++ph) {
            const size_t words = (size_t)(ph == 0 ? c->strideA : c->strideB) * 2 * (ph == 0 ? FA_COUNT : FB_LL_COUNT);
            HIPCK(hipExtMallocWithFlags((void**)&p.pll[ph], sizeof(u64) * words, hipDeviceMallocUncached));
            HIPCK(hipMemset(p.pll[ph], 0, sizeof(u64) * words));
        }
        const PartLists L = part_lists(t->segs, p.rank);
        if (int rc = ipc_pool().acquire(c->device, mbox_bytes(t->world) + sizeof(double) * std::max<long long>(L.recv_len, 1), true, &p.slab)) return rc;
        p.mbox = (u64*)p.slab;
        p.recv_cols = (double*)((char*)p.slab + mbox_bytes(t->world));
        (void)ml;
        for (hipEvent_t* e : {&p.ev_recA, &p.ev_gA, &p.ev_redge, &p.ev_recB, &p.ev_gB, &p.ev_halo}) HIPCK(hipEventCreateWithFlags(e, ev_flags));
        p.sends = L.sends; p.recvs = L.recvs; p.send_off = L.send_off; p.recv_off = L.recv_off;
        p.split = !p.sends.empty() || !p.recvs.empty();
        for (auto& s : p.sends) if (p.halo_to.empty() || p.halo_to.back() != s.dst) p.halo_to.push_back(s.dst);
        for (auto& s : p.recvs) if (p.halo_from.empty() || p.halo_from.back() != s.src) p.halo_from.push_back(s.src);
        p.pack = ColArgs{}; p.unpack = ColArgs{};
        p.pack.g = c->g; p.unpack.g = c->g; p.unpack.scatter = 1;
        for (size_t i = 0; i < p.sends.size(); ++i) if (p.sends[i].kind == 1) { if (p.pack.ns >= kMaxColSegs) return fail(MI355CG_ERR_INVALID, "too many column messages"); p.pack.s[p.pack.ns++] = ColSeg{p.sends[i].x0, p.sends[i].y0, (int)seg_count(p.sends[i]), 0, p.send_off[i]}; }
        for (size_t i = 0; i < p.recvs.size(); ++i) if (p.recvs[i].kind == 1) { if (p.unpack.ns >= kMaxColSegs) return fail(MI355CG_ERR_INVALID, "too many column messages"); p.unpack.s[p.unpack.ns++] = ColSeg{p.recvs[i].x0, p.recvs[i].y0, (int)seg_count(p.recvs[i]), 0, p.recv_off[i]}; }
        if (int rc = alloc_vec(&p.send_cols, std::max<long long>(L.send_len, 1))) return rc;
        p.pack.buf = p.send_cols; p.unpack.buf = p.recv_cols;
        p.peer_mbox.assign(t->world, nullptr); p.peer_cols.assign(t->world, nullptr); p.peer_r.assign(t->world, nullptr);
        p.peer_mbox[p.rank] = p.mbox; p.peer_cols[p.rank] = p.recv_cols; p.peer_r[p.rank] = c->r;
        HIPCK(hipDeviceSynchronize());
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
        for (auto& p : t->parts) for (auto& q : t->parts) { p.peer_mbox[q.rank] = q.mbox; p.peer_cols[q.rank] = q.recv_cols; p.peer_r[q.rank] = q.c->r; }
    }
    HIPCK(hipSetDevice(t->parts[0].c->device));
    HIPCK(hipStreamCreateWithFlags(&t->side, hipStreamNonBlocking));
    HIPCK(hipHostMalloc((void**)&t->stop_h, sizeof(int)));
    *t->stop_h = 0;
    return MI355CG_OK;
}

// The device-side destination tables of the records and the PUSH list, from peer_mbox / peer_cols.
int team_build_tables(mi355cg_team_s* t) {
    const MboxLayout ml{t->world};
    for (auto& p : t->parts) {
        HIPCK(hipSetDevice(p.c->device));
        for (int ph = 0; ph < 2; ++ph) {
            for (void* q : {(void*)p.dst_self[ph], (void*)p.dst_all[ph], (void*)p.flag_all[ph]}) if (q) hipFree(q);
            p.dst_self[ph] = p.dst_all[ph] = p.flag_all[ph] = nullptr;
            std::vector<u64*> self{p.mbox + ml.rec(ph, 0, p.rank)}, all, flags;
            for (int j = 0; j < t->world; ++j) if (p.peer_mbox[j] && j != p.rank) { all.push_back(p.peer_mbox[j] + ml.rec(ph, 0, p.rank)); flags.push_back(p.peer_mbox[j] + ml.flag(ph, p.rank)); }
            if (int rc = upload_ptrs(&p.dst_self[ph], self)) return rc;
            if (int rc = upload_ptrs(&p.dst_all[ph], all)) return rc;
            if (int rc = upload_ptrs(&p.flag_all[ph], flags)) return rc;
            p.ndst_all = (int)all.size();
        }
        // PUSH: every outgoing message as (source in this part, destination in the neighbour's memory); a second list for the fp32
        // residual of an F32_MIXED inner solve (rows only: the fp32 kernels run on row slabs)
        for (int f32 = 0; f32 < 2; ++f32) {
            PushArgs& pa = f32 ? p.push32 : p.push;
            const int esz = f32 ? 4 : 8;
            pa = PushArgs{};
            bool reach = true;
            for (size_t i = 0; i < p.sends.size(); ++i) {
                const Seg& s = p.sends[i];
                if (!p.peer_cols[s.dst] || !p.peer_r[s.dst] || (f32 && s.kind != 0)) { reach = false; break; }
                if (pa.ns >= kMaxPush) return fail(MI355CG_ERR_INVALID, "too many halo messages for one push launch");
                PushSeg ps{};
                ps.n = (int)seg_count(s);
                if (s.kind == 0) {                                             // a row: straight into the neighbour's ghost row
                    ps.src = seg_bytes(p.c, p.c->r, s, esz); ps.dst = seg_bytes_g(t->geoms[s.dst], p.peer_r[s.dst], s, esz);
                } else {                                                       // a column: packed, into the neighbour's receive buffer where IT expects the message
                    const PartLists Lq = part_lists(t->segs, s.dst);
                    long long off = -1;
                    for (size_t j = 0; j < Lq.recvs.size(); ++j) if (Lq.recvs[j].id == s.id) off = Lq.recv_off[j];
                    if (off < 0) return fail(MI355CG_ERR_STATE, "halo message %d has no receiver", s.id);
                    ps.src = p.send_cols + p.send_off[i]; ps.dst = p.peer_cols[s.dst] + off;
                }
                pa.s[pa.ns++] = ps;
            }
            if (!reach) pa.ns = -1;
            pa.ticket = p.push_ticket;
            pa.nflag = 0;
            if (reach) for (int j : p.halo_to) {
                if (pa.nflag >= kMaxPushPeers) return fail(MI355CG_ERR_INVALID, "a part pushes its halo to at most %d neighbours", kMaxPushPeers);
                pa.flag[pa.nflag++] = p.peer_mbox[j] + ml.halo(p.rank);
            }
        }
    }
    return MI355CG_OK;
}

// ---- waiting without hanging -------------------------------------------------------------------------------------------
// Every wait of the host on a team stream is bounded: a stream that does not drain within the team's timeout is taken to be
// stuck behind a peer that will never deliver, its stream-level waits are released (all-ones into the flag words, so the kernels
// behind them run, miss their records within their own budget and end the solve), and the solve returns MI355CG_ERR_STATE.
int bounded_sync(mi355cg_team_s* t, hipStream_t st, const volatile int* stop_flag, double seconds) {
    const auto t0 = std::chrono::steady_clock::now();
    bool patient = false;
    for (unsigned spins = 0;; ++spins) {
        if (stop_flag && *stop_flag) *t->stop_h = 1;
        const hipError_t q = hipStreamQuery(st);
        if (q == hipSuccess) return MI355CG_OK;
        if (q != hipErrorNotReady) HIPCK(q);
        // The first 20 ms are spent polling (a 20-iteration solve at N = 4096 is 2.7 ms: a sleep of 50 us that turns into 100 with the
        // scheduler's help would be 4 % of it); a wait that lasts longer does not care about 50 us and gives the core away.
        double waited = -1.0;
        if ((spins & 255u) == 255u) {
            waited = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            if (waited > seconds) return -1;
            if (waited > 0.02) patient = true;
        }
        if (stop_flag || !patient) std::this_thread::yield(); else std::this_thread::sleep_for(std::chrono::microseconds(50));
    }
}
int team_abandon(mi355cg_team_s* t, const char* what) {
    t->broken = true;
    const MboxLayout ml{t->world};
    for (auto& p : t->parts) {
        hipSetDevice(p.c->device);
        hipMemsetAsync(p.mbox + ml.flags_begin(), 0xff, sizeof(u64) * (ml.words() - ml.flags_begin()), t->side);
    }
    hipStreamSynchronize(t->side);
    for (auto& p : t->parts) { hipSetDevice(p.c->device); bounded_sync(t, p.c->stream, nullptr, 5.0 + 2e-8 * (double)t->budget_ticks); }
    (void)hipGetLastError();
    return fail(MI355CG_ERR_STATE, "team solve abandoned: %s (no progress within %.0f s; a peer did not deliver).  The team cannot be used again", what, t->timeout_s);
}

// ---- records ---------------------------------------------------------------------------------------------------------
// where the record of (which, seq) of part p goes
RecSpec team_rec_spec(mi355cg_team_s* t, TeamPart& p, int which, u64 seq) {
    const MboxLayout ml{t->world};
    RecSpec rs{};
    const bool everywhere = t->rec_mode != REC_RCCL;
    rs.ndst = everywhere ? p.ndst_all : 1;
    rs.dst = everywhere ? p.dst_all[which] : p.dst_self[which];
    rs.flag = (everywhere && t->rec_mode == REC_MAILBOX && t->wait_mode == WAIT_STREAM) ? p.flag_all[which] : nullptr;
    rs.slot_words = (int)(ml.rec(0, 1, 0) - ml.rec(0, 0, 0));
    rs.seq = stamp_of(seq); rs.slot = (int)(seq & 1); rs.flag_value = seq;
    return rs;
}
// what the producer launches of (which, seq) get: their blocks store the partials flagged as well -- when anybody outside the stream reads them
FlagSpec team_flag_spec(mi355cg_team_s* t, TeamPart& p, int which, u64 seq) {
    FlagSpec f{};
    if (t->world > 1) { f.part = p.pll[which]; f.stamp = stamp_of(seq); }
    return f;
}
// The reducer launch of (which, seq): flagged partials of part p -> its record in the other parts' mailboxes.  REC_MAILBOX: on the
// part's side stream, BESIDE the producer launch (it polls the flagged words as they arrive; the producer carries no tail work and
// the consumer launch behind it does not wait for this one).  REC_RCCL / REC_EVENTS: on the compute stream, behind the producer.
void team_reduce(mi355cg_team_s* t, TeamPart& p, int which, int nslots, u64 seq) {
    p.nslots[which] = nslots;
    if (t->world == 1) return;                                    // nobody else: the consumers reduce their own partials, as on a single GPU
    ReduceArgs a{};
    a.part = p.pll[which]; a.nslots = nslots; a.which = which; a.stamp = stamp_of(seq); a.budget = t->budget_ticks;
    a.rs = team_rec_spec(t, p, which, seq);
    hipLaunchKernelGGL(k_reduce_ll, dim3(1), dim3(kBlock), 0, t->rec_mode == REC_MAILBOX ? p.side : p.c->stream, a);
    if (t->rec_mode == REC_EVENTS) hipEventRecord(which == 0 ? p.ev_recA : p.ev_recB, p.c->stream);
}
// the record of the initialisation pass: a one-block launch behind it that reads its plain partials
void team_record(mi355cg_team_s* t, TeamPart& p, int which, int nslots, u64 seq) {
    mi355cg_ctx* c = p.c;
    p.nslots[which] = nslots;
    if (t->world == 1) return;
    TeamRecArgs a{};
    if (which == 0) { a.part = c->partA; a.stride = c->strideA; a.nsum = kNumSumsA; a.lo_off = FA_LO; a.max_first = 0; a.nmax = 0; }
    else { a.part = c->partB; a.stride = c->strideB; a.nsum = kNumSumsB; a.lo_off = FB_LO; a.max_first = FB_RMAX; a.nmax = 3; }
    a.stop_req = nullptr;        // the record of the initialisation pass never carries a stop request: the first one that can is iteration 1's
    a.n = nslots;
    a.rs = team_rec_spec(t, p, which, seq);
    hipLaunchKernelGGL(k_team_record, dim3(1), dim3(kBlock), 0, c->stream, a);
    if (t->rec_mode == REC_EVENTS) hipEventRecord(which == 0 ? p.ev_recA : p.ev_recB, c->stream);
}
// where a consumer launch of part p finds the scalars of phase `which` with sequence number seq: its own partials + the other parts' records
PartSrc team_gsrc(const mi355cg_team_s* t, TeamPart& p, int which, u64 seq) {
    const MboxLayout ml{t->world};
    PartSrc s{};
    s.ptr = which == 0 ? p.c->partA : p.c->partB; s.n = p.nslots[which]; s.fstride = which == 0 ? p.c->strideA : p.c->strideB; s.estride = 1;
    s.rec.mbox = p.mbox + ml.rec(which, (int)(seq & 1), 0); s.rec.world = t->world; s.rec.me = p.rank; s.rec.stamp = stamp_of(seq); s.rec.budget = t->budget_ticks;
    return s;
}
// REC_MAILBOX + WAIT_STREAM: the compute stream itself waits until every OTHER part's record of (which, seq) has been announced
int part_wait_records(mi355cg_team_s* t, TeamPart& p, int which, u64 seq) {
    if (t->rec_mode != REC_MAILBOX || t->wait_mode != WAIT_STREAM) return MI355CG_OK;
    const MboxLayout ml{t->world};
    for (int j = 0; j < t->world; ++j) if (j != p.rank) HIPCK(hipStreamWaitValue64(p.c->stream, p.mbox + ml.flag(which, j), seq, hipStreamWaitValueGte, ~0ull));
    return MI355CG_OK;
}

// after the producers of phase `which` (0 = A, 1 = B) have been enqueued: whatever the transport needs so that every part's
// compute stream may run the consumer launch
int team_exchange_records(mi355cg_team_s* t, int which, u64 seq) {
    if (t->rec_mode == REC_MAILBOX) return MI355CG_OK;                 // the reducer launches deliver; the consumers poll (or their streams wait)
    if (t->rec_mode == REC_RCCL) {
        // in-stream: the compute stream itself carries the all-gather between the producer and the consumer launch
        TeamPart& p = t->parts[0];
        const MboxLayout ml{t->world};
        u64* g = p.mbox + ml.rec(which, (int)(seq & 1), 0);
        hipEvent_t e0 = nullptr, e1 = nullptr;
        if (t->profiling) { e0 = p.c->events.get(); if (e0) hipEventRecord(e0, p.c->stream); }
        if (t->world > 1 || env_int("MI355CG_FORCE_COLLECTIVES", 0))
            NCCLCK(rccl_api()->AllGather(g + (size_t)p.rank * kLLWords, g, kLLWords, ncclUint64, t->comm, p.c->stream));     // in place
        if (t->profiling && e0) { e1 = p.c->events.get(); if (e1) { hipEventRecord(e1, p.c->stream); p.comm_pairs.push_back({e0, e1}); } }
        return MI355CG_OK;
    }
    if (t->parts.size() == 1) return MI355CG_OK;                   // same stream wrote the record
    for (auto& p : t->parts) HIPCK(hipStreamWaitEvent(t->hub, which == 0 ? p.ev_recA : p.ev_recB, 0));
    HIPCK(hipEventRecord(t->ev_hub, t->hub));
    for (auto& p : t->parts) HIPCK(hipStreamWaitEvent(p.c->stream, t->ev_hub, 0));
    return MI355CG_OK;
}

// ---- halo ------------------------------------------------------------------------------------------------------------
int part_halo_in(mi355cg_team_s* t, TeamPart& p);
// RCCL: this rank's messages of one exchange as ONE group on stream hs.  An error inside the group still closes it.
int rccl_halo_group(mi355cg_team_s* t, TeamPart& p, ncclComm_t comm, hipStream_t hs) {
    RcclApi* api = rccl_api();
    const int esz = halo_esz(t);
    NCCLCK(api->GroupStart());
    ncclResult_t first = ncclSuccess;
    for (size_t i = 0; i < p.sends.size() && first == ncclSuccess; ++i) {
        const Seg& s = p.sends[i];
        const void* src = s.kind == 0 ? seg_bytes(p.c, p.c->r, s, esz) : (void*)(p.send_cols + p.send_off[i]);
        first = api->Send(src, (size_t)seg_count(s), s.kind == 0 && t->f32 ? ncclFloat : ncclDouble, s.dst, comm, hs);
    }
    for (size_t i = 0; i < p.recvs.size() && first == ncclSuccess; ++i) {
        const Seg& s = p.recvs[i];
        void* dst = s.kind == 0 ? seg_bytes(p.c, p.c->r, s, esz) : (void*)(p.recv_cols + p.recv_off[i]);
        first = api->Recv(dst, (size_t)seg_count(s), s.kind == 0 && t->f32 ? ncclFloat : ncclDouble, s.src, comm, hs);
    }
    const ncclResult_t end = api->GroupEnd();
    if (first != ncclSuccess) return fail(MI355CG_ERR_HIP, "ncclSend/ncclRecv failed: %s", api->GetErrorString(first));
    NCCLCK(end);
    return MI355CG_OK;
}
// After the update launches (and the column packs) of iteration `seq` have been enqueued: get every part's boundary rows /
// columns of r into its neighbours' ghost cells.  ev_halo (LOCAL, RCCL_STREAM) fires when a part's ghost cells are in place.
int team_exchange_halo(mi355cg_team_s* t, u64 seq) {
    if (t->halo_mode == HALO_PUSH) return MI355CG_OK;                  // part_push_halo, right behind the update launch
    if (t->rccl) {
        TeamPart& p = t->parts[0];
        const bool inl = t->halo_mode == HALO_RCCL_INLINE;
        const hipStream_t hs = inl ? p.c->stream : p.comm;
        if (!inl) HIPCK(hipStreamWaitEvent(p.comm, p.ev_redge, 0));
        hipEvent_t e0 = nullptr, e1 = nullptr;
        if (t->profiling) { e0 = p.c->events.get(); if (e0) hipEventRecord(e0, hs); }
        if (!p.sends.empty() || !p.recvs.empty()) {
            if (int rc = rccl_halo_group(t, p, inl ? t->comm : t->comm_halo, hs)) return rc;
            if (p.unpack.ns) { ColArgs a = p.unpack; a.v = p.c->r; hipLaunchKernelGGL(k_cols, dim3(16), dim3(kBlock), 0, hs, a); }
        }
        if (t->profiling && e0) { e1 = p.c->events.get(); if (e1) { hipEventRecord(e1, hs); p.comm_pairs.push_back({e0, e1}); } }
        if (!inl) HIPCK(hipEventRecord(p.ev_halo, p.comm));
        HIPCK(hipGetLastError());
        return MI355CG_OK;
    }
    for (auto& p : t->parts) { HIPCK(hipSetDevice(p.c->device)); if (int rc = part_halo_in(t, p)) return rc; }      // p = destination
    return MI355CG_OK;
}
// HALO_PUSH, producer side: one launch stores this part's boundary cells into the neighbours' ghost rows / receive buffers and, when
// all of its workgroups have released their stores, the sequence number into the neighbours' halo words
int part_push_halo(mi355cg_team_s* t, TeamPart& p, u64 seq) {
    if (t->halo_mode != HALO_PUSH || p.sends.empty()) return MI355CG_OK;
    const MboxLayout ml{t->world};
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (t->profiling) { e0 = p.c->events.get(); if (e0) hipEventRecord(e0, p.c->stream); }
    PushArgs a = t->f32 ? p.push32 : p.push;
    a.flag_value = seq;
    // one workgroup per 8 KB, at most 32: every workgroup pays a system-scope release and a ticket, so a launch that moves two 32 KB
    // rows is quickest with 8 of them (two polling parts on one GPU, N = 4096: 0.1454 ms per iteration with 8, 0.1465 with 4,
    // 0.1485 - 0.1493 with 32, 0.1484 with 1)
    long long bytes = 0;
    for (int k = 0; k < a.ns; ++k) bytes += (long long)a.s[k].n * (t->f32 ? 4 : 8);
    const int grid = (int)std::min<long long>(32, std::max<long long>(1, (bytes + 8191) / 8192));
    if (t->f32) hipLaunchKernelGGL(k_push<float>, dim3(grid), dim3(kBlock), 0, p.c->stream, a);
    else hipLaunchKernelGGL(k_push<double>, dim3(grid), dim3(kBlock), 0, p.c->stream, a);
    (void)ml;
    if (t->profiling && e0) { e1 = p.c->events.get(); if (e1) { hipEventRecord(e1, p.c->stream); p.comm_pairs.push_back({e0, e1}); } }
    return MI355CG_OK;
}
// HALO_PUSH, consumer side: the compute stream waits for the neighbours' announcements of iteration `seq`, then scatters the columns
int part_wait_halo(mi355cg_team_s* t, TeamPart& p, u64 seq) {
    if (t->halo_mode != HALO_PUSH || p.recvs.empty()) return MI355CG_OK;
    const MboxLayout ml{t->world};
    for (int j : p.halo_from) HIPCK(hipStreamWaitValue64(p.c->stream, p.mbox + ml.halo(j), seq, hipStreamWaitValueGte, ~0ull));
    // row messages are in the ghost rows already; column messages wait in the receive buffer
    if (p.unpack.ns) { ColArgs a = p.unpack; a.v = p.c->r; hipLaunchKernelGGL(k_cols, dim3(16), dim3(kBlock), 0, p.c->stream, a); }
    return MI355CG_OK;
}
bool halo_uses_events(const mi355cg_team_s* t) { return t->halo_mode == HALO_LOCAL || t->halo_mode == HALO_RCCL_STREAM; }

// ---- one part's share of an iteration (called by the one driving thread, or by the part's own thread) --------------
// stencil phase of iteration seq: consumes the update records of seq - 1 (seqB) and the halo of r that followed them; the launch
// that ends it is followed by the part's reducer launch (team_reduce)
int part_stencil_phase(mi355cg_team_s* t, TeamPart& p, const IterCfg& cfg, u64 seqB, u64 seq) {
    mi355cg_ctx* c = p.c;
    hipEvent_t e0 = nullptr;
    const bool ev = halo_uses_events(t);
    const bool two = p.split && t->split_phases == 1 && !t->f32;        // interior launch, then the edge launch behind the halo (fp64 only)
    if (int rc = part_wait_records(t, p, 1, seqB)) return rc;
    if (p.split && !two) {                                               // one launch: the halo has to be there first
        if (ev) HIPCK(hipStreamWaitEvent(c->stream, p.ev_halo, 0));
        if (int rc = part_wait_halo(t, p, seqB)) return rc;
    }
    prof_begin(c, &e0);
    const FlagSpec fl = team_flag_spec(t, p, 0, seq);
    if (two) {
        launch_iteration_stencil<double, 2>(c, cfg, c->r, c->p, Where{c->stream, &c->interior, 0}, team_gsrc(t, p, 1, seqB), &fl);
        prof_end(c, 0, e0);
        if (ev) HIPCK(hipStreamWaitEvent(c->stream, p.ev_halo, 0));
        if (int rc = part_wait_halo(t, p, seqB)) return rc;
        prof_begin(c, &e0);
        launch_iteration_stencil<double, 2>(c, cfg, c->r, c->p, Where{c->stream, &c->edge, c->interior.grid}, team_gsrc(t, p, 1, seqB), &fl);
        prof_end(c, 0, e0);
        team_reduce(t, p, 0, c->interior.grid + c->edge.grid, seq);
    } else if (t->f32) {                                                 // an inner solve of F32_MIXED: the fp32 kernels on 256-column strips
        launch_iteration_stencil<float, 4>(c, cfg, c->rf, c->pf, Where{c->stream, &c->whole32, 0}, team_gsrc(t, p, 1, seqB), &fl);
        prof_end(c, 0, e0);
        team_reduce(t, p, 0, c->whole32.grid, seq);
    } else {
        launch_iteration_stencil<double, 2>(c, cfg, c->r, c->p, whole_part(c), team_gsrc(t, p, 1, seqB), &fl);
        prof_end(c, 0, e0);
        team_reduce(t, p, 0, c->whole.grid, seq);
    }
    c->cur = (c->cur + 1) % c->xsteps;
    return MI355CG_OK;
}
// update phase: edge items first (split), so the halo of r is on its way while the interior is updated
int part_update_phase(mi355cg_team_s* t, TeamPart& p, const IterCfg& cfg, u64 seq) {
    mi355cg_ctx* c = p.c;
    hipEvent_t e0 = nullptr;
    const bool ev = halo_uses_events(t);
    // (split_phases == 2: only this phase is split -- the edge items' rows leave ~60 us before the phase ends, so the neighbours'
    //  rows are there when the next stencil launch, ONE launch, is due; one more launch per iteration instead of two)
    const bool two = p.split && t->split_phases >= 1 && !t->f32;
    if (int rc = part_wait_records(t, p, 0, seq)) return rc;
    prof_begin(c, &e0);
    const FlagSpec fl = team_flag_spec(t, p, 1, seq);
    if (two) {
        launch_iteration_update<double, 2>(c, cfg, c->x, c->r, c->p, c->u, Where{c->stream, &c->edge, c->interior.grid}, team_gsrc(t, p, 0, seq), &fl);
        if (p.pack.ns) { ColArgs a = p.pack; a.v = c->r; hipLaunchKernelGGL(k_cols, dim3(16), dim3(kBlock), 0, c->stream, a); }
        prof_end(c, 1, e0);
        if (ev) HIPCK(hipEventRecord(p.ev_redge, c->stream));
        if (int rc = part_push_halo(t, p, seq)) return rc;
        prof_begin(c, &e0);
        launch_iteration_update<double, 2>(c, cfg, c->x, c->r, c->p, c->u, Where{c->stream, &c->interior, 0}, team_gsrc(t, p, 0, seq), &fl);
        prof_end(c, 1, e0);
        team_reduce(t, p, 1, c->interior.grid + c->edge.grid, seq);
    } else {
        if (t->f32) launch_iteration_update<float, 4>(c, cfg, c->xf, c->rf, c->pf, (const float*)nullptr, Where{c->stream, &c->whole32, 0}, team_gsrc(t, p, 0, seq), &fl);
        else launch_iteration_update<double, 2>(c, cfg, c->x, c->r, c->p, c->u, whole_part(c), team_gsrc(t, p, 0, seq), &fl);
        prof_end(c, 1, e0);
        team_reduce(t, p, 1, t->f32 ? c->whole32.grid : c->whole.grid, seq);
        if (p.pack.ns && !t->f32) { ColArgs a = p.pack; a.v = c->r; hipLaunchKernelGGL(k_cols, dim3(16), dim3(kBlock), 0, c->stream, a); }
        if (ev) HIPCK(hipEventRecord(p.ev_redge, c->stream));
        if (int rc = part_push_halo(t, p, seq)) return rc;
    }
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
        const int esz = s.kind == 0 ? halo_esz(t) : 8;
        const void* src = nullptr;
        if (s.kind == 0) src = seg_bytes(q->c, q->c->r, s, esz);
        else for (size_t j = 0; j < q->sends.size(); ++j) if (q->sends[j].id == s.id) src = q->send_cols + q->send_off[j];
        void* dst = s.kind == 0 ? seg_bytes(p.c, p.c->r, s, esz) : (void*)(p.recv_cols + p.recv_off[i]);
        HIPCK(hipMemcpyAsync(dst, src, (size_t)esz * seg_count(s), hipMemcpyDefault, p.comm));
    }
    if (p.unpack.ns) { ColArgs a = p.unpack; a.v = p.c->r; hipLaunchKernelGGL(k_cols, dim3(16), dim3(kBlock), 0, p.comm, a); }
    HIPCK(hipEventRecord(p.ev_halo, p.comm));
    HIPCK(hipGetLastError());
    return MI355CG_OK;
}
// the decision of the last iteration, for the host (every part evaluates it: all parts hold the same records)
int part_poll_enqueue(mi355cg_team_s* t, TeamPart& p, const IterCfg& cfg, u64 seqB) {
    if (int rc = part_wait_records(t, p, 1, seqB)) return rc;
    launch_check(p.c, cfg, p.c->stream, team_gsrc(t, p, 1, seqB));
    return MI355CG_OK;
}

// ---- LOCAL transport with one host thread per part -------------------------------------------------------------------
// One thread issues ~20 runtime calls per part and iteration; with the parts on 8 different GPUs that is the host, not the GPUs,
// setting the pace.  Here every part has its own thread; the threads meet at barriers inside an iteration, each right after the
// events the other parts are about to wait on have been recorded (an event has to be RECORDED before another thread may enqueue
// a wait on it).  With the records in mailboxes (REC_MAILBOX) only the halo events are left: one barrier per iteration.
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
    u64 chunk_seq0 = 0;
    bool chunk_poll = false, quit = false;

    void note_error(int rc) { int zero = 0; if (error.compare_exchange_strong(zero, rc)) { std::lock_guard<std::mutex> g(mu); error_text = g_err; } }
    // spinning barrier for the meeting points inside an iteration (microseconds apart); gives way when there are more
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
    // the iterations of one chunk as seen by part i; seq0 = sequence number of the last update records before the chunk
    int run_chunk(int i, int m, bool poll, u64 seq0) {
        TeamPart& p = t->parts[i];
        const bool ev_rec = t->rec_mode == REC_EVENTS;
        u64 seqB = seq0;
        for (int k = 0; k < m; ++k) {
            const u64 seq = seqB + 1;
            if (int rc = part_stencil_phase(t, p, cfg, seqB, seq)) return rc;
            if (ev_rec) {
                if (!barrier()) return MI355CG_ERR_STATE;
                for (auto& q : t->parts) if (&q != &p) HIPCK(hipStreamWaitEvent(p.c->stream, q.ev_recA, 0));
            }
            if (int rc = part_update_phase(t, p, cfg, seq)) return rc;
            if (!barrier()) return MI355CG_ERR_STATE;
            if (t->halo_mode == HALO_LOCAL) if (int rc = part_halo_in(t, p)) return rc;
            if (ev_rec) for (auto& q : t->parts) if (&q != &p) HIPCK(hipStreamWaitEvent(p.c->stream, q.ev_recB, 0));
            seqB = seq;
        }
        if (poll) if (int rc = part_poll_enqueue(t, p, cfg, seqB)) return rc;
        return MI355CG_OK;
    }
    void worker(int i) {
        if (hipSetDevice(t->parts[i].c->device) != hipSuccess) { fail(MI355CG_ERR_HIP, "hipSetDevice failed in a team thread"); note_error(MI355CG_ERR_HIP); }
        int seen = 0;
        for (;;) {
            int m; bool poll; u64 seq0;
            {
                std::unique_lock<std::mutex> g(mu);
                cv.wait(g, [&] { return quit || chunk_seq != seen; });
                if (quit) return;
                seen = chunk_seq; m = chunk_m; poll = chunk_poll; seq0 = chunk_seq0;
            }
            if (!error.load()) if (int rc = run_chunk(i, m, poll, seq0)) { if (rc != MI355CG_ERR_STATE || !error.load()) note_error(rc); }
            { std::lock_guard<std::mutex> g(mu); ++done_count; }
            cv.notify_all();
        }
    }
    // called by the solving thread (which is part 0's thread): run m iterations on every part, return when all are enqueued
    int chunk(int m, bool poll, u64 seq0) {
        { std::lock_guard<std::mutex> g(mu); chunk_m = m; chunk_poll = poll; chunk_seq0 = seq0; done_count = 0; ++chunk_seq; }
        cv.notify_all();
        if (hipSetDevice(t->parts[0].c->device) != hipSuccess) { fail(MI355CG_ERR_HIP, "hipSetDevice failed"); note_error(MI355CG_ERR_HIP); }
        if (!error.load()) if (int rc = run_chunk(0, m, poll, seq0)) { if (rc != MI355CG_ERR_STATE || !error.load()) note_error(rc); }
        { std::unique_lock<std::mutex> g(mu); cv.wait(g, [&] { return done_count == nthreads - 1; }); }
        if (const int rc = error.load()) { g_err = error_text; return rc; }
        return MI355CG_OK;
    }
};

int env_choice(const char* name, std::initializer_list<const char*> names, int dflt) {
    const char* v = getenv(name);
    if (!v || !*v) return dflt;
    int k = 0;
    for (const char* n : names) { if (strcmp(v, n) == 0) return k; ++k; }
    return dflt;
}

// How this solve moves records and halos (every rank reads the same environment and holds the same ipc_ok / shared_device):
//   MI355CG_TEAM_RECORDS = auto | rccl | mailbox | events      MI355CG_TEAM_WAIT = auto | kernel | stream
//   MI355CG_TEAM_HALO    = auto | inline | stream | push       MI355CG_TEAM_TIMEOUT_MS
int team_pick_modes(mi355cg_team_s* t) {
    bool several_devices = false;
    for (auto& p : t->parts) if (p.c->device != t->parts[0].c->device) several_devices = true;
    const int rec_env = env_choice("MI355CG_TEAM_RECORDS", {"auto", "rccl", "mailbox", "events"}, 0);
    const int wait_env = env_choice("MI355CG_TEAM_WAIT", {"auto", "kernel", "stream"}, 0);
    int halo_env = env_choice("MI355CG_TEAM_HALO", {"auto", "inline", "stream", "push"}, 0);
    if (halo_env == 0 && env_int("MI355CG_TEAM_HALO_INLINE", 0)) halo_env = 1;
    t->split_phases = std::min(2, std::max(0, env_int("MI355CG_TEAM_SPLIT", 0)));
    t->timeout_s = std::max(1, env_int("MI355CG_TEAM_TIMEOUT_MS", 30000)) * 1e-3;
    t->budget_ticks = (u64)(t->timeout_s * 1e8);
    if (t->rccl) {
        const bool solo = t->world == 1;
        t->rec_mode = rec_env == 1 ? REC_RCCL : rec_env == 2 ? REC_MAILBOX : ((t->ipc_ok || solo) ? REC_MAILBOX : REC_RCCL);
        if (rec_env == 3) return fail(MI355CG_ERR_INVALID, "MI355CG_TEAM_RECORDS=events is for one-process teams");
        if (t->rec_mode == REC_MAILBOX && !(t->ipc_ok || solo)) return fail(MI355CG_ERR_STATE, "MI355CG_TEAM_RECORDS=mailbox: the ranks could not map each other's mailboxes (IPC)");
        // kernels of ranks that share one physical GPU must not spin on each other (the waiting one may keep the producing one off the CUs)
        t->wait_mode = wait_env == 1 ? WAIT_KERNEL : wait_env == 2 ? WAIT_STREAM : (t->shared_device ? WAIT_STREAM : WAIT_KERNEL);
        t->halo_mode = halo_env == 1 ? HALO_RCCL_INLINE : halo_env == 2 ? HALO_RCCL_STREAM : halo_env == 3 ? HALO_PUSH : ((t->ipc_ok || solo) ? HALO_PUSH : HALO_RCCL_INLINE);
        if (t->halo_mode == HALO_PUSH && !(t->ipc_ok || solo)) return fail(MI355CG_ERR_STATE, "MI355CG_TEAM_HALO=push: the ranks could not map each other's vectors (IPC)");
        if (t->halo_mode == HALO_RCCL_STREAM && !solo && !t->comm_halo) return fail(MI355CG_ERR_STATE, "MI355CG_TEAM_HALO=stream needs the second communicator (create the team with MI355CG_TEAM_HALO=stream set)");
        if (t->split_phases && t->halo_mode == HALO_RCCL_INLINE) t->split_phases = 0;      // nothing to overlap with
    } else {
        // one process: parts that share a GPU order their streams with events (a kernel that polls would keep the producer off the
        // CUs); parts on GPUs of their own poll their mailboxes -- no hub stream, no event joins on the critical path
        if (rec_env == 1) return fail(MI355CG_ERR_INVALID, "MI355CG_TEAM_RECORDS=rccl is for one-process-per-GPU teams");
        const bool distinct = t->parts.size() == 1 || [&] { for (auto& a : t->parts) for (auto& b : t->parts) if (&a != &b && a.c->device == b.c->device) return false; return true; }();
        t->rec_mode = rec_env == 2 ? REC_MAILBOX : rec_env == 3 ? REC_EVENTS : (distinct ? REC_MAILBOX : REC_EVENTS);
        t->wait_mode = wait_env == 2 ? WAIT_STREAM : wait_env == 1 ? WAIT_KERNEL : (distinct ? WAIT_KERNEL : WAIT_STREAM);
        t->halo_mode = halo_env == 3 ? HALO_PUSH : HALO_LOCAL;
        (void)several_devices;
    }
    for (auto& p : t->parts) if (t->halo_mode == HALO_PUSH && p.push.ns < 0) return fail(MI355CG_ERR_STATE, "push halo: a neighbour's memory is not mapped");
    // WAIT_KERNEL: a consumer launch fills the CUs with workgroups that poll for the OTHER parts' records, and those parts wait for
    // THIS part's reducer launch -- which therefore has to become resident beside the consumer: 72 VGPRs per wave next to two consumer
    // waves per SIMD (192 each in the fp64 default launches, 216 in the fp32 ones: 128 / 80 of 512 left).  The 12-word update of MI355CG_XSTEPS=8 takes 264: no room,
    // the reducer would wait for a wave that never ends.  Those teams let their streams wait instead.
    if (t->rec_mode == REC_MAILBOX && t->wait_mode == WAIT_KERNEL && wait_env != 1)
        for (auto& p : t->parts) if (p.c->xsteps > 4) t->wait_mode = WAIT_STREAM;
    return MI355CG_OK;
}

int team_solve(mi355cg_team_s* t, const mi355cg_params* prm, mi355cg_iter_cb cb, void* user,
               const volatile int* stop_flag, mi355cg_results* out) {
    if (prm->rule != MI355CG_RULE_MSG_MAXNORM && prm->rule != MI355CG_RULE_REL_2NORM) return fail(MI355CG_ERR_INVALID, "unknown rule %d", prm->rule);
    if (prm->diagnostics) return fail(MI355CG_ERR_INVALID, "per-iteration diagnostics are not available on a team");
    if (t->broken) return fail(MI355CG_ERR_STATE, "this team abandoned an earlier solve and cannot be used again");
    if (t->f32 && (prm->rule != MI355CG_RULE_REL_2NORM || prm->use_true_solution)) return fail(MI355CG_ERR_INVALID, "the fp32 inner CG runs the REL_2NORM rule without a true solution");
    const bool msg = prm->rule == MI355CG_RULE_MSG_MAXNORM;
    const IterCfg cfg = make_cfg(prm);             // has_u: u is read on every iteration here (the single-GPU path skips it where unobservable)
    const auto t0 = std::chrono::steady_clock::now();
    *t->stop_h = 0;
    if (int rc = team_pick_modes(t)) return rc;
    const bool ev = halo_uses_events(t);

    // x = 0, r = b, z = 0; partial norms of r0; first record + halo of r0 = b
    const u64 seq_init = ++t->seq;
    for (auto& p : t->parts) {
        mi355cg_ctx* c = p.c;
        HIPCK(hipSetDevice(c->device));
        if (cfg.has_u) if (int rc = ensure_u_on_device(c)) return rc;
        c->events.reset(); c->ev_pairs[0].clear(); c->ev_pairs[1].clear(); p.comm_pairs.clear();
        c->profiling = t->profiling;
        // Block 0 of every update launch samples the pinned stop word (as mi355cg_solve's do): the sample goes into the part's state, for
        // its own next stencil prologue, and into its record, for the other parts'.  Every part ORs its own sample with the samples in the
        // other parts' records, so all of them take the decision INTERRUPTED in the same iteration.
        c->stop_dev = nullptr;
        HIPCK(hipHostGetDevicePointer((void**)&c->stop_dev, t->stop_h, 0));
        // one pass over the owned range (mi355cg_solve does the same); the ghost cells of the first direction are zeroed too:
        // they still hold the neighbours' last direction of the previous solve.  Row slabs: the two ghost rows; 2-D parts: the
        // whole vector (ghost columns are strided); a part that is the whole grid has no ghost cells.
        int nslots = c->whole.grid;
        if (t->f32) {
            // an inner solve of F32_MIXED (as inner_cg_f32): rf -- in r's memory -- holds its right-hand side; correction, direction
            // ring and A p start from zero; one flat pass measures r0 and writes the fresh state
            const size_t bytes = sizeof(float) * c->storage_len;
            HIPCK(hipMemsetAsync(c->xf, 0, bytes, c->stream));
            for (int k = 0; k < c->xsteps; ++k) HIPCK(hipMemsetAsync(c->pf[k], 0, bytes, c->stream));
            HIPCK(hipMemsetAsync(c->apf, 0, bytes, c->stream));
            c->cur = 0;
            launch_update_flat<float, 4>(c, cfg, c->xf, c->rf, c->pf[0], c->apf, (const float*)nullptr, c->stream, c->whole32.grid);
            nslots = c->whole32.grid;
        } else {
        if (c->is_slab && c->has_gc) HIPCK(hipMemsetAsync(c->p[0], 0, sizeof(double) * c->storage_len, c->stream));
        else if (c->is_slab) {
            const Geom& g = c->g;
            for (int y : {g.y_lo - 1, g.y_hi + 1})
                HIPCK(hipMemsetAsync(c->p[0] + (phys_start(g, y) - g.base0), 0, sizeof(double) * (size_t)(phys_end(g, y) - phys_start(g, y)), c->stream));
        }
        if (c->qctr && c->dyn_rows > 0) HIPCK(hipMemsetAsync(c->qctr, 0, sizeof(int) * 2 * kXcds * kQueueSubs * kQueuePitch, c->stream));
        c->cur = 0;
        {
            FreshArgs<double> f{};
            f.begin = c->g.own_begin / 2; f.nvec = c->g.own_len / 2;
            f.b = c->b; f.x = c->x; f.r = c->r; f.p0 = c->p[0]; f.u = c->u;
            f.partB = c->partB; f.strideB = c->strideB; f.s_out = c->sB;
            if (cfg.has_u) hipLaunchKernelGGL((k_init_fresh<double, 2, true>), dim3(c->whole.grid), dim3(kBlock), 0, c->stream, f);
            else hipLaunchKernelGGL((k_init_fresh<double, 2, false>), dim3(c->whole.grid), dim3(kBlock), 0, c->stream, f);
        }
        }
        if (p.pack.ns && !t->f32) { ColArgs a = p.pack; a.v = c->r; hipLaunchKernelGGL(k_cols, dim3(16), dim3(kBlock), 0, c->stream, a); }
        if (ev) HIPCK(hipEventRecord(p.ev_redge, c->stream));
        team_record(t, p, 1, nslots, seq_init);
        if (int rc = part_push_halo(t, p, seq_init)) return rc;
        HIPCK(hipGetLastError());
        c->solved = true;
    }
    if (int rc = team_exchange_halo(t, seq_init)) return rc;
    if (int rc = team_exchange_records(t, 1, seq_init)) return rc;

    TeamPart& lead = t->parts[0];
    u64 seqB = seq_init;                                           // sequence number of the newest update records
    auto poll_fetch = [&]() -> int {
        HIPCK(hipSetDevice(lead.c->device));
        HIPCK(hipMemcpyAsync(lead.c->summary_h, lead.c->summary, sizeof(CgState), hipMemcpyDeviceToHost, lead.c->stream));
        HIPCK(hipMemcpyAsync(lead.c->hist_h, lead.c->hist, sizeof(HistEntry) * kHist, hipMemcpyDeviceToHost, lead.c->stream));
        HIPCK(hipGetLastError());
        const int rc = bounded_sync(t, lead.c->stream, stop_flag, t->timeout_s + 2e-8 * (double)t->budget_ticks);
        if (rc < 0) return team_abandon(t, "the lead part's stream did not drain");
        return rc;
    };
    auto poll = [&]() -> int {
        for (auto& p : t->parts) { HIPCK(hipSetDevice(p.c->device)); if (int rc = part_poll_enqueue(t, p, cfg, seqB)) return rc; }
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
    // Never with stream-level value waits between parts that share a device: a process's streams are multiplexed onto a few hardware
    // queues, and a wait at the head of one of them holds back every stream behind it -- the producer of the awaited word too, unless
    // its packets were enqueued first.  The one-thread loop enqueues in that order; threads do not (seen as a hang inside the runtime
    // with MI355CG_TEAM_THREADS=1 + mailboxes + pushed halo on one GPU).  Rank processes have queues of their own.
    bool shares = false;
    for (auto& a : t->parts) for (auto& b : t->parts) if (&a != &b && a.c->device == b.c->device) shares = true;
    const bool value_waits = (t->rec_mode == REC_MAILBOX && t->wait_mode == WAIT_STREAM) || t->halo_mode == HALO_PUSH;
    if (!t->rccl && t->parts.size() > 1 && env_int("MI355CG_TEAM_THREADS", several_devices ? 1 : 0) != 0 && !(shares && value_waits)) {
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
    // The chunk schedule is a function of the PARAMETERS only, so it is the same on every rank whatever callbacks or stop flags the
    // ranks were given: all ranks enqueue the same launches and collectives, poll after the same iterations and leave the loop at the
    // same poll.  (MSG rule with a callback cadence: the first iteration is a chunk of its own, so the it = 1 callback is delivered
    // -- and a stop requested from it seen -- before more work is queued.)
    bool interrupted = false, first_chunk = msg && every > 0;
    const bool act_at_once = !t->rccl || t->world == 1;            // one process holds every part: nobody else to keep in step
    if (!lead.c->ev_loop[0]) { HIPCK(hipEventCreate(&lead.c->ev_loop[0])); HIPCK(hipEventCreate(&lead.c->ev_loop[1])); }
    HIPCK(hipSetDevice(lead.c->device));
    HIPCK(hipEventRecord(lead.c->ev_loop[0], lead.c->stream));
    while (!lead.c->summary_h->done) {
        // A stop request (msg_solver.cpp:82-87).  One process: act on it at once between chunks.  Always: raise the pinned word; it
        // travels with the next update record of this rank, every rank finds max(stop words) > 0 in the records of that iteration and
        // takes the decision INTERRUPTED in the same stencil prologue -- in the middle of a chunk, on all ranks at the same iteration.
        const bool want_stop = stop_flag && *stop_flag;
        if (want_stop && act_at_once) { interrupted = true; break; }
        if (want_stop) *t->stop_h = 1;
        int m = std::min(sync_every, prm->max_iterations - it_done);
        if (msg && every > 0) m = std::min(m, every - it_done % every);
        if (first_chunk) { m = 1; first_chunk = false; }
        if (m <= 0) m = 1;
        if (crew) {
            if (int rc = crew->chunk(m, true, seqB)) return rc;
            seqB += m; t->seq = seqB;
            if (int rc = poll_fetch()) return rc;
        } else {
            for (int k = 0; k < m; ++k) {
                const u64 seq = ++t->seq;
                for (auto& p : t->parts) { HIPCK(hipSetDevice(p.c->device)); if (int rc = part_stencil_phase(t, p, cfg, seqB, seq)) return rc; }
                if (int rc = team_exchange_records(t, 0, seq)) return rc;
                for (auto& p : t->parts) { HIPCK(hipSetDevice(p.c->device)); if (int rc = part_update_phase(t, p, cfg, seq)) return rc; }
                if (int rc = team_exchange_halo(t, seq)) return rc;
                if (int rc = team_exchange_records(t, 1, seq)) return rc;
                seqB = seq;
            }
            if (int rc = poll()) return rc;
        }
        if (lead.c->summary_h->done && lead.c->summary_h->reason == kReasonTransport) {
            t->broken = true;
            return fail(MI355CG_ERR_STATE, "a part's record did not arrive within %.0f s (MI355CG_TEAM_TIMEOUT_MS): a peer stopped delivering.  The team cannot be used again", t->timeout_s);
        }
        const int it_now = lead.c->summary_h->it;
        if (msg && cb) for (int it = it_done + 1; it <= it_now; ++it) {
            const int reason = lead.c->summary_h->reason;
            const bool stopped_here = lead.c->summary_h->done && reason != MI355CG_STOP_ITERATIONS && reason != MI355CG_STOP_INTERRUPTED && it == it_now;
            const HistEntry& h = lead.c->hist_h[it % kHist];
            if ((it == 1 || (every > 0 && it % every == 0)) && !stopped_here) cb(user, it, h.dmax, h.rmax, cfg.has_u ? h.emax : DBL_MAX);
        }
        it_done = it_now;
    }
    HIPCK(hipSetDevice(lead.c->device));
    HIPCK(hipEventRecord(lead.c->ev_loop[1], lead.c->stream));
    const CgState fin = *lead.c->summary_h;
    if (fin.done && fin.reason == MI355CG_STOP_INTERRUPTED) interrupted = true;
    for (auto& p : t->parts) {
        mi355cg_ctx* c = p.c;
        HIPCK(hipSetDevice(c->device));
        if (ev) HIPCK(hipStreamWaitEvent(c->stream, p.ev_halo, 0));                       // the last halo exchange writes this part's ghost cells
        if (int rc = part_wait_halo(t, p, seqB)) return rc;                               // (PUSH: the neighbours' last push has landed before anything else touches r)
        c->cur = fin.it % c->xsteps;
        if (cfg.x2 && t->f32) launch_flush_x<float, 4>(c, c->whole32, c->xf, c->pf, fin, c->stream);
        else if (cfg.x2) launch_flush_x<double, 2>(c, c->whole, c->x, c->p, fin, c->stream);
        HIPCK(hipGetLastError());
    }
    for (auto& p : t->parts) {
        HIPCK(hipSetDevice(p.c->device));
        for (hipStream_t st : {p.c->stream, p.comm, p.side}) {
            const int rc = bounded_sync(t, st, nullptr, t->timeout_s + 2e-8 * (double)t->budget_ticks);
            if (rc < 0) return team_abandon(t, "a part's stream did not drain after the last iteration");
            if (rc) return rc;
        }
    }
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

// ---- F32_MIXED on a team (BASELINE config 3 across GPUs): fp64 iterative refinement around the team's CG loop on fp32 vectors --------
// The algorithm is solve_mixed's (mi355cg.hip), statement for statement; what a team adds is the halo of x before the fp64 operator
// apply and a sum over the parts for the residual norm.  There is no reference twin (the reference is fp64 only).

// One exchange of the residual vector's ghost cells outside the CG loop (its sequence number is spent on the halo alone).
int team_halo_now(mi355cg_team_s* t) {
    const u64 seq = ++t->seq;
    const bool ev = halo_uses_events(t);
    for (auto& p : t->parts) {
        HIPCK(hipSetDevice(p.c->device));
        if (p.pack.ns) { ColArgs a = p.pack; a.v = p.c->r; hipLaunchKernelGGL(k_cols, dim3(16), dim3(kBlock), 0, p.c->stream, a); }
        if (ev) HIPCK(hipEventRecord(p.ev_redge, p.c->stream));
        if (int rc = part_push_halo(t, p, seq)) return rc;
    }
    if (int rc = team_exchange_halo(t, seq)) return rc;
    for (auto& p : t->parts) {
        HIPCK(hipSetDevice(p.c->device));
        if (ev) HIPCK(hipStreamWaitEvent(p.c->stream, p.ev_halo, 0));
        if (int rc = part_wait_halo(t, p, seq)) return rc;
    }
    return MI355CG_OK;
}
// every rank: the sum over all parts of one number per part, added in part order (the same bits on every rank).  Also a meeting
// point: when it returns, every rank has finished what it did before calling it.
int team_sum(mi355cg_team_s* t, const std::vector<double>& mine, double* total) {
    std::vector<double> all(t->world, 0.0);
    if (t->rccl && t->world > 1) { if (int rc = boot_all_gather(t, mine.data(), all.data(), sizeof(double))) return rc; }
    else for (size_t i = 0; i < t->parts.size(); ++i) all[t->parts[i].rank] = mine[i];
    double s = 0.0;
    for (double v : all) s += v;
    *total = s;
    return MI355CG_OK;
}

// every part's compute stream has drained, on every rank
int team_meet(mi355cg_team_s* t, const char* what) {
    for (auto& p : t->parts) {
        HIPCK(hipSetDevice(p.c->device));
        const int rc = bounded_sync(t, p.c->stream, nullptr, t->timeout_s);
        if (rc < 0) return team_abandon(t, what);
        if (rc) return rc;
    }
    std::vector<double> one(t->parts.size(), 1.0);
    double n = 0;
    return team_sum(t, one, &n);
}

int team_solve_mixed(mi355cg_team_s* t, const mi355cg_params* prm, mi355cg_iter_cb cb, void* user,
                     const volatile int* stop_flag, mi355cg_results* out) {
    if (prm->rule != MI355CG_RULE_REL_2NORM) return fail(MI355CG_ERR_INVALID, "F32_MIXED offers the REL_2NORM rule only");
    if (t->broken) return fail(MI355CG_ERR_STATE, "this team abandoned an earlier solve and cannot be used again");
    const auto t0 = std::chrono::steady_clock::now();
    if (int rc = team_pick_modes(t)) return rc;
    for (auto& p : t->parts) if (t->halo_mode == HALO_PUSH && p.push32.ns < 0) return fail(MI355CG_ERR_STATE, "push halo: a neighbour's memory is not mapped");
    const double inner_eps = prm->inner_eps > 0 ? prm->inner_eps : 1e-4;
    const int rgrid = 1024;
    // rf = (float)(b - ap) on every part (into r's memory: the fp64 residual is not kept between the stages), *norm = ||b - ap||_2 over the team.
    // The sum is a meeting point of the ranks: nobody's next halo message can land in ghost cells this pass is still writing.
    auto residual_pass = [&](double* norm) -> int {
        std::vector<double> mine(t->parts.size(), 0.0);
        for (auto& p : t->parts) {
            mi355cg_ctx* c = p.c;
            HIPCK(hipSetDevice(c->device));
            hipLaunchKernelGGL(k_residual_to_f32, dim3(rgrid), dim3(kBlock), 0, c->stream, c->storage_len, c->g.own_begin, c->g.own_len, c->b, c->ap, c->rf, c->partR);
            HIPCK(hipGetLastError());
            HIPCK(hipMemcpyAsync(c->partR_h, c->partR, sizeof(double) * rgrid, hipMemcpyDeviceToHost, c->stream));
        }
        for (size_t i = 0; i < t->parts.size(); ++i) {
            mi355cg_ctx* c = t->parts[i].c;
            HIPCK(hipSetDevice(c->device));
            const int rc = bounded_sync(t, c->stream, nullptr, t->timeout_s);
            if (rc < 0) return team_abandon(t, "a part's stream did not drain (residual pass)");
            if (rc) return rc;
            double s = 0; for (int k = 0; k < rgrid; ++k) s += c->partR_h[k];
            mine[i] = s;
        }
        double total = 0;
        if (int rc = team_sum(t, mine, &total)) return rc;
        *norm = std::sqrt(total);
        return MI355CG_OK;
    };
    for (auto& p : t->parts) {
        mi355cg_ctx* c = p.c;
        HIPCK(hipSetDevice(c->device));
        const size_t bytes64 = sizeof(double) * c->storage_len;
        HIPCK(hipMemsetAsync(c->x, 0, bytes64, c->stream));
        HIPCK(hipMemsetAsync(c->ap, 0, bytes64, c->stream));
    }
    double bnorm = 0, rnorm = 0;
    if (int rc = residual_pass(&bnorm)) return rc;       // x = 0: r = b
    rnorm = bnorm;
    int total = 0, outer = 0;
    bool interrupted = false, converged = bnorm == 0.0;
    double loop_s = 0;
    while (!converged && total < prm->max_iterations && !interrupted) {
        mi355cg_params ip = *prm;
        ip.eps_rel = inner_eps; ip.diagnostics = 0; ip.use_true_solution = 0; ip.callback_every = 0;
        ip.max_iterations = prm->max_iterations - total;              // restarted refinement: every inner solve starts from scratch with the remaining budget
        mi355cg_results ir{};
        t->f32 = true;
        const int rc_in = team_solve(t, &ip, nullptr, nullptr, stop_flag, &ir);
        t->f32 = false;
        if (rc_in) return rc_in;
        const int its = ir.iterations;
        interrupted = ir.stop_reason == MI355CG_STOP_INTERRUPTED;
        loop_s += ir.loop_seconds;
        total += its; ++outer;
        for (auto& p : t->parts) {
            mi355cg_ctx* c = p.c;
            HIPCK(hipSetDevice(c->device));
            hipLaunchKernelGGL(k_accumulate_f32, dim3(flat_grid(c->g.own_len)), dim3(kBlock), 0, c->stream, c->g.own_begin, c->g.own_len, c->x, c->xf);
            // The halo machinery is about the residual vector: x travels as its guest.  ALL of it is copied: r's memory held the fp32
            // residual until now, and the operator reads zeros in the boundary rows, the pads and the columns outside the domain.
            HIPCK(hipMemcpyAsync(c->r, c->x, sizeof(double) * c->storage_len, hipMemcpyDeviceToDevice, c->stream));
        }
        // (the copy also clears the ghost cells, so no neighbour may have delivered yet: meet first)
        if (int rc = team_meet(t, "the copy of x")) return rc;
        if (int rc = team_halo_now(t)) return rc;
        for (auto& p : t->parts) {
            HIPCK(hipSetDevice(p.c->device));
            launch_apply<double, 2>(p.c, p.c->r, p.c->ap, whole_part(p.c));
        }
        const double prev = rnorm;
        if (int rc = residual_pass(&rnorm)) return rc;
        if (cb) cb(user, total, 0.0, rnorm, 0.0);
        converged = !prm->fixed_iterations && rnorm <= prm->eps_rel * bnorm;
        if (prm->fixed_iterations || its == 0) break;
        if (!converged && rnorm > 0.5 * prev) break;      // fp32 cannot improve this x any further
    }
    // leave the fp64 residual of the returned x in r (mi355cg_team_get_vector(1))
    for (auto& p : t->parts) {
        mi355cg_ctx* c = p.c;
        HIPCK(hipSetDevice(c->device));
        HIPCK(hipMemsetAsync(c->r, 0, sizeof(double) * c->storage_len, c->stream));      // (boundary rows and pads held fp32 data: the fp64 kernels read zeros there)
        hipLaunchKernelGGL((k_sub<double>), dim3(flat_grid(c->g.own_len)), dim3(kBlock), 0, c->stream, c->g.own_begin, c->g.own_len, c->b, c->ap, c->r);
        HIPCK(hipGetLastError());
        const int rc = bounded_sync(t, c->stream, nullptr, t->timeout_s);
        if (rc < 0) return team_abandon(t, "a part's stream did not drain (end of the mixed solve)");
        if (rc) return rc;
        c->solved = true; c->cur = 0;
    }
    // nobody's next solve may write ghost cells a slower rank's last pass still reads
    if (int rc = team_meet(t, "the end of the mixed solve")) return rc;
    mi355cg_results res{};
    res.iterations = total; res.converged = converged ? 1 : 0;
    res.stop_reason = interrupted ? MI355CG_STOP_INTERRUPTED : (converged ? MI355CG_STOP_RESIDUAL : MI355CG_STOP_ITERATIONS);
    res.final_residual_norm = res.final_precision = res.final_error_norm = DBL_MAX;
    res.r_norm2 = rnorm; res.initial_r_norm2 = bnorm;
    res.refine_outer = outer; res.refine_true_rel = bnorm > 0 ? rnorm / bnorm : 0.0;
    res.solve_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    res.loop_seconds = loop_s;
    if (out) *out = res;
    return MI355CG_OK;
}

}  // namespace

extern "C" {

// MI355CG_F32_MIXED: the team's solves become fp64 iterative refinement around an fp32 inner CG (mi355cg_create's dtype, for a team;
// BASELINE config 3 across GPUs).  Row slabs only: the fp32 kernels march 256-column strips.  Collective: every rank makes the same call.
int mi355cg_team_set_dtype(mi355cg_team t, int dtype) {
    if (!t) return fail(MI355CG_ERR_INVALID, "null team");
    if (dtype != MI355CG_F64 && dtype != MI355CG_F32_MIXED) return fail(MI355CG_ERR_INVALID, "unknown dtype %d", dtype);
    if (dtype == MI355CG_F32_MIXED) {
        for (auto& p : t->parts) {
            HIPCK(hipSetDevice(p.c->device));
            if (int rc = ensure_f32_vectors(p.c)) return rc;
            p.c->rf = (float*)p.c->r;                    // the fp32 residual lives in the fp64 residual's memory: see seg_bytes
        }
    }
    t->dtype = dtype;
    return MI355CG_OK;
}

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
    if (int rc = team_build_tables(t)) { team_free(t); return rc; }
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

namespace {
// What every rank tells the others at team creation: IPC handles of the three allocations its neighbours write into, and the
// physical GPU it sits on.
struct BootRec { hipIpcMemHandle_t slab, r; char bus[32]; int ok, pad; unsigned long long nonce; };
// all-gather of `bytes` per rank through the team's communicator (host buffers; staged through device memory)
int boot_all_gather(mi355cg_team_s* t, const void* mine, void* all, size_t bytes) {
    TeamPart& p = t->parts[0];
    unsigned char* d = nullptr;
    HIPCK(hipMalloc((void**)&d, bytes * t->world));
    int rc = MI355CG_OK;
    if (hipMemcpy(d + bytes * p.rank, mine, bytes, hipMemcpyHostToDevice) != hipSuccess) rc = fail(MI355CG_ERR_HIP, "boot upload failed");
    if (!rc) {
        const ncclResult_t r = rccl_api()->AllGather(d + bytes * p.rank, d, bytes, ncclUint8, t->comm, p.comm);
        if (r != ncclSuccess) rc = fail(MI355CG_ERR_HIP, "ncclAllGather (team bootstrap) failed: %s", rccl_api()->GetErrorString(r));
    }
    if (!rc && (hipStreamSynchronize(p.comm) != hipSuccess || hipMemcpy(all, d, bytes * t->world, hipMemcpyDeviceToHost) != hipSuccess)) rc = fail(MI355CG_ERR_HIP, "boot download failed");
    hipFree(d);
    return rc;
}
// Map every other rank's mailbox / residual / column buffer (IPC).  Collective.  Leaves t->ipc_ok = false (and nothing mapped)
// when any rank could not export or open a handle, or does not read the owner's bytes through a mapping: the team then runs on
// RCCL alone.  The check is not paranoia: a handle of re-allocated memory has been seen to resolve to the memory that was there before.
int team_map_peers(mi355cg_team_s* t) {
    TeamPart& p = t->parts[0];
    const int W = t->world;
    const MboxLayout ml{W};
    BootRec mine{};
    char why[160] = "";
    mine.ok = env_int("MI355CG_TEAM_IPC", 1) != 0 ? 1 : 0;
    if (!mine.ok) std::snprintf(why, sizeof why, "MI355CG_TEAM_IPC=0");
    // Mapping another rank's 3.2 GB residual vector (N = 32768 cut in two) never came back from the runtime (minutes of system time;
    // 1.6 GB -- the same grid cut in four -- maps at once).  Parts that large spend milliseconds per launch: RCCL's tens of
    // microseconds for the records and the halo are below 1 % there, so they simply do not map.  (>=: 2^31 bytes is where a signed
    // 32-bit size would turn over.)
    const double r_gib = (double)sizeof(double) * (double)p.c->storage_len / (double)(1ull << 30);
    if (mine.ok && r_gib >= env_int("MI355CG_TEAM_IPC_MAX_GIB", 2)) { mine.ok = 0; std::snprintf(why, sizeof why, "a residual vector of %.1f GiB is not mapped (limit %d GiB)", r_gib, env_int("MI355CG_TEAM_IPC_MAX_GIB", 2)); }
    if (mine.ok) {
        hipError_t e = hipIpcGetMemHandle(&mine.slab, p.slab);
        if (e == hipSuccess) e = p.r_pooled ? hipIpcGetMemHandle(&mine.r, p.c->r) : hipErrorInvalidValue;
        if (e != hipSuccess) { mine.ok = 0; std::snprintf(why, sizeof why, "hipIpcGetMemHandle: %s", hipGetErrorString(e)); (void)hipGetLastError(); }
    }
    // a word only this rank could have written (in a cell nothing reads before the first solve writes it)
    const unsigned long long nonce = ((unsigned long long)getpid() << 32) ^ (unsigned long long)std::chrono::steady_clock::now().time_since_epoch().count() ^ ((unsigned long long)p.rank << 56);
    mine.nonce = nonce;
    const double nonce_d = __builtin_bit_cast(double, (nonce & 0x000fffffffffffffull) | 0x3ff0000000000000ull);      // the same bits as a finite double, for the vector
    HIPCK(hipMemcpy(p.mbox + ml.halo(p.rank), &nonce, sizeof nonce, hipMemcpyHostToDevice));
    HIPCK(hipMemcpy(p.c->r, &nonce_d, sizeof nonce_d, hipMemcpyHostToDevice));          // (first cell of the lower ghost / boundary row: zeroed again below)
    if (hipDeviceGetPCIBusId(mine.bus, sizeof mine.bus, p.c->device) != hipSuccess) { std::snprintf(mine.bus, sizeof mine.bus, "rank%d", p.rank); (void)hipGetLastError(); }
    std::vector<BootRec> all(W);
    if (int rc = boot_all_gather(t, &mine, all.data(), sizeof(BootRec))) return rc;
    t->shared_device = false;
    bool every = true;
    for (int i = 0; i < W; ++i) { if (!all[i].ok) every = false; for (int j = 0; j < i; ++j) if (std::strncmp(all[i].bus, all[j].bus, sizeof mine.bus) == 0) t->shared_device = true; }
    int opened = every ? 1 : 0;
    if (every) for (int j = 0; j < W && opened; ++j) {
        if (j == p.rank) continue;
        void *ps = nullptr, *pr = nullptr;
        hipError_t e = hipIpcOpenMemHandle(&ps, all[j].slab, hipIpcMemLazyEnablePeerAccess);
        if (e == hipSuccess) { p.ipc_opened.push_back(ps); e = hipIpcOpenMemHandle(&pr, all[j].r, hipIpcMemLazyEnablePeerAccess); }
        if (e != hipSuccess) { opened = 0; std::snprintf(why, sizeof why, "hipIpcOpenMemHandle(rank %d): %s", j, hipGetErrorString(e)); (void)hipGetLastError(); continue; }
        p.ipc_opened.push_back(pr);
        p.peer_mbox[j] = (u64*)ps; p.peer_cols[j] = (double*)((char*)ps + mbox_bytes(W)); p.peer_r[j] = (double*)pr;
        unsigned long long seen = 0; double seen_d = 0;                // do the mappings show what rank j wrote?
        const double want_d = __builtin_bit_cast(double, (all[j].nonce & 0x000fffffffffffffull) | 0x3ff0000000000000ull);
        if (hipMemcpy(&seen, p.peer_mbox[j] + ml.halo(j), sizeof seen, hipMemcpyDeviceToHost) != hipSuccess || seen != all[j].nonce ||
            hipMemcpy(&seen_d, p.peer_r[j], sizeof seen_d, hipMemcpyDeviceToHost) != hipSuccess || seen_d != want_d) {
            opened = 0; std::snprintf(why, sizeof why, "the mapping of rank %d's memory does not show what rank %d wrote (stale IPC mapping)", j, j); (void)hipGetLastError();
        }
    }
    std::vector<int> oks(W, 0);
    if (int rc = boot_all_gather(t, &opened, oks.data(), sizeof(int))) return rc;            // (also: nobody clears its words before everybody has looked)
    t->ipc_ok = true;
    for (int v : oks) if (!v) t->ipc_ok = false;
    const unsigned long long zero = 0;
    HIPCK(hipMemcpy(p.mbox + ml.halo(p.rank), &zero, sizeof zero, hipMemcpyHostToDevice));
    HIPCK(hipMemcpy(p.c->r, &zero, sizeof zero, hipMemcpyHostToDevice));
    if (!t->ipc_ok) {
        for (void* q : p.ipc_opened) hipIpcCloseMemHandle(q);
        p.ipc_opened.clear();
        for (int j = 0; j < W; ++j) if (j != p.rank) { p.peer_mbox[j] = nullptr; p.peer_cols[j] = nullptr; p.peer_r[j] = nullptr; }
        (void)hipGetLastError();
        t->ipc_note = why[0] ? why : "another rank could not map its peers";
        for (auto& ch : t->ipc_note) if (ch == ' ') ch = '_';
    }
    return MI355CG_OK;
}
}  // namespace

int mi355cg_team_create_rccl(int n, int m, double a, double b, double c_, double d, int world, int rank, int device,
                             const void* id128, int decomp, mi355cg_team* out) {
    if (!out) return fail(MI355CG_ERR_INVALID, "out is null");
    *out = nullptr;
    const bool dbg = env_int("MI355CG_TEAM_DEBUG", 0) != 0;                 // stage times of the creation on stderr
    const auto dbg_t0 = std::chrono::steady_clock::now();
    auto stage = [&](const char* what) { if (dbg) std::fprintf(stderr, "[mi355cg team rank %d +%.2f s] %s\n", rank, std::chrono::duration<double>(std::chrono::steady_clock::now() - dbg_t0).count(), what); };
    if (rank < 0 || rank >= world || !id128) return fail(MI355CG_ERR_INVALID, "bad rank %d of %d / null id", rank, world);
    if (world > kMaxRecDst) return fail(MI355CG_ERR_INVALID, "a team has at most %d parts", kMaxRecDst);
    RcclApi* api = rccl_api();
    if (!api) return fail(MI355CG_ERR_HIP, "librccl could not be loaded");
    mi355cg_team_s* t = nullptr;
    if (int rc = team_create_common(n, m, a, b, c_, d, world, decomp, &t)) return rc;
    t->rccl = true;
    const Box& bx = t->boxes[rank];
    TeamPart p; p.rank = rank;
    stage("decomposition done; creating the part");
    int rc = create_impl(n, m, a, b, c_, d, MI355CG_F64, device, bx.y_lo, bx.y_hi, bx.s_lo, bx.s_hi, world > 1, &p.c);
    if (rc) { team_free(t); return rc; }
    stage("part created (vectors, right-hand side)");
    if (world > 1) {
        // the residual vector is what the neighbours' push launches write into: it has to come from the pool (see IpcPool)
        void* pooled = nullptr;
        if ((rc = ipc_pool().acquire(device, sizeof(double) * p.c->storage_len, false, &pooled))) { mi355cg_destroy(p.c); team_free(t); return rc; }
        hipFree(p.c->r);
        p.c->r = (double*)pooled; p.r_pooled = true;
    }
    t->parts.push_back(p);
    stage("residual vector pooled");
    if ((rc = team_finish_setup(t))) { team_free(t); return rc; }
    stage("halo lists, mailbox, streams");
    ncclUniqueId id;
    std::memcpy(&id, id128, sizeof id);
    if (hipSetDevice(device) != hipSuccess) { team_free(t); return fail(MI355CG_ERR_HIP, "hipSetDevice(%d) failed", device); }
    const ncclResult_t nr = api->CommInitRank(&t->comm, world, id, rank);
    if (nr != ncclSuccess) { t->comm = nullptr; team_free(t); return fail(MI355CG_ERR_HIP, "ncclCommInitRank failed: %s", api->GetErrorString(nr)); }
    if (api->CommCount(t->comm, &t->rccl_nranks) != ncclSuccess) t->rccl_nranks = -1;
    stage("communicator up");
    if (world > 1) {
        if ((rc = team_map_peers(t))) { team_free(t); return rc; }
        stage("peers mapped");
        // A second communicator only where the halo is asked to travel on RCCL beside the records (MI355CG_TEAM_HALO=stream):
        // its id comes from rank 0 through the first one
        if (env_choice("MI355CG_TEAM_HALO", {"auto", "inline", "stream", "push"}, 0) == 2) {
            ncclUniqueId id2{};
            if (rank == 0) { const ncclResult_t r0 = api->GetUniqueId(&id2); if (r0 != ncclSuccess) { team_free(t); return fail(MI355CG_ERR_HIP, "ncclGetUniqueId failed: %s", api->GetErrorString(r0)); } }
            std::vector<ncclUniqueId> ids(world);
            if ((rc = boot_all_gather(t, &id2, ids.data(), sizeof id2))) { team_free(t); return rc; }
            const ncclResult_t n2 = api->CommInitRank(&t->comm_halo, world, ids[0], rank);
            if (n2 != ncclSuccess) { t->comm_halo = nullptr; team_free(t); return fail(MI355CG_ERR_HIP, "ncclCommInitRank (halo communicator) failed: %s", api->GetErrorString(n2)); }
        }
    }
    if ((rc = team_build_tables(t))) { team_free(t); return rc; }
    stage("tables built");
    *out = t;
    return MI355CG_OK;
}

// "transport=rccl records=mailbox wait=kernel halo=push ipc=1 shared_device=0 rccl_nranks=8 rccl_lib=librccl.so.1": what the next
// solve of this team will use (environment + what the ranks found out about each other at creation)
int mi355cg_team_describe(mi355cg_team t, char* buf, int len) {
    if (!t || !buf || len <= 0) return fail(MI355CG_ERR_INVALID, "null argument");
    if (int rc = team_pick_modes(t)) return rc;
    std::snprintf(buf, (size_t)len, "transport=%s records=%s wait=%s halo=%s split=%d ipc=%d shared_device=%d rccl_nranks=%d rccl_lib=%s ipc_note=%s",
                  t->rccl ? "rccl" : "local", rec_name(t->rec_mode), wait_name(t->wait_mode), halo_name(t->halo_mode), t->split_phases,
                  t->ipc_ok ? 1 : 0, t->shared_device ? 1 : 0, t->rccl_nranks, t->rccl && rccl_api() ? rccl_api()->lib_name.c_str() : "-", t->ipc_note.c_str());
    return MI355CG_OK;
}

void mi355cg_team_destroy(mi355cg_team t) { team_free(t); }

int mi355cg_team_solve(mi355cg_team t, const mi355cg_params* prm, mi355cg_iter_cb cb, void* user,
                       const volatile int* stop_flag, mi355cg_results* out) {
    if (!t || !prm) return fail(MI355CG_ERR_INVALID, "null argument");
    if (t->dtype == MI355CG_F32_MIXED) return team_solve_mixed(t, prm, cb, user, stop_flag, out);
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

// which: 2 right-hand side, 3 exact solution.  The caller's GLOBAL packed vector (length mi355cg_size) is cut up: every part this
// process drives takes the entries it owns (the Solver(a, b, ...) constructor takes any b, solver/solver.hpp:33-39; MSGSolver::solve
// any true_solution, msg_solver.cpp:64-72).  On an RCCL team every rank calls it with the same vector.
int mi355cg_team_set_vector(mi355cg_team t, int which, const double* global_packed) {
    if (!t || !global_packed) return fail(MI355CG_ERR_INVALID, "null argument");
    if (which != 2 && which != 3) return fail(MI355CG_ERR_INVALID, "vector %d cannot be set (2 right-hand side, 3 exact solution)", which);
    for (auto& p : t->parts) {
        mi355cg_ctx* c = p.c;
        HIPCK(hipSetDevice(c->device));
        std::vector<double> own(std::max<long long>(c->pk_len, 1));
        const PackGeom& pg = c->pg;
        long long i = 0;
        for (int k = 0; k < pg.nb_rows; ++k, i += pg.wb) std::memcpy(own.data() + i, global_packed + packed_index(t->gp, pg.xb0, pg.yb0 + k), sizeof(double) * pg.wb);
        for (int k = 0; k < pg.nu_rows; ++k, i += pg.wu) std::memcpy(own.data() + i, global_packed + packed_index(t->gp, pg.xu0, pg.yu0 + k), sizeof(double) * pg.wu);
        if (int rc = which == 2 ? mi355cg_set_rhs(c, own.data()) : mi355cg_set_true_solution(c, own.data())) return rc;
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
