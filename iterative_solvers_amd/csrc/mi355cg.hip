// mi355cg.hip -- C ABI (include/mi355cg.h) over the HIP kernels in cg_kernels.h.
// One context = one GPU = one part of the grid (the whole grid on a single GPU; a row slab or a 2-D block of a
// decomposed grid).  Teams of parts (native multi-GPU loop, RCCL / in-process transports) live in team.h.
// No CPU fallback: every compute entry point needs a working HIP device.
#include "../../include/mi355cg.h"
#include "cg_kernels.h"
#include "csr_kernels.h"
#include "grid_setup.h"

#include <algorithm>
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

using namespace mi355cg;

namespace {

thread_local std::string g_err;

int fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
    g_err = buf;
    return code;
}

#define HIPCK(expr)                                                                                  \
    do {                                                                                             \
        hipError_t e_ = (expr);                                                                      \
        if (e_ != hipSuccess)                                                                        \
            return fail(MI355CG_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

int env_int(const char* name, int dflt) {
    const char* v = getenv(name);
    return (v && *v) ? atoi(v) : dflt;
}

inline long long round_up(long long v, long long m) { return (v + m - 1) / m * m; }

constexpr int kRecHeader = kRecWords;     // doubles reserved for the sums (hi/lo pairs) and maxes at the head of a part's record
constexpr int kStripCols = 128;    // fp64 strip = 64 lanes x double2: the unit of the x-cuts of a 2-D decomposition

struct EventPool {
    std::vector<hipEvent_t> ev;
    size_t used = 0;
    hipEvent_t get() {
        if (used == ev.size()) { hipEvent_t e; if (hipEventCreate(&e) != hipSuccess) return nullptr; ev.push_back(e); }
        return ev[used++];
    }
    void reset() { used = 0; }
    void destroy() { for (auto e : ev) hipEventDestroy(e); ev.clear(); used = 0; }
};

// One launch shape: the work items of a set of rows x strips and the persistent grid that marches them.
struct Plan { WorkList wl{}; int grid = 0; int ty = 0; };

}  // namespace

struct mi355cg_ctx {
    int device = 0;
    int dtype = MI355CG_F64;
    hipStream_t stream = nullptr;
    GridParams gp;
    Geom g{};
    int s_lo = 0, s_hi = 0;             // owned fp64 strips [s_lo, s_hi) (kStripCols columns each); all of them unless 2-D
    long long storage_len = 0;          // elements per vector incl. ghost rows
    long long pk_begin = 0, pk_len = 0; // owned cells in the part's packed order (pk_begin: global packed index of the first; row slabs are contiguous)
    PackGeom pg{};
    Plan whole, interior, edge;         // whole part; rows / strips that need no ghost data; the rest (first + last row, edge strips)
    Plan whole32;                       // fp32 kernels (VEC = 4, 256-column strips): the whole grid, or a row slab of full width
    int depth = 2;                      // raw rows in flight per wave (env MI355CG_DEPTH: 2 or 3)
    bool has_gc = false;                // 2-D part with a neighbour in x: the stencil launches keep the ghost columns of the direction
    int nB_own = 0;                     // partB slots written by the last update-phase launch(es) of this context
    int strideA = 0, strideB = 0;

    // device vectors in storage layout (fp64 set always; fp32 set for F32_MIXED)
    double *x = nullptr, *r = nullptr, *p[kRing] = {}, *ap = nullptr, *b = nullptr, *u = nullptr;
    int xsteps = 4;                     // M: the direction ring has M buffers, x is touched every M-th iteration (env MI355CG_XSTEPS: 2 | 4 | 8)
    double* scratch[2] = {nullptr, nullptr};      // mi355cg_apply / true residual work space, allocated on first use: never a solver vector
    float *xf = nullptr, *rf = nullptr, *pf[kRing] = {}, *apf = nullptr;
    double* packed = nullptr;           // device scratch, pk_len doubles
    double *partA = nullptr, *partB = nullptr, *partR = nullptr;
    double *sumsA = nullptr, *sumsB = nullptr;   // slab mode: this rank's record = reduced partials [+ its two boundary rows] (feeds the all-gather)
    int rec_width = 0;
    mi355cg_params dist_prm{};                    // slab mode: parameters given to mi355cg_dist_begin
    bool dist_active = false, is_slab = false;
    CgState *sA = nullptr, *sB = nullptr, *summary = nullptr;
    int* qctr = nullptr;                // dynamic item queues: kXcds * kQueueSubs counters of the stencil launches, then as many of the update launches (QueueSpec)
    int dyn_rows = 0;                   // > 0: the whole-part launches cut their items this short and deal them through the queues
    int* stop_h = nullptr;              // pinned host word: a stop request, sampled by block 0 of every update launch (msg_solver.cpp:82-87)
    int* stop_dev = nullptr;            // the same word as the device addresses it (nullptr while no solve with a stop flag is running)
    HistEntry* hist = nullptr;
    CgState* summary_h = nullptr;       // pinned
    HistEntry* hist_h = nullptr;        // pinned
    double* partR_h = nullptr;          // pinned

    std::vector<double> rhs_h, u_h;     // host copies (owned cells, the part's packed order)
    bool host_rhs_valid = true, host_u_valid = true;     // false after mi355cg_setup_on_device until somebody asks for the host copy
    bool have_u_dev = false, solved = false;
    // generic CSR handle (mi355cg_create_csr): vectors are plain length-n arrays, the operator is this matrix
    bool is_csr = false;
    long long csr_n = 0, csr_nnz = 0;
    int *csr_row_map = nullptr, *csr_entries = nullptr; double* csr_values = nullptr;
    int grid_csr = 0, grid_update = 0;
    int cur = 0;                        // p[cur] holds the current direction after the last stencil: cur = (iterations done) % xsteps
    int nA_dist = 0;                    // slab mode: stencil partial slots written by the last stencil phase

    // hipGraph cache for launch-bound (small) grids: one instantiated graph per distinct chunk shape of a solve
    struct ChunkGraph { int m, cur; std::vector<char> flags; hipGraphExec_t exec; };
    std::vector<ChunkGraph> graphs;
    mi355cg_params graph_prm{};          // parameters the cached graphs were captured with
    bool graph_stop = false;            // ... and whether those solves sampled the stop word
    int use_graph = -1;                 // env MI355CG_GRAPH: -1 auto (small grids), 0 off, 1 on

    hipEvent_t ev_loop[2] = {nullptr, nullptr};   // brackets the iterations of the last solve (mi355cg_results::loop_seconds)
    bool profiling = false;
    EventPool events;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_pairs[2];
    double kernel_ms[2] = {0, 0};
    long long kernel_launches[2] = {0, 0};
};

namespace {

// ---- layout ----------------------------------------------------------------------------------------
long long phys_start(const Geom& g, int y) { return row_off(g, y) + (y <= g.half ? g.cb : 0); }
long long phys_end(const Geom& g, int y) { return phys_start(g, y) + (y <= g.half ? g.Pb : g.Pu); }
int strips_total(const GridParams& gp, int vec) { return (gp.n - 1) / (kWave * vec) + 1; }
int first_bottom_strip(const GridParams& gp, int vec) { return (gp.half + 1) / (kWave * vec); }

// first / one-past-last interior column of row y inside the part's strips
void own_cols(const mi355cg_ctx* c, int y, int* xa, int* xb) {
    const int lo = y <= c->gp.half ? c->gp.half + 1 : 1, hi = c->gp.n;       // interior columns [lo, hi)
    *xa = std::max(lo, c->s_lo * kStripCols);
    *xb = std::max(*xa, std::min(hi, c->s_hi * kStripCols));
}

void build_geom(mi355cg_ctx* c, int vec, int y_lo, int y_hi) {
    Geom& g = c->g;
    const GridParams& gp = c->gp;
    g.N = gp.n; g.half = gp.half;
    g.Pu = (int)round_up(g.N + 1, 32);
    g.cb = g.half & ~31;
    g.Pb = g.Pu - g.cb;
    g.xlim = (int)round_up(g.N + 1, vec);
    g.y_lo = y_lo; g.y_hi = y_hi;
    g.base0 = 0;
    g.base0 = phys_start(g, y_lo - 1);
    c->storage_len = phys_end(g, y_hi + 1) - g.base0;
    g.own_begin = phys_start(g, y_lo) - g.base0;
    g.own_len = phys_end(g, y_hi) - phys_start(g, y_lo);
    g.A = gp.A; g.xk = gp.x_k; g.yk = gp.y_k;
    // the part's packed order: its bottom-block rows (each restricted to the own columns), then its upper rows
    PackGeom& pg = c->pg;
    pg.g = g;
    const int yb0 = y_lo, yb1 = std::min(y_hi, gp.half), yu0 = std::max(y_lo, gp.half + 1), yu1 = y_hi;
    int xa = 0, xb = 0;
    pg.nb_rows = std::max(0, yb1 - yb0 + 1); pg.yb0 = yb0;
    own_cols(c, gp.half, &xa, &xb); pg.xb0 = xa; pg.wb = pg.nb_rows ? xb - xa : 0;
    pg.nu_rows = std::max(0, yu1 - yu0 + 1); pg.yu0 = yu0;
    own_cols(c, gp.half + 1, &xa, &xb); pg.xu0 = xa; pg.wu = pg.nu_rows ? xb - xa : 0;
    pg.pk_len = (long long)pg.nb_rows * pg.wb + (long long)pg.nu_rows * pg.wu;
    c->pk_len = pg.pk_len;
    c->pk_begin = packed_row_begin(gp, y_lo);        // meaningful for row slabs (contiguous global range)
}

// Rows [ya, yb] x strips [sa, sb) as up to two rectangles (bottom-right block rows, upper block rows) with their
// ghost-column flags (bit 0: a part to the left, bit 1: a part to the right).
struct Rect { int y0, y1, s0, s1, gc; };
int region_rects(const GridParams& gp, int vec, int ya, int yb, int sa, int sb, Rect out[2]) {
    const int ns_all = strips_total(gp, vec), s0b = first_bottom_strip(gp, vec);
    int nr = 0;
    sb = std::min(sb, ns_all);
    if (ya <= gp.half && yb >= 1) {
        const int s0 = std::max(sa, s0b);
        if (s0 < sb) out[nr++] = Rect{std::max(ya, 1), std::min(yb, gp.half), s0, sb, (sa > s0b ? 1 : 0) | (sb < ns_all ? 2 : 0)};
    }
    if (yb > gp.half && sa < sb)
        out[nr++] = Rect{std::max(ya, gp.half + 1), std::min(yb, gp.n - 1), sa, sb, (sa > 0 ? 1 : 0) | (sb < ns_all ? 2 : 0)};
    return nr;
}

// Append rows [y0, y1] x strips [s0, s1) cut into items of ~ty rows.  Returns the strip-rows added (ty <= 0: only count).
long long add_panel(WorkList& wl, int y0, int y1, int s0, int s1, int ty, int gc) {
    const int rows = y1 - y0 + 1, ns = s1 - s0;
    if (rows <= 0 || ns <= 0) return 0;
    if (ty <= 0) return (long long)rows * ns;
    if (wl.np >= kMaxPanels) return 0;
    Panel& P = wl.p[wl.np++];
    P.y0 = y0; P.y1 = y1; P.s0 = s0; P.ns = ns; P.gc = gc;
    P.nchunks = (rows + ty - 1) / ty;
    P.ty = (rows + P.nchunks - 1) / P.nchunks;            // rebalance
    P.nchunks = (rows + P.ty - 1) / P.ty;
    P.item0 = wl.nitems;
    wl.nitems += P.ns * P.nchunks;
    return (long long)rows * ns;
}

// Launch shape for a list of rectangles.  2 048 resident waves (2 workgroups per CU: 8 waves per CU already saturate the
// memory system, round 1) take the items round-robin.  Measured (tools/tune.py, profiles/r02_tune_notes.md): one round of
// items "as tall as it takes" is the best fp64 shape up to ~800 rows per item (N <= 16384 on one GPU); taller items (3 136 rows
// of a 262 KB pitch at N = 32768: every wave sweeps 0.8 GB per stream) lose 14-16 %, so the height is capped and the rest
// becomes further rounds -- which cost nothing since the load pipeline no longer drains between items.  The fp32 kernels
// (256-column strips) like 64 rows.  The height is then nudged so that the items fill a whole number of rounds: a last
// round with a few items would run at a fraction of the chip.
Plan make_plan(const std::vector<Rect>& rects, int max_rows, int fixed_ty = 0, int dyn_rows = 0) {
    Plan pl{};
    const int target_waves = std::max(kWaves, env_int("MI355CG_WAVES", 2048));
    const int max_blocks = std::max(1, env_int("MI355CG_BLOCKS", 512));
    const int waves = std::min(target_waves, max_blocks * kWaves);
    const long long item_rows = std::max(1, env_int("MI355CG_ITEM_ROWS", max_rows));
    long long strip_rows = 0;
    WorkList dry{};
    for (auto& r : rects) strip_rows += add_panel(dry, r.y0, r.y1, r.s0, r.s1, 0, 0);
    if (strip_rows == 0) return pl;
    // XCD classes (see WorkList): only for launches that fill the chip, never for the single-row edge launches
    const bool classes = env_int("MI355CG_XCD_CLASSES", 1) != 0 && fixed_ty == 0 && max_blocks >= kXcds && strip_rows >= 4LL * waves;
    auto build = [&](int ty) {
        WorkList wl{};
        wl.ncls = 1;
        for (auto& r : rects) add_panel(wl, r.y0, r.y1, r.s0, r.s1, ty, r.gc);
        if (classes) {
            wl.ncls = kXcds;
            for (int k = 0; k <= kXcds; ++k) wl.cls0[k] = (int)((long long)wl.nitems * k / kXcds);
        }
        return wl;
    };
    auto fits = [&](const WorkList& wl, long long rounds) {
        if (wl.ncls == kXcds) { for (int k = 0; k < kXcds; ++k) if (wl.cls0[k + 1] - wl.cls0[k] > rounds * (waves / kXcds)) return false; return true; }
        return wl.nitems <= rounds * waves;
    };
    int ty = fixed_ty;
    if (ty <= 0 && dyn_rows > 0 && classes) {
        // dynamic queues: short items, several per wave; no need to fill whole rounds -- whoever is early takes more
        ty = dyn_rows;
        pl.wl = build(ty);
    } else if (ty <= 0) {
        const long long rounds = std::max<long long>(1, (strip_rows + waves * item_rows - 1) / (waves * item_rows));
        ty = (int)std::max<long long>(std::min<long long>(8, item_rows), (strip_rows + rounds * waves - 1) / (rounds * waves));
        for (int tries = 0; tries < 64; ++tries) {
            pl.wl = build(ty);
            if (fits(pl.wl, rounds)) break;
            ++ty;
        }
    } else {
        pl.wl = build(ty);
    }
    pl.ty = ty;
    pl.grid = std::max(1, std::min(max_blocks, (pl.wl.nitems + kWaves - 1) / kWaves));
    if (pl.wl.ncls == kXcds) pl.grid = std::min(max_blocks / kXcds * kXcds, (pl.grid + kXcds - 1) / kXcds * kXcds);      // the classes take turns over the workgroups
    return pl;
}
constexpr int kMaxRowsF64 = 800, kMaxRowsF32 = 64;

void build_plans(mi355cg_ctx* c) {
    const Geom& g = c->g;
    const GridParams& gp = c->gp;
    Rect rr[2];
    const int nr = region_rects(gp, 2, g.y_lo, g.y_hi, c->s_lo, c->s_hi, rr);
    std::vector<Rect> whole(rr, rr + nr);
    // Static deal (one round of tall items) or run-time item queues (short items, QueueSpec)?  Measured (profiles/r03_tune_notes.md
    // section 5): the queues lose 1.5 % at N = 4096 (49-row items: the balancing gain and the cost of short items cancel) and win
    // +1-3 % at N = 8192, +3-5 % at N = 16384, +5-8 % at N = 32768, where the static items are hundreds of rows tall and the launch
    // tail is whole items.  Default: queues of 16-row items when the static items would be at least 96 rows tall.  MI355CG_DYN_ROWS = 0 | R overrides.
    c->whole = make_plan(whole, kMaxRowsF64);
    c->dyn_rows = std::max(0, env_int("MI355CG_DYN_ROWS", c->whole.ty >= 96 ? 16 : 0));
    if (c->dyn_rows > 0) {
        const Plan dyn = make_plan(whole, kMaxRowsF64, 0, c->dyn_rows);
        if (dyn.wl.ncls == kXcds && dyn.wl.nitems >= 2 * kWaves * dyn.grid) c->whole = dyn; else c->dyn_rows = 0;      // small launches keep the static deal
    }
    c->has_gc = false;
    for (auto& r : whole) if (r.gc) c->has_gc = true;
    // split for halo / compute overlap: `edge` = everything that reads ghost data of r (first and last owned row, the
    // strips beside a ghost column), `interior` = the rest
    std::vector<Rect> inner, rows, cols;
    for (auto& r : whole) {
        const int yi0 = std::max(r.y0, g.y_lo + 1), yi1 = std::min(r.y1, g.y_hi - 1);
        const int si0 = r.s0 + ((r.gc & 1) ? 1 : 0), si1 = r.s1 - ((r.gc & 2) ? 1 : 0);
        if (r.y0 <= g.y_lo && g.y_lo <= r.y1) rows.push_back(Rect{g.y_lo, g.y_lo, r.s0, r.s1, r.gc});
        if (g.y_hi > g.y_lo && r.y0 <= g.y_hi && g.y_hi <= r.y1) rows.push_back(Rect{g.y_hi, g.y_hi, r.s0, r.s1, r.gc});
        if (yi0 > yi1) continue;
        if (si0 < si1) inner.push_back(Rect{yi0, yi1, si0, si1, 0});
        if (si0 >= si1) { cols.push_back(Rect{yi0, yi1, r.s0, r.s1, r.gc}); continue; }     // too narrow to have an interior
        if (r.gc & 1) cols.push_back(Rect{yi0, yi1, r.s0, r.s0 + 1, 1});
        if (r.gc & 2) cols.push_back(Rect{yi0, yi1, r.s1 - 1, r.s1, 2});
    }
    c->interior = make_plan(inner, kMaxRowsF64);
    // edge rows are single-row items; edge strips are cut like the interior
    Plan e_rows = make_plan(rows, kMaxRowsF64, 1), e_cols = make_plan(cols, kMaxRowsF64, c->interior.ty > 0 ? c->interior.ty : 0);
    c->edge = e_rows;
    c->edge.wl.ncls = 1;
    for (int k = 0; k < e_cols.wl.np && c->edge.wl.np < kMaxPanels; ++k) {
        Panel P = e_cols.wl.p[k];
        P.item0 = c->edge.wl.nitems;
        c->edge.wl.p[c->edge.wl.np++] = P;
        c->edge.wl.nitems += P.ns * P.nchunks;
    }
    c->edge.grid = std::max(1, std::min(std::max(1, env_int("MI355CG_BLOCKS", 512)), (c->edge.wl.nitems + kWaves - 1) / kWaves));
    if (c->edge.wl.nitems == 0) c->edge.grid = 0;
    // the fp32 kernels use 256-column strips (float4 per lane) on the same pitches; a 2-D part is cut on 128 columns and has none
    if (c->dtype == MI355CG_F32_MIXED || (c->s_lo == 0 && c->s_hi == strips_total(gp, 2))) {
        Rect r4[2];
        const int n4 = region_rects(gp, 4, g.y_lo, g.y_hi, 0, strips_total(gp, 4), r4);
        c->whole32 = make_plan(std::vector<Rect>(r4, r4 + n4), kMaxRowsF32);
    }
    // rows in flight per wave: 3 measured +1.5 % at N = 4096 and neutral elsewhere (profiles/r02_tune_notes.md section 9); the
    // 12-word update of MI355CG_XSTEPS=8 would need 290 VGPRs at 3 (one wave per SIMD instead of two) and stays at 2
    c->depth = env_int("MI355CG_DEPTH", 3) == 2 ? 2 : 3;
    c->use_graph = env_int("MI355CG_GRAPH", -1);
    c->nB_own = c->whole.grid;
}

int flat_grid(long long n) { return (int)std::max<long long>(1, std::min<long long>(2048, (n + kBlock - 1) / kBlock)); }

// ---- launchers -------------------------------------------------------------------------------------
// Where a consumer kernel finds the partials it reduces in its prologue: the producer kernel's own
// field-major array (estride 1) or partials all-gathered across ranks, rank-major (estride = #fields).
struct PartSrc { const double* ptr; int n, fstride, estride; RecSrc rec; };     // rec.mbox != nullptr (teams): the parts' records instead of partials
// Which items a launch covers, where it runs, and the first partial slot it writes.
struct Where { hipStream_t stream; const Plan* plan; int slot; };

Where whole_part(const mi355cg_ctx* c) { return Where{c->stream, &c->whole, 0}; }
PartSrc own_partB(const mi355cg_ctx* c) { return PartSrc{c->partB, c->nB_own, c->strideB, 1}; }
PartSrc own_partA(const mi355cg_ctx* c) { return PartSrc{c->partA, c->whole.grid, c->strideA, 1}; }

template <typename T, int VEC> Geom kernel_geom(const mi355cg_ctx* c) { Geom g = c->g; g.xlim = (int)round_up(c->g.N + 1, VEC); return g; }

// y = A_h v (plain operator apply on storage-layout vectors)
template <typename T, int VEC>
void launch_apply(const mi355cg_ctx* c, const T* v, T* out, const Where& w) {
    if (w.plan->wl.nitems == 0) return;
    StencilArgs<T> a{};
    a.g = kernel_geom<T, VEC>(c); a.wl = w.plan->wl;
    a.pin = v; a.ap = out; a.partA = nullptr;
    hipLaunchKernelGGL((k_stencil<T, VEC, false, false, 2, false, false>), dim3(w.plan->grid), dim3(kBlock), 0, w.stream, a);
}

struct IterCfg { RuleParams rp; int want_diag; bool has_u; bool x2 = false; };

// Phase A'.  Does NOT flip c->cur (a part's interior and edge launches share one direction pair).
template <typename T, int VEC>
void launch_iteration_stencil(mi355cg_ctx* c, const IterCfg& cfg, const T* r, T* const p[kRing], const Where& w, const PartSrc& pb, const FlagSpec* fl = nullptr) {
    if (w.plan->wl.nitems == 0) return;
    StencilArgs<T> a{};
    a.g = kernel_geom<T, VEC>(c); a.wl = w.plan->wl;
    a.r = r; a.pin = p[c->cur]; a.pout = p[(c->cur + 1) % c->xsteps]; a.ap = nullptr;
    a.partB = pb.ptr; a.nB = pb.n; a.strideB = pb.fstride; a.esB = pb.estride; a.src = pb.rec;
    a.partA = c->partA; a.strideA = c->strideA; a.slotA = w.slot;
    a.s_in = c->sB; a.s_out = c->sA; a.hist = c->hist; a.rp = cfg.rp; a.want_diag = cfg.want_diag;
    a.store_ghosts = c->is_slab ? 1 : 0;
    if (fl) a.fl = *fl;
    if (VEC == 2 && c->dyn_rows > 0 && w.plan == &c->whole) a.dq = QueueSpec{c->qctr, c->qctr + kXcds * kQueueSubs * kQueuePitch};
    const dim3 grid(w.plan->grid), block(kBlock);
    const bool msg = cfg.rp.rule == MI355CG_RULE_MSG_MAXNORM, gc = c->has_gc, d3 = c->depth == 3;
#define MI355CG_ST(MSG, D, GC) hipLaunchKernelGGL((k_stencil<T, VEC, true, MSG, D, true, GC>), grid, block, 0, w.stream, a)
    if constexpr (VEC == 2) {
        if (msg) { if (gc) { if (d3) MI355CG_ST(true, 3, true); else MI355CG_ST(true, 2, true); } else { if (d3) MI355CG_ST(true, 3, false); else MI355CG_ST(true, 2, false); } }
        else     { if (gc) { if (d3) MI355CG_ST(false, 3, true); else MI355CG_ST(false, 2, true); } else { if (d3) MI355CG_ST(false, 3, false); else MI355CG_ST(false, 2, false); } }
    } else {
        if (d3) MI355CG_ST(false, 3, false); else MI355CG_ST(false, 2, false);       // fp32 inner CG: REL_2NORM, single GPU
    }
#undef MI355CG_ST
}

// Phase B on the stencil's work items, marched the other way (it starts on what the stencil launch touched last).
// c->cur was flipped after this iteration's stencil launch: it is the iteration number's parity.
template <typename T, int VEC>
void launch_iteration_update(mi355cg_ctx* c, const IterCfg& cfg, T* x, T* r, T* const p[kRing], const T* u, const Where& w, const PartSrc& pa, const FlagSpec* fl = nullptr) {
    if (w.plan->wl.nitems == 0) return;
    UpdateStArgs<T> a{};
    a.g = kernel_geom<T, VEC>(c); a.wl = w.plan->wl;
    a.p = p[c->cur]; a.r = r; a.x = x; a.u = u;
    for (int i = 0; i < kRing - 1; ++i) a.pprev[i] = p[(c->cur + 2 * kRing * c->xsteps - 1 - i) % c->xsteps];      // directions of iterations k-1, k-2, ...
    a.partA = pa.ptr; a.nA = pa.n; a.strideA = pa.fstride; a.esA = pa.estride; a.src = pa.rec;
    a.partB = c->partB; a.strideB = c->strideB; a.slotB = w.slot;
    a.s_in = c->sA; a.s_out = c->sB; a.rule = cfg.rp.rule; a.reverse = 1;
    if (fl) a.fl = *fl;
    if (VEC == 2 && c->dyn_rows > 0 && w.plan == &c->whole) a.dq = QueueSpec{c->qctr + kXcds * kQueueSubs * kQueuePitch, c->qctr};
    a.stop_req = w.slot == 0 ? c->stop_dev : nullptr;   // block 0 samples the pinned stop word once per iteration (of a phase in two launches: the one that runs last and owns slot 0)
    const dim3 grid(w.plan->grid), block(kBlock);
    const bool d3 = c->depth == 3 && !(cfg.x2 && c->cur == 0 && c->xsteps == 8);
#define MI355CG_UST(XM, HASU) do { if (d3) hipLaunchKernelGGL((k_update_st<T, VEC, XM, HASU, 3, true>), grid, block, 0, w.stream, a); \
                                   else hipLaunchKernelGGL((k_update_st<T, VEC, XM, HASU, 2, true>), grid, block, 0, w.stream, a); } while (0)
    if (cfg.x2) {                    // iterations k = 0 mod M carry all M x steps (c->cur = k % M)
        if (c->cur != 0) MI355CG_UST(0, false); else if (c->xsteps == 8) MI355CG_UST(8, false); else if (c->xsteps == 4) MI355CG_UST(4, false); else MI355CG_UST(2, false);
    }
    else if constexpr (VEC == 2) { if (cfg.has_u) MI355CG_UST(1, true); else MI355CG_UST(1, false); }
#undef MI355CG_UST
}

// Flat pass over the owned rows: state initialisation (x = 0, r = b: norms of r0) or the resume step of the mixed path.
template <typename T, int VEC>
void launch_update_flat(mi355cg_ctx* c, const IterCfg& cfg, T* x, T* r, const T* p, const T* ap, const T* u, hipStream_t stream,
                        int grid, double resume_r0norm = -1.0) {
    UpdateArgs<T> a{};
    a.begin = c->g.own_begin / VEC; a.nvec = c->g.own_len / VEC;
    a.x = x; a.r = r; a.p = p; a.ap = ap; a.u = u;
    a.partA = c->partA; a.nA = 0; a.strideA = c->strideA; a.esA = 1;
    a.partB = c->partB; a.strideB = c->strideB;
    a.s_in = c->sA; a.s_out = c->sB; a.rule = cfg.rp.rule; a.init = 1;
    if (resume_r0norm >= 0.0) { a.init = 2; a.r0norm_resume = resume_r0norm; a.s_in = c->sB; }   // measure + re-arm, see k_update
    if (cfg.has_u) hipLaunchKernelGGL((k_update<T, VEC, true>), dim3(grid), dim3(kBlock), 0, stream, a);
    else hipLaunchKernelGGL((k_update<T, VEC, false>), dim3(grid), dim3(kBlock), 0, stream, a);
}

// The x steps of the iterations after the last multiple of M (REL_2NORM folded update), oldest first.
template <typename T, int VEC>
void launch_flush_x(const mi355cg_ctx* c, const Plan& plan, T* x, T* const p[kRing], const CgState& fin, hipStream_t stream) {
    const int pending = fin.it % c->xsteps;
    if (plan.wl.nitems == 0 || pending == 0) return;
    FlushArgs<T> f{};
    f.n = pending;
    for (int j = 0; j < pending; ++j) { const int k = fin.it - pending + 1 + j; f.p[j] = p[k % c->xsteps]; f.a[j] = (T)fin.alpha_hist[k & (kRing - 1)]; }
    hipLaunchKernelGGL((k_flush_x<T, VEC>), dim3(std::max(1, std::min(1024, (plan.wl.nitems + kWaves - 1) / kWaves))), dim3(kBlock), 0, stream,
                       kernel_geom<T, VEC>(c), plan.wl, x, f);
}

void launch_check(mi355cg_ctx* c, const IterCfg& cfg, hipStream_t stream, const PartSrc& pb) {
    CheckArgs a{};
    a.partB = pb.ptr; a.nB = pb.n; a.strideB = pb.fstride; a.esB = pb.estride; a.src = pb.rec;
    a.s_in = c->sB; a.summary = c->summary; a.hist = c->hist; a.rp = cfg.rp; a.want_diag = cfg.want_diag;
    hipLaunchKernelGGL(k_check, dim3(1), dim3(kBlock), 0, stream, a);
}

// The launch shapes of an iteration: REL_2NORM without diagnostics = x folded every M-th iteration (M = 4: 7.25 words per unknown);
// MSG, and REL_2NORM with the reference's per-iteration diagnostics = x and its norms every iteration (8 words, + u when read).
IterCfg make_cfg(const mi355cg_params* prm) {
    IterCfg cfg{};
    const bool msg = prm->rule == MI355CG_RULE_MSG_MAXNORM;
    cfg.rp.rule = prm->rule; cfg.rp.max_iterations = prm->max_iterations;
    cfg.rp.eps_precision = prm->eps_precision; cfg.rp.eps_residual = prm->eps_residual;
    cfg.rp.eps_exact_error = prm->eps_exact_error; cfg.rp.eps_rel = prm->eps_rel;
    cfg.rp.fixed_iterations = prm->fixed_iterations;
    const bool diag = !msg && prm->diagnostics;
    cfg.has_u = (msg && prm->use_true_solution) || diag;
    cfg.rp.use_u = cfg.has_u ? 1 : 0;
    cfg.want_diag = diag ? 1 : 0;
    cfg.x2 = !msg && !diag;
    return cfg;
}

template <typename T>
int upload_packed(mi355cg_ctx* c, const double* host_packed, T* storage) {
    if (c->is_csr) {                                        // no layout conversion: the caller's order is the storage order
        HIPCK(hipMemcpyAsync(storage, host_packed, sizeof(double) * c->pk_len, hipMemcpyHostToDevice, c->stream));
        HIPCK(hipStreamSynchronize(c->stream));
        return MI355CG_OK;
    }
    if (c->is_slab) HIPCK(hipDeviceSynchronize());
    if (c->pk_len == 0) return MI355CG_OK;
    HIPCK(hipMemcpyAsync(c->packed, host_packed, sizeof(double) * c->pk_len, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL((k_unpack<T>), dim3(flat_grid(c->pk_len)), dim3(kBlock), 0, c->stream, c->pg, c->packed, storage);
    HIPCK(hipGetLastError());
    HIPCK(hipStreamSynchronize(c->stream));
    return MI355CG_OK;
}
template <typename T>
int download_packed(mi355cg_ctx* c, const T* storage, double* host_packed) {
    if (c->is_csr) {
        HIPCK(hipMemcpyAsync(host_packed, storage, sizeof(double) * c->pk_len, hipMemcpyDeviceToHost, c->stream));
        HIPCK(hipStreamSynchronize(c->stream));
        return MI355CG_OK;
    }
    if (c->is_slab) HIPCK(hipDeviceSynchronize());      // part phases run on the caller's / the team's streams
    if (c->pk_len == 0) return MI355CG_OK;
    hipLaunchKernelGGL((k_pack<T>), dim3(flat_grid(c->pk_len)), dim3(kBlock), 0, c->stream, c->pg, storage, c->packed);
    HIPCK(hipGetLastError());
    HIPCK(hipMemcpyAsync(host_packed, c->packed, sizeof(double) * c->pk_len, hipMemcpyDeviceToHost, c->stream));
    HIPCK(hipStreamSynchronize(c->stream));
    return MI355CG_OK;
}

int alloc_vec(double** p, long long n) {
    HIPCK(hipMalloc((void**)p, sizeof(double) * n));
    HIPCK(hipMemset(*p, 0, sizeof(double) * n));
    return MI355CG_OK;
}

int ensure_host_copies(mi355cg_ctx* c) {        // device-generated problem data: fetch the host copies on first use
    if (!c->host_rhs_valid) { c->rhs_h.resize(c->pk_len); if (int rc = download_packed<double>(c, c->b, c->rhs_h.data())) return rc; c->host_rhs_valid = true; }
    if (!c->host_u_valid) { c->u_h.resize(c->pk_len); if (int rc = download_packed<double>(c, c->u, c->u_h.data())) return rc; c->host_u_valid = true; }
    return MI355CG_OK;
}
// The fp32 vectors of a team's part (mi355cg_team_set_dtype): correction, direction ring and A p.  The fp32 residual is not among
// them: a part keeps it in the memory of its fp64 residual vector (what the neighbours' halo messages are addressed to).
int ensure_f32_vectors(mi355cg_ctx* c) {
    if (c->xsteps > 4) return fail(MI355CG_ERR_INVALID, "MI355CG_XSTEPS=8 is fp64 only");
    if (c->whole32.wl.nitems == 0) return fail(MI355CG_ERR_INVALID, "the fp32 kernels march 256-column strips: a part of a 2-D decomposition (cut on 128 columns) cannot run them; use row slabs");
    if (c->g.own_begin % 4 != 0 || c->g.own_len % 4 != 0) return fail(MI355CG_ERR_STATE, "a part's owned range is not a whole number of float4");
    float** fv[] = {&c->xf, &c->apf, &c->pf[0], &c->pf[1], &c->pf[2], &c->pf[3]};
    for (int k = 0; k < 2 + c->xsteps; ++k) {
        if (*fv[k]) continue;
        if (hipMalloc((void**)fv[k], sizeof(float) * c->storage_len) != hipSuccess || hipMemset(*fv[k], 0, sizeof(float) * c->storage_len) != hipSuccess)
            return fail(MI355CG_ERR_HIP, "fp32 vector allocation failed");
    }
    HIPCK(hipDeviceSynchronize());
    return MI355CG_OK;
}

int ensure_u_on_device(mi355cg_ctx* c) {
    if (c->have_u_dev) return MI355CG_OK;
    if (!c->host_u_valid) return fail(MI355CG_ERR_STATE, "no exact solution on host or device");
    if (int rc = upload_packed<double>(c, c->u_h.data(), c->u)) return rc;
    c->have_u_dev = true;
    return MI355CG_OK;
}
// Work space of mi355cg_apply / mi355cg_get_true_residual.  The reference's apply is const and may be called from an
// iteration callback in the middle of a solve, so it must never borrow a solver vector.
int ensure_scratch(mi355cg_ctx* c) {
    for (auto& s : c->scratch) if (!s) { if (int rc = alloc_vec(&s, c->storage_len)) return rc; }
    HIPCK(hipDeviceSynchronize());      // the zero-fill ran on the NULL stream
    return MI355CG_OK;
}

void clear_graphs(mi355cg_ctx* c) {
    for (auto& g : c->graphs) hipGraphExecDestroy(g.exec);
    c->graphs.clear();
}

void prof_begin(mi355cg_ctx* c, hipEvent_t* e0) {
    if (!c->profiling) return;
    *e0 = c->events.get();
    if (*e0) hipEventRecord(*e0, c->stream);
}
void prof_end(mi355cg_ctx* c, int k, hipEvent_t e0) {
    if (!c->profiling || !e0) return;
    hipEvent_t e1 = c->events.get();
    if (!e1) return;
    hipEventRecord(e1, c->stream);
    c->ev_pairs[k].push_back({e0, e1});
}
void prof_collect(mi355cg_ctx* c) {
    for (int k = 0; k < 2; ++k) {
        double tot = 0; long long n = 0;
        for (auto& pr : c->ev_pairs[k]) {
            float ms = 0;
            if (hipEventElapsedTime(&ms, pr.first, pr.second) == hipSuccess) { tot += ms; ++n; }
        }
        c->kernel_ms[k] = n ? tot / n : 0.0;
        c->kernel_launches[k] = n;
        c->ev_pairs[k].clear();
    }
    c->events.reset();
}

}  // namespace

// ---- generic CSR handles ----------------------------------------------------------------------------------
namespace {

CsrView csr_view(const mi355cg_ctx* c) { return CsrView{c->csr_n, c->csr_row_map, c->csr_entries, c->csr_values}; }

void launch_csr_spmv(mi355cg_ctx* c, const double* x, double* y, const double* r, const CgState* s_in, double* partA) {
    SpmvArgs a{};
    a.A = csr_view(c); a.x = x; a.y = y; a.r = r; a.s_in = s_in; a.partA = partA; a.strideA = c->strideA;
    hipLaunchKernelGGL(k_csr_spmv, dim3(c->grid_csr), dim3(kBlock), 0, c->stream, a);
}

// CG on a caller-supplied matrix: xpay (decision + direction), spmv (+ dots), update.  Same device-side state
// machine, stop rules, callback cadence and result fields as the stencil path.
int solve_csr(mi355cg_ctx* c, const mi355cg_params* prm, mi355cg_iter_cb cb, void* user,
              const volatile int* stop_flag, mi355cg_results* out) {
    if (prm->diagnostics) return fail(MI355CG_ERR_INVALID, "per-iteration diagnostics are not available on CSR handles");
    const bool msg = prm->rule == MI355CG_RULE_MSG_MAXNORM;
    IterCfg cfg = make_cfg(prm);
    if (cfg.has_u && !c->have_u_dev) { cfg.has_u = false; cfg.rp.use_u = 0; }      // no true solution was supplied
    const auto t0 = std::chrono::steady_clock::now();
    const size_t bytes = sizeof(double) * c->storage_len;
    HIPCK(hipMemsetAsync(c->x, 0, bytes, c->stream));
    HIPCK(hipMemsetAsync(c->p[0], 0, bytes, c->stream));
    HIPCK(hipMemsetAsync(c->ap, 0, bytes, c->stream));
    HIPCK(hipMemcpyAsync(c->r, c->b, bytes, hipMemcpyDeviceToDevice, c->stream));
    c->cur = 0;
    const PartSrc pA{c->partA, c->grid_csr, c->strideA, 1};
    auto update = [&](bool init, bool with_u) {
        UpdateArgs<double> a{};
        a.begin = 0; a.nvec = c->csr_n;
        a.x = c->x; a.r = c->r; a.p = c->p[0]; a.ap = c->ap; a.u = c->u;
        a.partA = pA.ptr; a.nA = pA.n; a.strideA = pA.fstride; a.esA = 1;
        a.partB = c->partB; a.strideB = c->strideB;
        a.s_in = c->sA; a.s_out = c->sB; a.rule = cfg.rp.rule; a.init = init ? 1 : 0;
        if (with_u) hipLaunchKernelGGL((k_update<double, 1, true>), dim3(c->grid_update), dim3(kBlock), 0, c->stream, a);
        else hipLaunchKernelGGL((k_update<double, 1, false>), dim3(c->grid_update), dim3(kBlock), 0, c->stream, a);
    };
    update(true, cfg.has_u);
    HIPCK(hipGetLastError());
    auto poll = [&]() -> int {
        launch_check(c, cfg, c->stream, own_partB(c));
        HIPCK(hipMemcpyAsync(c->summary_h, c->summary, sizeof(CgState), hipMemcpyDeviceToHost, c->stream));
        HIPCK(hipMemcpyAsync(c->hist_h, c->hist, sizeof(HistEntry) * kHist, hipMemcpyDeviceToHost, c->stream));
        HIPCK(hipStreamSynchronize(c->stream));
        return MI355CG_OK;
    };
    if (int rc = poll()) return rc;
    const double initial_rnorm2 = c->summary_h->rnorm2;
    if (msg && cb) cb(user, 0, DBL_MAX, c->summary_h->rmax, cfg.has_u ? c->summary_h->emax : DBL_MAX);
    const int every = prm->callback_every;
    const int sync_every = std::min(prm->sync_every > 0 ? prm->sync_every : (msg ? 100 : 200), kHist);
    int it_done = 0;
    bool first_chunk = (cb != nullptr || stop_flag != nullptr);
    bool interrupted = false;
    while (!c->summary_h->done) {
        if (stop_flag && *stop_flag) { interrupted = true; break; }
        int m = std::min(sync_every, prm->max_iterations - it_done);
        if (msg && every > 0) m = std::min(m, every - it_done % every);
        if (first_chunk) { m = 1; first_chunk = false; }     // deliver the it == 1 callback / honour a stop request before queueing more
        if (m <= 0) m = 1;
        for (int k = 0; k < m; ++k) {
            XpayArgs xa{};
            xa.n = c->csr_n; xa.r = c->r; xa.p = c->p[0];
            xa.partB = c->partB; xa.nB = c->grid_update; xa.strideB = c->strideB; xa.esB = 1;
            xa.s_in = c->sB; xa.s_out = c->sA; xa.hist = c->hist; xa.rp = cfg.rp; xa.want_diag = 0;
            hipLaunchKernelGGL(k_csr_xpay, dim3(c->grid_update), dim3(kBlock), 0, c->stream, xa);
            launch_csr_spmv(c, c->p[0], c->ap, c->r, c->sA, c->partA);
            update(false, cfg.has_u);
        }
        HIPCK(hipGetLastError());
        if (int rc = poll()) return rc;
        const int it_now = c->summary_h->it;
        if (msg && cb)
            for (int it = it_done + 1; it <= it_now; ++it) {
                const bool stopped_here = c->summary_h->done && c->summary_h->reason != MI355CG_STOP_ITERATIONS && it == it_now;
                if ((it == 1 || (every > 0 && it % every == 0)) && !stopped_here) {
                    const HistEntry& h = c->hist_h[it % kHist];
                    cb(user, it, h.dmax, h.rmax, cfg.has_u ? h.emax : DBL_MAX);
                }
            }
        it_done = it_now;
    }
    const CgState fin = *c->summary_h;
    c->solved = true;
    mi355cg_results res{};
    res.iterations = fin.it;
    res.converged = interrupted ? 0 : fin.converged;
    res.stop_reason = interrupted ? MI355CG_STOP_INTERRUPTED : fin.reason;
    res.final_residual_norm = fin.rmax;
    res.final_precision = fin.it > 0 ? fin.dmax : DBL_MAX;
    res.final_error_norm = cfg.has_u ? fin.emax : DBL_MAX;
    res.r_norm2 = fin.rnorm2; res.initial_r_norm2 = initial_rnorm2;
    res.solve_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (msg && cb) cb(user, res.iterations, res.final_precision, res.final_residual_norm, res.final_error_norm);
    if (out) *out = res;
    return MI355CG_OK;
}

}  // namespace


// ---- F32_MIXED: fp32 inner CG inside fp64 iterative refinement ------------------------------------
// No reference twin (the reference is fp64 only): the pin is the fp64 TRUE residual of the returned x,
// computed with the bit-exact fp64 operator.  Outer step: d = CG_fp32(A, (float) r) ; x += d ;
// r = b - A x in fp64.  Only the relative 2-norm rule is offered.
namespace {

int poll_summary(mi355cg_ctx* c, const IterCfg& cfg) {
    launch_check(c, cfg, c->stream, own_partB(c));
    HIPCK(hipMemcpyAsync(c->summary_h, c->summary, sizeof(CgState), hipMemcpyDeviceToHost, c->stream));
    HIPCK(hipStreamSynchronize(c->stream));
    return MI355CG_OK;
}

// fp32 CG on (xf, rf): xf = 0 on entry, rf holds the right-hand side and ends as the recursive residual.
// resume_r0norm < 0: fresh start (x = 0, p = 0).  resume_r0norm >= 0: residual replacement -- rf already holds the new true
// residual; keep the direction and the CG scalars, restart only the correction vector xf and the reference norm.
int inner_cg_f32(mi355cg_ctx* c, const IterCfg& cfg, int sync_every, const volatile int* stop_flag, int* its, bool* interrupted,
                 double resume_r0norm = -1.0) {
    const Where w{c->stream, &c->whole32, 0};
    const PartSrc pA{c->partA, c->whole32.grid, c->strideA, 1};
    c->nB_own = c->whole32.grid;
    const size_t bytes = sizeof(float) * c->storage_len;
    HIPCK(hipMemsetAsync(c->xf, 0, bytes, c->stream));
    int done_its = 0;
    if (resume_r0norm < 0.0) {
        for (int k = 0; k < c->xsteps; ++k) HIPCK(hipMemsetAsync(c->pf[k], 0, bytes, c->stream));
        HIPCK(hipMemsetAsync(c->apf, 0, bytes, c->stream));
        c->cur = 0;
        launch_update_flat<float, 4>(c, cfg, c->xf, c->rf, c->pf[0], c->apf, (const float*)nullptr, c->stream, c->whole32.grid);
    } else {
        done_its = c->summary_h->it;
        launch_update_flat<float, 4>(c, cfg, c->xf, c->rf, c->pf[c->cur], c->apf, (const float*)nullptr, c->stream, c->whole32.grid, resume_r0norm);
    }
    HIPCK(hipGetLastError());
    if (int rc = poll_summary(c, cfg)) return rc;
    while (!c->summary_h->done) {
        if (stop_flag && *stop_flag) { *interrupted = true; break; }
        const int m = std::max(1, std::min(sync_every, cfg.rp.max_iterations - done_its));
        for (int k = 0; k < m; ++k) {
            hipEvent_t e0 = nullptr;
            prof_begin(c, &e0);
            launch_iteration_stencil<float, 4>(c, cfg, c->rf, c->pf, w, own_partB(c));
            c->cur = (c->cur + 1) % c->xsteps;
            prof_end(c, 0, e0);
            prof_begin(c, &e0);
            launch_iteration_update<float, 4>(c, cfg, c->xf, c->rf, c->pf, (const float*)nullptr, w, pA);
            prof_end(c, 1, e0);
        }
        HIPCK(hipGetLastError());
        if (int rc = poll_summary(c, cfg)) return rc;
        done_its = c->summary_h->it;
    }
    *its = c->summary_h->it;
    c->cur = *its % c->xsteps;
    launch_flush_x<float, 4>(c, c->whole32, c->xf, c->pf, *c->summary_h, c->stream);      // x steps still pending after the last multiple of M
    HIPCK(hipGetLastError());
    return MI355CG_OK;
}

int solve_mixed(mi355cg_ctx* c, const mi355cg_params* prm, mi355cg_iter_cb cb, void* user,
                const volatile int* stop_flag, mi355cg_results* out) {
    if (prm->rule != MI355CG_RULE_REL_2NORM) return fail(MI355CG_ERR_INVALID, "F32_MIXED offers the REL_2NORM rule only");
    if (c->is_slab) return fail(MI355CG_ERR_INVALID, "F32_MIXED is single-GPU only");
    const auto t0 = std::chrono::steady_clock::now();
    c->events.reset(); c->ev_pairs[0].clear(); c->ev_pairs[1].clear();
    const double inner_eps = prm->inner_eps > 0 ? prm->inner_eps : 1e-4;
    // Restarted refinement is the default.  MI355CG_MIXED_RESTART=0 selects residual replacement (direction and CG scalars
    // kept across outer steps): it saves iterations on small grids but stalled at 2e-5 on N = 8192 in round 1
    // (profiles/r01_tune_notes.md), so it stays experimental and every stage is capped.
    const bool restart = env_int("MI355CG_MIXED_RESTART", 1) != 0;
    int stage_cap = 0;                       // replacement mode: iterations a later stage may spend (3x the first stage)
    const int sync_every = std::min(prm->sync_every > 0 ? prm->sync_every : 200, kHist);
    const int rgrid = 1024;
    auto residual_pass = [&](double* norm) -> int {      // rf = (float)(b - ap64), *norm = ||b - ap64||_2
        hipLaunchKernelGGL(k_residual_to_f32, dim3(rgrid), dim3(kBlock), 0, c->stream, c->storage_len, c->g.own_begin, c->g.own_len,
                           c->b, c->ap, c->rf, c->partR);
        HIPCK(hipGetLastError());
        HIPCK(hipMemcpyAsync(c->partR_h, c->partR, sizeof(double) * rgrid, hipMemcpyDeviceToHost, c->stream));
        HIPCK(hipStreamSynchronize(c->stream));
        double s = 0; for (int i = 0; i < rgrid; ++i) s += c->partR_h[i];
        *norm = std::sqrt(s);
        return MI355CG_OK;
    };
    const size_t bytes64 = sizeof(double) * c->storage_len;
    HIPCK(hipMemsetAsync(c->x, 0, bytes64, c->stream));
    HIPCK(hipMemsetAsync(c->ap, 0, bytes64, c->stream));
    double bnorm = 0, rnorm = 0;
    if (int rc = residual_pass(&bnorm)) return rc;       // x = 0: r = b
    rnorm = bnorm;
    int total = 0, outer = 0;
    bool interrupted = false, converged = bnorm == 0.0;
    while (!converged && total < prm->max_iterations && !interrupted) {
        mi355cg_params ip = *prm;
        ip.eps_rel = inner_eps; ip.diagnostics = 0;
        // restarted refinement: every inner solve starts from scratch with the remaining budget;
        // residual replacement: the iteration counter runs on across outer steps, so the cap is the global one
        ip.max_iterations = restart ? prm->max_iterations - total : prm->max_iterations;
        if (!restart && stage_cap > 0) ip.max_iterations = std::min(ip.max_iterations, total + stage_cap);
        const IterCfg cfg = make_cfg(&ip);
        int its = 0;
        const double resume = (!restart && outer > 0) ? rnorm : -1.0;
        if (int rc = inner_cg_f32(c, cfg, sync_every, stop_flag, &its, &interrupted, resume)) return rc;
        const int its_this = restart ? its : its - total;
        total = restart ? total + its : its; ++outer;
        if (!restart && stage_cap == 0) stage_cap = std::max(2000, 3 * its_this);
        hipLaunchKernelGGL(k_accumulate_f32, dim3(flat_grid(c->g.own_len)), dim3(kBlock), 0, c->stream, c->g.own_begin, c->g.own_len, c->x, c->xf);
        launch_apply<double, 2>(c, c->x, c->ap, whole_part(c));
        const double prev = rnorm;
        if (int rc = residual_pass(&rnorm)) return rc;
        if (cb) cb(user, total, 0.0, rnorm, 0.0);
        converged = !prm->fixed_iterations && rnorm <= prm->eps_rel * bnorm;
        if (prm->fixed_iterations || its_this == 0) break;
        if (!converged && rnorm > 0.5 * prev) break;      // fp32 cannot improve this x any further
    }
    // leave the fp64 residual of the returned x in c->r for mi355cg_get_recursive_residual
    hipLaunchKernelGGL((k_sub<double>), dim3(flat_grid(c->g.own_len)), dim3(kBlock), 0, c->stream, c->g.own_begin, c->g.own_len, c->b, c->ap, c->r);
    HIPCK(hipStreamSynchronize(c->stream));
    c->solved = true; c->cur = 0; c->nB_own = c->whole.grid;
    prof_collect(c);
    mi355cg_results res{};
    res.iterations = total; res.converged = converged ? 1 : 0;
    res.stop_reason = interrupted ? MI355CG_STOP_INTERRUPTED : (converged ? MI355CG_STOP_RESIDUAL : MI355CG_STOP_ITERATIONS);
    res.final_residual_norm = res.final_precision = res.final_error_norm = DBL_MAX;
    res.r_norm2 = rnorm; res.initial_r_norm2 = bnorm;
    res.refine_outer = outer; res.refine_true_rel = bnorm > 0 ? rnorm / bnorm : 0.0;
    res.solve_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (out) *out = res;
    return MI355CG_OK;
}

}  // namespace

// ====================================================================================================
extern "C" {

const char* mi355cg_last_error(void) { return g_err.c_str(); }
const char* mi355cg_version(void) { return "mi355cg 0.2 (gfx950)"; }

static int create_impl(int n, int m, double a, double b, double c_, double d, int dtype, int device,
                       int y_lo, int y_hi, int s_lo, int s_hi, bool part, mi355cg_handle* out) {
    if (!out) return fail(MI355CG_ERR_INVALID, "out is null");
    *out = nullptr;
    if (dtype != MI355CG_F64 && dtype != MI355CG_F32_MIXED) return fail(MI355CG_ERR_INVALID, "unknown dtype %d", dtype);
    GridParams gp;
    if (!grid_params_init(&gp, n, m, a, b, c_, d))
        return fail(MI355CG_ERR_INVALID, "grid %dx%d rejected: the L-shaped index map is only consistent for n == m, even, >= 6", n, m);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(MI355CG_ERR_HIP, "no HIP device available (libmi355cg has no CPU fallback)");
    if (device < 0 || device >= ndev) return fail(MI355CG_ERR_INVALID, "device %d out of range (0..%d)", device, ndev - 1);
    HIPCK(hipSetDevice(device));

    const int ns_all = strips_total(gp, 2);
    if (!part) { y_lo = 1; y_hi = gp.n - 1; s_lo = 0; s_hi = ns_all; }
    if (y_lo < 1 || y_hi > gp.n - 1 || y_lo > y_hi)
        return fail(MI355CG_ERR_INVALID, "part rows [%d, %d] outside 1..%d", y_lo, y_hi, gp.n - 1);
    if (s_lo < 0 || s_hi > ns_all || s_lo >= s_hi)
        return fail(MI355CG_ERR_INVALID, "part strips [%d, %d) outside 0..%d", s_lo, s_hi, ns_all);
    if (dtype == MI355CG_F32_MIXED && part) return fail(MI355CG_ERR_INVALID, "F32_MIXED is single-GPU only");
    mi355cg_ctx* c = new mi355cg_ctx();
    c->device = device; c->dtype = dtype; c->gp = gp; c->is_slab = part;
    c->s_lo = s_lo; c->s_hi = s_hi;
    build_geom(c, 2, y_lo, y_hi);           // fp64 layout; the fp32 kernels use VEC=4 on the same pitches
    build_plans(c);
    c->strideA = std::max({c->whole.grid, c->interior.grid + c->edge.grid, c->whole32.grid, 1});
    c->strideB = c->strideA;                // the update launches run on the stencil's grids

    int rc = MI355CG_OK;
    auto cleanup = [&]() { mi355cg_destroy(c); return rc; };
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { rc = fail(MI355CG_ERR_HIP, "hipStreamCreate failed"); return cleanup(); }
    const long long L = c->storage_len;
    // x is folded every M-th iteration: 4 by default.  8 saves another 0.125 words per iteration and measured +0.2-0.7 % for
    // four more vectors (fp64 only: the fp32 8-step launch needs 262 VGPRs and would halve the resident waves)
    { const int m = env_int("MI355CG_XSTEPS", 4); c->xsteps = m == 2 ? 2 : ((m == 8 && dtype == MI355CG_F64) ? 8 : 4); }
    double** vecs[] = {&c->x, &c->r, &c->p[0], &c->p[1], &c->ap, &c->b, &c->u, &c->p[2], &c->p[3], &c->p[4], &c->p[5], &c->p[6], &c->p[7]};
    for (int k = 0; k < 7 + (c->xsteps - 2); ++k) if ((rc = alloc_vec(vecs[k], L))) return cleanup();
    if (dtype == MI355CG_F32_MIXED) {
        float** fv[] = {&c->xf, &c->rf, &c->pf[0], &c->pf[1], &c->apf, &c->pf[2], &c->pf[3], &c->pf[4], &c->pf[5], &c->pf[6], &c->pf[7]};
        const int nfv = 5 + (c->xsteps - 2);
        for (int k = 0; k < nfv; ++k) {
            float** v = fv[k];
            if (hipMalloc((void**)v, sizeof(float) * L) != hipSuccess || hipMemset(*v, 0, sizeof(float) * L) != hipSuccess) {
                rc = fail(MI355CG_ERR_HIP, "fp32 vector allocation failed"); return cleanup();
            }
        }
    }
    if ((rc = alloc_vec(&c->packed, std::max<long long>(c->pk_len, 1)))) return cleanup();
    if ((rc = alloc_vec(&c->partA, (long long)FA_COUNT * c->strideA))) return cleanup();
    if ((rc = alloc_vec(&c->partB, (long long)FB_COUNT * c->strideB))) return cleanup();
    if ((rc = alloc_vec(&c->partR, 2048))) return cleanup();
    c->rec_width = kRecHeader + 2 * c->g.Pu;            // [sums | first owned row | last owned row]
    if ((rc = alloc_vec(&c->sumsA, c->rec_width))) return cleanup();
    if ((rc = alloc_vec(&c->sumsB, c->rec_width))) return cleanup();
    if (hipMalloc((void**)&c->sA, sizeof(CgState)) != hipSuccess || hipMalloc((void**)&c->sB, sizeof(CgState)) != hipSuccess ||
        hipMalloc((void**)&c->summary, sizeof(CgState)) != hipSuccess || hipMalloc((void**)&c->hist, sizeof(HistEntry) * kHist) != hipSuccess ||
        hipHostMalloc((void**)&c->summary_h, sizeof(CgState)) != hipSuccess || hipHostMalloc((void**)&c->hist_h, sizeof(HistEntry) * kHist) != hipSuccess ||
        hipHostMalloc((void**)&c->partR_h, sizeof(double) * 2048) != hipSuccess) {
        rc = fail(MI355CG_ERR_HIP, "state allocation failed"); return cleanup();
    }
    hipMemset(c->sA, 0, sizeof(CgState)); hipMemset(c->sB, 0, sizeof(CgState)); hipMemset(c->summary, 0, sizeof(CgState));
    hipMemset(c->hist, 0, sizeof(HistEntry) * kHist);
    if (hipMalloc((void**)&c->qctr, sizeof(int) * 2 * kXcds * kQueueSubs * kQueuePitch) != hipSuccess || hipMemset(c->qctr, 0, sizeof(int) * 2 * kXcds * kQueueSubs * kQueuePitch) != hipSuccess) { rc = fail(MI355CG_ERR_HIP, "queue counter allocation failed"); return cleanup(); }
    if (hipHostMalloc((void**)&c->stop_h, sizeof(int)) != hipSuccess) { rc = fail(MI355CG_ERR_HIP, "stop word allocation failed"); return cleanup(); }
    *c->stop_h = 0;
    // The zero-fills above run on the NULL stream and are asynchronous to the host; the context's own stream is
    // non-blocking and does not order with them.  Without this wait a delayed memset can land AFTER the first upload
    // or kernel of the context and wipe it (seen as a right-hand side of zeros -> "converged" at iteration 0).
    HIPCK(hipDeviceSynchronize());

    if (env_int("MI355CG_DEVICE_SETUP", 0)) {
        // opt-in: b and u generated on the device, no host pass and no upload; <= 1 ulp from the host values
        c->host_rhs_valid = false; c->host_u_valid = false;
        if ((rc = mi355cg_setup_on_device(c))) return cleanup();
    } else {
        // problem data on the host in the part's packed order (bit-identical to the reference on the same libm), then into
        // storage layout on the device
        c->rhs_h.resize(c->pk_len); c->u_h.resize(c->pk_len);
        grid_fill_box(gp, c->g.y_lo, c->g.y_hi, s_lo * kStripCols, s_hi == ns_all ? gp.n : s_hi * kStripCols, c->rhs_h.data(), c->u_h.data(), nullptr, nullptr);
        if ((rc = upload_packed<double>(c, c->rhs_h.data(), c->b))) return cleanup();
    }
    *out = c;
    return MI355CG_OK;
}

int mi355cg_create(int n, int m, double a, double b, double c_, double d, int dtype, int device, mi355cg_handle* out) {
    return create_impl(n, m, a, b, c_, d, dtype, device, 0, 0, 0, 0, false, out);
}
int mi355cg_create_slab(int n, int m, double a, double b, double c_, double d, int dtype, int device,
                        int y_lo, int y_hi, mi355cg_handle* out) {
    GridParams gp;
    if (!grid_params_init(&gp, n, m, a, b, c_, d)) return fail(MI355CG_ERR_INVALID, "grid %dx%d rejected: the L-shaped index map is only consistent for n == m, even, >= 6", n, m);
    return create_impl(n, m, a, b, c_, d, dtype, device, y_lo, y_hi, 0, strips_total(gp, 2), true, out);
}
int mi355cg_create_part(int n, int m, double a, double b, double c_, double d, int dtype, int device,
                        int y_lo, int y_hi, int x_lo, int x_hi, mi355cg_handle* out) {
    GridParams gp;
    if (!grid_params_init(&gp, n, m, a, b, c_, d)) return fail(MI355CG_ERR_INVALID, "grid %dx%d rejected: the L-shaped index map is only consistent for n == m, even, >= 6", n, m);
    const int ns_all = strips_total(gp, 2);
    if (x_lo % kStripCols != 0 || (x_hi % kStripCols != 0 && x_hi < gp.n))
        return fail(MI355CG_ERR_INVALID, "part columns [%d, %d): x-cuts must be multiples of %d", x_lo, x_hi, kStripCols);
    return create_impl(n, m, a, b, c_, d, dtype, device, y_lo, y_hi, x_lo / kStripCols, x_hi >= gp.n ? ns_all : x_hi / kStripCols, true, out);
}

int mi355cg_create_csr(long long nrows, const int* row_map, const int* entries, const double* values,
                       int device, mi355cg_handle* out) {
    if (!out) return fail(MI355CG_ERR_INVALID, "out is null");
    *out = nullptr;
    if (nrows <= 0 || !row_map || !entries || !values) return fail(MI355CG_ERR_INVALID, "bad CSR arguments");
    if (row_map[0] != 0) return fail(MI355CG_ERR_INVALID, "row_map[0] must be 0");
    const long long nnz = row_map[nrows];
    for (long long i = 0; i < nrows; ++i) if (row_map[i + 1] < row_map[i]) return fail(MI355CG_ERR_INVALID, "row_map is not monotone at row %lld", i);
    for (long long j = 0; j < nnz; ++j) if (entries[j] < 0 || entries[j] >= nrows) return fail(MI355CG_ERR_INVALID, "column index %d out of range at entry %lld", entries[j], j);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(MI355CG_ERR_HIP, "no HIP device available (libmi355cg has no CPU fallback)");
    if (device < 0 || device >= ndev) return fail(MI355CG_ERR_INVALID, "device %d out of range (0..%d)", device, ndev - 1);
    HIPCK(hipSetDevice(device));
    mi355cg_ctx* c = new mi355cg_ctx();
    c->device = device; c->dtype = MI355CG_F64; c->is_csr = true;
    c->csr_n = nrows; c->csr_nnz = nnz;
    c->gp.size = nrows; c->pk_begin = 0; c->pk_len = nrows; c->storage_len = nrows; c->pg.pk_len = nrows;
    c->g.own_begin = 0; c->g.own_len = nrows;
    const long long nblk = (nrows + kBlock - 1) / kBlock;
    c->grid_csr = (int)std::max<long long>(1, std::min<long long>(2048, nblk));
    c->grid_update = (int)std::max<long long>(1, std::min<long long>(512, nblk));
    c->strideA = c->grid_csr; c->strideB = c->grid_update; c->nB_own = c->grid_update;
    int rc = MI355CG_OK;
    auto cleanup = [&]() { mi355cg_destroy(c); return rc; };
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { rc = fail(MI355CG_ERR_HIP, "hipStreamCreate failed"); return cleanup(); }
    double** vecs[] = {&c->x, &c->r, &c->p[0], &c->p[1], &c->ap, &c->b, &c->u};
    for (auto v : vecs) if ((rc = alloc_vec(v, nrows))) return cleanup();
    if ((rc = alloc_vec(&c->partA, (long long)FA_COUNT * c->strideA))) return cleanup();
    if ((rc = alloc_vec(&c->partB, (long long)FB_COUNT * c->strideB))) return cleanup();
    if ((rc = alloc_vec(&c->partR, 2048))) return cleanup();
    if (hipMalloc((void**)&c->csr_row_map, sizeof(int) * (nrows + 1)) != hipSuccess || hipMalloc((void**)&c->csr_entries, sizeof(int) * std::max<long long>(nnz, 1)) != hipSuccess ||
        hipMalloc((void**)&c->csr_values, sizeof(double) * std::max<long long>(nnz, 1)) != hipSuccess ||
        hipMalloc((void**)&c->sA, sizeof(CgState)) != hipSuccess || hipMalloc((void**)&c->sB, sizeof(CgState)) != hipSuccess ||
        hipMalloc((void**)&c->summary, sizeof(CgState)) != hipSuccess || hipMalloc((void**)&c->hist, sizeof(HistEntry) * kHist) != hipSuccess ||
        hipHostMalloc((void**)&c->summary_h, sizeof(CgState)) != hipSuccess || hipHostMalloc((void**)&c->hist_h, sizeof(HistEntry) * kHist) != hipSuccess ||
        hipHostMalloc((void**)&c->partR_h, sizeof(double) * 2048) != hipSuccess) {
        rc = fail(MI355CG_ERR_HIP, "allocation failed"); return cleanup();
    }
    if (hipMemcpy(c->csr_row_map, row_map, sizeof(int) * (nrows + 1), hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(c->csr_entries, entries, sizeof(int) * nnz, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(c->csr_values, values, sizeof(double) * nnz, hipMemcpyHostToDevice) != hipSuccess) {
        rc = fail(MI355CG_ERR_HIP, "CSR upload failed"); return cleanup();
    }
    hipMemset(c->sA, 0, sizeof(CgState)); hipMemset(c->sB, 0, sizeof(CgState)); hipMemset(c->summary, 0, sizeof(CgState));
    hipMemset(c->hist, 0, sizeof(HistEntry) * kHist);
    // The zero-fills above run on the NULL stream and are asynchronous to the host; the context's own stream is
    // non-blocking and does not order with them.  Without this wait a delayed memset can land AFTER the first upload
    // or kernel of the context and wipe it (seen as a right-hand side of zeros -> "converged" at iteration 0).
    HIPCK(hipDeviceSynchronize());
    c->rhs_h.assign(nrows, 0.0); c->u_h.assign(nrows, 0.0);
    *out = c;
    return MI355CG_OK;
}

int mi355cg_set_true_solution(mi355cg_handle c, const double* u) {
    if (!c || !u) return fail(MI355CG_ERR_INVALID, "null argument");
    HIPCK(hipSetDevice(c->device));
    c->host_u_valid = true;
    c->u_h.resize(c->pk_len);
    std::memcpy(c->u_h.data(), u, sizeof(double) * c->pk_len);
    if (int rc = upload_packed<double>(c, c->u_h.data(), c->u)) return rc;
    c->have_u_dev = true;
    return MI355CG_OK;
}

void mi355cg_destroy(mi355cg_handle c) {
    if (!c) return;
    hipSetDevice(c->device);
    if (c->stream) hipStreamSynchronize(c->stream);
    void* dev[] = {c->x, c->r, c->p[0], c->p[1], c->p[2], c->p[3], c->p[4], c->p[5], c->p[6], c->p[7], c->ap, c->b, c->u, c->scratch[0], c->scratch[1], c->xf, c->rf,
                   c->pf[0], c->pf[1], c->pf[2], c->pf[3], c->pf[4], c->pf[5], c->pf[6], c->pf[7], c->apf,
                   c->packed, c->partA, c->partB, c->partR, c->sumsA, c->sumsB, c->sA, c->sB, c->summary, c->hist, c->qctr};
    for (void* p : dev) if (p) hipFree(p);
    if (c->csr_row_map) hipFree(c->csr_row_map);
    if (c->csr_entries) hipFree(c->csr_entries);
    if (c->csr_values) hipFree(c->csr_values);
    if (c->summary_h) hipHostFree(c->summary_h);
    if (c->hist_h) hipHostFree(c->hist_h);
    if (c->partR_h) hipHostFree(c->partR_h);
    if (c->stop_h) hipHostFree(c->stop_h);
    clear_graphs(c);
    c->events.destroy();
    for (hipEvent_t e : c->ev_loop) if (e) hipEventDestroy(e);
    if (c->stream) hipStreamDestroy(c->stream);
    delete c;
}

long long mi355cg_size(mi355cg_handle c) { return c ? c->gp.size : -1; }

int mi355cg_get_rhs(mi355cg_handle c, double* out) {
    if (!c || !out) return fail(MI355CG_ERR_INVALID, "null argument");
    if (int rc = ensure_host_copies(c)) return rc;
    std::memcpy(out, c->rhs_h.data(), sizeof(double) * c->pk_len);
    return MI355CG_OK;
}
int mi355cg_get_true_solution(mi355cg_handle c, double* out) {
    if (!c || !out) return fail(MI355CG_ERR_INVALID, "null argument");
    if (int rc = ensure_host_copies(c)) return rc;
    std::memcpy(out, c->u_h.data(), sizeof(double) * c->pk_len);
    return MI355CG_OK;
}
int mi355cg_get_node_coords(mi355cg_handle c, double* xs, double* ys) {
    if (!c || !xs || !ys) return fail(MI355CG_ERR_INVALID, "null argument");
    if (c->is_csr) return fail(MI355CG_ERR_INVALID, "a CSR handle has no grid coordinates");
    const int ns_all = strips_total(c->gp, 2);
    grid_fill_box(c->gp, c->g.y_lo, c->g.y_hi, c->s_lo * kStripCols, c->s_hi == ns_all ? c->gp.n : c->s_hi * kStripCols, nullptr, nullptr, xs, ys);
    return MI355CG_OK;
}
// regenerate b and u of this handle's cells ON THE DEVICE (k_setup).  Opt-in: the device exp() is within
// 1 ulp of glibc's, not identical, so the vectors -- and with them every later number -- may differ from the reference's
// in the last bits.  The host copies are fetched lazily.
int mi355cg_setup_on_device(mi355cg_handle c) {
    if (!c) return fail(MI355CG_ERR_INVALID, "null handle");
    if (c->is_csr) return fail(MI355CG_ERR_INVALID, "a CSR handle has no grid to set up");
    HIPCK(hipSetDevice(c->device));
    if (c->is_slab) HIPCK(hipDeviceSynchronize());
    SetupArgs s{};
    s.pg = c->pg; s.a = c->gp.a; s.c = c->gp.c; s.x_step = c->gp.x_step; s.y_step = c->gp.y_step; s.xk = c->gp.x_k; s.yk = c->gp.y_k;
    s.n = c->gp.n; s.m = c->gp.m;
    if (c->pk_len > 0) hipLaunchKernelGGL(k_setup, dim3(flat_grid(c->pk_len)), dim3(kBlock), 0, c->stream, s, c->b, c->u);
    HIPCK(hipGetLastError());
    HIPCK(hipStreamSynchronize(c->stream));
    c->have_u_dev = true; c->host_rhs_valid = false; c->host_u_valid = false;
    return MI355CG_OK;
}
int mi355cg_set_rhs(mi355cg_handle c, const double* b) {
    if (!c || !b) return fail(MI355CG_ERR_INVALID, "null argument");
    HIPCK(hipSetDevice(c->device));
    c->host_rhs_valid = true;
    c->rhs_h.resize(c->pk_len);
    std::memcpy(c->rhs_h.data(), b, sizeof(double) * c->pk_len);
    return upload_packed<double>(c, c->rhs_h.data(), c->b);
}

// The reference's apply is const (matrix_free_system.hpp:56) and may be called from an iteration callback in the middle
// of a solve: both entry points work on dedicated scratch vectors, never on a solver vector.
int mi355cg_apply_device(mi355cg_handle c, const double* x_dev, double* y_dev) {
    if (!c || !x_dev || !y_dev) return fail(MI355CG_ERR_INVALID, "null argument");
    if (c->is_slab) return fail(MI355CG_ERR_STATE, "this handle owns one part of a decomposed grid: use the mi355cg_team_* / mi355cg_dist_* entry points");
    HIPCK(hipSetDevice(c->device));
    if (c->is_csr) {
        launch_csr_spmv(c, x_dev, y_dev, nullptr, nullptr, nullptr);
        HIPCK(hipGetLastError());
        HIPCK(hipStreamSynchronize(c->stream));
        return MI355CG_OK;
    }
    if (int rc = ensure_scratch(c)) return rc;
    hipLaunchKernelGGL((k_unpack<double>), dim3(flat_grid(c->pk_len)), dim3(kBlock), 0, c->stream, c->pg, x_dev, c->scratch[0]);
    launch_apply<double, 2>(c, c->scratch[0], c->scratch[1], whole_part(c));
    hipLaunchKernelGGL((k_pack<double>), dim3(flat_grid(c->pk_len)), dim3(kBlock), 0, c->stream, c->pg, c->scratch[1], y_dev);
    HIPCK(hipGetLastError());
    HIPCK(hipStreamSynchronize(c->stream));
    return MI355CG_OK;
}

int mi355cg_apply(mi355cg_handle c, const double* x, double* y) {
    if (!c || !x || !y) return fail(MI355CG_ERR_INVALID, "null argument");
    if (c->is_slab) return fail(MI355CG_ERR_STATE, "this handle owns one part of a decomposed grid: use the mi355cg_team_* / mi355cg_dist_* entry points");
    HIPCK(hipSetDevice(c->device));
    if (int rc = ensure_scratch(c)) return rc;
    if (int rc = upload_packed<double>(c, x, c->scratch[0])) return rc;
    if (c->is_csr) launch_csr_spmv(c, c->scratch[0], c->scratch[1], nullptr, nullptr, nullptr);
    else launch_apply<double, 2>(c, c->scratch[0], c->scratch[1], whole_part(c));
    HIPCK(hipGetLastError());
    return download_packed<double>(c, c->scratch[1], y);
}

void mi355cg_default_params(mi355cg_params* p, int rule) {
    if (!p) return;
    std::memset(p, 0, sizeof *p);
    p->rule = rule;
    p->max_iterations = 10000;                 // solver.hpp:36, matrix_free_system.hpp:101
    p->eps_precision = 1e-6; p->eps_residual = 1e-6; p->eps_exact_error = 1e-6;   // msg_solver.hpp:52-56
    p->eps_rel = 1e-6;                         // matrix_free_system.hpp:100
    p->use_true_solution = 1;
    p->callback_every = rule == MI355CG_RULE_MSG_MAXNORM ? 100 : 1;
    p->diagnostics = 0;
    p->sync_every = 0;
    p->fixed_iterations = 0;
    p->inner_eps = 0.0;
}

int mi355cg_solve(mi355cg_handle c, const mi355cg_params* prm, mi355cg_iter_cb cb, void* user,
                  const volatile int* stop_flag, mi355cg_results* out) {
    if (!c || !prm) return fail(MI355CG_ERR_INVALID, "null argument");
    if (prm->rule != MI355CG_RULE_MSG_MAXNORM && prm->rule != MI355CG_RULE_REL_2NORM) return fail(MI355CG_ERR_INVALID, "unknown rule %d", prm->rule);
    if (c->is_slab) return fail(MI355CG_ERR_STATE, "this handle owns one part of a decomposed grid: use the mi355cg_team_* / mi355cg_dist_* entry points");
    HIPCK(hipSetDevice(c->device));
    if (c->is_csr) return solve_csr(c, prm, cb, user, stop_flag, out);
    if (c->dtype == MI355CG_F32_MIXED) return solve_mixed(c, prm, cb, user, stop_flag, out);
    const bool msg = prm->rule == MI355CG_RULE_MSG_MAXNORM;
    const IterCfg cfg = make_cfg(prm);
    const bool diag = cfg.want_diag != 0;
    c->nB_own = c->whole.grid;
    if (cfg.has_u) if (int rc = ensure_u_on_device(c)) return rc;
    if (diag) if (int rc = ensure_scratch(c)) return rc;

    const auto t0 = std::chrono::steady_clock::now();
    c->events.reset(); c->ev_pairs[0].clear(); c->ev_pairs[1].clear();

    // x = 0, r = b, z = 0 (the first stencil makes z = r + 0*z = r)    msg_solver.cpp:33-39
    // One pass over the owned range.  Everything outside it (pitch padding, the rows around the grid) was zeroed when the
    // vectors were allocated and no launch writes anything but zeros there; the other directions of the ring are written
    // (iterations 1 .. M-1) before the folded x update first reads them (iteration M).
    c->cur = 0;
    if (c->qctr && c->dyn_rows > 0) HIPCK(hipMemsetAsync(c->qctr, 0, sizeof(int) * 2 * kXcds * kQueueSubs * kQueuePitch, c->stream));
    {
        FreshArgs<double> f{};
        f.begin = c->g.own_begin / 2; f.nvec = c->g.own_len / 2;
        f.b = c->b; f.x = c->x; f.r = c->r; f.p0 = c->p[0]; f.u = c->u;
        f.partB = c->partB; f.strideB = c->strideB; f.s_out = c->sB;
        if (cfg.has_u) hipLaunchKernelGGL((k_init_fresh<double, 2, true>), dim3(c->whole.grid), dim3(kBlock), 0, c->stream, f);
        else hipLaunchKernelGGL((k_init_fresh<double, 2, false>), dim3(c->whole.grid), dim3(kBlock), 0, c->stream, f);
    }
    HIPCK(hipGetLastError());

    // The stop request: the reference tests its flag at the top of EVERY iteration (msg_solver.cpp:82-87).  Here block 0 of every
    // update launch samples a pinned word and the next stencil prologue turns it into INTERRUPTED, so a request is honoured at the
    // next iteration boundary of the DEVICE, however many iterations the host has already queued.  The caller's flag lives in
    // ordinary host memory; while this thread waits for a chunk it forwards the flag to the pinned word.
    *c->stop_h = 0;
    c->stop_dev = nullptr;
    if (stop_flag) HIPCK(hipHostGetDevicePointer((void**)&c->stop_dev, c->stop_h, 0));
    auto wait_stream = [&]() -> int {
        if (!stop_flag) { HIPCK(hipStreamSynchronize(c->stream)); return MI355CG_OK; }
        for (;;) {
            if (*stop_flag) *c->stop_h = 1;
            const hipError_t q = hipStreamQuery(c->stream);
            if (q == hipSuccess) break;
            if (q != hipErrorNotReady) HIPCK(q);
            std::this_thread::yield();
        }
        return MI355CG_OK;
    };
    auto poll = [&]() -> int {
        launch_check(c, cfg, c->stream, own_partB(c));
        HIPCK(hipMemcpyAsync(c->summary_h, c->summary, sizeof(CgState), hipMemcpyDeviceToHost, c->stream));
        HIPCK(hipMemcpyAsync(c->hist_h, c->hist, sizeof(HistEntry) * kHist, hipMemcpyDeviceToHost, c->stream));
        return wait_stream();
    };
    // The state of iteration 0 is only fetched when somebody looks at it (the it = 0 callback, msg_solver.cpp:75-77);
    // ||r0|| travels in the state and is read with the last poll.
    *c->summary_h = CgState{};
    if (msg && cb) {
        if (int rc = poll()) return rc;
        cb(user, 0, DBL_MAX, c->summary_h->rmax, cfg.has_u ? c->summary_h->emax : DBL_MAX);
    }
    if (!c->ev_loop[0]) { HIPCK(hipEventCreate(&c->ev_loop[0])); HIPCK(hipEventCreate(&c->ev_loop[1])); }
    HIPCK(hipEventRecord(c->ev_loop[0], c->stream));

    // The reference recomputes ||x - u|| every iteration (msg_solver.cpp:132-139), but the value is only
    // observable through the exact-error criterion, the periodic callbacks and the final report: read u
    // on exactly those iterations (same values), and once more after the loop if the last one skipped it.
    const int every = prm->callback_every;
    auto need_u = [&](int it) { return diag || cfg.rp.eps_exact_error > 0 || it == 1 || (every > 0 && it % every == 0); };
    // iterations enqueued between two host polls (the reference polls its stop flag every iteration, msg_solver.cpp:82)
    int sync_every = prm->sync_every > 0 ? prm->sync_every : (msg ? 100 : 200);
    sync_every = std::min(sync_every, kHist);
    int it_done = 0;
    bool interrupted = false;
    // A caller that watches the solve (callback or stop flag) gets the first iteration on its own: the it == 1 callback
    // is delivered, and a stop requested from it is honoured, before any further work is queued.
    bool first_chunk = cb != nullptr || stop_flag != nullptr;
    if (std::memcmp(&c->graph_prm, prm, sizeof *prm) != 0 || c->graph_stop != (stop_flag != nullptr)) {         // kernel arguments embed the solve's parameters
        clear_graphs(c); c->graph_prm = *prm; c->graph_stop = stop_flag != nullptr;                                // (and whether the stop word is sampled): graphs live as long as those do
    }
    const bool graph_ok = !c->profiling && !diag &&
                          (c->use_graph == 1 || (c->use_graph < 0 && c->g.own_len < (4LL << 20) && prm->max_iterations >= 4 * sync_every));
    while (!c->summary_h->done) {
        if (stop_flag && *stop_flag) { interrupted = true; break; }            // msg_solver.cpp:82-87
        int m = std::min(sync_every, prm->max_iterations - it_done);
        if (msg && every > 0) m = std::min(m, every - it_done % every);        // land on the callback iterations
        if (first_chunk) { m = 1; first_chunk = false; }
        if (m <= 0) m = 1;                                                    // lets the kernels record ITERATIONS
        auto enqueue_chunk = [&]() -> int {
            for (int k = 0; k < m; ++k) {
                hipEvent_t e0 = nullptr;
                prof_begin(c, &e0);
                launch_iteration_stencil<double, 2>(c, cfg, c->r, c->p, whole_part(c), own_partB(c));
                c->cur = (c->cur + 1) % c->xsteps;
                prof_end(c, 0, e0);
                prof_begin(c, &e0);
                IterCfg ucfg = cfg;
                ucfg.has_u = cfg.has_u && need_u(it_done + k + 1);      // skip the u stream when nothing reads the error norm
                launch_iteration_update<double, 2>(c, ucfg, c->x, c->r, c->p, c->u, whole_part(c), own_partA(c));
                prof_end(c, 1, e0);
                if (diag) {
                    // MatrixFreeSolver's per-iteration report needs the TRUE residual (matrix_free_system.cpp:457-463): a
                    // second apply and its norm, all in-stream; the value lands in the history entry of this iteration.
                    launch_apply<double, 2>(c, c->x, c->scratch[0], whole_part(c));
                    hipLaunchKernelGGL((k_resid2<double>), dim3(1024), dim3(kBlock), 0, c->stream, c->g.own_begin, c->g.own_len, c->b, c->scratch[0], c->partR);
                    hipLaunchKernelGGL(k_resid2_hist, dim3(1), dim3(kBlock), 0, c->stream, c->partR, 1024, c->sB, c->hist);
                }
            }
            return MI355CG_OK;
        };
        if (graph_ok && m >= 8) {
            // Launch-bound grids: replay the chunk as one hipGraph.  The chunk's kernel arguments depend only on
            // (m, direction-buffer parity, which iterations read u), so equal shapes share an instantiated graph.
            std::vector<char> flags(m);
            for (int k = 0; k < m; ++k) flags[k] = cfg.has_u && need_u(it_done + k + 1);
            mi355cg_ctx::ChunkGraph* hit = nullptr;
            for (auto& g : c->graphs) if (g.m == m && g.cur == c->cur && g.flags == flags) hit = &g;
            if (!hit) {
                hipGraph_t graph = nullptr; hipGraphExec_t exec = nullptr;
                const int cur0 = c->cur;
                HIPCK(hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal));
                const int rc = enqueue_chunk();
                const hipError_t e1 = hipStreamEndCapture(c->stream, &graph);
                c->cur = cur0;                                   // the capture only recorded; nothing ran
                if (rc) return rc;
                HIPCK(e1);
                HIPCK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
                HIPCK(hipGraphDestroy(graph));
                c->graphs.push_back({m, cur0, flags, exec});
                hit = &c->graphs.back();
            }
            HIPCK(hipGraphLaunch(hit->exec, c->stream));
            c->cur = (c->cur + m) % c->xsteps;                    // what enqueue_chunk would have left behind
        } else {
            if (int rc = enqueue_chunk()) return rc;
        }
        HIPCK(hipGetLastError());
        if (int rc = poll()) return rc;
        const int it_now = c->summary_h->it;
        if (cb) for (int it = it_done + 1; it <= it_now; ++it) {
            const HistEntry& h = c->hist_h[it % kHist];
            if (diag) {
                cb(user, it - 1, std::sqrt(h.d2), std::sqrt(h.tr2), std::sqrt(h.e2));     // matrix_free_system.cpp:466-468 (0-based index)
            } else if (msg) {
                // callbacks only on iterations that did NOT stop (msg_solver.cpp:172-183 sits after the breaks)
                // (an interruption is noticed at the top of the NEXT iteration, :82-87, i.e. after this iteration's callback)
                const bool stopped_here = c->summary_h->done && c->summary_h->reason != MI355CG_STOP_ITERATIONS &&
                                          c->summary_h->reason != MI355CG_STOP_INTERRUPTED && it == it_now;
                if ((it == 1 || (every > 0 && it % every == 0)) && !stopped_here) cb(user, it, h.dmax, h.rmax, cfg.has_u ? h.emax : DBL_MAX);
            }
        }
        it_done = it_now;
    }
    HIPCK(hipEventRecord(c->ev_loop[1], c->stream));
    CgState fin = *c->summary_h;
    c->stop_dev = nullptr;
    if (fin.done && fin.reason == MI355CG_STOP_INTERRUPTED) interrupted = true;      // the device saw the request in the middle of a chunk
    // Launches enqueued after the stop decision return in their prologue but still flipped c->cur on the
    // host: the direction of the last REAL iteration is p[it % M] (the solve starts with cur = 0).
    c->cur = fin.it % c->xsteps;
    if (cfg.x2) {                                   // folded x update: the steps after the last multiple of M are still pending
        launch_flush_x<double, 2>(c, c->whole, c->x, c->p, fin, c->stream);
        HIPCK(hipGetLastError());
        HIPCK(hipStreamSynchronize(c->stream));
    }
    if (cfg.has_u && fin.it > 0 && !need_u(fin.it)) {
        hipLaunchKernelGGL((k_err_maxnorm<double>), dim3(1024), dim3(kBlock), 0, c->stream, c->g.own_begin, c->g.own_len, c->x, c->u, c->partR);
        HIPCK(hipGetLastError());
        HIPCK(hipMemcpyAsync(c->partR_h, c->partR, sizeof(double) * 1024, hipMemcpyDeviceToHost, c->stream));
        HIPCK(hipStreamSynchronize(c->stream));
        double m = 0; for (int i = 0; i < 1024; ++i) m = std::max(m, c->partR_h[i]);
        fin.emax = m;
    }
    c->solved = true;
    prof_collect(c);
#ifdef MI355CG_WAVE_TIMING
    if (const char* path = getenv("MI355CG_WAVE_TIMING_OUT")) {     // diagnostic build: per-wave timestamps of the last launch of each kernel
        std::vector<unsigned long long> buf(2 * kWtStamps * kWtWaves);
        HIPCK(hipMemcpyFromSymbol(buf.data(), HIP_SYMBOL(g_wave_dbg), buf.size() * sizeof(unsigned long long)));
        if (FILE* f = fopen(path, "wb")) { fwrite(buf.data(), sizeof(unsigned long long), buf.size(), f); fclose(f); }
    }
#endif
    mi355cg_results res{};
    res.iterations = fin.it;
    res.converged = interrupted ? 0 : fin.converged;
    res.stop_reason = interrupted ? MI355CG_STOP_INTERRUPTED : fin.reason;
    res.final_residual_norm = fin.rmax;
    res.final_precision = fin.it > 0 ? fin.dmax : DBL_MAX;
    res.final_error_norm = cfg.has_u ? fin.emax : DBL_MAX;
    res.r_norm2 = fin.rnorm2;
    res.initial_r_norm2 = fin.r0norm;
    res.solve_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    { float ms = 0; HIPCK(hipEventSynchronize(c->ev_loop[1])); if (hipEventElapsedTime(&ms, c->ev_loop[0], c->ev_loop[1]) == hipSuccess) res.loop_seconds = 1e-3 * ms; }
    if (msg && cb) cb(user, res.iterations, res.final_precision, res.final_residual_norm, res.final_error_norm);   // msg_solver.cpp:193-195
    if (out) *out = res;
    return MI355CG_OK;
}

int mi355cg_get_solution(mi355cg_handle c, double* x) {
    if (!c || !x) return fail(MI355CG_ERR_INVALID, "null argument");
    if (!c->solved) return fail(MI355CG_ERR_STATE, "no solve has run on this handle");
    HIPCK(hipSetDevice(c->device));
    return download_packed<double>(c, c->x, x);
}
int mi355cg_get_recursive_residual(mi355cg_handle c, double* r) {
    if (!c || !r) return fail(MI355CG_ERR_INVALID, "null argument");
    if (!c->solved) return fail(MI355CG_ERR_STATE, "no solve has run on this handle");
    HIPCK(hipSetDevice(c->device));
    return download_packed<double>(c, c->r, r);
}
int mi355cg_get_true_residual(mi355cg_handle c, double* out) {
    if (!c || !out) return fail(MI355CG_ERR_INVALID, "null argument");
    if (c->is_slab) return fail(MI355CG_ERR_STATE, "this handle owns one part of a decomposed grid: use the mi355cg_team_* / mi355cg_dist_* entry points");
    if (!c->solved) return fail(MI355CG_ERR_STATE, "no solve has run on this handle");
    HIPCK(hipSetDevice(c->device));
    if (int rc = ensure_scratch(c)) return rc;
    // residual = A x - b   (dirichlet_solver.cpp:147-161)
    if (c->is_csr) launch_csr_spmv(c, c->x, c->scratch[0], nullptr, nullptr, nullptr);
    else launch_apply<double, 2>(c, c->x, c->scratch[0], whole_part(c));
    hipLaunchKernelGGL((k_sub<double>), dim3(flat_grid(c->g.own_len)), dim3(kBlock), 0, c->stream, c->g.own_begin, c->g.own_len, c->scratch[0], c->b, c->scratch[1]);
    HIPCK(hipGetLastError());
    return download_packed<double>(c, c->scratch[1], out);
}

int mi355cg_set_profiling(mi355cg_handle c, int enable) {
    if (!c) return fail(MI355CG_ERR_INVALID, "null handle");
    c->profiling = enable != 0;
    return MI355CG_OK;
}
int mi355cg_get_kernel_time(mi355cg_handle c, int kernel, double* avg_ms, long long* launches) {
    if (!c || kernel < 0 || kernel > 1) return fail(MI355CG_ERR_INVALID, "bad argument");
    if (avg_ms) *avg_ms = c->kernel_ms[kernel];
    if (launches) *launches = c->kernel_launches[kernel];
    return MI355CG_OK;
}
int mi355cg_get_layout(mi355cg_handle c, long long* padded_len, int* pitch_bottom, int* pitch_upper,
                       int* grid_stencil, int* grid_update, int* rows_per_item) {
    if (!c) return fail(MI355CG_ERR_INVALID, "null handle");
    if (padded_len) *padded_len = c->g.own_len;
    if (pitch_bottom) *pitch_bottom = c->g.Pb;
    if (pitch_upper) *pitch_upper = c->g.Pu;
    if (grid_stencil) *grid_stencil = c->whole.grid;
    if (grid_update) *grid_update = c->whole.grid;
    if (rows_per_item) *rows_per_item = c->whole.ty;
    return MI355CG_OK;
}

// ---- checksums (tests of large decomposed grids compare these instead of host copies of the vectors) -----------------
namespace {
struct hdd { double hi, lo; };
inline hdd hdd_add(hdd a, hdd b) {                  // same TwoSum-based addition as the device's dd_add (no contraction: -ffp-contract=off)
    const double s = a.hi + b.hi, bb = s - a.hi, e = (a.hi - (s - bb)) + (b.hi - bb);
    const double lo = e + (a.lo + b.lo), hi = s + lo;
    return hdd{hi, lo - (hi - s)};
}
// out[0] += sum of v, out[1] += sum of v^2 over the part's own cells, as double-double pairs.  Synchronises the part's stream.
int ctx_checksum(mi355cg_ctx* c, int which, hdd out[2]) {
    const double* v = which == 0 ? c->x : which == 1 ? c->r : which == 2 ? c->b : which == 3 ? c->u : nullptr;
    if (!v) return fail(MI355CG_ERR_INVALID, "checksum: vector %d (0 x, 1 r, 2 b, 3 u)", which);
    if (c->is_csr) return fail(MI355CG_ERR_INVALID, "checksum: grid handles only");
    HIPCK(hipSetDevice(c->device));
    if (c->is_slab) HIPCK(hipDeviceSynchronize());
    if (which == 3) if (int rc = ensure_u_on_device(c)) return rc;
    const int grid = 512;
    hipLaunchKernelGGL((k_checksum<double, 2>), dim3(grid), dim3(kBlock), 0, c->stream, kernel_geom<double, 2>(c), c->whole.wl, v, c->partR);
    HIPCK(hipGetLastError());
    HIPCK(hipMemcpyAsync(c->partR_h, c->partR, sizeof(double) * 4 * grid, hipMemcpyDeviceToHost, c->stream));
    HIPCK(hipStreamSynchronize(c->stream));
    for (int b = 0; b < grid; ++b) {
        out[0] = hdd_add(out[0], hdd{c->partR_h[b], c->partR_h[grid + b]});
        out[1] = hdd_add(out[1], hdd{c->partR_h[2 * grid + b], c->partR_h[3 * grid + b]});
    }
    return MI355CG_OK;
}
}  // namespace

int mi355cg_checksum(mi355cg_handle c, int which, double* out2) {
    if (!c || !out2) return fail(MI355CG_ERR_INVALID, "null argument");
    hdd s[2] = {{0, 0}, {0, 0}};
    if (int rc = ctx_checksum(c, which, s)) return rc;
    out2[0] = s[0].hi + s[0].lo; out2[1] = s[1].hi + s[1].lo;
    return MI355CG_OK;
}

// ---- slab (one rank of a row-decomposed grid), driven phase by phase by the caller (iterative_solvers_amd/distributed.py) ----
int mi355cg_slab_rows(int n, int world, int rank, int* y_lo, int* y_hi) {
    GridParams gp;
    if (!grid_params_init(&gp, n, n, 0, 1, 0, 1)) return fail(MI355CG_ERR_INVALID, "grid %d rejected", n);
    if (world < 1 || rank < 0 || rank >= world || world > n - 1) return fail(MI355CG_ERR_INVALID, "bad world/rank %d/%d", rank, world);
    // contiguous row slabs balanced by unknown count (bottom rows hold n/2-1 unknowns, upper rows n-1)
    std::vector<int> first(world + 1);
    first[0] = 1; first[world] = n;
    for (int k = 1; k < world; ++k) {                         // smallest y with (unknowns in rows 1..y-1) >= k*U/world
        const long long target = (gp.size * k + world / 2) / world;
        int lo = 1, hi = n;
        while (lo < hi) { const int mid = (lo + hi) / 2; if (packed_row_begin(gp, mid) >= target) hi = mid; else lo = mid + 1; }
        first[k] = lo;
    }
    for (int k = 1; k < world; ++k) first[k] = std::max(first[k], first[k - 1] + 1);        // every rank non-empty
    for (int k = world - 1; k >= 1; --k) first[k] = std::min(first[k], first[k + 1] - 1);
    const int a = first[rank], b = first[rank + 1] - 1;
    if (y_lo) *y_lo = a;
    if (y_hi) *y_hi = b;
    return MI355CG_OK;
}

int mi355cg_owned_range(mi355cg_handle c, long long* packed_begin, long long* packed_len, int* y_lo, int* y_hi) {
    if (!c) return fail(MI355CG_ERR_INVALID, "null handle");
    if (packed_begin) *packed_begin = c->pk_begin;
    if (packed_len) *packed_len = c->pk_len;
    if (y_lo) *y_lo = c->g.y_lo;
    if (y_hi) *y_hi = c->g.y_hi;
    return MI355CG_OK;
}

// The dist entry points enqueue on the caller's stream, taken literally (NULL = HIP's default stream,
// which is also torch's default stream), so they order with the caller's collectives and copies.
static hipStream_t pick_stream(mi355cg_ctx*, void* stream) { return (hipStream_t)stream; }
// slots of a part's split launches: interior first, then edge
static Where part_where(const mi355cg_ctx* c, hipStream_t st, int rows) {
    if (rows == 1) return Where{st, &c->interior, 0};
    if (rows == 2) return Where{st, &c->edge, c->interior.grid};
    return Where{st, &c->whole, 0};
}
static int part_slots(const mi355cg_ctx* c, int rows) { return rows == 0 ? c->whole.grid : c->interior.grid + c->edge.grid; }

int mi355cg_dist_begin(mi355cg_handle c, const mi355cg_params* prm, void* stream) {
    if (!c || !prm) return fail(MI355CG_ERR_INVALID, "null argument");
    if (c->dtype != MI355CG_F64) return fail(MI355CG_ERR_INVALID, "slab mode is fp64 only");
    if (prm->diagnostics) return fail(MI355CG_ERR_INVALID, "per-iteration diagnostics are not available in slab mode");
    HIPCK(hipSetDevice(c->device));
    c->dist_prm = *prm; c->dist_active = true;
    const IterCfg cfg = make_cfg(prm);
    if (cfg.has_u) if (int rc = ensure_u_on_device(c)) return rc;
    hipStream_t st = pick_stream(c, stream);
    const size_t bytes = sizeof(double) * c->storage_len;
    HIPCK(hipMemsetAsync(c->x, 0, bytes, st));
    for (int k = 0; k < c->xsteps; ++k) HIPCK(hipMemsetAsync(c->p[k], 0, bytes, st));
    HIPCK(hipMemsetAsync(c->ap, 0, bytes, st));
    HIPCK(hipMemcpyAsync(c->r, c->b, bytes, hipMemcpyDeviceToDevice, st));
    c->cur = 0;
    c->nB_own = c->whole.grid;
    launch_update_flat<double, 2>(c, cfg, c->x, c->r, c->p[0], c->ap, c->u, st, c->whole.grid);
    HIPCK(hipGetLastError());
    c->solved = true;
    return MI355CG_OK;
}

// which: 0 = stencil partials (fields FA_*), 1 = update partials (fields FB_*).  Reduces this rank's
// partials into the head of its record (device).  with_rows != 0 also copies the rank's first and
// last owned row of the vector the neighbours need next (which 0: the direction the stencil just
// wrote, call after mi355cg_dist_flip; which 1: the residual) behind the sums, so ONE all-gather
// carries both the scalars and the halo.
int mi355cg_dist_reduce(mi355cg_handle c, int which, int with_rows, void* stream) {
    if (!c || !c->dist_active) return fail(MI355CG_ERR_STATE, "mi355cg_dist_begin has not run");
    hipStream_t st = pick_stream(c, stream);
    const Geom& g = c->g;
    RecordArgs a{};
    a.rec = which == 0 ? c->sumsA : c->sumsB; a.header = kRecHeader; a.row_slot = g.Pu;
    if (which == 0) { a.part = c->partA; a.n = c->nA_dist; a.stride = c->strideA; a.nsum = kNumSumsA; a.lo_off = FA_LO; a.max_first = 0; a.nmax = 0; }
    else { a.part = c->partB; a.n = c->nB_own; a.stride = c->strideB; a.nsum = kNumSumsB; a.lo_off = FB_LO; a.max_first = FB_RMAX; a.nmax = 3; }
    if (with_rows) {
        a.v = which == 0 ? c->p[c->cur] : c->r;
        a.off_lo = phys_start(g, g.y_lo) - g.base0; a.len_lo = g.y_lo <= g.half ? g.Pb : g.Pu;
        a.off_hi = phys_start(g, g.y_hi) - g.base0; a.len_hi = g.y_hi <= g.half ? g.Pb : g.Pu;
    }
    hipLaunchKernelGGL(k_make_record, dim3(with_rows ? 1 + 32 : 1), dim3(kBlock), 0, st, a);
    HIPCK(hipGetLastError());
    return MI355CG_OK;
}
int mi355cg_dist_record_layout(mi355cg_handle c, int* header, int* row_slot, int* width) {
    if (!c) return fail(MI355CG_ERR_INVALID, "null handle");
    if (header) *header = kRecHeader;
    if (row_slot) *row_slot = c->g.Pu;
    if (width) *width = c->rec_width;
    return MI355CG_OK;
}
int mi355cg_dist_sums_ptr(mi355cg_handle c, int which, void** dev_ptr, int* count) {
    if (!c) return fail(MI355CG_ERR_INVALID, "null handle");
    if (dev_ptr) *dev_ptr = which == 0 ? (void*)c->sumsA : (void*)c->sumsB;
    if (count) *count = c->rec_width;
    return MI355CG_OK;
}
// After all-gathering full records ([rank][rec_width]): copy the neighbours' boundary rows into this
// rank's ghost rows of `vector` (0 = r from update records, 1 = current direction from stencil records).
int mi355cg_dist_scatter_ghosts(mi355cg_handle c, int vector, const double* gathered, int nranks, int rank, void* stream) {
    if (!c || !gathered) return fail(MI355CG_ERR_INVALID, "null argument");
    if (rank < 0 || rank >= nranks) return fail(MI355CG_ERR_INVALID, "rank %d outside 0..%d", rank, nranks - 1);
    hipStream_t st = pick_stream(c, stream);
    const Geom& g = c->g;
    double* v = vector == 0 ? c->r : c->p[c->cur];
    const size_t W = (size_t)c->rec_width;
    ScatterArgs a{};
    if (rank > 0) {            // ghost row y_lo-1 = the lower neighbour's LAST owned row
        a.src_lo = gathered + W * (rank - 1) + kRecHeader + g.Pu;
        a.dst_lo = v + (phys_start(g, g.y_lo - 1) - g.base0);
        a.len_lo = (g.y_lo - 1) <= g.half ? g.Pb : g.Pu;
    }
    if (rank < nranks - 1) {   // ghost row y_hi+1 = the upper neighbour's FIRST owned row
        a.src_hi = gathered + W * (rank + 1) + kRecHeader;
        a.dst_hi = v + (phys_start(g, g.y_hi + 1) - g.base0);
        a.len_hi = (g.y_hi + 1) <= g.half ? g.Pb : g.Pu;
    }
    if (a.len_lo || a.len_hi) hipLaunchKernelGGL(k_scatter_ghosts, dim3(32), dim3(kBlock), 0, st, a);
    HIPCK(hipGetLastError());
    return MI355CG_OK;
}

// rows: 0 = whole slab, 1 = interior rows only, 2 = the first and last owned row (they read the
// neighbours' ghost rows).  A full stencil phase is either {0} or {1, 2}; call mi355cg_dist_flip once after it.
// `estride` = doubles between consecutive ranks' sums in `gathered_B` (FB_COUNT, or the record width).
int mi355cg_dist_stencil(mi355cg_handle c, const double* gathered_B, int nranks, int estride, int rows, void* stream) {
    if (!c || !c->dist_active || !gathered_B) return fail(MI355CG_ERR_STATE, "mi355cg_dist_begin has not run / null partials");
    const IterCfg cfg = make_cfg(&c->dist_prm);
    const PartSrc pb{gathered_B, nranks, 1, estride};
    launch_iteration_stencil<double, 2>(c, cfg, c->r, c->p, part_where(c, pick_stream(c, stream), rows), pb);
    c->nA_dist = part_slots(c, rows);
    HIPCK(hipGetLastError());
    return MI355CG_OK;
}
int mi355cg_dist_flip(mi355cg_handle c) {
    if (!c) return fail(MI355CG_ERR_INVALID, "null handle");
    c->cur = (c->cur + 1) % c->xsteps;
    return MI355CG_OK;
}
// rows: as in mi355cg_dist_stencil.  The update rebuilds A p from the stored direction, so its first and last owned row
// read the direction's ghost rows, which the stencil launch keeps up to date itself; a full update phase is {0} or {1, 2}.
int mi355cg_dist_update(mi355cg_handle c, const double* gathered_A, int nranks, int estride, int rows, void* stream) {
    if (!c || !c->dist_active || !gathered_A) return fail(MI355CG_ERR_STATE, "mi355cg_dist_begin has not run / null partials");
    IterCfg cfg = make_cfg(&c->dist_prm);
    const PartSrc pa{gathered_A, nranks, 1, estride};
    c->nB_own = part_slots(c, rows);
    launch_iteration_update<double, 2>(c, cfg, c->x, c->r, c->p, c->u, part_where(c, pick_stream(c, stream), rows), pa);
    HIPCK(hipGetLastError());
    return MI355CG_OK;
}
// Non-zero if the caller has to deliver the direction's ghost rows before the update's edge rows run.  Always 0: the
// stencil launch of a part keeps the new direction in its ghost rows itself (StencilArgs::store_ghosts).
int mi355cg_dist_update_reads_ghosts(mi355cg_handle c) {
    (void)c;
    return 0;
}
// Asynchronous: after the stream reaches this point the summary is in pinned host memory.
int mi355cg_dist_check(mi355cg_handle c, const double* gathered_B, int nranks, int estride, void* stream) {
    if (!c || !c->dist_active || !gathered_B) return fail(MI355CG_ERR_STATE, "mi355cg_dist_begin has not run / null partials");
    const IterCfg cfg = make_cfg(&c->dist_prm);
    hipStream_t st = pick_stream(c, stream);
    launch_check(c, cfg, st, PartSrc{gathered_B, nranks, 1, estride});
    HIPCK(hipMemcpyAsync(c->summary_h, c->summary, sizeof(CgState), hipMemcpyDeviceToHost, st));
    HIPCK(hipMemcpyAsync(c->hist_h, c->hist, sizeof(HistEntry) * kHist, hipMemcpyDeviceToHost, st));
    return MI355CG_OK;
}
// Call once after the loop (after the last mi355cg_dist_check has been synchronised): applies the x update
// that is still pending after an odd iteration count (REL_2NORM).  No-op otherwise.
int mi355cg_dist_finish(mi355cg_handle c, void* stream) {
    if (!c || !c->dist_active) return fail(MI355CG_ERR_STATE, "mi355cg_dist_begin has not run");
    const IterCfg cfg = make_cfg(&c->dist_prm);
    const CgState fin = *c->summary_h;
    c->cur = fin.it % c->xsteps;         // launches after the stop decision were no-ops but advanced the host-side index
    if (cfg.x2) {
        launch_flush_x<double, 2>(c, c->whole, c->x, c->p, fin, pick_stream(c, stream));
        HIPCK(hipGetLastError());
    }
    return MI355CG_OK;
}
int mi355cg_dist_summary(mi355cg_handle c, mi355cg_results* out, int* done) {
    if (!c || !out) return fail(MI355CG_ERR_INVALID, "null argument");
    const CgState fin = *c->summary_h;
    const IterCfg cfg = make_cfg(&c->dist_prm);
    mi355cg_results res{};
    res.iterations = fin.it; res.converged = fin.converged; res.stop_reason = fin.reason;
    res.final_residual_norm = fin.rmax;
    res.final_precision = fin.it > 0 ? fin.dmax : DBL_MAX;
    res.final_error_norm = cfg.has_u ? fin.emax : DBL_MAX;
    res.r_norm2 = fin.rnorm2; res.initial_r_norm2 = fin.r0norm;
    *out = res;
    if (done) *done = fin.done;
    return MI355CG_OK;
}
int mi355cg_dist_history(mi355cg_handle c, int iteration, double* precision, double* residual, double* error) {
    if (!c) return fail(MI355CG_ERR_INVALID, "null handle");
    const HistEntry& h = c->hist_h[((iteration % kHist) + kHist) % kHist];
    if (precision) *precision = h.dmax;
    if (residual) *residual = h.rmax;
    if (error) *error = h.emax;
    return MI355CG_OK;
}
// Boundary rows for the halo exchange.  vector: 0 = residual r, 1 = the CURRENT direction (the one
// the last stencil wrote).  *_lo: row y_lo (send) and ghost row y_lo-1 (recv); *_hi: row y_hi and
// ghost row y_hi+1.  Counts are in doubles (the full stored row, pads included).
int mi355cg_dist_halo(mi355cg_handle c, int vector, void** send_lo, void** recv_lo, long long* n_lo,
                      void** send_hi, void** recv_hi, long long* n_hi) {
    if (!c) return fail(MI355CG_ERR_INVALID, "null handle");
    double* v = vector == 0 ? c->r : c->p[c->cur];
    const Geom& g = c->g;
    auto row_ptr = [&](int y) { return v + (phys_start(g, y) - g.base0); };
    auto row_len = [&](int y) -> long long { return y <= g.half ? g.Pb : g.Pu; };
    if (send_lo) *send_lo = row_ptr(g.y_lo);
    if (recv_lo) *recv_lo = row_ptr(g.y_lo - 1);
    if (send_hi) *send_hi = row_ptr(g.y_hi);
    if (recv_hi) *recv_hi = row_ptr(g.y_hi + 1);
    // a message is the SENDER's row; the receiver's ghost row has the same global index, hence the same length
    if (n_lo) *n_lo = row_len(g.y_lo);          // what this rank sends down; it receives row_len(y_lo-1)
    if (n_hi) *n_hi = row_len(g.y_hi);          // what this rank sends up;   it receives row_len(y_hi+1)
    return MI355CG_OK;
}
int mi355cg_dist_halo_recv_counts(mi355cg_handle c, long long* n_from_lo, long long* n_from_hi) {
    if (!c) return fail(MI355CG_ERR_INVALID, "null handle");
    const Geom& g = c->g;
    if (n_from_lo) *n_from_lo = (g.y_lo - 1) <= g.half ? g.Pb : g.Pu;
    if (n_from_hi) *n_from_hi = (g.y_hi + 1) <= g.half ? g.Pb : g.Pu;
    return MI355CG_OK;
}

}  // extern "C"

#include "team.h"
