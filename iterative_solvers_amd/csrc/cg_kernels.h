// cg_kernels.h -- hand-written HIP kernels (gfx950 / CDNA4, wave64) for the matrix-free CG path.
//
//   k_stencil   (phase A')  p_new = r + beta*p_old (fused, halo recomputed), A_h p_new evaluated in registers and
//                           reduced into (Ap,p) [and (r,p)]; A p itself is not stored on the default path.  Replaces
//                           MatrixFreeSystem::apply (matrix_free_system.cpp:203-340) / KokkosSparse::spmv
//                           (msg_solver.cpp:93), the direction update (matrix_free_system.cpp:436-438,
//                           msg_solver.cpp:167-169) and the two dots (matrix_free_system.cpp:417, msg_solver.cpp:96,99).
//   k_update_st (phase B)   A p rebuilt from three rows of the stored direction, r -= alpha Ap, x update (every fourth
//                           iteration, four steps at once; every iteration for the MSG rule), partial sums / maxes of
//                           r.r, |r|, |dx|, |x-u|.  Replaces matrix_free_system.cpp:422-455 / msg_solver.cpp:105-139.
//   k_update                flat variant: state initialisation, resume step of the mixed-precision path, CSR handles.
//   k_check, k_flush_x, k_make_record, k_scatter_ghosts, k_pack/k_unpack, k_sub, k_resid2, ...: small helpers.
//
// Bandwidth-bound: no MFMA.  All arithmetic that the reference does element-wise is done in the reference's
// operation order without FMA contraction (-ffp-contract=off), so vectors are bit-identical to the CPU oracle for
// equal scalars; only the inner products differ: they are accumulated as double-double pairs (see dd below).
#pragma once
#include <hip/hip_runtime.h>
#include <cfloat>
#include <cmath>

namespace mi355cg {

constexpr int kBlock = 256;           // threads per workgroup = 4 wave64
constexpr int kWave = 64;
constexpr int kWaves = kBlock / kWave;
constexpr int kHist = 512;            // per-iteration norm history ring (>= sync_every)
constexpr int kMaxPanels = 8;
constexpr int kRing = 8;              // direction buffers of a context (ring; M = xsteps <= kRing of them are in use); also the alpha history depth

// ---- per-wave timing probe (diagnostic build only: -DMI355CG_WAVE_TIMING, tools/wave_timing.py) -------------------
// Every wave of the two iteration kernels records wall_clock64() (100 MHz) at entry, after the prologue and at exit.
#ifdef MI355CG_WAVE_TIMING
constexpr int kWtWaves = 16384, kWtStamps = 6;
__device__ unsigned long long g_wave_dbg[2][kWtStamps * kWtWaves];
#define MI355CG_WT_BEGIN unsigned long long wt_[kWtStamps]; for (int i_ = 0; i_ < kWtStamps; ++i_) wt_[i_] = 0; wt_[0] = wall_clock64();
#define MI355CG_WT_STAMP(I) wt_[I] = wall_clock64();
#define MI355CG_WT_MID wt_[kWtStamps - 2] = wall_clock64();
#define MI355CG_WT_END(K) do { if ((threadIdx.x & 63) == 0) { const int w_ = blockIdx.x * 4 + threadIdx.x / 64; \
    if (w_ < kWtWaves) { wt_[kWtStamps - 1] = wall_clock64(); for (int i_ = 0; i_ < kWtStamps; ++i_) g_wave_dbg[K][kWtStamps * w_ + i_] = wt_[i_]; } } } while (0)
#else
#define MI355CG_WT_BEGIN
#define MI355CG_WT_STAMP(I)
#define MI355CG_WT_MID
#define MI355CG_WT_END(K)
#endif

// ---- storage layout -------------------------------------------------------------------------------
// Node (x, y) of the (N+1)x(N+1) bounding grid lives at  row_off(y) + x - base0.  Rows y <= N/2
// (bottom-right block + its boundary row 0) only store columns [cb, cb+Pb), cb = N/2 rounded down
// to 32; rows above store columns [0, Pu).  Pb, Pu, cb are multiples of 32 elements so every row
// starts 256-B aligned.  Boundary nodes and pads hold 0 and stay 0; Dirichlet data is in the RHS.
struct Geom {
    int N, half, cb, xlim;            // xlim: columns [0, xlim) are touched by the stencil strips
    int Pb, Pu;                       // row pitches (elements) of bottom / upper rows
    int y_lo, y_hi;                   // owned rows (inclusive); rows y_lo-1 and y_hi+1 are ghosts
    long long base0;                  // physical offset of the first stored element (row y_lo-1)
    long long own_begin, own_len;     // flat [begin, begin+len) of the owned rows, local offsets
    double A, xk, yk;                 // stencil coefficients (grid_system.cpp:316-318)
};

__host__ __device__ inline long long row_off(const Geom& g, int y) {
    return y <= g.half ? (long long)y * g.Pb - g.cb
                       : (long long)(g.half + 1) * g.Pb + (long long)(y - g.half - 1) * g.Pu;
}
__host__ __device__ inline bool node_interior(const Geom& g, int x, int y) {
    return y >= 1 && y <= g.N - 1 && x <= g.N - 1 && x >= (y <= g.half ? g.half + 1 : 1);
}

// A panel is a rectangle of owned rows x column strips, cut into row chunks; one (chunk, strip)
// pair is one work item = one wave marching `ty` rows of a 64*VEC-column strip.
// gc (2-D decomposition): bit 0 = the column left of the panel's first strip belongs to another part (a ghost
// column of this part), bit 1 = the same on the right of its last strip.
struct Panel { int y0, y1, s0, ns, ty, nchunks, item0, gc; };
// XCD classes: workgroups land on XCD blockIdx % 8, and every XCD has its own L2.  With ncls == 8 the items are cut into
// eight contiguous ranges [cls0[k], cls0[k+1]) -- whole bands of chunk rows -- and range k is served by the workgroups with
// blockIdx % 8 == k only: the strips left and right of a workgroup (whose edge columns it reads) and the chunk rows above and
// below (whose halo rows it reads) are then work of the SAME XCD and those reads hit its L2 instead of going to the fabric
// (PMC: read traffic of the stencil launch 1.10 x -> see profiles/r02_tune_notes.md).  A wrong guess about the placement costs
// those hits, never correctness.
constexpr int kXcds = 8;
struct WorkList { Panel p[kMaxPanels]; int np; int nitems; int ncls; int cls0[kXcds + 1]; };
// the item indices a wave takes: first, first + step, ... < end
struct ItemSeq { int first, step, begin, end; };
__device__ inline ItemSeq item_seq(const WorkList& wl, int wave) {
    if (wl.ncls == kXcds) {
        const int cls = blockIdx.x % kXcds, nb = (gridDim.x - cls + kXcds - 1) / kXcds;
        int begin = wl.cls0[0], end = wl.cls0[1];
#pragma unroll
        for (int k = 1; k < kXcds; ++k) if (k == cls) { begin = wl.cls0[k]; end = wl.cls0[k + 1]; }     // constant indices only
        return ItemSeq{begin + (int)(blockIdx.x / kXcds) * kWaves + wave, nb * kWaves, begin, end};
    }
    return ItemSeq{(int)blockIdx.x * kWaves + wave, (int)gridDim.x * kWaves, 0, wl.nitems};
}

// ---- CG state carried on the device ----------------------------------------------------------------
struct CgState {
    double alpha, beta;
    double rr;          // (r, r) of the current residual
    double rr_prev;     // (r, r) one decision earlier (lets a stopped solve resume with the right beta denominator)
    double rz;          // MSG: (r, z) of the current iteration (denominator of the next beta)
    double r0norm;      // ||r0||_2
    double rnorm2;      // ||r||_2
    double rmax, dmax, emax, d2, e2;
    int it, done, reason, converged, first;
    int stop;           // a stop request was pending when the last update launch ended (single context: read from the pinned word; msg_solver.cpp:82-87)
    double alpha_hist[kRing];   // step length of iteration k at [k % kRing]: the folded x update (XM >= 2) applies up to kRing - 1 earlier steps at once
};
struct HistEntry { double dmax, rmax, emax, rnorm2, d2, e2, tr2; };   // tr2: ||b - A x||_2^2 (REL_2NORM diagnostics mode, written by k_resid2_hist)

// What every wave needs from the state, fetched with SCALAR loads (s_load through the constant address space, one
// request per scalar cache instead of one per wave).  Copying the whole struct made hipcc fetch half of it with
// per-lane global loads of one address: ~4000 waves x 3 loads of the same line queued at one L2 channel and the
// median wave waited 10-14 us of a 65 us launch for its copy of the state (tools/wave_timing.py, profiles/r01_tune_notes.md).
// Safe because no kernel writes the state object it reads (s_in != s_out) and the scalar cache is invalidated at
// every kernel start.
struct StateLite { double alpha, rr, rr_prev, rz, r0norm; int it, done, first, stop; };
template <typename V> __device__ inline V scalar_load(const V* p) {
    return *(const __attribute__((address_space(4))) V*)p;
}
// State object -> state object, by the one thread of the grid that forwards it.  Not inlined: hipcc otherwise hoists
// the loads in front of the branch that selects that thread, and every wave of the grid fetches all 128 bytes.
__device__ __attribute__((noinline)) void copy_state(CgState* dst, const CgState* src) { *dst = *src; }
__device__ inline StateLite load_state_lite(const CgState* p) {
    StateLite L;
    L.alpha = scalar_load(&p->alpha); L.rr = scalar_load(&p->rr); L.rr_prev = scalar_load(&p->rr_prev);
    L.rz = scalar_load(&p->rz); L.r0norm = scalar_load(&p->r0norm);
    L.it = scalar_load(&p->it); L.done = scalar_load(&p->done); L.first = scalar_load(&p->first); L.stop = scalar_load(&p->stop);
    return L;
}

struct RuleParams {
    int rule;                 // MI355CG_RULE_*
    int max_iterations;
    double eps_precision, eps_residual, eps_exact_error, eps_rel;
    int use_u;
    int fixed_iterations;
};

// partial-sum fields, field-major: part[field * stride + block]
// Sums are carried as double-double pairs (hi at the field, lo at field + *_LO): see dd below.
enum { FA_PAP = 0, FA_RZ = 1, FA_LO = 2, FA_COUNT = 4 };
enum { FB_RR = 0, FB_D2 = 1, FB_E2 = 2, FB_LO = 3, FB_RMAX = 6, FB_DMAX = 7, FB_EMAX = 8, FB_COUNT = 9 };
constexpr int kNumSumsA = 2, kNumSumsB = 3;          // sum fields come first, then their lo words, then the max fields
constexpr int FB_STOP = FB_COUNT, FB_LL_COUNT = FB_COUNT + 1;   // flagged update partials carry one more field: the stop request the launch sampled (its block 0; 0 elsewhere)

// ---- double-double accumulation of the inner products ------------------------------------------------
// Every inner product is accumulated as an unevaluated pair hi + lo (Knuth TwoSum, FMA TwoProduct): the
// result carries ~100 significant bits, so after the final rounding to double it no longer depends on the
// order in which lanes, waves, blocks or GPUs were combined (two orders can differ only if the exact sum
// sits within ~1e-30 of a rounding boundary).  That is what makes 1-, 2-, 4- and 8-GPU runs take bit-identical
// steps.  It costs VALU work only -- the kernels are an order of magnitude below the VALU roof.
struct dd { double hi, lo; };
__device__ inline dd dd_zero() { return dd{0.0, 0.0}; }
__device__ inline dd two_sum(double a, double b) { const double s = a + b, bb = s - a; return dd{s, (a - (s - bb)) + (b - bb)}; }
__device__ inline dd dd_add(dd a, dd b) {
    const dd s = two_sum(a.hi, b.hi);
    const double lo = s.lo + (a.lo + b.lo);
    const double hi = s.hi + lo;
    return dd{hi, lo - (hi - s.hi)};
}
__device__ inline void dd_acc_prod(dd& acc, double a, double b) {          // acc += a*b, product exact
#ifdef MI355CG_PLAIN_DOT                                                   // A/B build only: plain double accumulation
    acc.hi += a * b;
#elif defined(MI355CG_DOT2)                                                // A/B build: Ogita-Rump-Oishi Dot2 (no renormalisation per step)
    const double p = a * b;
    const dd s = two_sum(acc.hi, p);
    acc.hi = s.hi;
    acc.lo += s.lo + fma(a, b, -p);
#else
    const double p = a * b;
    acc = dd_add(acc, dd{p, fma(a, b, -p)});
#endif
}
__device__ inline double dd_value(dd a) { return a.hi + a.lo; }

// ---- block-level deterministic reductions ----------------------------------------------------------
__device__ inline double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, kWave);
    return v;
}
__device__ inline double wave_max(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_down(v, o, kWave));
    return v;
}
// All threads get the block total.  Fixed tree: lane tree inside a wave, then waves 0..3 in order.
template <bool IS_MAX>
__device__ inline double block_reduce(double v, double* lds /* >= kWaves doubles */) {
    v = IS_MAX ? wave_max(v) : wave_sum(v);
    const int lane = threadIdx.x & (kWave - 1), w = threadIdx.x / kWave;
    __syncthreads();                       // protect lds from the previous use
    if (lane == 0) lds[w] = v;
    __syncthreads();
    double t = lds[0];
#pragma unroll
    for (int k = 1; k < kWaves; ++k) t = IS_MAX ? fmax(t, lds[k]) : t + lds[k];
    return t;
}
// Reduce `n` partials of one field (thread t takes t, t+256, ... in ascending order).
// `es` = element stride: 1 for a kernel's own field-major partials, the field count for partials
// all-gathered rank-major across GPUs ([rank][field]).
template <bool IS_MAX>
__device__ inline double reduce_parts(const double* __restrict__ part, int n, int es, double* lds) {
    double v = 0.0;
    for (int i = threadIdx.x; i < n; i += kBlock) v = IS_MAX ? fmax(v, part[(long long)i * es]) : v + part[(long long)i * es];
    return block_reduce<IS_MAX>(v, lds);
}

__device__ inline dd wave_sum_dd(dd v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = dd_add(v, dd{__shfl_down(v.hi, o, kWave), __shfl_down(v.lo, o, kWave)});
    return v;
}
// lds: >= 2*kWaves doubles
__device__ inline dd block_reduce_dd(dd v, double* lds) {
    v = wave_sum_dd(v);
    const int lane = threadIdx.x & (kWave - 1), w = threadIdx.x / kWave;
    __syncthreads();
    if (lane == 0) { lds[2 * w] = v.hi; lds[2 * w + 1] = v.lo; }
    __syncthreads();
    dd t{lds[0], lds[1]};
#pragma unroll
    for (int k = 1; k < kWaves; ++k) t = dd_add(t, dd{lds[2 * k], lds[2 * k + 1]});
    return t;
}
// The first two partial pairs of a thread (i = t, t + 256), loaded EARLY: the consumer kernels issue these loads before they
// wait for the CG state (which says whether the solve is already done), so the two dependent ~1.5 us round trips of the
// prologue -- state, then partials -- overlap.  reduce_parts_dd_pre consumes them in exactly reduce_parts_dd's order.
struct PreParts { double hi0, lo0, hi1, lo1; };
__device__ inline PreParts prefetch_parts(const double* __restrict__ part_hi, const double* __restrict__ part_lo, int n, int es) {
    PreParts p{0.0, 0.0, 0.0, 0.0};
    const int i0 = threadIdx.x, i1 = threadIdx.x + kBlock;
    if (i0 < n) { p.hi0 = part_hi[(long long)i0 * es]; p.lo0 = part_lo[(long long)i0 * es]; }
    if (i1 < n) { p.hi1 = part_hi[(long long)i1 * es]; p.lo1 = part_lo[(long long)i1 * es]; }
    return p;
}
__device__ inline dd reduce_parts_dd_pre(const PreParts& pre, const double* __restrict__ part_hi, const double* __restrict__ part_lo, int n, int es, double* lds) {
    dd v = dd_zero();
    if ((int)threadIdx.x < n) v = dd_add(v, dd{pre.hi0, pre.lo0});
    if ((int)threadIdx.x + kBlock < n) v = dd_add(v, dd{pre.hi1, pre.lo1});
    for (int i = threadIdx.x + 2 * kBlock; i < n; i += kBlock) v = dd_add(v, dd{part_hi[(long long)i * es], part_lo[(long long)i * es]});
    return block_reduce_dd(v, lds);
}
// the same for one max field
struct PreMax { double v0, v1; };
__device__ inline PreMax prefetch_max(const double* __restrict__ part, int n, int es) {
    PreMax p{0.0, 0.0};
    if ((int)threadIdx.x < n) p.v0 = part[(long long)threadIdx.x * es];
    if ((int)threadIdx.x + kBlock < n) p.v1 = part[(long long)(threadIdx.x + kBlock) * es];
    return p;
}
template <bool IS_MAX> __device__ inline double block_reduce(double v, double* lds);
__device__ inline double reduce_max_pre(const PreMax& pre, const double* __restrict__ part, int n, int es, double* lds) {
    double v = fmax(pre.v0, pre.v1);
    for (int i = threadIdx.x + 2 * kBlock; i < n; i += kBlock) v = fmax(v, part[(long long)i * es]);
    return block_reduce<true>(v, lds);
}
// partial pairs: hi words at part_hi[i*es], lo words at part_lo[i*es]
__device__ inline dd reduce_parts_dd(const double* __restrict__ part_hi, const double* __restrict__ part_lo, int n, int es, double* lds) {
    dd v = dd_zero();
    for (int i = threadIdx.x; i < n; i += kBlock) v = dd_add(v, dd{part_hi[(long long)i * es], part_lo[(long long)i * es]});
    return block_reduce_dd(v, lds);
}

// ---- the decision taken after every update: convergence tests, beta ------------------------------
// Mirrors msg_solver.cpp:144-165 / matrix_free_system.cpp:409,432-433,441,472.  Every block
// evaluates it from the same reduced numbers, so all blocks agree.
struct Decision { int done, reason, converged; double beta, rr, rnorm2, r0norm, rmax, dmax, emax, d2, e2; };

// `stop`: a stop request is pending (this context's pinned word as sampled by the last update launch, or the max over the
// parts' records of a team).  It is looked at AFTER the convergence tests of the iteration just finished, i.e. where the
// reference's loop tests its flag: at the top of the next iteration (msg_solver.cpp:82-87).
__device__ inline Decision decide_after_update(const StateLite& s, const RuleParams& rp, double rr, double rmax,
                                               double dmax, double emax, double d2, double e2, bool stop = false) {
    Decision d;
    d.rr = rr; d.rnorm2 = sqrt(rr); d.rmax = rmax; d.dmax = dmax; d.emax = emax; d.d2 = d2; d.e2 = e2;
    d.r0norm = s.first ? d.rnorm2 : s.r0norm;
    d.done = 0; d.reason = 0 /*ITERATIONS*/; d.converged = 0; d.beta = 0.0;
    if (rp.rule == 1 /*REL_2NORM*/) {
        // for (...; iterations < maxIterations && r_norm > eps * initial_r_norm; ...)  :409
        const bool go = s.it < rp.max_iterations && (rp.fixed_iterations || d.rnorm2 > rp.eps_rel * d.r0norm);
        if (!go) { d.done = 1; d.converged = d.rnorm2 <= rp.eps_rel * d.r0norm; }     // :472
        if (!s.first) d.beta = rr / s.rr;                                               // :432-433
    } else {
        if (s.it >= 1 && !rp.fixed_iterations) {
            if (rp.eps_precision > 0 && dmax < rp.eps_precision) { d.done = 1; d.converged = 1; d.reason = 1; }
            else if (rp.eps_residual > 0 && rmax < rp.eps_residual) { d.done = 1; d.converged = 1; d.reason = 2; }
            else if (rp.eps_exact_error > 0 && rp.use_u && emax < rp.eps_exact_error) { d.done = 1; d.converged = 1; d.reason = 3; }
        }
        if (!d.done && !(s.it < rp.max_iterations)) d.done = 1;                         // while (it < maxIterations) :80
        if (!s.first) d.beta = (d.rnorm2 * d.rnorm2) / s.rz;                            // :165
    }
    if (!d.done && (stop || s.stop)) { d.done = 1; d.reason = 4 /*INTERRUPTED*/; d.converged = 0; }   // msg_solver.cpp:82-87
    return d;
}

// One thread of the grid: the full state object travels through this thread only.
__device__ inline void write_state_after_decision(CgState* out, HistEntry* hist, const CgState* in, const StateLite& s, const Decision& d) {
    copy_state(out, in);
    CgState* o = out;
    o->rr_prev = s.rr; o->rr = d.rr; o->rnorm2 = d.rnorm2; o->r0norm = d.r0norm; o->beta = d.beta;
    o->rmax = d.rmax; o->dmax = d.dmax; o->emax = d.emax; o->d2 = d.d2; o->e2 = d.e2;
    o->done = d.done; o->reason = d.reason; o->converged = d.converged;
    if (hist) {
        HistEntry* h = hist + (s.it % kHist);      // field by field: tr2 of this entry belongs to k_resid2_hist
        h->dmax = d.dmax; h->rmax = d.rmax; h->emax = d.emax; h->rnorm2 = d.rnorm2; h->d2 = d.d2; h->e2 = d.e2;
    }
}

// Reduce the update kernel's partials (only the fields the rule needs) and decide.
__device__ inline Decision reduce_and_decide(const StateLite& s, const RuleParams& rp, const double* partB,
                                             int nB, int strideB, int esB, int want_diag, double* lds, const PreParts* pre_rr = nullptr,
                                             const PreMax* pre_max = nullptr /* rmax, dmax, emax */) {
    const double rr = pre_rr ? dd_value(reduce_parts_dd_pre(*pre_rr, partB + FB_RR * strideB, partB + (FB_RR + FB_LO) * strideB, nB, esB, lds))
                             : dd_value(reduce_parts_dd(partB + FB_RR * strideB, partB + (FB_RR + FB_LO) * strideB, nB, esB, lds));
    double rmax = 0, dmax = 0, emax = 0, d2 = 0, e2 = 0;
    if (rp.rule == 0 || want_diag) {
        if (pre_max) {
            rmax = reduce_max_pre(pre_max[0], partB + FB_RMAX * strideB, nB, esB, lds);
            dmax = reduce_max_pre(pre_max[1], partB + FB_DMAX * strideB, nB, esB, lds);
            if (rp.use_u) emax = reduce_max_pre(pre_max[2], partB + FB_EMAX * strideB, nB, esB, lds);
        } else {
            rmax = reduce_parts<true>(partB + FB_RMAX * strideB, nB, esB, lds);
            dmax = reduce_parts<true>(partB + FB_DMAX * strideB, nB, esB, lds);
            if (rp.use_u) emax = reduce_parts<true>(partB + FB_EMAX * strideB, nB, esB, lds);
        }
    }
    if (want_diag) {
        d2 = dd_value(reduce_parts_dd(partB + FB_D2 * strideB, partB + (FB_D2 + FB_LO) * strideB, nB, esB, lds));
        if (rp.use_u) e2 = dd_value(reduce_parts_dd(partB + FB_E2 * strideB, partB + (FB_E2 + FB_LO) * strideB, nB, esB, lds));
    }
    return decide_after_update(s, rp, rr, rmax, dmax, emax, d2, e2);
}
// ---- what crosses the parts of a team after each phase (csrc/team.h) -------------------------------------------------
// A part's RECORD = its partials reduced in slot order (sums as hi/lo pairs, then the maxes) + its stop request: 16 doubles.
// Everything a launch hands to another launch that is NOT ordered behind it by a stream travels in FLAGGED form: a double is
// cut into two words of 32 data bits, each stored as one 64-bit word {stamp << 32 | data} with a single-copy-atomic
// system-scope store.  A reader knows a word has arrived when it carries the stamp it expects (the team-wide iteration
// sequence number), so no fence orders the words with anything.  Two hops use it:
//   * every block of a producer launch also stores its partials flagged (FlagSpec).  A one-block REDUCER launch (k_reduce_ll)
//     runs beside the producer on another stream, polls those words, reduces them in the consumers' order and stores the
//     part's record -- flagged -- straight into the mailbox of every OTHER part (a peer GPU's memory over xGMI; IPC-mapped
//     when that part is another process).  The producer launch carries no tail work for it and ends when its items end.
//   * a consumer launch reduces its OWN part's partials itself (they were written by the launch before it on its stream),
//     polls its mailbox for the other parts' records (bounded) and combines all parts in part order, so every part takes
//     the same decision.  By the time it is launched the neighbours' reducers have usually delivered.
// No collective, no event and no extra launch sits on the path producer -> consumer.  (RCCL's LL protocol is the same idea; with
// the RCCL all-gather as the transport the flagged records are what is gathered and the poll succeeds at once.)
typedef unsigned long long u64;
constexpr int kRecWords = 16;                     // doubles per record
constexpr int kLLWords = 2 * kRecWords;           // 64-bit flagged words per record
constexpr int kRecStopWord = 9;                   // word that carries a rank's stop request (max over ranks = stop everywhere)
constexpr int kMaxRecDst = 16;
constexpr int kReasonTransport = 5;               // internal stop reason: a record did not arrive within the budget (-> MI355CG_ERR_STATE)
constexpr int kMaxPartSlots = 1024;               // partial slots of one phase the reducer can take (interior + edge launches <= 2 x 512 blocks)
__device__ inline u64 ld_sys(const u64* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ inline void st_sys(u64* p, u64 v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ inline u64 flagged(unsigned stamp, unsigned data) { return ((u64)stamp << 32) | data; }

// A producer launch's flagged partials: slot-major, [slot][2 * fields] words (field f of a block: words 2f = low half, 2f + 1 = high half)
struct FlagSpec {
    u64* part;                                    // nullptr: this launch has no reader outside its stream
    unsigned stamp;
};
__device__ inline void store_flagged(const FlagSpec& fs, int slot, int nfields, const double* v) {      // one thread; v: the block's partials, field order
    u64* w = fs.part + (long long)slot * (2 * nfields);
    for (int f = 0; f < nfields; ++f) {
        const u64 bits = __builtin_bit_cast(u64, v[f]);
        st_sys(w + 2 * f, flagged(fs.stamp, (unsigned)bits));
        st_sys(w + 2 * f + 1, flagged(fs.stamp, (unsigned)(bits >> 32)));
    }
}
// a launch that ends in its prologue (the solve is over) still owes the reducer its words: zeros
__device__ inline void store_flagged_zero(const FlagSpec& fs, int slot, int nfields) {
    if (!fs.part) return;
    u64* w = fs.part + (long long)slot * (2 * nfields);
    for (int k = 0; k < 2 * nfields; ++k) st_sys(w + k, flagged(fs.stamp, 0u));
}

// Where a part's record goes.
struct RecSpec {
    int ndst;
    u64* const* dst;                              // DEVICE array: this part's record (slot 0) in the mailbox of every destination part
    u64* const* flag;                             // DEVICE array or null: per destination, the word a STREAM-level wait of the consumer watches
    int slot_words;                               // 64-bit words from slot 0 to slot 1 of a mailbox
    int slot;                                     // the slot of this record (iteration sequence number & 1)
    unsigned seq;                                 // stamp of this record (never 0)
    u64 flag_value;                               // what the announcement words get: the sequence number itself
};
// Where a consumer launch finds the OTHER parts' records of the phase before it.
struct RecSrc {
    const u64* mbox;                              // [world][kLLWords] (the slot this launch reads); nullptr: not a team launch
    int world, me;                                // me: this part (its own record is not awaited: the launch reduces its own partials)
    unsigned stamp;
    u64 budget;                                   // wall_clock64 ticks (100 MHz) this launch may wait for a word
};
// All threads of a block.  recs: LDS, [world][kRecWords] doubles; slot `me` is left alone.  Returns false when a word did not
// show up in time.  Ends with a barrier: what other threads wrote into recs before the call is visible after it.
__device__ inline bool gather_records(const RecSrc& src, double* recs) {
    unsigned* out = reinterpret_cast<unsigned*>(recs);
    const int n = src.world * kLLWords;
    int bad = 0;
    for (int i = threadIdx.x; i < n; i += kBlock) {
        if (i / kLLWords == src.me) continue;
        u64 v = ld_sys(src.mbox + i);
        if ((unsigned)(v >> 32) != src.stamp) {
            const u64 t0 = wall_clock64();
            unsigned spins = 0;
            for (;;) {
                __builtin_amdgcn_s_sleep(2);
                v = ld_sys(src.mbox + i);
                if ((unsigned)(v >> 32) == src.stamp) break;
                if ((++spins & 31u) == 0 && wall_clock64() - t0 > src.budget) { bad = 1; break; }
            }
        }
        out[i] = (unsigned)v;                     // word 2k + h of a record = half h of its double k (little endian)
    }
    return __syncthreads_or(bad) == 0;
}
// rec (LDS, kRecWords doubles) -> flagged words in every destination's mailbox
__device__ inline void publish_record(const double* rec, const RecSpec& rs) {
    const unsigned* half = reinterpret_cast<const unsigned*>(rec);
    const long long slot = (long long)rs.slot * rs.slot_words;
    for (int i = threadIdx.x; i < rs.ndst * kLLWords; i += kBlock)
        st_sys(rs.dst[i / kLLWords] + slot + i % kLLWords, flagged(rs.seq, half[i % kLLWords]));
    if (rs.flag && (int)threadIdx.x < rs.ndst) st_sys(rs.flag[threadIdx.x], rs.flag_value);
}

// ---- the reducer: flagged partials of one phase -> the part's record in the other parts' mailboxes --------------------
// One block.  Runs beside the producer launch(es) of the phase (its own stream) or behind them (same stream); either way it
// only believes stamped words.  The reduction is reduce_parts_dd's / reduce_parts<true>'s, slot for slot, so the record carries
// exactly the pairs a consumer of the same part computes for itself from the plain partials.
struct ReduceArgs {
    const u64* part; int nslots, which;           // which: 0 = stencil partials (FA_*), 1 = update partials (FB_*)
    unsigned stamp; u64 budget;
    RecSpec rs;
};
// Small on purpose (a few dozen VGPRs, no LDS staging): it has to become resident beside a producer launch that fills the CUs.
__global__ __launch_bounds__(kBlock) void k_reduce_ll(const ReduceArgs a) {
    __shared__ double lds[2 * kWaves];
    __shared__ double rec[kRecWords];
    const int nf = a.which == 0 ? FA_COUNT : FB_LL_COUNT, n = a.nslots;
    const int nsum = a.which == 0 ? kNumSumsA : kNumSumsB, lo_off = a.which == 0 ? FA_LO : FB_LO;
    if (threadIdx.x < kRecWords) rec[threadIdx.x] = 0.0;
    // Thread t takes slots t, t + 256, ... in ascending order -- reduce_parts_dd's assignment -- and polls each slot's words itself.
    dd sum[kNumSumsB] = {dd_zero(), dd_zero(), dd_zero()};
    double mx[4] = {0.0, 0.0, 0.0, 0.0};                  // update phase: rmax, dmax, emax, stop
    int bad = 0;
    const u64 t0 = wall_clock64();
#pragma unroll 1
    for (int slot = threadIdx.x; slot < n && !bad; slot += kBlock) {
        const u64* w = a.part + (long long)slot * (2 * nf);
        double val[FB_LL_COUNT];
#pragma unroll
        for (int f = 0; f < FB_LL_COUNT; ++f) {
            val[f] = 0.0;
            if (f < nf && !bad) {
                unsigned half[2] = {0u, 0u};
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    u64 v = ld_sys(w + 2 * f + h);
                    unsigned spins = 0;
                    for (;;) {
                        // a stamp is 0x80000000 | (sequence number mod 2^31); 0 = never written.  A NEWER stamp: the launches of a later
                        // iteration have already overwritten the words -- only possible once the solve is over and launches return in
                        // their prologue -- so nobody waits for this record any more.
                        const unsigned st = (unsigned)(v >> 32);
                        const int ahead = (st & 0x80000000u) ? (int)((st - a.stamp) << 1) : -1;
                        if (ahead == 0) break;
                        if (ahead > 0) { bad = 1; break; }
                        __builtin_amdgcn_s_sleep(8);
                        v = ld_sys(w + 2 * f + h);
                        if ((++spins & 31u) == 0 && wall_clock64() - t0 > a.budget) { bad = 1; break; }
                    }
                    half[h] = (unsigned)v;
                }
                val[f] = __builtin_bit_cast(double, ((u64)half[1] << 32) | half[0]);
            }
        }
#pragma unroll
        for (int f = 0; f < kNumSumsB; ++f) if (f < nsum) sum[f] = dd_add(sum[f], dd{val[f], val[f + lo_off]});
        if (a.which == 1) {
#pragma unroll
            for (int k = 0; k < 4; ++k) mx[k] = fmax(mx[k], val[FB_RMAX + k]);       // FB_RMAX, FB_DMAX, FB_EMAX, FB_STOP are consecutive
        }
    }
    if (__syncthreads_or(bad)) return;            // superseded, or timed out: then the consumers miss the record too and end the solve
#pragma unroll
    for (int f = 0; f < kNumSumsB; ++f) if (f < nsum) {
        const dd t = block_reduce_dd(sum[f], lds);
        if (threadIdx.x == 0) { rec[f] = t.hi; rec[f + lo_off] = t.lo; }
    }
    if (a.which == 1) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const double t = block_reduce<true>(mx[k], lds);
            // k = 3: the stop request as the update launch itself sampled it (the launch's own part acts on the same sample through its state)
            if (threadIdx.x == 0) rec[k < 3 ? FB_RMAX + k : kRecStopWord] = t;
        }
    }
    __syncthreads();
    publish_record(rec, a.rs);
}

// ---- the consumer side: the phase's scalars from the parts' records, in part order -------------------------------
// recs[me] was filled by the caller from the part's own partials; use_max: the rule (or the summary) looks at the max-norms
__device__ inline Decision decide_from_records(const StateLite& s, const RuleParams& rp, const double* recs, int world, bool use_max, bool want_diag) {
    dd rr = dd_zero(), d2 = dd_zero(), e2 = dd_zero();
    double rmax = 0, dmax = 0, emax = 0, stop = 0;
    for (int j = 0; j < world; ++j) {
        const double* R = recs + j * kRecWords;
        rr = dd_add(rr, dd{R[FB_RR], R[FB_RR + FB_LO]});
        d2 = dd_add(d2, dd{R[FB_D2], R[FB_D2 + FB_LO]});
        e2 = dd_add(e2, dd{R[FB_E2], R[FB_E2 + FB_LO]});
        rmax = fmax(rmax, R[FB_RMAX]); dmax = fmax(dmax, R[FB_DMAX]); emax = fmax(emax, R[FB_EMAX]);
        stop = fmax(stop, R[kRecStopWord]);
    }
    if (!use_max) { rmax = 0; dmax = 0; emax = 0; }
    if (!rp.use_u) { emax = 0; e2 = dd_zero(); }
    if (!want_diag) { d2 = dd_zero(); e2 = dd_zero(); }
    return decide_after_update(s, rp, dd_value(rr), rmax, dmax, emax, dd_value(d2), dd_value(e2), stop > 0.0);
}
__device__ inline void alpha_from_records(const StateLite& s, int rule, const double* recs, int world, double* alpha, double* rz_out) {
    dd pap = dd_zero(), rz = dd_zero();
    for (int j = 0; j < world; ++j) {
        const double* R = recs + j * kRecWords;
        pap = dd_add(pap, dd{R[FA_PAP], R[FA_PAP + FA_LO]});
        rz = dd_add(rz, dd{R[FA_RZ], R[FA_RZ + FA_LO]});
    }
    if (rule == 0) { *rz_out = dd_value(rz); *alpha = *rz_out / dd_value(pap); }       // msg_solver.cpp:102
    else { *rz_out = 0.0; *alpha = s.rr / dd_value(pap); }                             // matrix_free_system.cpp:419
}
// This part's own update-phase record fields from its plain partials (the reductions of reduce_and_decide, kept as pairs), into
// recs[me].  level 0: r.r only; 1: + the max-norms; 2: everything (summary).  All threads of the block.
__device__ inline void own_record_B(double* recs, int me, const double* partB, int nB, int strideB, int level, double* lds,
                                    const PreParts* pre_rr = nullptr, const PreMax* pre_max = nullptr) {
    double* R = recs + me * kRecWords;
    if (threadIdx.x < kRecWords) R[threadIdx.x] = 0.0;
    const dd rr = pre_rr ? reduce_parts_dd_pre(*pre_rr, partB + FB_RR * strideB, partB + (FB_RR + FB_LO) * strideB, nB, 1, lds)
                         : reduce_parts_dd(partB + FB_RR * strideB, partB + (FB_RR + FB_LO) * strideB, nB, 1, lds);
    double rmax = 0, dmax = 0, emax = 0;
    dd d2 = dd_zero(), e2 = dd_zero();
    if (level >= 1) {
        if (pre_max) {
            rmax = reduce_max_pre(pre_max[0], partB + FB_RMAX * strideB, nB, 1, lds);
            dmax = reduce_max_pre(pre_max[1], partB + FB_DMAX * strideB, nB, 1, lds);
            emax = reduce_max_pre(pre_max[2], partB + FB_EMAX * strideB, nB, 1, lds);
        } else {
            rmax = reduce_parts<true>(partB + FB_RMAX * strideB, nB, 1, lds);
            dmax = reduce_parts<true>(partB + FB_DMAX * strideB, nB, 1, lds);
            emax = reduce_parts<true>(partB + FB_EMAX * strideB, nB, 1, lds);
        }
    }
    if (level >= 2) {
        d2 = reduce_parts_dd(partB + FB_D2 * strideB, partB + (FB_D2 + FB_LO) * strideB, nB, 1, lds);
        e2 = reduce_parts_dd(partB + FB_E2 * strideB, partB + (FB_E2 + FB_LO) * strideB, nB, 1, lds);
    }
    __syncthreads();                                   // the zero fill above is done
    if (threadIdx.x == 0) {
        R[FB_RR] = rr.hi; R[FB_RR + FB_LO] = rr.lo; R[FB_D2] = d2.hi; R[FB_D2 + FB_LO] = d2.lo; R[FB_E2] = e2.hi; R[FB_E2 + FB_LO] = e2.lo;
        R[FB_RMAX] = rmax; R[FB_DMAX] = dmax; R[FB_EMAX] = emax;
    }
}
// a launch whose records did not arrive: the solve ends here with an internal reason the host turns into an error
__device__ inline void fail_transport(CgState* out, const CgState* in) {
    copy_state(out, in);
    out->done = 1; out->reason = kReasonTransport; out->converged = 0;
}

// ---- dynamic item queues (MI355CG_DYN_ROWS; experimental, see profiles/r03_tune_notes.md section 5) ------------------
// With one tall item per wave a launch ends when its SLOWEST wave ends.  Queues even that out: the items are cut shorter, every
// wave starts on its static first item and then takes tickets from a counter of its group (XCD class x 8 sub-groups: one address
// sustains only ~20 M atomics/s); ticket t of sub-group g is item begin + W + 8 t + g of the class's range, so the groups sweep
// the class's band together.  The ticket is asked for when an item is ENTERED and used at the next switch.  The counters of a
// launch kind are zeroed by the launches of the other kind, which run between two of its launches.
constexpr int kQueueSubs = 8;
constexpr int kQueuePitch = 64;        // ints between two counters: every counter has a 256-B line (an L2 channel) of its own
struct QueueSpec {
    int* mine;            // kXcds * kQueueSubs counters of this launch kind, 0 at launch; nullptr: static round-robin (w, w + W, ...)
    int* other;           // the other kind's counters, zeroed by this launch (may be null)
};
// lane 0 takes a ticket; the result stays in lane 0's register until the next item switch (the compiler waits for it there with a
// counted vmcnt: by then it is an old operation).  Built with -amdgpu-atomic-optimizer-strategy=None (build.py): hipcc's atomic
// optimizer would rewrite the one-lane atomic into its wave-aggregated form, which reads the result back at once -- a drain of the
// rows in flight at every item.
__device__ inline int queue_take(int* q, int lane) {
    int v = 0;
    if (lane == 0) v = __hip_atomic_fetch_add(q, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return v;
}

// ---- phase A': fused direction update + 5-point stencil + dots ------------------------------------
template <typename T>
struct StencilArgs {
    Geom g;
    WorkList wl;
    const T* r;          // FUSED: residual (with ghost rows / columns); PLAIN: unused
    const T* pin;        // FUSED: previous direction; PLAIN: the vector to apply the operator to
    T* pout;             // FUSED: new direction (ping-pong partner of pin)
    T* ap;               // PLAIN: A_h * input vector
    const double* partB; int nB, strideB, esB;  // update-kernel partials to reduce in the prologue (count, field stride, element stride)
    RecSrc src;                                 // team: the parts' update records instead (src.mbox != nullptr)
    double* partA; int strideA, slotA;          // this kernel's partials (field-major); first slot of this launch
    const CgState* s_in; CgState* s_out;        // state written by the update kernel / by this kernel
    HistEntry* hist;
    RuleParams rp;
    int want_diag;
    int store_ghosts;    // part of a decomposed grid: also store p_new of the ghost rows (recomputed from the local ghost copies
                         // of r and p_old, bit-identical to the neighbour's rows), so the direction never has to cross ranks
    FlagSpec fl;         // team: every block also stores its partials flagged, for the reducer launch that runs beside this one
    QueueSpec dq;        // dynamic item queues (see QueueSpec)
};

template <typename T, int VEC> struct VecOf { typedef T type __attribute__((ext_vector_type(VEC))); };

// ---- wave-uniform addressing --------------------------------------------------------------------------------
// Every stream is a buffer resource (base = first row of the work item, wave-uniform, in SGPRs), a row is an SGPR byte
// offset that advances by the row pitch, and the lane contributes ONE 32-bit byte offset that never changes while the
// item is marched.  A lane that must not touch memory carries an offset beyond num_records: the hardware range check
// returns 0 for its load and drops its store.  No load or store sits in an exec-masked region, so hipcc waits with
// counted vmcnt(N) and the rows prefetched for later steps really stay in flight.  In-row neighbours come through DPP
// wave shifts, not LDS permutes.  (Round 1: ~120 issued instructions per 128-column row instead of ~216 with flat
// addressing; profiles/r01_tune_notes.md.)
constexpr int kOob = (int)0x80000000;
typedef __amdgpu_buffer_rsrc_t rsrc_t;
__device__ inline rsrc_t make_rsrc(const void* base) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, 0x7fffffff, 0x00020000);
}
template <typename V> __device__ inline V buf_load(rsrc_t r, int voff, int soff) {
    if constexpr (sizeof(V) == 16) return __builtin_bit_cast(V, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
    else if constexpr (sizeof(V) == 8) return __builtin_bit_cast(V, __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0));
    else { static_assert(sizeof(V) == 4, "buf_load: 4, 8 or 16 bytes"); return __builtin_bit_cast(V, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0)); }
}
// HAZARD (measured on gfx950, ROCm 7.2; profiles/r02_tune_notes.md section 10): a VALU instruction that overwrites a data
// register of a 128-bit buffer store in the very next slot is seen by the store -- the lanes read last (12..15 of every row of
// 16) go to memory with the NEW value.  The ISA lists this hazard (VMEM store of more than 64 bits, then a VALU write of its
// vdata: 1 wait state) and hipcc pads it for FLAT stores and for MUBUF stores WITHOUT a register soffset; with the soffset in
// an SGPR (every store here: the row offset) it pads nothing.  The empty asm below keeps the data registers alive across two
// wait states after the store, so nothing can be scheduled into that window that writes them.
template <typename V> __device__ inline void buf_store(V v, rsrc_t r, int voff, int soff) {
    typedef unsigned int u4 __attribute__((ext_vector_type(4)));
    typedef unsigned int u2 __attribute__((ext_vector_type(2)));
    if constexpr (sizeof(V) == 16) {
        u4 d = __builtin_bit_cast(u4, v);
        __builtin_amdgcn_raw_buffer_store_b128(d, r, voff, soff, 0);
        asm volatile("s_nop 1" : "+v"(d));
    }
    else if constexpr (sizeof(V) == 8) __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u2, v), r, voff, soff, 0);
    else { static_assert(sizeof(V) == 4, "buf_store: 4, 8 or 16 bytes"); __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned int, v), r, voff, soff, 0); }
}
// value of the lane below (lane - 1) / above (lane + 1); lane 0 / lane 63 get 0 and are overridden by the caller
__device__ inline int dpp_from_below(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x138 /*wave_shr:1*/, 0xf, 0xf, false); }
__device__ inline int dpp_from_above(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x130 /*wave_shl:1*/, 0xf, 0xf, false); }
__device__ inline float lane_below(float v) { return __builtin_bit_cast(float, dpp_from_below(__builtin_bit_cast(int, v))); }
__device__ inline float lane_above(float v) { return __builtin_bit_cast(float, dpp_from_above(__builtin_bit_cast(int, v))); }
__device__ inline double lane_below(double v) {
    const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
    const unsigned lo = (unsigned)dpp_from_below((int)(unsigned)u), hi = (unsigned)dpp_from_below((int)(unsigned)(u >> 32));
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
__device__ inline double lane_above(double v) {
    const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
    const unsigned lo = (unsigned)dpp_from_above((int)(unsigned)u), hi = (unsigned)dpp_from_above((int)(unsigned)(u >> 32));
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
// bytes from row y to row y + 1 of the storage layout (rows <= half are Pb wide and start at column cb)
template <typename T> __device__ inline int row_step(const Geom& g, int y) { return (y + 1 <= g.half ? g.Pb : g.Pu) * (int)sizeof(T); }

// Byte offset of a lane's first column x in a row of the bottom block (columns >= cb stored) or of the upper block,
// kOob for a lane outside the stored columns; the same for the single element beyond a wave-edge lane.
template <typename T> __device__ inline int lane_off(const Geom& g, int x, bool bottom_row) {
    return (x < g.xlim && x >= (bottom_row ? g.cb : 0)) ? x * (int)sizeof(T) : kOob;
}
template <typename T, int VEC> __device__ inline int edge_off(const Geom& g, int x, int lane, bool bottom_row) {
    const int xe = lane == 0 ? x - 1 : x + VEC;
    const bool edge = lane == 0 || lane == kWave - 1;
    return (edge && xe >= (bottom_row ? g.cb : 0) && xe < g.xlim) ? xe * (int)sizeof(T) : kOob;
}

// ---- work items ------------------------------------------------------------------------------------------------
// Items are enumerated panel by panel, chunk-major (strip fastest), and dealt to the resident waves round-robin:
// wave w takes items w, w + W, w + 2W, ... (W = waves of the launch).  With SHORT items (a few dozen rows) the waves
// of one round therefore work inside one compact band of rows: at N = 16384 / 32768, where a row is 131 / 262 KB, that
// keeps concurrent accesses within a few MB per stream and measured +17-18 % over 2 048 independent waves each marching
// one tall item down the whole grid (tools/hbm_probe.hip, profiles/r02_hbm_probe_*.txt).
struct Item { int strip, ya, yb, gc; };
__device__ inline Item decode_item(const WorkList& wl, int item) {
    // constant indices only: a run-time index into the by-value argument struct makes hipcc spill the whole work
    // list into per-thread LDS
    Panel P = wl.p[0];
#pragma unroll
    for (int k = 1; k < kMaxPanels; ++k) if (k < wl.np && item >= wl.p[k].item0) P = wl.p[k];
    const int local = item - P.item0;
    const int chunk = local / P.ns;
    Item it;
    it.strip = P.s0 + (local - chunk * P.ns);
    it.ya = P.y0 + chunk * P.ty;
    it.yb = min(P.y1, it.ya + P.ty - 1);
    it.gc = (it.strip == P.s0 ? (P.gc & 1) : 0) | (it.strip == P.s0 + P.ns - 1 ? (P.gc & 2) : 0);
    return it;
}

// Addressing data of one work item, constant while the item is marched (wave-uniform unless noted).  It is computed
// when the FETCH cursor of a wave enters the item and handed to the COMPUTE cursor when that gets there: the fetch
// cursor runs DEPTH rows ahead of the compute cursor ACROSS item boundaries, so the load pipeline never drains between
// items (round 1 refilled it per item, which is what made short items lose there).
struct ItemAddr {
    int nrows, ystart;          // rows of the item; its first row in march order
    long long base_el;          // element offset of row y0 = ya - 1 from the start of a vector
    int so_first, so_c0;        // byte offsets from row y0 of the halo row behind the first own row / of the first own row
    int gc; bool bot;
    int x;                                        // per lane: first column
    int vo_own, ve_own, vo_first, vo_last;        // per lane: byte offsets within a row (lane_off / edge_off)
};
template <typename T, int VEC, bool DESC>
__device__ inline ItemAddr item_addr(const Geom& g, const Item& it, int lane) {
    constexpr int DIR = DESC ? -1 : 1;
    ItemAddr A;
    A.x = it.strip * (kWave * VEC) + lane * VEC;
    A.nrows = it.yb - it.ya + 1;
    A.ystart = DESC ? it.yb : it.ya;
    const int y0 = it.ya - 1;                                   // lowest row the item touches (halo)
    A.base_el = row_off(g, y0) - g.base0;
    A.bot = it.ya <= g.half;                                    // all own rows lie in one block of the L
    A.vo_own = lane_off<T>(g, A.x, A.bot);                      // the item's own rows (loads and stores)
    A.ve_own = edge_off<T, VEC>(g, A.x, lane, A.bot);
    A.vo_first = lane_off<T>(g, A.x, A.ystart - DIR <= g.half); // the halo row behind the first own row
    A.vo_last = lane_off<T>(g, A.x, A.ystart + DIR * A.nrows <= g.half);   // the halo row ahead of the last one
    A.so_first = DESC ? (int)((row_off(g, it.yb + 1) - row_off(g, y0)) * (long long)sizeof(T)) : 0;
    A.so_c0 = DESC ? (int)((row_off(g, it.yb) - row_off(g, y0)) * (long long)sizeof(T)) : row_step<T>(g, y0);
    A.gc = it.gc;
    return A;
}

// One wave marches (64*VEC)-column strips: per item the rows ya-1 .. yb+1 pass through a DEPTH-deep queue of raw rows in
// flight; three converted rows stay in registers (behind / centre / ahead in march order); in-row neighbours come from the
// adjacent lane (DPP wave shift), the two wave-edge lanes load their outside neighbour themselves.  The role of a
// dequeued row follows from its index in its item: -1 = halo row behind the first own row, 0 = first centre row,
// k >= 1 = the row ahead of centre row k-1, whose 5-point formula can now be evaluated.
// NOAP: A p is only reduced into (Ap, p), not stored -- the update phase recomputes it from the stored direction.
// GC (2-D decomposition): the strips at the left / right end of this part also store the new direction of the ghost
// COLUMN beside them (the edge lanes compute it anyway), so the direction never crosses ranks in x either.
template <typename T, int VEC, bool FUSED, bool MSG, int DEPTH, bool NOAP, bool GC>
__global__ __launch_bounds__(kBlock) void k_stencil(const StencilArgs<T> a) {
    static_assert(DEPTH >= 1 && DEPTH <= 3, "the compute cursor takes its item from the fetch cursor: DEPTH <= rows fetched per item (>= 3)");
    typedef typename VecOf<T, VEC>::type vec_t;
    __shared__ double lds[2 * kWaves];
    const Geom& g = a.g;
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);   // wave-uniform -> SGPR item decode
    const ItemSeq seq = item_seq(a.wl, wave);
    MI355CG_WT_BEGIN
    struct Raw { vec_t r, p; T re, pe; };

    // ---- fetch cursor ----
    int f_item = seq.first;
    bool f_have = f_item < seq.end;
    ItemAddr F{};
    int f_idx = 0, f_so = 0, f_y = 0;
    rsrc_t rs_p = make_rsrc(a.pin), rs_r = make_rsrc(a.pin);
    const int dq_nsub = min(kQueueSubs, (int)(gridDim.x / kXcds)), dq_sub = (int)(blockIdx.x / kXcds) % dq_nsub;      // (every sub-group has a workgroup)
    int* const dq = a.dq.mine ? a.dq.mine + ((int)(blockIdx.x % kXcds) * kQueueSubs + dq_sub) * kQueuePitch : nullptr;
    if (a.dq.other && blockIdx.x == 0 && threadIdx.x < kXcds * kQueueSubs) a.dq.other[threadIdx.x * kQueuePitch] = 0;
    int dq_ticket = 0;                                          // lane 0: this wave's next ticket of its group's queue
    auto enter = [&](int item) {
        if (dq) dq_ticket = queue_take(dq, lane);               // asked for now, needed at the next switch
        F = item_addr<T, VEC, false>(g, decode_item(a.wl, item), lane);
        rs_p = make_rsrc(a.pin + F.base_el);
        rs_r = make_rsrc(FUSED ? a.r + F.base_el : a.pin + F.base_el);
        f_idx = -1; f_so = F.so_first; f_y = F.ystart - 1;
    };
    auto fetch = [&]() -> Raw {
        if (f_have && f_idx > F.nrows) {                         // lazily: the compute cursor may still need F (see promote)
            f_item = dq ? seq.begin + seq.step + __builtin_amdgcn_readfirstlane(dq_ticket) * dq_nsub + dq_sub : f_item + seq.step;
            f_have = f_item < seq.end;
            if (f_have) enter(f_item);
        }
        const bool own = f_idx >= 0 && f_idx < F.nrows;
        int vo = own ? F.vo_own : (f_idx < 0 ? F.vo_first : F.vo_last);
        int ve = own ? F.ve_own : kOob;                          // only centre rows need the element beyond the wave edge
        if (!f_have) { vo = kOob; ve = kOob; }                   // past the last item: the loads touch nothing
        Raw w;
        w.p = buf_load<vec_t>(rs_p, vo, f_so);
        if (FUSED) w.r = buf_load<vec_t>(rs_r, vo, f_so);
        else for (int j = 0; j < VEC; ++j) w.r[j] = (T)0;
        w.pe = buf_load<T>(rs_p, ve, f_so);
        if (FUSED) w.re = buf_load<T>(rs_r, ve, f_so); else w.re = (T)0;
        f_so += row_step<T>(g, f_y); ++f_y; ++f_idx;
        return w;
    };

    // ---- compute cursor ----
    bool c_have = f_have;
    ItemAddr C{};
    int c_idx = -1, c_so = 0, c_y = 0, c_ve_gc = kOob;
    bool in_j[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) in_j[j] = false;
    rsrc_t rs_po = make_rsrc(a.pin), rs_ap = make_rsrc(a.pin);
    // The compute cursor enters the item the fetch cursor is in.  That is the right one: when the compute cursor finishes
    // item i the fetch cursor has fetched 1 .. DEPTH rows beyond it, all of them rows of item i+1 (every item fetches
    // nrows + 2 >= 3 >= DEPTH rows), and it only moves on to item i+2 at the fetch after the last row of item i+1.
    auto promote = [&]() {
        C = F;
        rs_po = make_rsrc(FUSED ? a.pout + C.base_el : a.pin + C.base_el);
        rs_ap = make_rsrc(NOAP ? a.pin + C.base_el : a.ap + C.base_el);
        c_idx = -1; c_so = C.so_c0; c_y = C.ystart;
        const int xint0 = C.bot ? g.half + 1 : 1;               // first interior column of the own rows
#pragma unroll
        for (int j = 0; j < VEC; ++j) in_j[j] = (C.x + j >= xint0) && (C.x + j <= g.N - 1);
        if (GC) c_ve_gc = ((lane == 0 && (C.gc & 1)) || (lane == kWave - 1 && (C.gc & 2))) ? C.ve_own : kOob;
    };

    // The first rows are requested BEFORE the prologue: they do not depend on beta, and the state load and the
    // reduction of the partials then run under their latency instead of in front of it.
    if (f_have) { enter(f_item); promote(); }
    Raw q[DEPTH];
#pragma unroll
    for (int k = 0; k < DEPTH; ++k) q[k] = fetch();

    T beta = (T)0;
    if (FUSED && a.src.mbox) {
        // team: this part's share of the sums comes from its own partials (as on a single GPU), the other parts' from their records
        __shared__ double recs[kMaxRecDst * kRecWords];
        const PreParts pre = prefetch_parts(a.partB + FB_RR * a.strideB, a.partB + (FB_RR + FB_LO) * a.strideB, a.nB, 1);
        PreMax pmax[3] = {{0, 0}, {0, 0}, {0, 0}};
        if (MSG) {
#pragma unroll
            for (int k = 0; k < 3; ++k) pmax[k] = prefetch_max(a.partB + (FB_RMAX + k) * a.strideB, a.nB, 1);
        }
        const StateLite s = load_state_lite(a.s_in);
        if (s.done) {
            if (threadIdx.x == 0) { store_flagged_zero(a.fl, a.slotA + blockIdx.x, FA_COUNT); if (blockIdx.x == 0) copy_state(a.s_out, a.s_in); }
            return;
        }
        MI355CG_WT_STAMP(1)
        own_record_B(recs, a.src.me, a.partB, a.nB, a.strideB, MSG ? 1 : 0, lds, &pre, MSG ? pmax : nullptr);
        if (!gather_records(a.src, recs)) {
            if (threadIdx.x == 0) { store_flagged_zero(a.fl, a.slotA + blockIdx.x, FA_COUNT); if (blockIdx.x == 0) fail_transport(a.s_out, a.s_in); }
            return;
        }
        const Decision d = decide_from_records(s, a.rp, recs, a.src.world, MSG, false);
        MI355CG_WT_STAMP(2)
        if (blockIdx.x == 0 && threadIdx.x == 0) write_state_after_decision(a.s_out, a.hist, a.s_in, s, d);
        if (d.done) { if (threadIdx.x == 0) store_flagged_zero(a.fl, a.slotA + blockIdx.x, FA_COUNT); return; }
        beta = (T)d.beta;
    } else if (FUSED) {
        const PreParts pre = prefetch_parts(a.partB + FB_RR * a.strideB, a.partB + (FB_RR + FB_LO) * a.strideB, a.nB, a.esB);
        PreMax pmax[3] = {{0, 0}, {0, 0}, {0, 0}};
        if (MSG) {
#pragma unroll
            for (int k = 0; k < 3; ++k) pmax[k] = prefetch_max(a.partB + (FB_RMAX + k) * a.strideB, a.nB, a.esB);
        }
        const StateLite s = load_state_lite(a.s_in);
        if (s.done) { if (blockIdx.x == 0 && threadIdx.x == 0) copy_state(a.s_out, a.s_in); return; }
        MI355CG_WT_STAMP(1)
        const Decision d = reduce_and_decide(s, a.rp, a.partB, a.nB, a.strideB, a.esB, a.want_diag, lds, &pre, MSG ? pmax : nullptr);
        MI355CG_WT_STAMP(2)
        if (blockIdx.x == 0 && threadIdx.x == 0) write_state_after_decision(a.s_out, a.hist, a.s_in, s, d);
        if (d.done) return;
        beta = (T)d.beta;
    }

    MI355CG_WT_MID
    const T cA = (T)g.A, cxk = (T)g.xk, cyk = (T)g.yk;
    dd acc_pap = dd_zero(), acc_rz = dd_zero();
    // a part's ghost row (not a physical boundary row): keep the new direction there too
    auto is_ghost = [&](int yy) { return (yy == g.y_lo - 1 || yy == g.y_hi + 1) && yy >= 1 && yy <= g.N - 1; };
    vec_t pn_b, pn_c, r_c;                 // behind / centre rows of the new direction, residual of the centre row
    T pne_c = (T)0;
#pragma unroll
    for (int j = 0; j < VEC; ++j) { pn_b[j] = (T)0; pn_c[j] = (T)0; r_c[j] = (T)0; }

    while (c_have) {
#pragma unroll
        for (int k = 0; k < DEPTH; ++k) {
            if (c_have) {
                const Raw w = q[k];
                q[k] = fetch();
                vec_t pn; T pne;
                if (FUSED) {
#pragma unroll
                    for (int j = 0; j < VEC; ++j) pn[j] = w.r[j] + beta * w.p[j];     // z = r + beta*z
                    pne = w.re + beta * w.pe;
                } else { pn = w.p; pne = w.pe; }

                if (c_idx < 0) {                                   // halo row behind the first own row
                    pn_b = pn;
                    if (FUSED && a.store_ghosts && is_ghost(C.ystart - 1)) buf_store(pn, rs_po, C.vo_first, C.so_first);
                } else if (c_idx == 0) {                           // first centre row
                    pn_c = pn; pne_c = pne; r_c = w.r;
                } else {                                           // pn = row ahead of centre row c_idx - 1
                    // in-row neighbours: from the adjacent lane, wave-edge lanes use their edge load
                    T left0 = lane_below(pn_c[VEC - 1]);
                    T rightL = lane_above(pn_c[0]);
                    if (lane == 0) left0 = pne_c;
                    if (lane == kWave - 1) rightL = pne_c;
                    vec_t out;
#pragma unroll
                    for (int j = 0; j < VEC; ++j) {
                        const T c = pn_c[j];
                        const T L = j == 0 ? left0 : pn_c[j - 1];
                        const T R = j == VEC - 1 ? rightL : pn_c[j + 1];
                        // y[row] += A*x[row]; += x_k*left; += x_k*right; += y_k*top; += y_k*bottom
                        T v = cA * c;
                        v = v + cxk * L;
                        v = v + cxk * R;
                        v = v + cyk * pn[j];                       // row y+1
                        v = v + cyk * pn_b[j];                     // row y-1
                        out[j] = in_j[j] ? v : (T)0;
                        dd_acc_prod(acc_pap, (double)c, (double)out[j]);
                        if (MSG) dd_acc_prod(acc_rz, (double)r_c[j], (double)c);
                    }
                    if (!NOAP) buf_store(out, rs_ap, C.vo_own, c_so);
                    if (FUSED) buf_store(pn_c, rs_po, C.vo_own, c_so);
                    if (FUSED && GC) buf_store(pne_c, rs_po, c_ve_gc, c_so);   // new direction of the ghost column beside the part
                    c_so += row_step<T>(g, c_y); ++c_y;
                    pn_b = pn_c; pn_c = pn; pne_c = pne; r_c = w.r;
                }
                if (c_idx == C.nrows) {                            // that was the halo row ahead of the last own row
                    if (FUSED && a.store_ghosts && is_ghost(c_y)) buf_store(pn, rs_po, C.vo_last, c_so);
                    c_have = f_have;                          // the next item is the one the fetch cursor is in by now (1 .. DEPTH rows into it), if any
                    if (c_have) promote();
                } else ++c_idx;
            }
        }
    }

    MI355CG_WT_END(0);
    const dd tp = block_reduce_dd(acc_pap, lds);
    dd tz = dd_zero();
    if (MSG) tz = block_reduce_dd(acc_rz, lds);
    if (threadIdx.x == 0 && a.partA) {
        const int b = a.slotA + blockIdx.x, st = a.strideA;
        a.partA[FA_PAP * st + b] = tp.hi; a.partA[(FA_PAP + FA_LO) * st + b] = tp.lo;
        a.partA[FA_RZ * st + b] = tz.hi;  a.partA[(FA_RZ + FA_LO) * st + b] = tz.lo;
        if (FUSED && a.fl.part) { const double v[FA_COUNT] = {tp.hi, tz.hi, tp.lo, tz.lo}; store_flagged(a.fl, b, FA_COUNT, v); }     // field order FA_*
    }
}

// ---- flat update: state initialisation, resume step of the mixed-precision path, generic CSR path -----------------
template <typename T>
struct UpdateArgs {
    long long begin, nvec;     // owned flat range in units of VEC elements (begin is a vec index)
    T* x; T* r; const T* p; const T* ap; const T* u;
    const double* partA; int nA, strideA, esA;
    double* partB; int strideB;
    const CgState* s_in; CgState* s_out;
    int rule;                  // MSG: alpha = rz / Azz ; REL2: alpha = rr / pAp
    int init;                  // 1: alpha := 0, state initialisation (x = 0, r = b); 2: resume (see below)
    double r0norm_resume;      // init == 2 (resume after a residual replacement): the new reference norm
};

template <typename T, int VEC, bool HAS_U>
__global__ __launch_bounds__(kBlock) void k_update(const UpdateArgs<T> a) {
    typedef typename VecOf<T, VEC>::type vec_t;
    __shared__ double lds[2 * kWaves];
    StateLite s{};
    double alpha_d = 0.0, rz = 0.0;
    if (a.init == 1) {
        s.first = 1; s.it = 0;
    } else if (a.init == 2) {
        // Resume after a residual replacement (mixed precision): r was overwritten with the freshly computed true
        // residual.  Measure it (alpha = 0 leaves x and r untouched) and re-arm the state WITHOUT restarting CG: the
        // direction is kept, the iteration count continues, the pending x update is gone (flushed by the host), and
        // the beta of the next step divides by the (r, r) the interrupted step would have used (see the state write below).
        s = load_state_lite(a.s_in);
    } else {
        s = load_state_lite(a.s_in);
        if (s.done) { if (blockIdx.x == 0 && threadIdx.x == 0) copy_state(a.s_out, a.s_in); return; }
        const double pap = dd_value(reduce_parts_dd(a.partA + FA_PAP * a.strideA, a.partA + (FA_PAP + FA_LO) * a.strideA, a.nA, a.esA, lds));
        if (a.rule == 0) {
            rz = dd_value(reduce_parts_dd(a.partA + FA_RZ * a.strideA, a.partA + (FA_RZ + FA_LO) * a.strideA, a.nA, a.esA, lds));
            alpha_d = rz / pap;                       // msg_solver.cpp:102
        } else {
            alpha_d = s.rr / pap;                     // matrix_free_system.cpp:419
        }
    }
    const T alpha = (T)alpha_d;

    dd s_rr = dd_zero(), s_d2 = dd_zero(), s_e2 = dd_zero();
    double s_rmax = 0, s_dmax = 0, s_emax = 0;
    const long long stride = (long long)gridDim.x * kBlock;
    vec_t* __restrict__ X = reinterpret_cast<vec_t*>(a.x);                 // five distinct vectors: no aliasing
    vec_t* __restrict__ R = reinterpret_cast<vec_t*>(a.r);
    const vec_t* __restrict__ Pp = reinterpret_cast<const vec_t*>(a.p);
    const vec_t* __restrict__ Q = reinterpret_cast<const vec_t*>(a.ap);
    const vec_t* __restrict__ Uu = reinterpret_cast<const vec_t*>(a.u);

    auto elem = [&](long long i, const vec_t& x0, const vec_t& pv, const vec_t& r0, const vec_t& qv, const vec_t& uv) {
        vec_t xn, rn;
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            xn[j] = x0[j] + alpha * pv[j];            // x = x + alpha*z        msg_solver.cpp:105-107
            rn[j] = r0[j] - alpha * qv[j];            // r = r - alpha*A_z      msg_solver.cpp:110-112
            const double rd = (double)rn[j];
            dd_acc_prod(s_rr, rd, rd);
            s_rmax = fmax(s_rmax, fabs(rd));
            const double dx = (double)(xn[j] - x0[j]); // diff = x - x_prev     msg_solver.cpp:124-127
            s_dmax = fmax(s_dmax, fabs(dx));
            dd_acc_prod(s_d2, dx, dx);
            if (HAS_U) {
                const double ee = (double)(xn[j] - uv[j]);   // error = x - u   msg_solver.cpp:132-136
                s_emax = fmax(s_emax, fabs(ee));
                dd_acc_prod(s_e2, ee, ee);
            }
        }
        X[i] = xn; R[i] = rn;
    };
    long long i = a.begin + (long long)blockIdx.x * kBlock + threadIdx.x;
    const long long end = a.begin + a.nvec;
    // two elements per lane and trip: both sets of loads are issued before the first store
    for (; i + stride < end; i += 2 * stride) {
        const long long j0 = i, j1 = i + stride;
        const vec_t xa = X[j0], pa = Pp[j0], ra = R[j0], qa = Q[j0];
        const vec_t xb = X[j1], pb = Pp[j1], rb = R[j1], qb = Q[j1];
        vec_t ua{}, ub{}; if (HAS_U) { ua = Uu[j0]; ub = Uu[j1]; }
        elem(j0, xa, pa, ra, qa, ua);
        elem(j1, xb, pb, rb, qb, ub);
    }
    if (i < end) {
        const vec_t x0 = X[i], pv = Pp[i], r0 = R[i], qv = Q[i];
        vec_t uv{}; if (HAS_U) uv = Uu[i];
        elem(i, x0, pv, r0, qv, uv);
    }

    const dd t_rr = block_reduce_dd(s_rr, lds);
    const double t_rmax = block_reduce<true>(s_rmax, lds);
    const double t_dmax = block_reduce<true>(s_dmax, lds);
    const dd t_d2 = block_reduce_dd(s_d2, lds);
    double t_emax = 0; dd t_e2 = dd_zero();
    if (HAS_U) { t_emax = block_reduce<true>(s_emax, lds); t_e2 = block_reduce_dd(s_e2, lds); }
    if (threadIdx.x == 0) {
        const int b = blockIdx.x, st = a.strideB;
        a.partB[FB_RR * st + b] = t_rr.hi; a.partB[(FB_RR + FB_LO) * st + b] = t_rr.lo;
        a.partB[FB_D2 * st + b] = t_d2.hi; a.partB[(FB_D2 + FB_LO) * st + b] = t_d2.lo;
        a.partB[FB_E2 * st + b] = t_e2.hi; a.partB[(FB_E2 + FB_LO) * st + b] = t_e2.lo;
        a.partB[FB_RMAX * st + b] = t_rmax; a.partB[FB_DMAX * st + b] = t_dmax; a.partB[FB_EMAX * st + b] = t_emax;
        if (blockIdx.x == 0) {
            CgState* o = a.s_out;
            if (a.init == 1) {
                *o = CgState{};
                o->first = 1;
            } else {
                copy_state(a.s_out, a.s_in);
                if (a.init == 2) {
                    if (s.done) o->rr = s.rr_prev;
                    o->done = 0; o->reason = 0; o->converged = 0; o->alpha = 0.0; o->r0norm = a.r0norm_resume;
                    for (int i = 0; i < kRing; ++i) o->alpha_hist[i] = 0.0;
                } else { o->it = s.it + 1; o->first = 0; o->alpha = alpha_d; o->rz = rz; o->alpha_hist[(s.it + 1) & (kRing - 1)] = alpha_d; }
            }
        }
    }
}

// ---- start of a solve in one pass: x = 0, r = b, z = 0 (msg_solver.cpp:33-39, matrix_free_system.cpp:392-401), the norms of
// r0 (and of x - u = -u) and a fresh state.  Same partial-sum layout and the same element-to-thread map as k_update with
// init == 1, which it replaces on the fp64 grid path: 4 words per unknown instead of 6 memsets + a copy + a 6-word pass.
template <typename T>
struct FreshArgs {
    long long begin, nvec;     // owned flat range in units of VEC elements
    const T* b; T* x; T* r; T* p0; const T* u;
    double* partB; int strideB;
    CgState* s_out;
};
template <typename T, int VEC, bool HAS_U>
__global__ __launch_bounds__(kBlock) void k_init_fresh(const FreshArgs<T> a) {
    typedef typename VecOf<T, VEC>::type vec_t;
    __shared__ double lds[2 * kWaves];
    dd s_rr = dd_zero(), s_e2 = dd_zero();
    double s_rmax = 0, s_emax = 0;
    const long long stride = (long long)gridDim.x * kBlock;
    const vec_t* __restrict__ B = reinterpret_cast<const vec_t*>(a.b);
    const vec_t* __restrict__ Uu = reinterpret_cast<const vec_t*>(a.u);
    vec_t* __restrict__ X = reinterpret_cast<vec_t*>(a.x);
    vec_t* __restrict__ R = reinterpret_cast<vec_t*>(a.r);
    vec_t* __restrict__ P = reinterpret_cast<vec_t*>(a.p0);
    vec_t zero;
#pragma unroll
    for (int j = 0; j < VEC; ++j) zero[j] = (T)0;
    const long long end = a.begin + a.nvec;
    for (long long i = a.begin + (long long)blockIdx.x * kBlock + threadIdx.x; i < end; i += stride) {
        const vec_t bv = B[i];
        vec_t uv; if (HAS_U) uv = Uu[i];
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            const double rd = (double)bv[j];
            dd_acc_prod(s_rr, rd, rd);
            s_rmax = fmax(s_rmax, fabs(rd));
            if (HAS_U) {
                const double ee = (double)((T)0 - uv[j]);     // error = x - u with x = 0
                s_emax = fmax(s_emax, fabs(ee));
                dd_acc_prod(s_e2, ee, ee);
            }
        }
        X[i] = zero; R[i] = bv; P[i] = zero;
    }
    const dd t_rr = block_reduce_dd(s_rr, lds);
    const double t_rmax = block_reduce<true>(s_rmax, lds);
    double t_emax = 0; dd t_e2 = dd_zero();
    if (HAS_U) { t_emax = block_reduce<true>(s_emax, lds); t_e2 = block_reduce_dd(s_e2, lds); }
    if (threadIdx.x == 0) {
        const int b = blockIdx.x, st = a.strideB;
        a.partB[FB_RR * st + b] = t_rr.hi; a.partB[(FB_RR + FB_LO) * st + b] = t_rr.lo;
        a.partB[FB_D2 * st + b] = 0.0; a.partB[(FB_D2 + FB_LO) * st + b] = 0.0;
        a.partB[FB_E2 * st + b] = t_e2.hi; a.partB[(FB_E2 + FB_LO) * st + b] = t_e2.lo;
        a.partB[FB_RMAX * st + b] = t_rmax; a.partB[FB_DMAX * st + b] = 0.0; a.partB[FB_EMAX * st + b] = t_emax;
        if (blockIdx.x == 0) { *a.s_out = CgState{}; a.s_out->first = 1; }
    }
}

// ---- phase B: r -= alpha * (A_h p) with A_h p rebuilt from the stored direction, x update, norms ----------------
// This kernel walks the same (chunk, strip) items as the stencil launch -- in the opposite order and direction, so it
// starts on the rows that launch touched last -- keeps three rows of p in registers and evaluates the 5-point formula
// again: same operands, same operation order, hence the same bits as the values the stencil launch reduced into
// (Ap, p).  A p never touches HBM.  XM selects what happens to x in this launch:
//   0  nothing (the iterations between two folded updates, see M below);
//   1  x += alpha p plus the norms |dx|, |x - u| (MSG rule every iteration; REL_2NORM with per-iteration diagnostics);
//   M = 2 or 4  folded update on iterations k = 0 mod M: x = (..(x + alpha_{k-M+1} p_{k-M+1}) + ..) + alpha_k p_k.  The M - 1 earlier
//      directions are still intact in the other buffers of the direction ring, their step lengths are in the state
//      (alpha_hist).  Same operations in the same order as M single updates, but x is read and written once per M iterations:
//      M = 4: stencil launch 3 words (r, p in, p out); this launch 3 words on three iterations and 8 on the fourth
//      (p, r, x, three old directions in; r, x out): 7.25 words per unknown and iteration on average (M = 2: 7.5).
template <typename T>
struct UpdateStArgs {
    Geom g;
    WorkList wl;
    const T* p;          // current direction, ghost rows / columns valid
    const T* pprev[kRing - 1];   // XM >= 2: the directions of the 1, 2, ... iterations before (the other buffers of the ring)
    T* r; T* x; const T* u;
    const double* partA; int nA, strideA, esA;
    RecSrc src;          // team: the parts' stencil records instead (src.mbox != nullptr)
    double* partB; int strideB, slotB;
    const CgState* s_in; CgState* s_out;
    int rule;
    int reverse;         // take the items from the last to the first (start where the stencil launch ended)
    const int* stop_req; // pinned host word, sampled once per iteration by block 0 -> CgState::stop (msg_solver.cpp:82-87) and, in a team, -> the
                         // part's record (every part ORs its own sample with the other parts' records, so all decide alike); may be null
    FlagSpec fl;         // team: every block also stores its partials flagged, for the reducer launch that runs beside this one
    QueueSpec dq;        // dynamic item queues (see QueueSpec)
};

template <typename T, int VEC, int XM, bool HAS_U, int DEPTH, bool DESC>
__global__ __launch_bounds__(kBlock) void k_update_st(const UpdateStArgs<T> a) {
    static_assert(DEPTH >= 1 && DEPTH <= 3, "see k_stencil");
    typedef typename VecOf<T, VEC>::type vec_t;
    __shared__ double lds[2 * kWaves];
    const Geom& g = a.g;
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
    const ItemSeq seq = item_seq(a.wl, wave);
    MI355CG_WT_BEGIN
    // the stop request (a pinned HOST word) is sampled once per iteration by block 0 -- a scalar load issued here and consumed
    // in the epilogue, so its PCIe round trip runs under the whole launch
    int stop_word = 0;
    if (blockIdx.x == 0 && a.stop_req) stop_word = scalar_load(a.stop_req);
    constexpr bool FULL = XM == 1;
    constexpr int NP = XM >= 2 ? XM - 1 : 0;          // earlier directions folded into this launch's x update
    struct Raw { vec_t p, r, x, u, pp[NP > 0 ? NP : 1]; T pe; };

    // ---- fetch cursor (see k_stencil) ----
    int f_item = seq.first;
    bool f_have = f_item < seq.end;
    ItemAddr F{};
    int f_idx = 0, f_so = 0, f_y = 0;
    rsrc_t rs_p = make_rsrc(a.p), rs_r = rs_p, rs_x = rs_p, rs_u = rs_p;
    rsrc_t rs_pp[NP > 0 ? NP : 1] = {rs_p};
    const int dq_nsub = min(kQueueSubs, (int)(gridDim.x / kXcds)), dq_sub = (int)(blockIdx.x / kXcds) % dq_nsub;      // (every sub-group has a workgroup)
    int* const dq = a.dq.mine ? a.dq.mine + ((int)(blockIdx.x % kXcds) * kQueueSubs + dq_sub) * kQueuePitch : nullptr;
    if (a.dq.other && blockIdx.x == 0 && threadIdx.x < kXcds * kQueueSubs) a.dq.other[threadIdx.x * kQueuePitch] = 0;
    int dq_ticket = 0;
    auto enter = [&](int idx) {
        if (dq) dq_ticket = queue_take(dq, lane);
        F = item_addr<T, VEC, DESC>(g, decode_item(a.wl, a.reverse ? seq.end - 1 - (idx - seq.begin) : idx), lane);   // reversed within the wave's class
        rs_p = make_rsrc(a.p + F.base_el);
        rs_r = make_rsrc(a.r + F.base_el);
        rs_x = make_rsrc(XM != 0 ? a.x + F.base_el : a.p + F.base_el);
#pragma unroll
        for (int i = 0; i < NP; ++i) rs_pp[i] = make_rsrc(a.pprev[i] + F.base_el);
        rs_u = make_rsrc((FULL && HAS_U) ? a.u + F.base_el : a.p + F.base_el);
        f_idx = -1; f_so = F.so_first; f_y = DESC ? F.ystart + 1 : F.ystart - 1;
    };
    // `own`: the row is one of this item's rows (its r / x / u / previous direction are needed, and its edge element)
    auto fetch = [&]() -> Raw {
        if (f_have && f_idx > F.nrows) {
            f_item = dq ? seq.begin + seq.step + __builtin_amdgcn_readfirstlane(dq_ticket) * dq_nsub + dq_sub : f_item + seq.step;
            f_have = f_item < seq.end;
            if (f_have) enter(f_item);
        }
        const bool own = f_have && f_idx >= 0 && f_idx < F.nrows;
        int vo = own ? F.vo_own : (f_idx < 0 ? F.vo_first : F.vo_last);
        if (!f_have) vo = kOob;
        const int vown = own ? F.vo_own : kOob;
        Raw w;
        w.p = buf_load<vec_t>(rs_p, vo, f_so);
        w.r = buf_load<vec_t>(rs_r, vown, f_so);
        if (XM != 0) w.x = buf_load<vec_t>(rs_x, vown, f_so);
#pragma unroll
        for (int i = 0; i < NP; ++i) w.pp[i] = buf_load<vec_t>(rs_pp[i], vown, f_so);
        if (FULL && HAS_U) w.u = buf_load<vec_t>(rs_u, vown, f_so);
        w.pe = buf_load<T>(rs_p, own ? F.ve_own : kOob, f_so);
        if (DESC) { f_so -= row_step<T>(g, f_y - 1); --f_y; } else { f_so += row_step<T>(g, f_y); ++f_y; }
        ++f_idx;
        return w;
    };

    // ---- compute cursor ----
    bool c_have = f_have;
    ItemAddr C{};
    int c_idx = -1, c_so = 0, c_y = 0;
    bool in_j[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) in_j[j] = false;
    rsrc_t rs_ro = rs_p, rs_xo = rs_p;
    auto promote = [&]() {
        C = F;
        rs_ro = make_rsrc(a.r + C.base_el);
        rs_xo = make_rsrc(XM != 0 ? a.x + C.base_el : a.p + C.base_el);
        c_idx = -1; c_so = C.so_c0; c_y = C.ystart;
        const int xint0 = C.bot ? g.half + 1 : 1;
#pragma unroll
        for (int j = 0; j < VEC; ++j) in_j[j] = (C.x + j >= xint0) && (C.x + j <= g.N - 1);
    };

    if (f_have) { enter(f_item); promote(); }
    Raw q[DEPTH];
#pragma unroll
    for (int k = 0; k < DEPTH; ++k) q[k] = fetch();

    double alpha_d, rz = 0.0;
    StateLite s;
    if (a.src.mbox) {
        __shared__ double recs[kMaxRecDst * kRecWords];
        const PreParts pre = prefetch_parts(a.partA + FA_PAP * a.strideA, a.partA + (FA_PAP + FA_LO) * a.strideA, a.nA, 1);
        PreParts pre_rz{0, 0, 0, 0};
        if (FULL) pre_rz = prefetch_parts(a.partA + FA_RZ * a.strideA, a.partA + (FA_RZ + FA_LO) * a.strideA, a.nA, 1);
        s = load_state_lite(a.s_in);
        if (s.done) {
            if (threadIdx.x == 0) { store_flagged_zero(a.fl, a.slotB + blockIdx.x, FB_LL_COUNT); if (blockIdx.x == 0) copy_state(a.s_out, a.s_in); }
            return;
        }
        {   // this part's own (Ap, p) [and (r, p)] from its partials, into its slot of the records
            double* R = recs + a.src.me * kRecWords;
            if (threadIdx.x < kRecWords) R[threadIdx.x] = 0.0;
            const dd pap = reduce_parts_dd_pre(pre, a.partA + FA_PAP * a.strideA, a.partA + (FA_PAP + FA_LO) * a.strideA, a.nA, 1, lds);
            dd rzd = dd_zero();
            if (a.rule == 0) rzd = FULL ? reduce_parts_dd_pre(pre_rz, a.partA + FA_RZ * a.strideA, a.partA + (FA_RZ + FA_LO) * a.strideA, a.nA, 1, lds)
                                        : reduce_parts_dd(a.partA + FA_RZ * a.strideA, a.partA + (FA_RZ + FA_LO) * a.strideA, a.nA, 1, lds);
            __syncthreads();
            if (threadIdx.x == 0) { R[FA_PAP] = pap.hi; R[FA_PAP + FA_LO] = pap.lo; R[FA_RZ] = rzd.hi; R[FA_RZ + FA_LO] = rzd.lo; }
        }
        if (!gather_records(a.src, recs)) {
            if (threadIdx.x == 0) { store_flagged_zero(a.fl, a.slotB + blockIdx.x, FB_LL_COUNT); if (blockIdx.x == 0) fail_transport(a.s_out, a.s_in); }
            return;
        }
        alpha_from_records(s, a.rule, recs, a.src.world, &alpha_d, &rz);
    } else {
        const PreParts pre = prefetch_parts(a.partA + FA_PAP * a.strideA, a.partA + (FA_PAP + FA_LO) * a.strideA, a.nA, a.esA);
        PreParts pre_rz{0, 0, 0, 0};
        if (FULL) pre_rz = prefetch_parts(a.partA + FA_RZ * a.strideA, a.partA + (FA_RZ + FA_LO) * a.strideA, a.nA, a.esA);
        s = load_state_lite(a.s_in);
        if (s.done) { if (blockIdx.x == 0 && threadIdx.x == 0) copy_state(a.s_out, a.s_in); return; }
        const double pap = dd_value(reduce_parts_dd_pre(pre, a.partA + FA_PAP * a.strideA, a.partA + (FA_PAP + FA_LO) * a.strideA, a.nA, a.esA, lds));
        if (a.rule == 0) {
            rz = FULL ? dd_value(reduce_parts_dd_pre(pre_rz, a.partA + FA_RZ * a.strideA, a.partA + (FA_RZ + FA_LO) * a.strideA, a.nA, a.esA, lds))
                      : dd_value(reduce_parts_dd(a.partA + FA_RZ * a.strideA, a.partA + (FA_RZ + FA_LO) * a.strideA, a.nA, a.esA, lds));
            alpha_d = rz / pap;                       // msg_solver.cpp:102
        } else {
            alpha_d = s.rr / pap;                     // matrix_free_system.cpp:419
        }
    }
    const T alpha = (T)alpha_d;
    MI355CG_WT_MID
    T aprev[NP > 0 ? NP : 1] = {(T)0};     // XM >= 2: step lengths of iterations k-1, k-2, k-3 (k = s.it + 1; 0 after init / resume)
#pragma unroll
    for (int i = 0; i < NP; ++i) aprev[i] = (T)scalar_load(&a.s_in->alpha_hist[(s.it - i) & (kRing - 1)]);
    const T cA = (T)g.A, cxk = (T)g.xk, cyk = (T)g.yk;
    dd s_rr = dd_zero(), s_d2 = dd_zero(), s_e2 = dd_zero();
    double s_rmax = 0, s_dmax = 0, s_emax = 0;
    vec_t p_b;
    Raw c;                                 // centre row
#pragma unroll
    for (int j = 0; j < VEC; ++j) { p_b[j] = (T)0; c.p[j] = (T)0; c.r[j] = (T)0; c.x[j] = (T)0; c.u[j] = (T)0; for (int i = 0; i < (NP > 0 ? NP : 1); ++i) c.pp[i][j] = (T)0; }
    c.pe = (T)0;

    while (c_have) {
#pragma unroll
        for (int k = 0; k < DEPTH; ++k) {
            if (c_have) {
                const Raw w = q[k];
                q[k] = fetch();
                if (c_idx < 0) {
                    p_b = w.p;
                } else if (c_idx == 0) {
                    c = w;
                } else {                                           // w.p = row ahead of the centre row
                    T left0 = lane_below(c.p[VEC - 1]);
                    T rightL = lane_above(c.p[0]);
                    if (lane == 0) left0 = c.pe;
                    if (lane == kWave - 1) rightL = c.pe;
                    const vec_t& top = DESC ? p_b : w.p;           // row y+1
                    const vec_t& bot = DESC ? w.p : p_b;           // row y-1
                    vec_t rn, xn;
#pragma unroll
                    for (int j = 0; j < VEC; ++j) {
                        const T cc = c.p[j];
                        const T L = j == 0 ? left0 : c.p[j - 1];
                        const T R = j == VEC - 1 ? rightL : c.p[j + 1];
                        T v = cA * cc;                                   // the stencil launch's formula, verbatim
                        v = v + cxk * L;
                        v = v + cxk * R;
                        v = v + cyk * top[j];
                        v = v + cyk * bot[j];
                        const T apj = in_j[j] ? v : (T)0;
                        rn[j] = c.r[j] - alpha * apj;                    // r = r - alpha*A_z      msg_solver.cpp:110-112
                        const double rd = (double)rn[j];
                        dd_acc_prod(s_rr, rd, rd);
                        s_rmax = fmax(s_rmax, fabs(rd));
                        if (XM >= 2) {                                   // M single x = x + alpha*z steps, oldest first
                            T t = c.x[j];
#pragma unroll
                            for (int i = NP - 1; i >= 0; --i) t = t + aprev[i] * c.pp[i][j];
                            xn[j] = t + alpha * cc;
                        }
                        if (FULL) {
                            xn[j] = c.x[j] + alpha * cc;                 // x = x + alpha*z        msg_solver.cpp:105-107
                            const double dx = (double)(xn[j] - c.x[j]);  // diff = x - x_prev      msg_solver.cpp:124-127
                            s_dmax = fmax(s_dmax, fabs(dx));
                            dd_acc_prod(s_d2, dx, dx);
                            if (HAS_U) {
                                const double ee = (double)(xn[j] - c.u[j]);   // error = x - u     msg_solver.cpp:132-136
                                s_emax = fmax(s_emax, fabs(ee));
                                dd_acc_prod(s_e2, ee, ee);
                            }
                        }
                    }
                    buf_store(rn, rs_ro, C.vo_own, c_so);
                    if (XM != 0) buf_store(xn, rs_xo, C.vo_own, c_so);
                    if (DESC) { c_so -= row_step<T>(g, c_y - 1); --c_y; } else { c_so += row_step<T>(g, c_y); ++c_y; }
                    p_b = c.p; c = w;
                }
                if (c_idx == C.nrows) {
                    c_have = f_have;                          // see k_stencil
                    if (c_have) promote();
                } else ++c_idx;
            }
        }
    }

    MI355CG_WT_END(1);
    const dd t_rr = block_reduce_dd(s_rr, lds);
    const double t_rmax = block_reduce<true>(s_rmax, lds);
    double t_dmax = 0, t_emax = 0; dd t_d2 = dd_zero(), t_e2 = dd_zero();
    if (FULL) { t_dmax = block_reduce<true>(s_dmax, lds); t_d2 = block_reduce_dd(s_d2, lds); }
    if (FULL && HAS_U) { t_emax = block_reduce<true>(s_emax, lds); t_e2 = block_reduce_dd(s_e2, lds); }
    if (threadIdx.x == 0) {
        const int b = a.slotB + blockIdx.x, st = a.strideB;
        a.partB[FB_RR * st + b] = t_rr.hi; a.partB[(FB_RR + FB_LO) * st + b] = t_rr.lo;
        a.partB[FB_D2 * st + b] = t_d2.hi; a.partB[(FB_D2 + FB_LO) * st + b] = t_d2.lo;
        a.partB[FB_E2 * st + b] = t_e2.hi; a.partB[(FB_E2 + FB_LO) * st + b] = t_e2.lo;
        a.partB[FB_RMAX * st + b] = t_rmax; a.partB[FB_DMAX * st + b] = t_dmax; a.partB[FB_EMAX * st + b] = t_emax;
        if (a.fl.part) {                                                  // field order FB_*, then the stop request this launch sampled
            const double v[FB_LL_COUNT] = {t_rr.hi, t_d2.hi, t_e2.hi, t_rr.lo, t_d2.lo, t_e2.lo, t_rmax, t_dmax, t_emax, (blockIdx.x == 0 && stop_word) ? 1.0 : 0.0};
            store_flagged(a.fl, b, FB_LL_COUNT, v);
        }
        if (blockIdx.x == 0) {
            copy_state(a.s_out, a.s_in);
            CgState* o = a.s_out;
            o->it = s.it + 1; o->first = 0; o->alpha = alpha_d; o->rz = rz; o->alpha_hist[(s.it + 1) & (kRing - 1)] = alpha_d;
            o->stop = stop_word != 0;              // the reference tests its flag once per iteration (msg_solver.cpp:82-87)
        }
    }
}

// ---- end-of-chunk check: same decision as the next stencil prologue, without advancing -------------
struct CheckArgs {
    const double* partB; int nB, strideB, esB;
    RecSrc src;               // team: the parts' update records instead
    const CgState* s_in;      // state written by the last update kernel
    CgState* summary;         // device copy that the host reads
    HistEntry* hist;
    RuleParams rp;
    int want_diag;
};
__global__ __launch_bounds__(kBlock) void k_check(const CheckArgs a) {
    __shared__ double lds[2 * kWaves];
    const StateLite s = load_state_lite(a.s_in);
    if (s.done) { if (threadIdx.x == 0) copy_state(a.summary, a.s_in); return; }
    Decision d;
    if (a.src.mbox) {
        __shared__ double recs[kMaxRecDst * kRecWords];
        own_record_B(recs, a.src.me, a.partB, a.nB, a.strideB, 2, lds);
        if (!gather_records(a.src, recs)) { if (threadIdx.x == 0) fail_transport(a.summary, a.s_in); return; }
        d = decide_from_records(s, a.rp, recs, a.src.world, true, true);
    } else d = reduce_and_decide(s, a.rp, a.partB, a.nB, a.strideB, a.esB, 1, lds);
    if (threadIdx.x == 0) write_state_after_decision(a.summary, a.hist, a.s_in, s, d);
}

// The x updates still pending when the folded loop ends on a count that is not a multiple of M (up to three): over the
// part's own cells, x = ((x + a[0] p[0]) + a[1] p[1]) + ..., oldest step first.
template <typename T> struct FlushArgs { const T* p[kRing - 1]; T a[kRing - 1]; int n; };
template <typename T, int VEC>
__global__ __launch_bounds__(kBlock) void k_flush_x(const Geom g, const WorkList wl, T* x, const FlushArgs<T> f) {
    typedef typename VecOf<T, VEC>::type vec_t;
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
    for (int item = blockIdx.x * kWaves + wave; item < wl.nitems; item += gridDim.x * kWaves) {
        const Item it = decode_item(wl, item);
        const int x0 = it.strip * (kWave * VEC) + lane * VEC;
        if (x0 < (it.ya <= g.half ? g.cb : 0) || x0 >= g.xlim) continue;
        for (int y = it.ya; y <= it.yb; ++y) {
            const long long off = row_off(g, y) - g.base0 + x0;
            vec_t xn = *reinterpret_cast<const vec_t*>(x + off);
#pragma unroll
            for (int k = 0; k < kRing - 1; ++k) if (k < f.n) {
                const vec_t pv = *reinterpret_cast<const vec_t*>(f.p[k] + off);
#pragma unroll
                for (int j = 0; j < VEC; ++j) xn[j] = xn[j] + f.a[k] * pv[j];          // x = x + alpha*z
            }
            *reinterpret_cast<vec_t*>(x + off) = xn;
        }
    }
}

// Deterministic checksums of a vector over the part's own cells: per-block double-double partials of sum(v) and sum(v*v)
// (fields: 0 sum hi, 1 sum lo, 2 squares hi, 3 squares lo; field stride = gridDim.x).  Lets tests compare decompositions
// of grids whose vectors are too large to bring to the host.
template <typename T, int VEC>
__global__ __launch_bounds__(kBlock) void k_checksum(const Geom g, const WorkList wl, const T* v, double* part) {
    typedef typename VecOf<T, VEC>::type vec_t;
    __shared__ double lds[2 * kWaves];
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
    dd s1 = dd_zero(), s2 = dd_zero();
    for (int item = blockIdx.x * kWaves + wave; item < wl.nitems; item += gridDim.x * kWaves) {
        const Item it = decode_item(wl, item);
        const int x0 = it.strip * (kWave * VEC) + lane * VEC;
        if (x0 < (it.ya <= g.half ? g.cb : 0) || x0 >= g.xlim) continue;
        for (int y = it.ya; y <= it.yb; ++y) {
            const vec_t t = *reinterpret_cast<const vec_t*>(v + (row_off(g, y) - g.base0 + x0));
#pragma unroll
            for (int j = 0; j < VEC; ++j) { s1 = dd_add(s1, dd{(double)t[j], 0.0}); dd_acc_prod(s2, (double)t[j], (double)t[j]); }
        }
    }
    const dd t1 = block_reduce_dd(s1, lds), t2 = block_reduce_dd(s2, lds);
    if (threadIdx.x == 0) {
        const int n = gridDim.x, b = blockIdx.x;
        part[b] = t1.hi; part[n + b] = t1.lo; part[2 * n + b] = t2.hi; part[3 * n + b] = t2.lo;
    }
}

// per-block max |x - u| over the owned range (MSG rule: the error norm of an iteration whose update
// kernel skipped the u stream because no criterion or callback needed it; msg_solver.cpp:132-139)
template <typename T>
__global__ __launch_bounds__(kBlock) void k_err_maxnorm(long long begin, long long len, const T* x, const T* u, double* part) {
    __shared__ double lds[2 * kWaves];
    const long long stride = (long long)gridDim.x * kBlock;
    double m = 0.0;
    for (long long i = begin + (long long)blockIdx.x * kBlock + threadIdx.x; i < begin + len; i += stride)
        m = fmax(m, fabs((double)(x[i] - u[i])));
    const double t = block_reduce<true>(m, lds);
    if (threadIdx.x == 0) part[blockIdx.x] = t;
}

// ---- mixed precision (fp32 inner CG inside fp64 iterative refinement; no reference twin) ------------
// rf = (float)(b - ax) over the whole stored range (pads/ghosts stay 0) and partial sums of (b - ax)^2
// over the owned range: forms the fp64 true residual, its norm and the fp32 right-hand side in one pass.
__global__ __launch_bounds__(kBlock) void k_residual_to_f32(long long total, long long own_begin, long long own_len,
                                                           const double* b, const double* ax, float* rf, double* part) {
    __shared__ double lds[2 * kWaves];
    const long long stride = (long long)gridDim.x * kBlock;
    double s = 0.0;
    for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < total; i += stride) {
        const double d = b[i] - ax[i];
        rf[i] = (float)d;
        if (i >= own_begin && i < own_begin + own_len) s += d * d;
    }
    const double t = block_reduce<false>(s, lds);
    if (threadIdx.x == 0) part[blockIdx.x] = t;
}
// x64 += (double) d32 over the owned range
__global__ __launch_bounds__(kBlock) void k_accumulate_f32(long long begin, long long len, double* x, const float* d) {
    const long long stride = (long long)gridDim.x * kBlock;
    for (long long i = begin + (long long)blockIdx.x * kBlock + threadIdx.x; i < begin + len; i += stride) x[i] += (double)d[i];
}

// Slab record = [sums (block 0 reduces the partials) | first owned row | last owned row] in ONE launch.
struct RecordArgs {
    const double* part; int n, stride;
    int nsum, lo_off;                                  // sum fields 0..nsum-1 (hi) with lo words at field + lo_off
    int max_first, nmax;                               // max fields max_first .. max_first+nmax-1
    double* rec; int header, row_slot;                 // rows land at rec + header and rec + header + row_slot
    const double* v; long long off_lo, off_hi; int len_lo, len_hi;   // v == nullptr: sums only
};
__global__ __launch_bounds__(kBlock) void k_make_record(const RecordArgs a) {
    __shared__ double lds[2 * kWaves];
    if (blockIdx.x == 0) {
        for (int f = 0; f < a.nsum; ++f) {
            const dd t = reduce_parts_dd(a.part + f * a.stride, a.part + (f + a.lo_off) * a.stride, a.n, 1, lds);
            if (threadIdx.x == 0) { a.rec[f] = t.hi; a.rec[f + a.lo_off] = t.lo; }
        }
        for (int f = a.max_first; f < a.max_first + a.nmax; ++f) {
            const double t = reduce_parts<true>(a.part + f * a.stride, a.n, 1, lds);
            if (threadIdx.x == 0) a.rec[f] = t;
        }
        return;
    }
    if (!a.v) return;
    const int nb = gridDim.x - 1, b = blockIdx.x - 1;
    for (int i = b * kBlock + threadIdx.x; i < a.len_lo; i += nb * kBlock) a.rec[a.header + i] = a.v[a.off_lo + i];
    for (int i = b * kBlock + threadIdx.x; i < a.len_hi; i += nb * kBlock) a.rec[a.header + a.row_slot + i] = a.v[a.off_hi + i];
}
// Neighbours' boundary rows out of the all-gathered records into this rank's two ghost rows.
struct ScatterArgs { const double* src_lo; const double* src_hi; double* dst_lo; double* dst_hi; int len_lo, len_hi; };
__global__ __launch_bounds__(kBlock) void k_scatter_ghosts(const ScatterArgs a) {
    const int stride = gridDim.x * kBlock;
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < a.len_lo; i += stride) a.dst_lo[i] = a.src_lo[i];
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < a.len_hi; i += stride) a.dst_hi[i] = a.src_hi[i];
}

// ---- packed (reference order) <-> storage layout ---------------------------------------------------
// A part's packed order = the reference's unknown order (grid_system.cpp:84-111) restricted to the part's cells: its
// bottom-block rows, then its upper rows, each row restricted to the part's columns.  For the whole grid and for row
// slabs that is a contiguous range of the reference's vector.
struct PackGeom {
    Geom g;
    int nb_rows, yb0, xb0, wb;      // bottom-block rows of the part: count, first row, first own column, own columns per row
    int nu_rows, yu0, xu0, wu;      // upper rows
    long long pk_len;
};

__device__ inline long long packed_to_storage(const PackGeom& pg, long long i) {
    const Geom& g = pg.g;
    const long long nb = (long long)pg.nb_rows * pg.wb;
    int x, y;
    if (i < nb) { const int k = (int)(i / pg.wb); y = pg.yb0 + k; x = pg.xb0 + (int)(i - (long long)k * pg.wb); }
    else { const long long j = i - nb; const int k = (int)(j / pg.wu); y = pg.yu0 + k; x = pg.xu0 + (int)(j - (long long)k * pg.wu); }
    return row_off(g, y) - g.base0 + x;
}
template <typename T>
__global__ __launch_bounds__(kBlock) void k_unpack(const PackGeom pg, const double* __restrict__ packed, T* __restrict__ storage) {
    const long long stride = (long long)gridDim.x * kBlock;
    for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < pg.pk_len; i += stride)
        storage[packed_to_storage(pg, i)] = (T)packed[i];
}
template <typename T>
__global__ __launch_bounds__(kBlock) void k_pack(const PackGeom pg, const T* __restrict__ storage, double* __restrict__ packed) {
    const long long stride = (long long)gridDim.x * kBlock;
    for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < pg.pk_len; i += stride)
        packed[i] = (double)storage[packed_to_storage(pg, i)];
}

// ---- device-side problem setup (opt-in) ---------------------------------------------------------------
// Right-hand side and exact solution of the part's own cells straight into storage layout, in the reference's expression
// order (grid_system.cpp:8-15,45-67,69-77).  The one difference from the host path: exp() is the device library's
// (<= 1 ulp from glibc's), so b and u can differ from the reference's in the last bit -- hence opt-in.
struct SetupArgs { PackGeom pg; double a, c, x_step, y_step, xk, yk; int n, m; };
__device__ inline double setup_u(double x, double y) { return exp(x * x - y * y); }                              // solution()  :12-15
__device__ inline double setup_f(double x, double y) { return 4 * (x * x + y * y) * exp(x * x - y * y); }        // function()  :8-10
__global__ __launch_bounds__(kBlock) void k_setup(const SetupArgs s, double* __restrict__ b, double* __restrict__ u) {
    const Geom& g = s.pg.g;
    const long long stride = (long long)gridDim.x * kBlock;
    const long long nb = (long long)s.pg.nb_rows * s.pg.wb;
    for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < s.pg.pk_len; i += stride) {
        int x, y;
        if (i < nb) { const int k = (int)(i / s.pg.wb); y = s.pg.yb0 + k; x = s.pg.xb0 + (int)(i - (long long)k * s.pg.wb); }
        else { const long long j = i - nb; const int k = (int)(j / s.pg.wu); y = s.pg.yu0 + k; x = s.pg.xu0 + (int)(j - (long long)k * s.pg.wu); }
        const double xp = s.a + x * s.x_step, yp = s.c + y * s.y_step;                                          // calculate_x / calculate_y :69-77
        double value = setup_f(xp, yp);                                                                          // calculate_value :45-67
        const bool left_b = (x - 1 == 0 && y >= s.m / 2 && y <= s.m) || (x - 1 == s.n / 2 && y >= 0 && y <= s.m / 2);      // is_left_boundary(x-1, y) :17-22
        const bool bottom_b = (y - 1 == 0 && x >= s.n / 2 && x <= s.n) || (y - 1 == s.m / 2 && x >= 0 && x <= s.n / 2);    // is_bottom_boundary(x, y-1) :38-43
        if (left_b) value -= s.xk * setup_u(s.a + (x - 1) * s.x_step, yp);
        if (x + 1 == s.n) value -= s.xk * setup_u(s.a + (x + 1) * s.x_step, yp);
        if (y + 1 == s.m) value -= s.yk * setup_u(xp, s.c + (y + 1) * s.y_step);
        if (bottom_b) value -= s.yk * setup_u(xp, s.c + (y - 1) * s.y_step);
        const long long off = row_off(g, y) - g.base0 + x;
        b[off] = value;
        u[off] = setup_u(xp, yp);
    }
}

// out = a - b over the owned flat range (true residual A x - b; dirichlet_solver.cpp:156-158)
template <typename T>
__global__ __launch_bounds__(kBlock) void k_sub(long long begin, long long len, const T* a, const T* b, T* out) {
    const long long stride = (long long)gridDim.x * kBlock;
    for (long long i = begin + (long long)blockIdx.x * kBlock + threadIdx.x; i < begin + len; i += stride) out[i] = a[i] - b[i];
}
// partial sums of (b - ax)^2 over the owned range: the REL_2NORM diagnostic residual
// (matrix_free_system.cpp:457-463).  One partial per block.
template <typename T>
__global__ __launch_bounds__(kBlock) void k_resid2(long long begin, long long len, const T* b, const T* ax, double* part) {
    __shared__ double lds[2 * kWaves];
    const long long stride = (long long)gridDim.x * kBlock;
    double s = 0.0;
    for (long long i = begin + (long long)blockIdx.x * kBlock + threadIdx.x; i < begin + len; i += stride) {
        const double d = (double)(b[i] - ax[i]);
        s += d * d;
    }
    const double t = block_reduce<false>(s, lds);
    if (threadIdx.x == 0) part[blockIdx.x] = t;
}

// REL_2NORM diagnostics mode: fold k_resid2's partials into the history entry of the iteration the state is at
// (the update launch just advanced it), so the per-iteration true residual needs no host round trip.
__global__ __launch_bounds__(kBlock) void k_resid2_hist(const double* part, int n, const CgState* s_in, HistEntry* hist) {
    __shared__ double lds[2 * kWaves];
    const double t = reduce_parts<false>(part, n, 1, lds);
    if (threadIdx.x == 0) hist[scalar_load(&s_in->it) % kHist].tr2 = t;
}

}  // namespace mi355cg
