// cg_kernels.h -- hand-written HIP kernels (gfx950 / CDNA4, wave64) for the matrix-free CG path.
//
//   k_stencil   (phase A')  p_new = r + beta*p_old (fused, halo recomputed), A_h p_new evaluated in registers and
//                           reduced into (Ap,p) [and (r,p)]; A p itself is not stored on the default path.  Replaces
//                           MatrixFreeSystem::apply (matrix_free_system.cpp:203-340) / KokkosSparse::spmv
//                           (msg_solver.cpp:93), the direction update (matrix_free_system.cpp:436-438,
//                           msg_solver.cpp:167-169) and the two dots (matrix_free_system.cpp:417, msg_solver.cpp:96,99).
//   k_update_st (phase B)   A p rebuilt from three rows of the stored direction, r -= alpha Ap, x update (every second
//                           iteration, two steps at once; every iteration for the MSG rule), partial sums / maxes of
//                           r.r, |r|, |dx|, |x-u|.  Replaces matrix_free_system.cpp:422-455 / msg_solver.cpp:105-139.
//   k_update                flat variant: state initialisation and the fall-back that streams a stored A p.
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
constexpr int kMaxPanels = 6;

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
// pair is one work item = one wave marching down `ty` rows of a 64*VEC-column strip.
struct Panel { int y0, y1, s0, ns, ty, nchunks, item0; };
struct WorkList { Panel p[kMaxPanels]; int np; int nitems; };

// ---- CG state carried on the device ----------------------------------------------------------------
struct CgState {
    double alpha, beta;
    double rr;          // (r, r) of the current residual
    double rr_prev;     // (r, r) one decision earlier (lets a stopped solve resume with the right beta denominator)
    double rz;          // MSG: (r, z) of the current iteration (denominator of the next beta)
    double r0norm;      // ||r0||_2
    double rnorm2;      // ||r||_2
    double rmax, dmax, emax, d2, e2;
    int it, done, reason, converged, first, pad_;
};
struct HistEntry { double dmax, rmax, emax, rnorm2, d2, e2; };

// What every wave needs from the state, fetched with SCALAR loads (s_load through the constant address space, one
// request per scalar cache instead of one per wave).  Copying the whole struct made hipcc fetch half of it with
// per-lane global loads of one address: ~4000 waves x 3 loads of the same line queued at one L2 channel and the
// median wave waited 10-14 us of a 65 us launch for its copy of the state (tools/wave_timing.py, profiles/r01_tune_notes.md).
// Safe because no kernel writes the state object it reads (s_in != s_out) and the scalar cache is invalidated at
// every kernel start.
struct StateLite { double alpha, rr, rr_prev, rz, r0norm; int it, done, first; };
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
    L.it = scalar_load(&p->it); L.done = scalar_load(&p->done); L.first = scalar_load(&p->first);
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

__device__ inline Decision decide_after_update(const StateLite& s, const RuleParams& rp, double rr, double rmax,
                                               double dmax, double emax, double d2, double e2) {
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
        HistEntry h; h.dmax = d.dmax; h.rmax = d.rmax; h.emax = d.emax; h.rnorm2 = d.rnorm2; h.d2 = d.d2; h.e2 = d.e2;
        hist[s.it % kHist] = h;
    }
}

// Reduce the update kernel's partials (only the fields the rule needs) and decide.
__device__ inline Decision reduce_and_decide(const StateLite& s, const RuleParams& rp, const double* partB,
                                             int nB, int strideB, int esB, int want_diag, double* lds) {
    const double rr = dd_value(reduce_parts_dd(partB + FB_RR * strideB, partB + (FB_RR + FB_LO) * strideB, nB, esB, lds));
    double rmax = 0, dmax = 0, emax = 0, d2 = 0, e2 = 0;
    if (rp.rule == 0 || want_diag) {
        rmax = reduce_parts<true>(partB + FB_RMAX * strideB, nB, esB, lds);
        dmax = reduce_parts<true>(partB + FB_DMAX * strideB, nB, esB, lds);
        if (rp.use_u) emax = reduce_parts<true>(partB + FB_EMAX * strideB, nB, esB, lds);
    }
    if (want_diag) {
        d2 = dd_value(reduce_parts_dd(partB + FB_D2 * strideB, partB + (FB_D2 + FB_LO) * strideB, nB, esB, lds));
        if (rp.use_u) e2 = dd_value(reduce_parts_dd(partB + FB_E2 * strideB, partB + (FB_E2 + FB_LO) * strideB, nB, esB, lds));
    }
    return decide_after_update(s, rp, rr, rmax, dmax, emax, d2, e2);
}

// ---- phase A': fused direction update + 5-point stencil + dots ------------------------------------
template <typename T>
struct StencilArgs {
    Geom g;
    WorkList wl;
    const T* r;          // FUSED: residual (with ghost rows); PLAIN: unused
    const T* pin;        // FUSED: previous direction; PLAIN: the vector to apply the operator to
    T* pout;             // FUSED: new direction (ping-pong partner of pin)
    T* ap;               // A_h * (new direction | input vector)
    T* x;                // XUPD: solution vector updated with the previous iteration's step
    const double* partB; int nB, strideB, esB;  // update-kernel partials to reduce in the prologue (count, field stride, element stride)
    double* partA; int strideA, slotA;          // this kernel's partials (field-major); first slot of this launch
    const CgState* s_in; CgState* s_out;        // state written by the update kernel / by this kernel
    HistEntry* hist;
    RuleParams rp;
    int want_diag;
    int store_ghosts;    // slab mode: also store p_new of the two ghost rows (recomputed from the local ghost copies of r and
                         // p_old, bit-identical to the neighbour's rows), so the direction never has to cross ranks
};

template <typename T, int VEC> struct VecOf { typedef T type __attribute__((ext_vector_type(VEC))); };

// Cache-policy experiment knobs for the update kernel (bit mask, wave-uniform; env MI355CG_NT).
// Non-temporal accesses on the once-per-iteration streams were measured in the real CG loop
// (profiles/r01_tune_notes.md): no consistent gain at N = 4096, so the default mask is 0.
enum { NT_B_X = 1, NT_B_AP = 2, NT_B_P = 16, NT_B_R = 32, NT_B_U = 256 };
template <typename V> __device__ inline V ld_pol(const V* p, bool nt) { return nt ? __builtin_nontemporal_load(p) : *p; }
template <typename V> __device__ inline void st_pol(V* p, V v, bool nt) { if (nt) __builtin_nontemporal_store(v, p); else *p = v; }

// ---- wave-uniform addressing --------------------------------------------------------------------------------
// The iteration kernels were instruction-issue bound, not bandwidth bound (SQ counters + tools/wave_timing.py: ~216
// issued instructions per 128-column row, of which ~60 are the fp64 arithmetic; the SIMDs' issue slots ~100 % busy and
// the waves dispatched last onto a CU a third slower than the first).  Most of the overhead was 64-bit per-lane address
// arithmetic and pointer selects.  Here every stream is a buffer resource (base = first row of the item, wave-uniform,
// in SGPRs), a row is an SGPR byte offset that advances by the row pitch, and the lane contributes ONE 32-bit byte
// offset that never changes while the item is marched.  A lane that must not touch memory carries an offset beyond
// num_records: the hardware range check returns 0 for its load and drops its store.  No load or store sits in an
// exec-masked region, so hipcc waits with counted vmcnt(N) and the rows prefetched for later iterations really stay
// in flight.  In-row neighbours come through DPP wave shifts, not LDS permutes.
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
template <typename V> __device__ inline void buf_store(V v, rsrc_t r, int voff, int soff) {
    typedef unsigned int u4 __attribute__((ext_vector_type(4)));
    typedef unsigned int u2 __attribute__((ext_vector_type(2)));
    if constexpr (sizeof(V) == 16) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, v), r, voff, soff, 0);
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

// Work-item decode shared by the stencil and the 2-D update kernel.
struct Item { int strip, ya, yb; };
__device__ inline Item decode_item(const WorkList& wl, int item) {
    // constant indices only: a run-time index into the by-value argument struct makes hipcc spill the whole work
    // list into per-thread LDS (14 KB per block)
    Panel P = wl.p[0];
#pragma unroll
    for (int k = 1; k < kMaxPanels; ++k) if (k < wl.np && item >= wl.p[k].item0) P = wl.p[k];
    const int local = item - P.item0;
    const int chunk = local / P.ns;
    Item it;
    it.strip = P.s0 + (local - chunk * P.ns);
    it.ya = P.y0 + chunk * P.ty;
    it.yb = min(P.y1, it.ya + P.ty - 1);
    return it;
}

// One wave marches a (64*VEC)-column strip over rows ya..yb, ascending (DESC=false) or
// descending (DESC=true), keeping three converted rows in registers and DEPTH raw rows in
// flight.  In-row neighbours come from the adjacent lane (DPP wave shift); the two wave-edge
// lanes load their outside neighbour themselves.  The update kernel marches the same chunks
// in the opposite direction, so each kernel starts on the rows the previous one touched last
// (they are still in the 256 MiB Infinity Cache when the vectors are ~100 MB each).
// XUPD (relative-2-norm rule only): the x update of the PREVIOUS iteration, x += alpha_{k-1} p_{k-1}, rides
// along here -- p_{k-1} is this kernel's input direction and is in registers anyway -- so the update kernel
// shrinks to r -= alpha*Ap (3 words) and an iteration moves 9 words per unknown instead of 10.  Element-wise
// arithmetic and order are unchanged; the last pending x update is flushed by k_flush_x after the loop.
// NOAP: A p is only reduced into (Ap, p), not stored -- the update phase recomputes it from the stored direction
// (k_update_st), which removes one written and one read word per unknown and iteration.
template <typename T, int VEC, bool FUSED, bool MSG, int DEPTH, bool DESC, bool XUPD = false, bool NOAP = false>
__global__ __launch_bounds__(kBlock) void k_stencil(const StencilArgs<T> a) {
    typedef typename VecOf<T, VEC>::type vec_t;
    __shared__ double lds[2 * kWaves];
    const Geom& g = a.g;
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);   // wave-uniform -> SGPR item decode
    MI355CG_WT_BEGIN
    constexpr int DIR = DESC ? -1 : 1;
    struct Raw { vec_t r, p, x; T re, pe; };

    // ---- per-item addressing state (wave-uniform unless noted) ----
    int nrows = 0, ystart = 0, y0 = 0, yf = 0, so_f = 0, so_first = 0, fidx = 0, yc = 0, so_c = 0;
    int vo_own = kOob, ve_own = kOob, vo_first = kOob, vo_last = kOob;      // per lane
    bool in_j[VEC];                                                         // per lane: column is an interior node
    rsrc_t rs_p, rs_r, rs_x, rs_po, rs_ap;
    auto setup = [&](int item) {
        const Item it = decode_item(a.wl, item);
        const int x = it.strip * (kWave * VEC) + lane * VEC;
        nrows = it.yb - it.ya + 1;
        ystart = DESC ? it.yb : it.ya;
        y0 = it.ya - 1;                                             // lowest row the item touches (halo)
        const long long base_el = row_off(g, y0) - g.base0;
        rs_p = make_rsrc(a.pin + base_el);
        rs_r = make_rsrc(FUSED ? a.r + base_el : a.pin + base_el);
        rs_x = make_rsrc(XUPD ? a.x + base_el : a.pin + base_el);
        rs_po = make_rsrc(FUSED ? a.pout + base_el : a.pin + base_el);
        rs_ap = make_rsrc(NOAP ? a.pin + base_el : a.ap + base_el);
        const bool bot_item = it.ya <= g.half;                      // all own rows lie in one block of the L
        vo_own = lane_off<T>(g, x, bot_item);                       // lane offset of the item's own rows (loads and stores)
        ve_own = edge_off<T, VEC>(g, x, lane, bot_item);
        vo_first = lane_off<T>(g, x, ystart - DIR <= g.half);       // the halo row behind the first own row
        vo_last = lane_off<T>(g, x, ystart + DIR * nrows <= g.half);   // the halo row ahead of the last one
        const int xint0 = bot_item ? g.half + 1 : 1;                // first interior column of the own rows
#pragma unroll
        for (int j = 0; j < VEC; ++j) in_j[j] = (x + j >= xint0) && (x + j <= g.N - 1);
        // rows are fetched strictly in march order: yf / so_f = next row to fetch and its byte offset from row y0
        yf = ystart - DIR;
        so_f = DESC ? (int)((row_off(g, it.yb + 1) - row_off(g, y0)) * (long long)sizeof(T)) : 0;
        so_first = so_f;
        fidx = -1;                                                  // march index of row yf (-1: the row behind the first)
        yc = ystart;                                                // the centre row and its byte offset
        so_c = DESC ? (int)((row_off(g, it.yb) - row_off(g, y0)) * (long long)sizeof(T)) : row_step<T>(g, y0);
    };
    auto fetch = [&]() -> Raw {
        Raw w;
        const bool own = fidx >= 0 && fidx < nrows;
        const int vo = own ? vo_own : (fidx < 0 ? vo_first : (fidx == nrows ? vo_last : kOob));
        const int ve = own ? ve_own : kOob;                        // only centre rows need the element beyond the wave edge
        w.p = buf_load<vec_t>(rs_p, vo, so_f);
        if (FUSED) w.r = buf_load<vec_t>(rs_r, vo, so_f);
        else for (int j = 0; j < VEC; ++j) w.r[j] = (T)0;
        if (XUPD) w.x = buf_load<vec_t>(rs_x, own ? vo : kOob, so_f);
        w.pe = buf_load<T>(rs_p, ve, so_f);
        if (FUSED) w.re = buf_load<T>(rs_r, ve, so_f); else w.re = (T)0;
        if (DESC) { so_f -= row_step<T>(g, yf - 1); --yf; } else { so_f += row_step<T>(g, yf); ++yf; }
        ++fidx;
        return w;
    };

    // The first item's leading rows are requested BEFORE the prologue: they do not depend on beta, and the state load
    // and the reduction of the partials (4 us) then run under their latency instead of in front of it.
    int item = blockIdx.x * kWaves + wave;
    bool have = item < a.wl.nitems;
    Raw wb, wc, q[DEPTH];
    if (have) {
        setup(item);
        wb = fetch(); wc = fetch();
#pragma unroll
        for (int k = 0; k < DEPTH; ++k) q[k] = fetch();
    }

    T beta = (T)0, alpha_prev = (T)0;
    if (FUSED) {
        const StateLite s = load_state_lite(a.s_in);
        if (s.done) { if (blockIdx.x == 0 && threadIdx.x == 0) copy_state(a.s_out, a.s_in); return; }
        MI355CG_WT_STAMP(1)
        const Decision d = reduce_and_decide(s, a.rp, a.partB, a.nB, a.strideB, a.esB, a.want_diag, lds);
        MI355CG_WT_STAMP(2)
        if (blockIdx.x == 0 && threadIdx.x == 0) write_state_after_decision(a.s_out, a.hist, a.s_in, s, d);
        if (d.done) return;
        beta = (T)d.beta;
        alpha_prev = (T)s.alpha;          // step length of the iteration whose x update is still pending (0 at the start)
    }

    MI355CG_WT_MID
    const T cA = (T)g.A, cxk = (T)g.xk, cyk = (T)g.yk;
    dd acc_pap = dd_zero(), acc_rz = dd_zero();
    auto conv = [&](const Raw& w, vec_t& pn, T& pne) {
        if (FUSED) {
#pragma unroll
            for (int j = 0; j < VEC; ++j) pn[j] = w.r[j] + beta * w.p[j];     // z = r + beta*z
            pne = w.re + beta * w.pe;
        } else { pn = w.p; pne = w.pe; }
    };

    while (have) {
        vec_t pn_b, pn_c, pn_a, r_c;       // behind / centre / ahead rows in march order
        vec_t x_c, pold_c;                 // XUPD: x and the input direction of the centre row
        T pne_c, pne_a, dummy;
        conv(wb, pn_b, dummy);
        conv(wc, pn_c, pne_c);
        r_c = wc.r;
        if (XUPD) { x_c = wc.x; pold_c = wc.p; }
        // a slab's ghost row (not a physical boundary row): keep the new direction there too
        auto is_ghost = [&](int yy) { return (yy == g.y_lo - 1 || yy == g.y_hi + 1) && yy >= 1 && yy <= g.N - 1; };
        if (FUSED && a.store_ghosts && is_ghost(ystart - DIR)) buf_store(pn_b, rs_po, vo_first, so_first);

        for (int i0 = 0; i0 < nrows; i0 += DEPTH) {
#pragma unroll
            for (int k = 0; k < DEPTH; ++k) {
                const int i = i0 + k;
                if (i < nrows) {
                    const Raw w = q[k];
                    q[k] = fetch();
                    conv(w, pn_a, pne_a);

                    // in-row neighbours: from the adjacent lane, wave-edge lanes use their edge load
                    T left0 = lane_below(pn_c[VEC - 1]);
                    T rightL = lane_above(pn_c[0]);
                    if (lane == 0) left0 = pne_c;
                    if (lane == kWave - 1) rightL = pne_c;

                    const vec_t& top = DESC ? pn_b : pn_a;              // row y+1
                    const vec_t& bot = DESC ? pn_a : pn_b;              // row y-1
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
                        v = v + cyk * top[j];
                        v = v + cyk * bot[j];
                        out[j] = in_j[j] ? v : (T)0;
                        dd_acc_prod(acc_pap, (double)c, (double)out[j]);
                        if (MSG) dd_acc_prod(acc_rz, (double)r_c[j], (double)c);
                    }
                    if (!NOAP) buf_store(out, rs_ap, vo_own, so_c);
                    if (FUSED) buf_store(pn_c, rs_po, vo_own, so_c);
                    if (XUPD) {
                        vec_t xn;
#pragma unroll
                        for (int j = 0; j < VEC; ++j) xn[j] = x_c[j] + alpha_prev * pold_c[j];   // x = x + alpha*z
                        buf_store(xn, rs_x, vo_own, so_c);
                    }
                    if (DESC) { so_c -= row_step<T>(g, yc - 1); --yc; } else { so_c += row_step<T>(g, yc); ++yc; }
                    pn_b = pn_c; pn_c = pn_a; pne_c = pne_a; r_c = w.r;
                    if (XUPD) { x_c = w.x; pold_c = w.p; }
                }
            }
        }
        if (FUSED && a.store_ghosts && is_ghost(yc)) buf_store(pn_c, rs_po, vo_last, so_c);   // yc / so_c / pn_c: the row ahead of the last own row
        item += gridDim.x * kWaves;
        have = item < a.wl.nitems;
        if (have) {
            setup(item);
            wb = fetch(); wc = fetch();
#pragma unroll
            for (int k = 0; k < DEPTH; ++k) q[k] = fetch();
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
    }
}

// ---- phase B: fused x / r update + norms -----------------------------------------------------------
template <typename T>
struct UpdateArgs {
    long long begin, nvec;     // owned flat range in units of VEC elements (begin is a vec index)
    T* x; T* r; const T* p; const T* ap; const T* u;
    const double* partA; int nA, strideA, esA;
    double* partB; int strideB;
    const CgState* s_in; CgState* s_out;
    int rule;                  // MSG: alpha = rz / Azz ; REL2: alpha = rr / pAp
    int init;                  // 1: alpha := 0, state initialisation (x = 0, r = b)
    int reverse;               // flat kernel: sweep the range from its end to its start
    int nt;                    // NT_B_* cache-policy bits
    int light;                 // 1: r -= alpha*Ap only (the x update rides in the next stencil launch, see XUPD)
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

    const long long last = a.begin + a.nvec - 1;
    const bool nt_x = a.nt & NT_B_X, nt_ap = a.nt & NT_B_AP, nt_p = a.nt & NT_B_P, nt_r = a.nt & NT_B_R, nt_u = a.nt & NT_B_U;
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
        st_pol(X + i, xn, nt_x); st_pol(R + i, rn, nt_r);
    };
    auto body = [&](long long i_fwd) {
        const long long i = a.reverse ? last - (i_fwd - a.begin) : i_fwd;
        const vec_t x0 = ld_pol(X + i, nt_x), pv = ld_pol(Pp + i, nt_p), r0 = ld_pol(R + i, nt_r), qv = ld_pol(Q + i, nt_ap);
        vec_t uv; if (HAS_U) uv = ld_pol(Uu + i, nt_u);
        elem(i, x0, pv, r0, qv, uv);
    };
    long long i = a.begin + (long long)blockIdx.x * kBlock + threadIdx.x;
    const long long end = a.begin + a.nvec;
    auto at = [&](long long f) { return a.reverse ? last - (f - a.begin) : f; };
    if (a.light) {
        // r = r - alpha*A_z only (matrix_free_system.cpp:427-429).  Full groups first: all loads of a group are
        // issued before any store (no bounds branch inside), so 8 x 16 B per lane are in flight.
        constexpr int U = 4;
        for (; i + (U - 1) * stride < end; i += U * stride) {
            vec_t r0[U], qv[U];
#pragma unroll
            for (int k = 0; k < U; ++k) { const long long j = at(i + k * stride); r0[k] = R[j]; qv[k] = Q[j]; }
            __builtin_amdgcn_sched_barrier(0);      // keep all 2U loads ahead of the arithmetic (hipcc otherwise re-interleaves them)
#pragma unroll
            for (int k = 0; k < U; ++k) {
                vec_t rn;
#pragma unroll
                for (int j = 0; j < VEC; ++j) {
                    rn[j] = r0[k][j] - alpha * qv[k][j];
                    const double rd = (double)rn[j];
                    dd_acc_prod(s_rr, rd, rd);
                    s_rmax = fmax(s_rmax, fabs(rd));
                }
                R[at(i + k * stride)] = rn;
            }
        }
        for (; i < end; i += stride) {
            const long long j2 = at(i);
            const vec_t r0 = R[j2], qv = Q[j2];
            vec_t rn;
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
                rn[j] = r0[j] - alpha * qv[j];
                const double rd = (double)rn[j];
                dd_acc_prod(s_rr, rd, rd);
                s_rmax = fmax(s_rmax, fabs(rd));
            }
            R[j2] = rn;
        }
    } else {
        // two elements per lane and trip: both sets of loads are issued before the first store
        for (; i + stride < end; i += 2 * stride) {
            const long long j0 = at(i), j1 = at(i + stride);
            const vec_t xa = ld_pol(X + j0, nt_x), pa = ld_pol(Pp + j0, nt_p), ra = ld_pol(R + j0, nt_r), qa = ld_pol(Q + j0, nt_ap);
            const vec_t xb = ld_pol(X + j1, nt_x), pb = ld_pol(Pp + j1, nt_p), rb = ld_pol(R + j1, nt_r), qb = ld_pol(Q + j1, nt_ap);
            vec_t ua, ub; if (HAS_U) { ua = ld_pol(Uu + j0, nt_u); ub = ld_pol(Uu + j1, nt_u); }
            elem(j0, xa, pa, ra, qa, ua);
            elem(j1, xb, pb, rb, qb, ub);
        }
        if (i < end) body(i);
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
                } else { o->it = s.it + 1; o->first = 0; o->alpha = alpha_d; o->rz = rz; }
            }
        }
    }
}

// ---- phase B, recomputing variant: r -= alpha * (A_h p) with A_h p rebuilt from the stored direction -----------
// The flat update streams A p back in (one word per unknown) after the stencil launch streamed it out (another
// word).  This kernel walks the same (chunk, strip) items as the stencil, keeps three rows of p in registers and
// evaluates the 5-point formula again -- same operands, same operation order, hence the same bits as the values
// the stencil launch reduced into (Ap, p) -- so A p never touches HBM.  An iteration moves 8 words per unknown
// (stencil launch: r, p_old, x in; p, x out; this launch: p, r in; r out) instead of 9 (REL_2NORM) or 10 (MSG).
// XM selects what happens to x in this launch:
//   0  nothing (odd iterations of the two-step scheme below);
//   1  x += alpha p plus the MSG norms (|dx|, |x - u|), element-wise identical to k_update;
//   2  two-step update on even iterations k: x = (x + alpha_{k-1} p_{k-1}) + alpha_k p_k.  p_{k-1} is still intact in the
//      other direction buffer, alpha_{k-1} is in the state.  Same operations in the same order as two single updates,
//      but x is read and written once per two iterations: 7.5 words per unknown and iteration on average
//      (stencil launch 3: r, p in, p out; this launch 3 on odd iterations, 6 on even ones).
template <typename T>
struct UpdateStArgs {
    Geom g;
    WorkList wl;
    const T* p;          // current direction, ghost rows valid
    const T* pprev;      // XM == 2: the previous direction (the other ping-pong buffer)
    T* r; T* x; const T* u;
    const double* partA; int nA, strideA, esA;
    double* partB; int strideB, slotB;
    const CgState* s_in; CgState* s_out;
    int rule;
    int reverse;         // take the items from the last to the first (start where the stencil launch ended)
};

template <typename T, int VEC, int XM, bool HAS_U, int DEPTH, bool DESC>
__global__ __launch_bounds__(kBlock) void k_update_st(const UpdateStArgs<T> a) {
    typedef typename VecOf<T, VEC>::type vec_t;
    __shared__ double lds[2 * kWaves];
    const Geom& g = a.g;
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
    MI355CG_WT_BEGIN
    constexpr bool FULL = XM == 1;
    constexpr int DIR = DESC ? -1 : 1;
    struct Raw { vec_t p, r, x, u, pp; T pe; };

    // ---- per-item addressing state, as in k_stencil ----
    int nrows = 0, ystart = 0, y0 = 0, yf = 0, so_f = 0, fidx = 0, yc = 0, so_c = 0;
    int vo_own = kOob, ve_own = kOob, vo_first = kOob, vo_last = kOob;
    bool in_j[VEC];
    rsrc_t rs_p, rs_r, rs_x, rs_pp, rs_u;
    auto setup = [&](int idx) {
        const Item it = decode_item(a.wl, a.reverse ? a.wl.nitems - 1 - idx : idx);
        const int x = it.strip * (kWave * VEC) + lane * VEC;
        nrows = it.yb - it.ya + 1;
        ystart = DESC ? it.yb : it.ya;
        y0 = it.ya - 1;
        const long long base_el = row_off(g, y0) - g.base0;
        rs_p = make_rsrc(a.p + base_el);
        rs_r = make_rsrc(a.r + base_el);
        rs_x = make_rsrc(XM != 0 ? a.x + base_el : a.p + base_el);
        rs_pp = make_rsrc(XM == 2 ? a.pprev + base_el : a.p + base_el);
        rs_u = make_rsrc((FULL && HAS_U) ? a.u + base_el : a.p + base_el);
        const bool bot_item = it.ya <= g.half;
        vo_own = lane_off<T>(g, x, bot_item);
        ve_own = edge_off<T, VEC>(g, x, lane, bot_item);
        vo_first = lane_off<T>(g, x, ystart - DIR <= g.half);
        vo_last = lane_off<T>(g, x, ystart + DIR * nrows <= g.half);
        const int xint0 = bot_item ? g.half + 1 : 1;
#pragma unroll
        for (int j = 0; j < VEC; ++j) in_j[j] = (x + j >= xint0) && (x + j <= g.N - 1);
        yf = ystart - DIR;
        so_f = DESC ? (int)((row_off(g, it.yb + 1) - row_off(g, y0)) * (long long)sizeof(T)) : 0;
        fidx = -1;
        yc = ystart;
        so_c = DESC ? (int)((row_off(g, it.yb) - row_off(g, y0)) * (long long)sizeof(T)) : row_step<T>(g, y0);
    };
    // `own`: the row is one of this item's rows (its r / x / u / previous direction are needed, and its edge element)
    auto fetch = [&]() -> Raw {
        Raw w;
        const bool own = fidx >= 0 && fidx < nrows;
        const int vo = own ? vo_own : (fidx < 0 ? vo_first : (fidx == nrows ? vo_last : kOob));
        const int vown = own ? vo_own : kOob;
        w.p = buf_load<vec_t>(rs_p, vo, so_f);
        w.r = buf_load<vec_t>(rs_r, vown, so_f);
        if (XM != 0) w.x = buf_load<vec_t>(rs_x, vown, so_f);
        if (XM == 2) w.pp = buf_load<vec_t>(rs_pp, vown, so_f);
        if (FULL && HAS_U) w.u = buf_load<vec_t>(rs_u, vown, so_f);
        w.pe = buf_load<T>(rs_p, own ? ve_own : kOob, so_f);
        if (DESC) { so_f -= row_step<T>(g, yf - 1); --yf; } else { so_f += row_step<T>(g, yf); ++yf; }
        ++fidx;
        return w;
    };

    // first item's leading rows requested before the prologue (see k_stencil)
    int idx = blockIdx.x * kWaves + wave;
    bool have = idx < a.wl.nitems;
    Raw wb, c, q[DEPTH];                   // c: centre row
    if (have) {
        setup(idx);
        wb = fetch(); c = fetch();
#pragma unroll
        for (int k = 0; k < DEPTH; ++k) q[k] = fetch();
    }

    const StateLite s = load_state_lite(a.s_in);
    if (s.done) { if (blockIdx.x == 0 && threadIdx.x == 0) copy_state(a.s_out, a.s_in); return; }
    double alpha_d, rz = 0.0;
    {
        const double pap = dd_value(reduce_parts_dd(a.partA + FA_PAP * a.strideA, a.partA + (FA_PAP + FA_LO) * a.strideA, a.nA, a.esA, lds));
        if (a.rule == 0) {
            rz = dd_value(reduce_parts_dd(a.partA + FA_RZ * a.strideA, a.partA + (FA_RZ + FA_LO) * a.strideA, a.nA, a.esA, lds));
            alpha_d = rz / pap;                       // msg_solver.cpp:102
        } else {
            alpha_d = s.rr / pap;                     // matrix_free_system.cpp:419
        }
    }
    const T alpha = (T)alpha_d;
    MI355CG_WT_MID
    const T alpha_prev = (T)s.alpha;       // XM == 2: step length of the previous iteration (0 after init / resume)
    const T cA = (T)g.A, cxk = (T)g.xk, cyk = (T)g.yk;
    dd s_rr = dd_zero(), s_d2 = dd_zero(), s_e2 = dd_zero();
    double s_rmax = 0, s_dmax = 0, s_emax = 0;

    while (have) {
        vec_t p_b = wb.p, p_a;
        for (int i0 = 0; i0 < nrows; i0 += DEPTH) {
#pragma unroll
            for (int k = 0; k < DEPTH; ++k) {
                const int i = i0 + k;
                if (i < nrows) {
                    const Raw w = q[k];
                    q[k] = fetch();
                    p_a = w.p;

                    T left0 = lane_below(c.p[VEC - 1]);
                    T rightL = lane_above(c.p[0]);
                    if (lane == 0) left0 = c.pe;
                    if (lane == kWave - 1) rightL = c.pe;

                    const vec_t& top = DESC ? p_b : p_a;
                    const vec_t& bot = DESC ? p_a : p_b;
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
                        if (XM == 2) xn[j] = (c.x[j] + alpha_prev * c.pp[j]) + alpha * cc;   // two x = x + alpha*z steps
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
                    buf_store(rn, rs_r, vo_own, so_c);
                    if (XM != 0) buf_store(xn, rs_x, vo_own, so_c);
                    if (DESC) { so_c -= row_step<T>(g, yc - 1); --yc; } else { so_c += row_step<T>(g, yc); ++yc; }
                    p_b = c.p; c = w;
                }
            }
        }
        idx += gridDim.x * kWaves;
        have = idx < a.wl.nitems;
        if (have) {
            setup(idx);
            wb = fetch(); c = fetch();
#pragma unroll
            for (int k = 0; k < DEPTH; ++k) q[k] = fetch();
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
        if (blockIdx.x == 0) {
            copy_state(a.s_out, a.s_in);
            CgState* o = a.s_out;
            o->it = s.it + 1; o->first = 0; o->alpha = alpha_d; o->rz = rz;
        }
    }
}

// 2-D variant of the update: same (chunk, strip) work items as the stencil, marched in the
// opposite direction (see k_stencil).  Element-wise arithmetic identical to k_update.
template <typename T>
struct Update2DArgs {
    Geom g;
    WorkList wl;
    UpdateArgs<T> u;
};

template <typename T, int VEC, bool HAS_U, int UNROLL, bool DESC>
__global__ __launch_bounds__(kBlock) void k_update2d(const Update2DArgs<T> aa) {
    typedef typename VecOf<T, VEC>::type vec_t;
    __shared__ double lds[2 * kWaves];
    const UpdateArgs<T>& a = aa.u;
    const Geom& g = aa.g;
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
    CgState s;
    double alpha_d = 0.0, rz = 0.0;
    if (a.init) {
        s = CgState{}; s.first = 1; s.it = 0;
    } else {
        s = *a.s_in;
        if (s.done) { if (blockIdx.x == 0 && threadIdx.x == 0) *a.s_out = s; return; }
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
    constexpr int DIR = DESC ? -1 : 1;

    for (int item = blockIdx.x * kWaves + wave; item < aa.wl.nitems; item += gridDim.x * kWaves) {
        const Item it = decode_item(aa.wl, item);
        const int x = it.strip * (kWave * VEC) + lane * VEC;
        const bool xin = x < g.xlim;
        const int nrows = it.yb - it.ya + 1;
        const int ystart = DESC ? it.yb : it.ya;
        for (int i0 = 0; i0 < nrows; i0 += UNROLL) {
            vec_t x0[UNROLL], pv[UNROLL], r0[UNROLL], qv[UNROLL], uv[UNROLL];
            long long off[UNROLL];
            bool ok[UNROLL];
#pragma unroll
            for (int k = 0; k < UNROLL; ++k) {
                const int y = ystart + DIR * (i0 + k);
                ok[k] = (i0 + k < nrows) && xin && x >= (y <= g.half ? g.cb : 0);
                off[k] = row_off(g, y) - g.base0 + x;
                if (ok[k]) {
                    x0[k] = *reinterpret_cast<const vec_t*>(a.x + off[k]);
                    pv[k] = *reinterpret_cast<const vec_t*>(a.p + off[k]);
                    r0[k] = *reinterpret_cast<const vec_t*>(a.r + off[k]);
                    qv[k] = *reinterpret_cast<const vec_t*>(a.ap + off[k]);
                    if (HAS_U) uv[k] = *reinterpret_cast<const vec_t*>(a.u + off[k]);
                }
            }
#pragma unroll
            for (int k = 0; k < UNROLL; ++k) {
                if (ok[k]) {
                    vec_t xn, rn;
#pragma unroll
                    for (int j = 0; j < VEC; ++j) {
                        xn[j] = x0[k][j] + alpha * pv[k][j];        // x = x + alpha*z     msg_solver.cpp:105-107
                        rn[j] = r0[k][j] - alpha * qv[k][j];        // r = r - alpha*A_z   msg_solver.cpp:110-112
                        const double rd = (double)rn[j];
                        dd_acc_prod(s_rr, rd, rd);
                        s_rmax = fmax(s_rmax, fabs(rd));
                        const double dx = (double)(xn[j] - x0[k][j]);   // diff = x - x_prev   :124-127
                        s_dmax = fmax(s_dmax, fabs(dx));
                        dd_acc_prod(s_d2, dx, dx);
                        if (HAS_U) {
                            const double ee = (double)(xn[j] - uv[k][j]);   // error = x - u    :132-136
                            s_emax = fmax(s_emax, fabs(ee));
                            dd_acc_prod(s_e2, ee, ee);
                        }
                    }
                    *reinterpret_cast<vec_t*>(a.x + off[k]) = xn;
                    *reinterpret_cast<vec_t*>(a.r + off[k]) = rn;
                }
            }
        }
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
            CgState o = s;
            if (!a.init) { o.it = s.it + 1; o.first = 0; o.alpha = alpha_d; o.rz = rz; }
            *a.s_out = o;
        }
    }
}

// ---- end-of-chunk check: same decision as the next stencil prologue, without advancing -------------
struct CheckArgs {
    const double* partB; int nB, strideB, esB;
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
    const Decision d = reduce_and_decide(s, a.rp, a.partB, a.nB, a.strideB, a.esB, 1, lds);
    if (threadIdx.x == 0) write_state_after_decision(a.summary, a.hist, a.s_in, s, d);
}

// x += alpha * p over the owned range: the x update still pending when an XUPD loop ends.
template <typename T>
__global__ __launch_bounds__(kBlock) void k_flush_x(long long begin, long long len, T* x, const T* p, T alpha) {
    const long long stride = (long long)gridDim.x * kBlock;
    for (long long i = begin + (long long)blockIdx.x * kBlock + threadIdx.x; i < begin + len; i += stride) x[i] = x[i] + alpha * p[i];
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
// Packed index i (relative to the first owned row) of the reference's unknown vector
// (grid_system.cpp:84-111) <-> node (x, y) <-> storage offset.
struct PackGeom { Geom g; long long pk_begin, pk_len, bottom_size; };

__device__ inline long long packed_to_storage(const PackGeom& pg, long long i_global) {
    const Geom& g = pg.g;
    int x, y;
    if (i_global < pg.bottom_size) { const int w = g.half - 1; y = (int)(i_global / w) + 1; x = (int)(i_global - (long long)(y - 1) * w) + g.half + 1; }
    else { const long long j = i_global - pg.bottom_size; const int w = g.N - 1; y = (int)(j / w) + g.half + 1; x = (int)(j - (long long)(y - g.half - 1) * w) + 1; }
    return row_off(g, y) - g.base0 + x;
}
template <typename T>
__global__ __launch_bounds__(kBlock) void k_unpack(const PackGeom pg, const double* __restrict__ packed, T* __restrict__ storage) {
    const long long stride = (long long)gridDim.x * kBlock;
    for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < pg.pk_len; i += stride)
        storage[packed_to_storage(pg, pg.pk_begin + i)] = (T)packed[i];
}
template <typename T>
__global__ __launch_bounds__(kBlock) void k_pack(const PackGeom pg, const T* __restrict__ storage, double* __restrict__ packed) {
    const long long stride = (long long)gridDim.x * kBlock;
    for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < pg.pk_len; i += stride)
        packed[i] = (double)storage[packed_to_storage(pg, pg.pk_begin + i)];
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

}  // namespace mi355cg
