// csr_kernels.h -- generic CSR path behind Solver(const KokkosCrsMatrix&, const KokkosVector&, ...)
// (solver/solver.hpp:33-39): any caller-supplied matrix, not just the grid's 5-point operator.
// Replaces KokkosSparse::spmv("N", 1, A, z, 0, A_z) (solver/msg_solver.cpp:93) with a CSR-stream kernel and reuses
// the CG state machine, the flat update kernel and the double-double reductions of cg_kernels.h.  The gather in
// A*p needs the finished direction vector, so the direction update cannot be fused into the SpMV: an iteration is
// three launches (xpay, spmv, update): 9 vector words plus 12 B per stored non-zero per row.  It is the general
// path, not the fast one.
#pragma once
#include "cg_kernels.h"

namespace mi355cg {

struct CsrView { long long n; const int* row_map; const int* entries; const double* values; };

// ---- p = r + beta*p, with the same prologue as the fused stencil (stop decision, beta, state forward) -----------
struct XpayArgs {
    long long n; const double* r; double* p;
    const double* partB; int nB, strideB, esB;
    const CgState* s_in; CgState* s_out; HistEntry* hist; RuleParams rp; int want_diag;
};
__global__ __launch_bounds__(kBlock) void k_csr_xpay(const XpayArgs a) {
    __shared__ double lds[2 * kWaves];
    const StateLite s = load_state_lite(a.s_in);
    if (s.done) { if (blockIdx.x == 0 && threadIdx.x == 0) copy_state(a.s_out, a.s_in); return; }
    const Decision d = reduce_and_decide(s, a.rp, a.partB, a.nB, a.strideB, a.esB, a.want_diag, lds);
    if (blockIdx.x == 0 && threadIdx.x == 0) write_state_after_decision(a.s_out, a.hist, a.s_in, s, d);
    if (d.done) return;
    const double beta = d.beta;
    const long long stride = (long long)gridDim.x * kBlock;
    // 16 bytes per lane (the vectors come from hipMalloc: 256-byte aligned); an odd last element goes to one thread
    typedef double v2 __attribute__((ext_vector_type(2)));
    const v2* __restrict__ R = reinterpret_cast<const v2*>(a.r);
    v2* __restrict__ P = reinterpret_cast<v2*>(a.p);
    const long long nv = a.n / 2;
    for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < nv; i += stride) {
        const v2 rv = R[i], pv = P[i];
        v2 o;
        o[0] = rv[0] + beta * pv[0]; o[1] = rv[1] + beta * pv[1];                                                          // z = r + beta*z
        P[i] = o;
    }
    if ((a.n & 1) && blockIdx.x == 0 && threadIdx.x == 0) a.p[a.n - 1] = a.r[a.n - 1] + beta * a.p[a.n - 1];
}

// ---- y = A x (CSR-stream): a block owns 256 consecutive rows, stages their products v_j * x[col_j] through LDS with
// coalesced loads, then every lane sums ITS row's products in entry order -- the same order as a serial CSR loop, so
// the result is bit-identical to the CPU oracle.  Optional partial sums (x, y) and (r, x) for the CG loop. ---------
struct SpmvArgs {
    CsrView A; const double* x; double* y;
    const double* r;                        // may be null
    const CgState* s_in;                    // may be null (plain apply); done -> nothing to do
    double* partA; int strideA;             // may be null
};
constexpr int kCsrChunk = 2048;
__global__ __launch_bounds__(kBlock) void k_csr_spmv(const SpmvArgs a) {
    __shared__ double prod[kCsrChunk];
    __shared__ double lds[2 * kWaves];
    if (a.s_in && scalar_load(&a.s_in->done)) return;
    dd acc_xy = dd_zero(), acc_rx = dd_zero();
    const long long nblk = (a.A.n + kBlock - 1) / kBlock;
    for (long long rb = blockIdx.x; rb < nblk; rb += gridDim.x) {
        const long long row0 = rb * kBlock, row = row0 + threadIdx.x;
        const long long rend = row0 + kBlock < a.A.n ? row0 + kBlock : a.A.n;
        const int jb = a.A.row_map[row0], je = a.A.row_map[rend];
        const int my_end = row < a.A.n ? a.A.row_map[row + 1] : je;
        int pos = row < a.A.n ? a.A.row_map[row] : je;
        double sum = 0.0;
        for (int base = jb; base < je; base += kCsrChunk) {
            const int lim = base + kCsrChunk < je ? base + kCsrChunk : je;
            for (int j = base + threadIdx.x; j < lim; j += kBlock) prod[j - base] = a.A.values[j] * a.x[a.A.entries[j]];
            __syncthreads();
            const int mine = my_end < lim ? my_end : lim;
            while (pos < mine) { sum += prod[pos - base]; ++pos; }
            __syncthreads();
        }
        if (row < a.A.n) {
            a.y[row] = sum;
            if (a.partA) { dd_acc_prod(acc_xy, a.x[row], sum); if (a.r) dd_acc_prod(acc_rx, a.r[row], a.x[row]); }
        }
    }
    if (a.partA) {
        const dd t0 = block_reduce_dd(acc_xy, lds), t1 = block_reduce_dd(acc_rx, lds);
        if (threadIdx.x == 0) {
            const int b = blockIdx.x, st = a.strideA;
            a.partA[FA_PAP * st + b] = t0.hi; a.partA[(FA_PAP + FA_LO) * st + b] = t0.lo;
            a.partA[FA_RZ * st + b] = t1.hi;  a.partA[(FA_RZ + FA_LO) * st + b] = t1.lo;
        }
    }
}

}  // namespace mi355cg
