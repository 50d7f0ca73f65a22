/*
 * mi355cg.h -- C ABI of libmi355cg.so: the MI355X (gfx950) matrix-free conjugate-gradient
 * path for the 2-D Dirichlet Poisson problem on the reference's L-shaped grid.
 *
 * This is the drop-in boundary underneath the reference's C++ plug-in interfaces.  Each entry
 * point names the reference interface it replaces (paths relative to the reference checkout):
 *
 *   mi355cg_create / _destroy      GridSystem::GridSystem(m,n,a,b,c,d)      solver/grid_system.cpp:301-322
 *                                  MatrixFreeSystem::MatrixFreeSystem       solver/matrix_free_system.cpp:144-159
 *   mi355cg_size                   MatrixFreeSystem::size / matrix.numRows  solver/matrix_free_system.hpp:66
 *   mi355cg_get_rhs                GridSystem::get_rhs / MatrixFreeSystem::get_rhs   grid_system.h:73, matrix_free_system.hpp:49
 *   mi355cg_get_true_solution      get_true_solution_vector                 grid_system.cpp:276-299, matrix_free_system.cpp:162-199
 *   mi355cg_get_node_coords        GridSystem::get_x_coords/get_y_coords    grid_system.cpp:189-190,235-236
 *   mi355cg_set_rhs                Solver(a, b, ...) caller-supplied b      solver/solver.hpp:33-39
 *   mi355cg_apply                  MatrixFreeSystem::apply(x, y)            solver/matrix_free_system.cpp:203-340
 *                                  KokkosSparse::spmv("N",1,A,z,0,A_z)      solver/msg_solver.cpp:93
 *   mi355cg_solve                  MSGSolver::solve(true_solution)          solver/msg_solver.cpp:10-212   (rule MSG_MAXNORM)
 *                                  MatrixFreeSolver::solve(true_solution)   solver/matrix_free_system.cpp:383-482 (rule REL_2NORM)
 *   mi355cg_get_solution/_residual DirichletSolver::getSolution, computeResidual (A x - b)   solver/dirichlet_solver.cpp:147-161,183-191
 *   mi355cg_iter_cb                Solver::setIterationCallback             solver/solver.hpp:46-50
 *   stop_flag                      MSGSolver::requestStop (atomic flag)     solver/msg_solver.hpp:35,76; msg_solver.cpp:82-87
 *
 * Plain pointers and sizes only; no C++/torch types.  All host vectors are in the reference's
 * PACKED unknown order (bottom-right block row-major, then the upper block row-major;
 * grid_system.cpp:84-111) and are caller-owned.  Every function returns MI355CG_OK (0) or an
 * error code; mi355cg_last_error() gives the text (thread-local).  There is no CPU fallback:
 * without a usable HIP device every compute entry point fails with MI355CG_ERR_HIP.
 */
#ifndef MI355CG_H
#define MI355CG_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MI355CG_OK            0
#define MI355CG_ERR_INVALID   1   /* bad argument (std::invalid_argument in the C++ wrapper)    */
#define MI355CG_ERR_HIP       2   /* HIP runtime failure / no device (std::runtime_error)       */
#define MI355CG_ERR_STATE     3   /* call order (e.g. get_solution before solve)                */

/* storage / arithmetic type of the CG vectors */
#define MI355CG_F64           0   /* fp64 storage and arithmetic: the parity path               */
#define MI355CG_F32_MIXED     1   /* fp32 inner CG, fp64 residual refinement (no reference twin) */

/* stop rule */
#define MI355CG_RULE_MSG_MAXNORM  0   /* MSGSolver: absolute max-norm criteria, strict <        */
#define MI355CG_RULE_REL_2NORM    1   /* MatrixFreeSolver: ||r||_2 > eps * ||r0||_2             */

/* StopCriterion, solver/msg_solver.hpp:9-15 (same order) */
#define MI355CG_STOP_ITERATIONS   0
#define MI355CG_STOP_PRECISION    1
#define MI355CG_STOP_RESIDUAL     2
#define MI355CG_STOP_EXACT_ERROR  3
#define MI355CG_STOP_INTERRUPTED  4

typedef struct mi355cg_ctx *mi355cg_handle;

/* (iteration, ||x_n - x_{n-1}||, ||residual||, ||x - u||), invoked on the thread that called
 * mi355cg_solve.  MSG rule: max-norms, at it = 0, 1, every callback_every-th and the final one
 * (msg_solver.cpp:75-77,172-183,193-195).  REL_2NORM rule with diagnostics: 2-norms, the TRUE
 * residual b - A x, every iteration, 0-based index (matrix_free_system.cpp:466-468). */
typedef void (*mi355cg_iter_cb)(void *user, int iteration, double precision, double residual, double error);

typedef struct mi355cg_params {
    int    rule;               /* MI355CG_RULE_*                                                 */
    int    max_iterations;     /* Solver::maxIterations                                          */
    double eps_precision;      /* MSG: <= 0 disables (msg_solver.cpp:144)                        */
    double eps_residual;       /* MSG: <= 0 disables (:151)                                      */
    double eps_exact_error;    /* MSG: <= 0 disables (:158)                                      */
    double eps_rel;            /* REL_2NORM: eps of matrix_free_system.cpp:409                   */
    int    use_true_solution;  /* MSG: true_solution.extent(0) > 0 (:64,132,158)                 */
    int    callback_every;     /* MSG cadence (reference: 100); 0 = only it 0/1/final            */
    int    diagnostics;        /* REL_2NORM: reproduce the per-iteration diagnostics + callback  */
    int    sync_every;         /* iterations enqueued between host polls; 0 = automatic          */
    int    fixed_iterations;   /* bench mode: ignore every convergence test, run max_iterations  */
    double inner_eps;          /* F32_MIXED: relative tolerance of each fp32 inner solve (0 = 1e-4) */
} mi355cg_params;

typedef struct mi355cg_results {
    int    iterations;
    int    converged;
    int    stop_reason;           /* MI355CG_STOP_*                                              */
    double final_residual_norm;   /* MSG: max-norm of the recursive residual (msg_solver.cpp:188) */
    double final_precision;       /* MSG: max-norm of x_n - x_{n-1} (:189); DBL_MAX if none      */
    double final_error_norm;      /* MSG: max-norm of x - u (:190); DBL_MAX without u            */
    double r_norm2;               /* Euclidean norm of the recursive residual                    */
    double initial_r_norm2;       /* ||r0||_2                                                    */
    double solve_seconds;         /* wall time of the device loop (init .. last poll)            */
    double refine_true_rel;       /* F32_MIXED: final fp64 ||b-Ax||_2/||b||_2, else 0            */
    int    refine_outer;          /* F32_MIXED: outer refinement steps, else 0                   */
    double loop_seconds;          /* device time of the iterations alone (HIP events on the solve stream: after the
                                     initialisation pass .. after the last launch); 0 where not measured          */
} mi355cg_results;

/* ---- lifetime -------------------------------------------------------------------------------- */
/* Argument order follows DirichletSolver(n, m, a, b, c, d) (dirichlet_solver.cpp:11); only
 * n == m, even, >= 6 is accepted (the reference's index map is only consistent there).         */
int  mi355cg_create(int n, int m, double a, double b, double c, double d,
                    int dtype, int device, mi355cg_handle *out);
/* Generic operator: any caller-supplied CSR matrix (int32 row_map[nrows+1], entries, fp64 values), the contract of
 * Solver(const KokkosCrsMatrix& a, const KokkosVector& b, ...) (solver/solver.hpp:33-39).  Vectors of such a handle are
 * plain length-nrows arrays; set_rhs / set_true_solution / apply / solve / get_solution work as on a grid handle
 * (both stop rules; no per-iteration diagnostics).  A x = KokkosSparse::spmv("N",1,A,x,0,y) (solver/msg_solver.cpp:93),
 * summed per row in entry order. */
int  mi355cg_create_csr(long long nrows, const int *row_map, const int *entries, const double *values,
                        int device, mi355cg_handle *out);
int  mi355cg_set_true_solution(mi355cg_handle h, const double *u);   /* u of the error criterion / norm (MSG rule) */
void mi355cg_destroy(mi355cg_handle h);
const char *mi355cg_last_error(void);
const char *mi355cg_version(void);

/* ---- setup data (host, packed order, length mi355cg_size) ------------------------------------ */
long long mi355cg_size(mi355cg_handle h);
int  mi355cg_get_rhs(mi355cg_handle h, double *out);
int  mi355cg_get_true_solution(mi355cg_handle h, double *out);
int  mi355cg_get_node_coords(mi355cg_handle h, double *xs, double *ys);
int  mi355cg_set_rhs(mi355cg_handle h, const double *b);
/* Opt-in (replaces the host loops of GridSystem::calculate_value / get_true_solution_vector,
 * solver/grid_system.cpp:45-67,276-299): regenerate the right-hand side and the exact solution of the handle's cells on
 * the GPU.  Same expression order; exp() is the device library's (<= 1 ulp from glibc's), so the vectors may differ from
 * the reference's in the last bit, which is why mi355cg_create does NOT do this by default.                              */
int  mi355cg_setup_on_device(mi355cg_handle h);

/* ---- operator -------------------------------------------------------------------------------- */
int  mi355cg_apply(mi355cg_handle h, const double *x, double *y);            /* host buffers   */
int  mi355cg_apply_device(mi355cg_handle h, const double *x_dev, double *y_dev); /* packed, device */

/* ---- solver ---------------------------------------------------------------------------------- */
void mi355cg_default_params(mi355cg_params *p, int rule);
int  mi355cg_solve(mi355cg_handle h, const mi355cg_params *params,
                   mi355cg_iter_cb cb, void *user, const volatile int *stop_flag,
                   mi355cg_results *out);
int  mi355cg_get_solution(mi355cg_handle h, double *x);         /* packed x of the last solve   */
int  mi355cg_get_recursive_residual(mi355cg_handle h, double *r);
int  mi355cg_get_true_residual(mi355cg_handle h, double *ax_minus_b); /* A x - b, one more apply */

/* ---- measurement hooks (bench.py) ------------------------------------------------------------ */
/* Per-kernel device time of the last mi355cg_solve, measured with HIP events on the solve
 * stream when profiling was enabled.  kernel: 0 = fused stencil (A'), 1 = fused update (B).   */
int  mi355cg_set_profiling(mi355cg_handle h, int enable);
int  mi355cg_get_kernel_time(mi355cg_handle h, int kernel, double *avg_ms, long long *launches);
/* launch geometry: bytes of storage per vector, padded length, grid sizes (for DESIGN/bench)   */
int  mi355cg_get_layout(mi355cg_handle h, long long *padded_len, int *pitch_bottom, int *pitch_upper,
                        int *grid_stencil, int *grid_update, int *rows_per_item);

/* ---- multi-GPU: one context per rank, one contiguous slab of grid rows per context --------------
 * The reference is single-process (no collectives exist in it); this is the scaling
 * surface.  The grid is cut into row slabs balanced by unknown count.  The library runs the
 * kernels; the caller (iterative_solvers_amd/distributed.py over torch.distributed = RCCL)
 * moves what crosses ranks each phase: the per-rank record = reduced partials, optionally followed
 * by the rank's two boundary rows (mi355cg_dist_sums_ptr -> all_gather -> gathered_* arguments,
 * mi355cg_dist_scatter_ghosts), or the boundary rows as point-to-point messages (mi355cg_dist_halo).
 * Every rank reduces the gathered partials in rank order, so all ranks take identical decisions.
 * All dist calls are asynchronous on `stream`, a hipStream_t taken literally (NULL = HIP's default
 * stream, which is torch's default stream too), so they order with the caller's collectives.
 * Host vectors of a slab context (get_rhs, get_solution, ...) cover only its owned packed range. */
int  mi355cg_slab_rows(int n, int world, int rank, int *y_lo, int *y_hi);
/* a part of a 2-D decomposition: rows [y_lo, y_hi] x columns [x_lo, x_hi); x-cuts are multiples of 128 (one wave's strip)   */
int  mi355cg_create_part(int n, int m, double a, double b, double c, double d, int dtype, int device,
                         int y_lo, int y_hi, int x_lo, int x_hi, mi355cg_handle *out);
/* out2[0] = sum of v, out2[1] = sum of v^2 over the handle's own cells, accumulated in double-double on the device (the value
 * does not depend on the decomposition).  which: 0 x, 1 recursive residual, 2 right-hand side, 3 exact solution.             */
int  mi355cg_checksum(mi355cg_handle h, int which, double *out2);
int  mi355cg_create_slab(int n, int m, double a, double b, double c, double d, int dtype, int device,
                         int y_lo, int y_hi, mi355cg_handle *out);
int  mi355cg_owned_range(mi355cg_handle h, long long *packed_begin, long long *packed_len, int *y_lo, int *y_hi);
int  mi355cg_dist_begin(mi355cg_handle h, const mi355cg_params *params, void *stream);
int  mi355cg_dist_reduce(mi355cg_handle h, int which /*0 stencil, 1 update*/, int with_rows, void *stream);
int  mi355cg_dist_sums_ptr(mi355cg_handle h, int which, void **dev_ptr, int *count /*record width*/);
int  mi355cg_dist_record_layout(mi355cg_handle h, int *header, int *row_slot, int *width);   /* doubles */
int  mi355cg_dist_scatter_ghosts(mi355cg_handle h, int vector /*0 r, 1 current direction*/,
                                 const double *gathered_records, int nranks, int rank, void *stream);
int  mi355cg_dist_stencil(mi355cg_handle h, const double *gathered_update_sums, int nranks, int estride,
                          int rows /*0 all, 1 interior, 2 edge rows*/, void *stream);
int  mi355cg_dist_flip(mi355cg_handle h);          /* once per stencil phase: new direction becomes current */
/* rows as in mi355cg_dist_stencil (a full update phase is {0} or {1, 2}).  The default update rebuilds A p from the
 * stored direction, so its first and last owned row read the direction's ghost rows -- which the stencil launch of a
 * slab keeps up to date by itself (it recomputes p_new on its halo anyway and stores it there, bit-identical to the
 * neighbour's rows): the direction never has to cross ranks and mi355cg_dist_update_reads_ghosts() returns 0.        */
int  mi355cg_dist_update(mi355cg_handle h, const double *gathered_stencil_sums, int nranks, int estride,
                         int rows /*0 all, 1 interior, 2 edge rows*/, void *stream);
int  mi355cg_dist_update_reads_ghosts(mi355cg_handle h);   /* non-zero: the caller must deliver the direction halo first (never, see above) */
int  mi355cg_dist_check(mi355cg_handle h, const double *gathered_update_sums, int nranks, int estride, void *stream);
int  mi355cg_dist_summary(mi355cg_handle h, mi355cg_results *out, int *done);   /* after a stream sync */
int  mi355cg_dist_finish(mi355cg_handle h, void *stream);   /* once after the loop: flush the pending x update */
int  mi355cg_dist_history(mi355cg_handle h, int iteration, double *precision, double *residual, double *error);
int  mi355cg_dist_halo(mi355cg_handle h, int vector /*0 r, 1 current direction*/,
                       void **send_lo, void **recv_lo, long long *n_send_lo,
                       void **send_hi, void **recv_hi, long long *n_send_hi);
int  mi355cg_dist_halo_recv_counts(mi355cg_handle h, long long *n_from_lo, long long *n_from_hi);

/* ---- teams: the native multi-GPU loop (csrc/team.h) ---------------------------------------------------------------
 * A team = a decomposition of the grid into `world` parts + a transport.  MI355CG_DECOMP_ROWS: row slabs balanced by
 * unknown count.  MI355CG_DECOMP_2D: (world/2) x 2 blocks -- BASELINE config 4's "2 x 2" for world = 4: y-cuts where the
 * slabs hold equal unknowns, every slab cut in x where ITS unknowns halve, x-cuts on 128-column strip boundaries.
 * Per iteration every part needs every part's 16-double record of partial sums twice, and its neighbours' boundary rows /
 * columns of the residual once.  Default transport: a one-workgroup reducer launch beside every producer launch stores the part's
 * record straight into every other part's mailbox (peer memory over xGMI, IPC-mapped across processes); the consumer launch reduces
 * its own partials and polls its own mailbox for the others; the halo is pushed into the neighbours' ghost cells by one small launch
 * whose last workgroup announces it with a 64-bit store the neighbour's stream waits for.
 * RCCL (ncclAllGather / ncclSend / ncclRecv) carries the bootstrap and is the fallback for both (environment:
 * MI355CG_TEAM_RECORDS = auto | rccl | mailbox | events, MI355CG_TEAM_WAIT = auto | kernel | stream, MI355CG_TEAM_HALO = auto |
 * inline | stream | push, MI355CG_TEAM_TIMEOUT_MS; mi355cg_team_describe says what a team uses).  Results are bit-identical to
 * the single-GPU solve for every decomposition and transport.  The reference has no counterpart (single process,
 * solver/solver.hpp:13 HostSpace only); the entry points mirror mi355cg_create / mi355cg_solve.                          */
#define MI355CG_DECOMP_ROWS 0
#define MI355CG_DECOMP_2D   1
typedef struct mi355cg_team_s *mi355cg_team;
typedef struct mi355cg_halo_msg {
    int id;          /* position in the global message order (every rank enumerates the same list)     */
    int peer;        /* the other part                                                                  */
    int send;        /* 1: this part sends, 0: it receives                                              */
    int kind;        /* 0: cells [x0, x1) of row y0;  1: column x0 over rows y0..y1                     */
    int y0, y1, x0, x1;
    long long count; /* doubles                                                                         */
} mi355cg_halo_msg;
/* pure host arithmetic (no GPU needed): the box of `rank`, and its halo messages per iteration */
int  mi355cg_decompose(int n, int world, int decomp, int rank, int *y_lo, int *y_hi, int *x_lo, int *x_hi);
int  mi355cg_halo_plan(int n, int world, int decomp, int rank, int max_msgs, int *n_msgs, mi355cg_halo_msg *msgs);
/* for tests (host arithmetic only): the work items of a part's launches.  which: 0 whole part, 1 interior, 2 edge.
 * panels: 8 ints per panel {y0, y1, first strip, strips, item rows, chunks, first item, ghost-column flags};
 * cls: [0] = number of XCD classes, [1..9] = their item boundaries                                                    */
int  mi355cg_debug_plan(int n, int world, int decomp, int rank, int which, int *n_panels, int *panels,
                        int *grid, int *n_items, int *cls);
/* LOCAL transport: this process drives all `world` parts; part r runs on devices[r % ndevices] (NULL: device 0).
 * Parts on different GPUs need peer access (xGMI).                                                                   */
int  mi355cg_team_create_local(int n, int m, double a, double b, double c, double d, int world,
                               const int *devices, int ndevices, int decomp, mi355cg_team *out);
/* RCCL transport: one process per part.  Rank 0 obtains a 128-byte id (ncclGetUniqueId) and hands it to the others by
 * any means (MPI, torch.distributed, a file); every rank then creates its side of the team (ncclCommInitRank inside).   */
int  mi355cg_team_unique_id(void *id128);
int  mi355cg_team_create_rccl(int n, int m, double a, double b, double c, double d, int world, int rank, int device,
                              const void *id128, int decomp, mi355cg_team *out);
void mi355cg_team_destroy(mi355cg_team t);
/* as mi355cg_solve (no per-iteration diagnostics).  Collective over the team: every rank calls it with the same params; the
 * callback and the stop flag may differ from rank to rank (the schedule of launches, polls and collectives depends on the
 * parameters only).  A stop request on any rank travels with that rank's next update record and every rank takes the decision
 * INTERRUPTED in the same iteration (msg_solver.cpp:82-87).  A record that does not arrive within MI355CG_TEAM_TIMEOUT_MS
 * (30 s) ends the solve with MI355CG_ERR_STATE on every rank instead of a hang; the team cannot be used after that.        */
int  mi355cg_team_solve(mi355cg_team t, const mi355cg_params *params, mi355cg_iter_cb cb, void *user,
                        const volatile int *stop_flag, mi355cg_results *out);
int  mi355cg_team_info(mi355cg_team t, int *world, int *nlocal, int *decomp, long long *size);
/* "transport=rccl records=mailbox wait=kernel halo=push split=0 ipc=1 shared_device=0 rccl_nranks=8 rccl_lib=...": what the next
 * solve of this team uses (rccl_nranks = what ncclCommCount reports for the team's communicator; 0 for a LOCAL team)        */
int  mi355cg_team_describe(mi355cg_team t, char *buf, int len);
int  mi355cg_team_part(mi355cg_team t, int local_index, mi355cg_handle *part, int *rank);
/* which as in mi355cg_checksum.  get_vector fills the entries of the caller's GLOBAL packed vector (length size) owned by
 * this process's parts; checksum covers this process's parts.                                                           */
int  mi355cg_team_get_vector(mi355cg_team t, int which, double *global_packed);
int  mi355cg_team_checksum(mi355cg_team t, int which, double *out2);
/* which: 2 right-hand side, 3 exact solution: every local part takes its entries of the caller's GLOBAL packed vector -- the b of
 * Solver(a, b, ...) (solver/solver.hpp:33-39) and the true_solution of MSGSolver::solve (msg_solver.cpp:64-72) may be anything.   */
int  mi355cg_team_set_vector(mi355cg_team t, int which, const double *global_packed);
/* MI355CG_F32_MIXED: the team's solves become mi355cg_create(..., MI355CG_F32_MIXED)'s algorithm -- fp64 iterative refinement around
 * an fp32 inner CG (REL_2NORM only) -- across the parts: BASELINE config 3 on more than one GPU.  Row slabs only (the fp32 kernels
 * march 256-column strips).  Collective.  No reference twin: the reference is fp64 only.                                          */
int  mi355cg_team_set_dtype(mi355cg_team t, int dtype);
int  mi355cg_team_setup_on_device(mi355cg_team t);                     /* mi355cg_setup_on_device for every local part */
int  mi355cg_team_set_profiling(mi355cg_team t, int enable);
int  mi355cg_team_phase_times(mi355cg_team t, double *kernel_ms, double *comm_ms, double *wall_ms);   /* per iteration */

#ifdef __cplusplus
}
#endif
#endif /* MI355CG_H */
