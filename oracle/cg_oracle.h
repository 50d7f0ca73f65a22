/*
 * cg_oracle.h -- CPU restatement (plain C) of the reference's matrix-free 5-point-stencil
 * operator and its two conjugate-gradient loops.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load it, and only as the checker / the timed CPU baseline.
 * The product path (iterative_solvers_amd/, libmi355cg.so) never links, imports or calls it.
 *
 * Pinning status: PINNED against the reference's own known-answer data (check.py's 16x16
 * operator, check_debug.py's right-hand side, py_debug.txt's two-iteration CG trace; see
 * tests/golden/ and tests/test_oracle_golden.py).  The reference's C++ path itself is
 * unbuildable in this image (needs Kokkos 4.0.01 headers fetched from the network; the dead
 * matrix_free_system.cpp only compiles next to a stand-in header), so there is no oracle/_ref.
 *
 * Every function cites the reference file:line it restates (paths relative to the reference
 * checkout).  Arithmetic order is the reference's; build with -ffp-contract=off.
 */
#ifndef CG_ORACLE_H
#define CG_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

typedef struct og_grid {
    int n, m;                 /* intervals in x, y (matrix_free_system.hpp:14)                 */
    double a, b, c, d;        /* domain [a,b]x[c,d] (:15)                                      */
    double x_step, y_step;    /* (:16)                                                         */
    double A, x_k, y_k;       /* stencil coefficients (:17)                                    */
    int size;                 /* number of unknowns                                            */
} og_grid;

/* stop criteria of MSGSolver, solver/msg_solver.hpp:9-15 (same numeric order) */
enum { OG_STOP_ITERATIONS = 0, OG_STOP_PRECISION = 1, OG_STOP_RESIDUAL = 2,
       OG_STOP_EXACT_ERROR = 3, OG_STOP_INTERRUPTED = 4 };

typedef void (*og_iter_cb)(void *user, int it, double precision, double residual, double error);

typedef struct og_msg_result {
    int iterations;
    int converged;
    int stop_reason;
    double final_residual_norm;   /* max-norm of the recursive residual (msg_solver.cpp:188)   */
    double final_precision;       /* max-norm of x_n - x_{n-1}          (:189)                 */
    double final_error_norm;      /* max-norm of x - u                  (:190)                 */
    double r_norm2;               /* Euclidean norm of the recursive residual (:120)           */
    double initial_r_norm2;       /* (:44)                                                     */
} og_msg_result;

typedef struct og_mf_result {
    int iterations;
    int converged;
    double r_norm;                /* sqrt(new_r_dot_r) of the last iteration (:441)            */
    double initial_r_norm;        /* (:400)                                                    */
} og_mf_result;

/* geometry / setup ------------------------------------------------------------------------- */
void og_grid_init(og_grid *g, int m, int n, double a, double b, double c, double d);
int  og_is_boundary(const og_grid *g, int x, int y);
int  og_position(const og_grid *g, int x, int y);          /* -1 where the reference throws   */
void og_rhs(const og_grid *g, double *rhs);
void og_true_solution(const og_grid *g, double *u);
void og_node_coords(const og_grid *g, double *xs, double *ys);

/* operator --------------------------------------------------------------------------------- */
void og_apply(const og_grid *g, const double *x, double *y);

/* CSR assembly as GridSystem::initiate_matrix does it (row order diag,left,right,top,bottom).
 * row_map has size+1 entries, entries/values have room for 5*size.  Returns nnz.             */
long og_assemble_csr(const og_grid *g, int *row_map, int *entries, double *values);

/* solvers ---------------------------------------------------------------------------------- */
/* MatrixFreeSolver::solve, solver/matrix_free_system.cpp:383-482.
 * diagnostics!=0 computes the per-iteration precision / true residual / error norms exactly as
 * the reference does (second apply per iteration) and fires cb; diagnostics==0 skips that
 * unobservable work (same x, same iteration count).                                          */
void og_mf_solve(const og_grid *g, const double *b, const double *true_solution,
                 double eps, int max_iterations, int diagnostics,
                 og_iter_cb cb, void *user, double *x_out, og_mf_result *res);

/* MSGSolver::solve, solver/msg_solver.cpp:10-212.  true_solution may be NULL (extent 0).
 * stop_flag may be NULL; it is polled once per iteration like msg_solver.cpp:82.             */
void og_msg_solve(const og_grid *g, const double *b, const double *true_solution,
                  double eps_precision, double eps_residual, double eps_exact_error,
                  int max_iterations, og_iter_cb cb, void *user, const volatile int *stop_flag,
                  double *x_out, double *r_out, og_msg_result *res);

/* helpers exposed for tests */
double og_dot(const double *a, const double *b, long n);       /* serial ascending, init 0.0  */
typedef struct { int iterations, outer, converged; double rnorm, bnorm; } og_mixed_result;
/* tests only: CPU statement of the library's mixed-precision algorithm (config 3; see cg_oracle.c) */
void og_mixed_solve(const og_grid *g, const double *b, double eps, int max_iterations, double inner_eps, double *x, og_mixed_result *res);
void og_set_exact_dots(int on);   /* tests only: inner products as if in twice the working precision (see cg_oracle.c) */
double og_max_norm(const double *a, long n);

#ifdef __cplusplus
}
#endif
#endif
