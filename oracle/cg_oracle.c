/*
 * cg_oracle.c -- CPU restatement of the reference's hot path.  TEST INFRASTRUCTURE ONLY
 * (see cg_oracle.h).  Build: gcc -O2 -ffp-contract=off -fPIC -shared (oracle/Makefile).
 *
 * Reference files restated (paths relative to the reference checkout):
 *   solver/matrix_free_system.cpp   geometry, RHS, apply(), MatrixFreeSolver::solve
 *   solver/grid_system.cpp          same geometry, CSR assembly order, node coordinates
 *   solver/msg_solver.cpp           MSGSolver::solve and its dot / norm / max_norm helpers
 */
#include "cg_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ---- PDE data: matrix_free_system.cpp:10-16 == grid_system.cpp:8-15 ---------------------- */
static double og_function(double x, double y) { return 4 * (x * x + y * y) * exp(x * x - y * y); }
static double og_solution(double x, double y) { return exp(x * x - y * y); }

/* ---- boundary predicates: matrix_free_system.cpp:18-36 == grid_system.cpp:17-43 ---------- */
static int is_left(const og_grid *g, int x, int y)
{
    int top = x == 0 && (y >= g->m / 2 && y <= g->m);
    int bot = x == g->n / 2 && (y >= 0 && y <= g->m / 2);
    return top || bot;
}
static int is_right(const og_grid *g, int x, int y) { (void)y; return x == g->n; }
static int is_top(const og_grid *g, int x, int y) { (void)x; return y == g->m; }
static int is_bottom(const og_grid *g, int x, int y)
{
    int right = y == 0 && (x >= g->n / 2 && x <= g->n);
    int left = y == g->m / 2 && (x >= 0 && x <= g->n / 2);
    return right || left;
}
/* matrix_free_system.cpp:65-67 */
int og_is_boundary(const og_grid *g, int x, int y)
{
    return is_left(g, x, y) || is_right(g, x, y) || is_top(g, x, y) || is_bottom(g, x, y);
}

/* matrix_free_system.cpp:57-63 */
static double calc_x(const og_grid *g, int x) { return g->a + x * g->x_step; }
static double calc_y(const og_grid *g, int y) { return g->c + y * g->y_step; }

/* matrix_free_system.cpp:84-90 (note the n/2 where m/2 is meant, kept as is) */
static int pos_upper(const og_grid *g, int x, int y) { return (y - g->n / 2 - 1) * (g->n - 1) + x - 1; }
static int pos_bottom(const og_grid *g, int x, int y) { return (g->n / 2 - 1) * (y - 1) + x - g->n / 2 - 1; }

/* matrix_free_system.cpp:69-82; the two `throw std::invalid_argument` become -1 */
int og_position(const og_grid *g, int x, int y)
{
    if (x < g->n / 2 && y < g->m / 2) return -1;
    if (x == 0 || y == 0 || x == g->n || y == g->m) return -1;
    if (y <= g->m / 2) return pos_bottom(g, x, y);
    return pos_upper(g, x, y) + pos_bottom(g, g->n - 1, g->m / 2) + 1;
}

/* matrix_free_system.cpp:93-101 */
static int system_size(const og_grid *g)
{
    int p = og_position(g, g->n - 1, g->m - 1);
    if (p < 0) return (g->n * g->m) / 2;
    return p + 1;
}

/* constructor: matrix_free_system.cpp:144-159 == grid_system.cpp:301-322; argument order (m, n, ...) */
void og_grid_init(og_grid *g, int m, int n, double a, double b, double c, double d)
{
    g->n = n; g->m = m; g->a = a; g->b = b; g->c = c; g->d = d;
    g->x_step = (b - a) / (n);
    g->y_step = (d - c) / (m);
    g->A = -2 * (1 / (g->x_step * g->x_step) + 1 / (g->y_step * g->y_step));
    g->x_k = 1 / (g->x_step * g->x_step);
    g->y_k = 1 / (g->y_step * g->y_step);
    g->size = system_size(g);
}

/* matrix_free_system.cpp:38-55 == grid_system.cpp:45-67 */
static double calc_value(const og_grid *g, int x, int y)
{
    double value = og_function(calc_x(g, x), calc_y(g, y));
    if (is_left(g, x - 1, y)) value -= g->x_k * og_solution(calc_x(g, x - 1), calc_y(g, y));
    if (is_right(g, x + 1, y)) value -= g->x_k * og_solution(calc_x(g, x + 1), calc_y(g, y));
    if (is_top(g, x, y + 1)) value -= g->y_k * og_solution(calc_x(g, x), calc_y(g, y + 1));
    if (is_bottom(g, x, y - 1)) value -= g->y_k * og_solution(calc_x(g, x), calc_y(g, y - 1));
    return value;
}

/* The reference walks "bottom-right part, then upper part" in every setup/apply routine
 * (matrix_free_system.cpp:109-140,167-196,210-339; grid_system.cpp:181-270).  phase 0/1. */
#define OG_FOR_EACH_NODE(g, x, y)                                                               \
    for (int _ph = 0; _ph < 2; ++_ph)                                                           \
        for (int y = _ph ? (g)->m / 2 + 1 : 1; _ph ? y < (g)->m : y <= (g)->m / 2; ++y)         \
            for (int x = _ph ? 1 : (g)->n / 2 + 1; x < (g)->n; ++x)

/* matrix_free_system.cpp:104-141 */
void og_rhs(const og_grid *g, double *rhs)
{
    for (int i = 0; i < g->size; ++i) rhs[i] = 0.0;
    OG_FOR_EACH_NODE(g, x, y) {
        if (og_is_boundary(g, x, y)) continue;
        int row = og_position(g, x, y);
        if (row >= 0 && row < g->size) rhs[row] = calc_value(g, x, y);
    }
}

/* matrix_free_system.cpp:162-199 (== grid_system.cpp:276-299 through node_x/y_coords) */
void og_true_solution(const og_grid *g, double *u)
{
    for (int i = 0; i < g->size; ++i) u[i] = 0.0;
    OG_FOR_EACH_NODE(g, x, y) {
        if (og_is_boundary(g, x, y)) continue;
        int row = og_position(g, x, y);
        if (row >= 0 && row < g->size) u[row] = og_solution(calc_x(g, x), calc_y(g, y));
    }
}

/* grid_system.cpp:189-190,235-236 */
void og_node_coords(const og_grid *g, double *xs, double *ys)
{
    for (int i = 0; i < g->size; ++i) { xs[i] = 0.0; ys[i] = 0.0; }
    OG_FOR_EACH_NODE(g, x, y) {
        if (og_is_boundary(g, x, y)) continue;
        int row = og_position(g, x, y);
        if (row >= 0 && row < g->size) { xs[row] = calc_x(g, x); ys[row] = calc_y(g, y); }
    }
}

/* matrix_free_system.cpp:203-340: y = A_h x, accumulation order diag, left, right, top, bottom */
void og_apply(const og_grid *g, const double *x, double *y)
{
    const int sz = g->size;
    for (int i = 0; i < sz; ++i) y[i] = 0.0;                       /* :207 */
    OG_FOR_EACH_NODE(g, xi, yi) {
        if (og_is_boundary(g, xi, yi)) continue;
        int row = og_position(g, xi, yi);
        if (!(row >= 0 && row < sz)) continue;
        y[row] += g->A * x[row];                                   /* :217 */
        if (!is_left(g, xi - 1, yi)) {                             /* :221-230 */
            int col = og_position(g, xi - 1, yi);
            if (col >= 0 && col < sz) y[row] += g->x_k * x[col];
        }
        if (!is_right(g, xi + 1, yi)) {                            /* :233-242 */
            int col = og_position(g, xi + 1, yi);
            if (col >= 0 && col < sz) y[row] += g->x_k * x[col];
        }
        if (!is_top(g, xi, yi + 1)) {                              /* :245-254 */
            int col = og_position(g, xi, yi + 1);
            if (col >= 0 && col < sz) y[row] += g->y_k * x[col];
        }
        if (!is_bottom(g, xi, yi - 1)) {                           /* :257-266 */
            int col = og_position(g, xi, yi - 1);
            if (col >= 0 && col < sz) y[row] += g->y_k * x[col];
        }
    }
}

/* grid_system.cpp:157-274 (add_matrix_entry :114-119, finalize_matrix prefix sum :122-127) */
long og_assemble_csr(const og_grid *g, int *row_map, int *entries, double *values)
{
    long nnz = 0;
    for (int i = 0; i <= g->size; ++i) row_map[i] = 0;
    OG_FOR_EACH_NODE(g, x, y) {
        if (og_is_boundary(g, x, y)) continue;
        int row = og_position(g, x, y);
#define OG_ADD(col, val) do { entries[nnz] = (col); values[nnz] = (val); ++nnz; row_map[row + 1]++; } while (0)
        OG_ADD(row, g->A);
        if (!is_left(g, x - 1, y)) OG_ADD(og_position(g, x - 1, y), g->x_k);
        if (!is_right(g, x + 1, y)) OG_ADD(og_position(g, x + 1, y), g->x_k);
        if (!is_top(g, x, y + 1)) OG_ADD(og_position(g, x, y + 1), g->y_k);
        if (!is_bottom(g, x, y - 1)) OG_ADD(og_position(g, x, y - 1), g->y_k);
#undef OG_ADD
    }
    for (int i = 1; i <= g->size; ++i) row_map[i] += row_map[i - 1];
    return nnz;
}

/* NOT the reference: the same inner product evaluated as if in twice the working precision (Dot2 of Ogita, Rump & Oishi:
 * error-free product and sum transformations, the rounding errors carried in a second double).  Switched on by the tests
 * that separate "the reference's arithmetic" from "the reference's summation order": with exact inner products every other
 * operation of the CG loop is elementwise and identical on both sides, so the GPU has to reproduce the oracle to the bit
 * at any size, while against the serial sums below it can only agree to ~ U * 2^-53. */
static int g_exact_dots = 0;
void og_set_exact_dots(int on) { g_exact_dots = on; }
static double og_dot2(const double *a, const double *b, long n)
{
    double s = 0.0, c = 0.0;
    for (long i = 0; i < n; ++i) {
        const double p = a[i] * b[i];
        const double e = fma(a[i], b[i], -p);             /* a*b = p + e exactly */
        const double t = s + p;
        const double z = t - s;
        const double q = (s - (t - z)) + (p - z);         /* s + p = t + q exactly */
        s = t;
        c += q + e;
    }
    return s + c;
}

/* std::inner_product(v1, v1+n, v2, 0.0) (matrix_free_system.cpp:364-366) and
 * MSGSolver::dot (msg_solver.cpp:215-229): serial ascending sum of products */
double og_dot(const double *a, const double *b, long n)
{
    if (g_exact_dots) return og_dot2(a, b, n);
    double result = 0.0;
    for (long i = 0; i < n; ++i) result += a[i] * b[i];
    return result;
}

/* MSGSolver::max_norm, msg_solver.cpp:247-258 */
double og_max_norm(const double *a, long n)
{
    double max_val = 0.0;
    for (long i = 0; i < n; ++i) {
        double v = fabs(a[i]);
        max_val = max_val < v ? v : max_val;          /* std::max(max_val, |a_i|) */
    }
    return max_val;
}

static double og_norm(const double *a, long n) { return sqrt(og_dot(a, a, n)); }

/* MatrixFreeSolver::solve, matrix_free_system.cpp:383-482 */
void og_mf_solve(const og_grid *g, const double *b, const double *true_solution,
                 double eps, int max_iterations, int diagnostics,
                 og_iter_cb cb, void *user, double *x, og_mf_result *res)
{
    const int n = g->size;
    double *r = malloc(sizeof(double) * n), *Ax = malloc(sizeof(double) * n);
    double *p = malloc(sizeof(double) * n), *Ap = malloc(sizeof(double) * n);
    double *prev_x = malloc(sizeof(double) * n), *tmp = malloc(sizeof(double) * n);
    int iterations;

    for (int i = 0; i < n; ++i) x[i] = 0.0;                         /* :387 */
    og_apply(g, x, Ax);                                             /* :392 */
    for (int i = 0; i < n; ++i) r[i] = 1.0 * b[i] + -1.0 * Ax[i];   /* :393, axpby :377-379 */
    memcpy(p, r, sizeof(double) * n);                               /* :396 */
    double r_norm = og_norm(r, n);                                  /* :399 */
    double initial_r_norm = r_norm;                                 /* :400 */
    memcpy(prev_x, x, sizeof(double) * n);                          /* :403 */

    for (iterations = 0; iterations < max_iterations && r_norm > eps * initial_r_norm; ++iterations) { /* :409 */
        memcpy(prev_x, x, sizeof(double) * n);                      /* :411 */
        og_apply(g, p, Ap);                                         /* :414 */
        double p_dot_Ap = og_dot(p, Ap, n);                         /* :417 */
        double r_dot_r = og_dot(r, r, n);                           /* :418 */
        double alpha = r_dot_r / p_dot_Ap;                          /* :419 */
        for (int i = 0; i < n; ++i) x[i] += alpha * p[i];           /* :422-424 */
        for (int i = 0; i < n; ++i) r[i] -= alpha * Ap[i];          /* :427-429 */
        double new_r_dot_r = og_dot(r, r, n);                       /* :432 */
        double beta = new_r_dot_r / r_dot_r;                        /* :433 */
        for (int i = 0; i < n; ++i) p[i] = r[i] + beta * p[i];      /* :436-438 */
        r_norm = sqrt(new_r_dot_r);                                 /* :441 */
        if (diagnostics) {
            for (int i = 0; i < n; ++i) tmp[i] = x[i] - prev_x[i];  /* :444-447 */
            double precision = og_norm(tmp, n);                     /* :448 */
            for (int i = 0; i < n; ++i) tmp[i] = x[i] - true_solution[i]; /* :451-454 */
            double error_norm = og_norm(tmp, n);                    /* :455 */
            og_apply(g, x, Ax);                                     /* :459 */
            for (int i = 0; i < n; ++i) tmp[i] = b[i] - Ax[i];      /* :460-462 */
            double residual_norm = og_norm(tmp, n);                 /* :463 */
            if (cb) cb(user, iterations, precision, residual_norm, error_norm); /* :466-468 */
        }
    }
    if (res) {
        res->iterations = iterations;
        res->converged = r_norm <= eps * initial_r_norm;            /* :472 */
        res->r_norm = r_norm;
        res->initial_r_norm = initial_r_norm;
    }
    free(r); free(Ax); free(p); free(Ap); free(prev_x); free(tmp);
}

/* MSGSolver::solve, msg_solver.cpp:10-212.  A z is the 5-point operator (the assembled CSR of
 * grid_system.cpp:157-274 has exactly og_apply's entries in og_apply's order; KokkosSparse::spmv's
 * intra-row summation order is third-party and unpinned). */
void og_msg_solve(const og_grid *g, const double *b, const double *true_solution,
                  double eps_precision, double eps_residual, double eps_exact_error,
                  int max_iterations, og_iter_cb cb, void *user, const volatile int *stop_flag,
                  double *x, double *r_out, og_msg_result *res)
{
    const int n = g->size;
    double *x_prev = malloc(sizeof(double) * n), *r = malloc(sizeof(double) * n);
    double *z = malloc(sizeof(double) * n), *A_z = malloc(sizeof(double) * n);
    double *tmp = malloc(sizeof(double) * n);
    int converged = 0;

    for (int i = 0; i < n; ++i) x[i] = 0.0;                         /* :33 */
    memcpy(r, b, sizeof(double) * n);                               /* :36 */
    memcpy(z, r, sizeof(double) * n);                               /* :39 */
    double r_norm = og_norm(r, n);                                  /* :42 */
    double r_max_norm = og_max_norm(r, n);                          /* :43 */
    double initial_r_norm = r_norm;                                 /* :44 */
    int it = 0;                                                     /* :47 */
    int stop_reason = OG_STOP_ITERATIONS;                           /* :53 */
    double precision_max_norm = DBL_MAX;                            /* :56-57 */
    double error_max_norm = DBL_MAX;                                /* :60-61 */
    if (true_solution) {                                            /* :64-72 */
        for (int i = 0; i < n; ++i) tmp[i] = x[i] - true_solution[i];
        error_max_norm = og_max_norm(tmp, n);
    }
    if (cb) cb(user, 0, precision_max_norm, r_max_norm, error_max_norm);   /* :75-77 */

    while (it < max_iterations) {                                   /* :80 */
        if (stop_flag && *stop_flag) {                              /* :82-87 */
            stop_reason = OG_STOP_INTERRUPTED;
            converged = 0;
            break;
        }
        memcpy(x_prev, x, sizeof(double) * n);                      /* :90 */
        og_apply(g, z, A_z);                                        /* :93 */
        double rz = og_dot(r, z, n);                                /* :96 */
        double Az_z = og_dot(A_z, z, n);                            /* :99 */
        double alpha = rz / Az_z;                                   /* :102 */
        for (int i = 0; i < n; ++i) x[i] = x[i] + alpha * z[i];     /* :105-107 */
        for (int i = 0; i < n; ++i) r[i] = r[i] - alpha * A_z[i];   /* :110-112 */
        it++;                                                       /* :115 */
        r_norm = og_norm(r, n);                                     /* :120 */
        r_max_norm = og_max_norm(r, n);                             /* :121 */
        for (int i = 0; i < n; ++i) tmp[i] = x[i] - x_prev[i];      /* :124-127 */
        precision_max_norm = og_max_norm(tmp, n);                   /* :129 */
        if (true_solution) {                                        /* :132-139 */
            for (int i = 0; i < n; ++i) tmp[i] = x[i] - true_solution[i];
            error_max_norm = og_max_norm(tmp, n);
        }
        if (eps_precision > 0 && precision_max_norm < eps_precision) {      /* :144-148 */
            converged = 1; stop_reason = OG_STOP_PRECISION; break;
        }
        if (eps_residual > 0 && r_max_norm < eps_residual) {                /* :151-155 */
            converged = 1; stop_reason = OG_STOP_RESIDUAL; break;
        }
        if (eps_exact_error > 0 && true_solution && error_max_norm < eps_exact_error) { /* :158-162 */
            converged = 1; stop_reason = OG_STOP_EXACT_ERROR; break;
        }
        double beta = (r_norm * r_norm) / (rz);                     /* :165 */
        for (int i = 0; i < n; ++i) z[i] = r[i] + beta * z[i];      /* :167-169 */
        if (it % 100 == 0 || it == 1) {                             /* :172-183 */
            if (cb) cb(user, it, precision_max_norm, r_max_norm, error_max_norm);
        }
    }
    if (cb) cb(user, it, precision_max_norm, r_max_norm, error_max_norm);  /* :193-195 */
    if (res) {
        res->iterations = it;                                       /* :187 */
        res->converged = converged;
        res->stop_reason = stop_reason;
        res->final_residual_norm = r_max_norm;                      /* :188 */
        res->final_precision = precision_max_norm;                  /* :189 */
        res->final_error_norm = error_max_norm;                     /* :190 */
        res->r_norm2 = r_norm;
        res->initial_r_norm2 = initial_r_norm;
    }
    if (r_out) memcpy(r_out, r, sizeof(double) * n);
    free(x_prev); free(r); free(z); free(A_z); free(tmp);
}


/* ---------------------------------------------------------------------------------------------------------------------
 * NOT in the reference (it has no fp32 path): a CPU statement of the mixed-precision algorithm that BASELINE config 3 asks of
 * the GPU library (csrc/mi355cg.hip solve_mixed / inner_cg_f32) -- fp32-storage CG inside fp64 iterative refinement -- so that
 * the fp32 kernels can be checked value for value against an independent implementation, not only through the fp64 residual
 * of the answer.  Arithmetic contract (stated in DESIGN.md section 3): vectors of the inner CG are float; the stencil, the
 * direction and the updates are evaluated in float in the operation order of the fp64 path; inner products are exact sums of
 * the double products of the float values, rounded once; alpha and beta are formed in double and rounded to float; the
 * refinement (x += correction, r = b - A x, its norm) is fp64. */
static void og_apply_f32(const og_grid *g, const float *x, float *y)
{
    const int sz = g->size;
    const float A = (float)g->A, xk = (float)g->x_k, yk = (float)g->y_k;
    OG_FOR_EACH_NODE(g, xi, yi) {
        if (og_is_boundary(g, xi, yi)) continue;
        int row = og_position(g, xi, yi);
        if (!(row >= 0 && row < sz)) continue;
        float v = A * x[row];
        if (!is_left(g, xi - 1, yi)) { int col = og_position(g, xi - 1, yi); if (col >= 0 && col < sz) v = v + xk * x[col]; }
        if (!is_right(g, xi + 1, yi)) { int col = og_position(g, xi + 1, yi); if (col >= 0 && col < sz) v = v + xk * x[col]; }
        if (!is_top(g, xi, yi + 1)) { int col = og_position(g, xi, yi + 1); if (col >= 0 && col < sz) v = v + yk * x[col]; }
        if (!is_bottom(g, xi, yi - 1)) { int col = og_position(g, xi, yi - 1); if (col >= 0 && col < sz) v = v + yk * x[col]; }
        y[row] = v;
    }
}
static double og_dot_f32_exact(const float *a, const float *b, long n)      /* products of floats are exact in double */
{
    double s = 0.0, c = 0.0;
    for (long i = 0; i < n; ++i) {
        const double p = (double)a[i] * (double)b[i];
        const double t = s + p, z = t - s;
        c += (s - (t - z)) + (p - z);
        s = t;
    }
    return s + c;
}
/* inner CG in float on (x = 0, r = right-hand side); returns the iterations taken */
static int og_inner_cg_f32(const og_grid *g, float *x, float *r, float *p, float *Ap, double eps, int max_iterations)
{
    const int n = g->size;
    for (int i = 0; i < n; ++i) { x[i] = 0.0f; p[i] = 0.0f; }
    double rr = og_dot_f32_exact(r, r, n), rr_prev = 0.0;
    const double r0norm = sqrt(rr);
    int it = 0;
    for (;;) {
        if (!(it < max_iterations && sqrt(rr) > eps * r0norm)) break;
        const float beta = (float)(it == 0 ? 0.0 : rr / rr_prev);
        for (int i = 0; i < n; ++i) p[i] = r[i] + beta * p[i];
        og_apply_f32(g, p, Ap);
        const double pAp = og_dot_f32_exact(Ap, p, n);
        const float alpha = (float)(rr / pAp);
        for (int i = 0; i < n; ++i) { x[i] = x[i] + alpha * p[i]; r[i] = r[i] - alpha * Ap[i]; }
        rr_prev = rr;
        rr = og_dot_f32_exact(r, r, n);
        ++it;
    }
    return it;
}
void og_mixed_solve(const og_grid *g, const double *b, double eps, int max_iterations, double inner_eps, double *x,
                    og_mixed_result *res)
{
    const int n = g->size;
    float *xf = malloc(sizeof(float) * n), *rf = malloc(sizeof(float) * n), *pf = malloc(sizeof(float) * n), *Apf = malloc(sizeof(float) * n);
    double *Ax = malloc(sizeof(double) * n), *r64 = malloc(sizeof(double) * n);
    const int saved = g_exact_dots;
    g_exact_dots = 1;                                              /* fp64 norms of the refinement: exact sums as well */
    for (int i = 0; i < n; ++i) { x[i] = 0.0; r64[i] = b[i] - 0.0; rf[i] = (float)r64[i]; }
    const double bnorm = sqrt(og_dot(r64, r64, n));
    double rnorm = bnorm;
    int total = 0, outer = 0, converged = bnorm == 0.0;
    if (inner_eps <= 0) inner_eps = 1e-4;
    while (!converged && total < max_iterations) {
        const int its = og_inner_cg_f32(g, xf, rf, pf, Apf, inner_eps, max_iterations - total);
        total += its; ++outer;
        for (int i = 0; i < n; ++i) x[i] += (double)xf[i];
        og_apply(g, x, Ax);
        for (int i = 0; i < n; ++i) { r64[i] = b[i] - Ax[i]; rf[i] = (float)r64[i]; }
        const double prev = rnorm;
        rnorm = sqrt(og_dot(r64, r64, n));
        converged = rnorm <= eps * bnorm;
        if (its == 0) break;
        if (!converged && rnorm > 0.5 * prev) break;               /* fp32 cannot improve this x any further */
    }
    g_exact_dots = saved;
    if (res) { res->iterations = total; res->outer = outer; res->converged = converged; res->rnorm = rnorm; res->bnorm = bnorm; }
    free(xf); free(rf); free(pf); free(Apf); free(Ax); free(r64);
}

/* ---- all-cores variant (timing baseline only) ----------------------------------------------------------------
 * Same per-element arithmetic as og_apply / og_mf_solve, rows distributed over OpenMP threads; the dot products use
 * an OpenMP reduction, so their summation order (hence the last bits) differs from the serial reference-faithful
 * code above.  Built only into libcg_oracle_omp.so (-fopenmp); bench.py times it as the "all host cores" figure
 * BASELINE.md section 4 calls ref-omp.  Never used as a parity checker. */
#ifdef _OPENMP
#include <omp.h>
int og_omp_threads(void) { return omp_get_max_threads(); }
void og_omp_set_threads(int n) { if (n > 0) omp_set_num_threads(n); }

static void og_apply_row(const og_grid *g, int yi, const double *x, double *y)
{
    const int sz = g->size;
    const int x0 = yi <= g->m / 2 ? g->n / 2 + 1 : 1;
    for (int xi = x0; xi < g->n; ++xi) {
        if (og_is_boundary(g, xi, yi)) continue;
        int row = og_position(g, xi, yi);
        if (!(row >= 0 && row < sz)) continue;
        double v = 0.0;
        v += g->A * x[row];
        if (!is_left(g, xi - 1, yi)) { int col = og_position(g, xi - 1, yi); if (col >= 0 && col < sz) v += g->x_k * x[col]; }
        if (!is_right(g, xi + 1, yi)) { int col = og_position(g, xi + 1, yi); if (col >= 0 && col < sz) v += g->x_k * x[col]; }
        if (!is_top(g, xi, yi + 1)) { int col = og_position(g, xi, yi + 1); if (col >= 0 && col < sz) v += g->y_k * x[col]; }
        if (!is_bottom(g, xi, yi - 1)) { int col = og_position(g, xi, yi - 1); if (col >= 0 && col < sz) v += g->y_k * x[col]; }
        y[row] = v;
    }
}
void og_apply_omp(const og_grid *g, const double *x, double *y)
{
#pragma omp parallel for schedule(static)
    for (int yi = 1; yi < g->m; ++yi) og_apply_row(g, yi, x, y);
}
static double og_dot_omp(const double *a, const double *b, long n)
{
    double result = 0.0;
#pragma omp parallel for schedule(static) reduction(+ : result)
    for (long i = 0; i < n; ++i) result += a[i] * b[i];
    return result;
}
/* MatrixFreeSolver loop without the diagnostics, all cores.  Returns iterations done. */
int og_mf_solve_omp(const og_grid *g, const double *b, double eps, int max_iterations, double *x)
{
    const int n = g->size;
    double *r = malloc(sizeof(double) * n), *p = malloc(sizeof(double) * n), *Ap = malloc(sizeof(double) * n);
#pragma omp parallel for schedule(static)
    for (int i = 0; i < n; ++i) { x[i] = 0.0; r[i] = b[i]; p[i] = b[i]; }
    double r_norm = sqrt(og_dot_omp(r, r, n));
    const double initial = r_norm;
    int it;
    for (it = 0; it < max_iterations && r_norm > eps * initial; ++it) {
        og_apply_omp(g, p, Ap);
        const double pAp = og_dot_omp(p, Ap, n), rr = og_dot_omp(r, r, n), alpha = rr / pAp;
#pragma omp parallel for schedule(static)
        for (int i = 0; i < n; ++i) { x[i] += alpha * p[i]; r[i] -= alpha * Ap[i]; }
        const double rr_new = og_dot_omp(r, r, n), beta = rr_new / rr;
#pragma omp parallel for schedule(static)
        for (int i = 0; i < n; ++i) p[i] = r[i] + beta * p[i];
        r_norm = sqrt(rr_new);
    }
    free(r); free(p); free(Ap);
    return it;
}
#endif
