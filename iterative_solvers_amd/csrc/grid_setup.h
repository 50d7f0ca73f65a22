// grid_setup.h -- host-side problem setup for the L-shaped Dirichlet grid (product code).
// Restates GridSystem's geometry / RHS / exact solution / node coordinates with closed-form
// index arithmetic instead of per-node predicate calls.  Reference: solver/grid_system.cpp
// :8-15 (f, u), :17-43 (boundary predicates), :45-67 (RHS), :69-77 (coordinates),
// :84-111 (packed index), :301-322 (coefficients).
#pragma once
#include <cstdint>

namespace mi355cg {

struct GridParams {
    int n = 0, m = 0;                 // intervals in x and y (n == m, even, >= 6)
    double a = 0, b = 1, c = 0, d = 1;
    double x_step = 0, y_step = 0;    // grid_system.cpp:314-315
    double A = 0, x_k = 0, y_k = 0;   // grid_system.cpp:316-318
    int half = 0;                     // n / 2
    long long size = 0;               // U = (n/2-1)(3n/2-1)
    long long bottom_size = 0;        // (n/2-1) * (n/2): unknowns of the bottom-right block
};

// Returns false (and leaves *gp untouched) unless n == m, even and >= 6.
bool grid_params_init(GridParams* gp, int n, int m, double a, double b, double c, double d);

// Packed index of interior node (x, y)  (grid_system.cpp:84-111 for n == m).
inline long long packed_index(const GridParams& g, int x, int y) {
    return y <= g.half ? (long long)(g.half - 1) * (y - 1) + (x - g.half - 1)
                       : g.bottom_size + (long long)(y - g.half - 1) * (g.n - 1) + (x - 1);
}
// First packed index of row y (rows 1 .. n-1), and one-past-the-end for y = n.
inline long long packed_row_begin(const GridParams& g, int y) {
    return y <= g.half ? (long long)(g.half - 1) * (y - 1)
                       : g.bottom_size + (long long)(y - g.half - 1) * (g.n - 1);
}

// Fill rhs / u_true / xs / ys (any may be null) for the rows y_begin..y_end (inclusive);
// outputs are indexed relative to packed_row_begin(y_begin).  Multi-threaded over rows; each
// element is computed by the same expression sequence as the reference (bit-identical given
// the same libm).
void grid_fill_rows(const GridParams& g, int y_begin, int y_end,
                    double* rhs, double* u_true, double* xs, double* ys);

// The same for one part of a decomposed grid: rows y_begin..y_end, each restricted to the interior columns inside
// [x_begin, x_end); outputs in the part's packed order (its bottom-block rows, then its upper rows).
void grid_fill_box(const GridParams& g, int y_begin, int y_end, int x_begin, int x_end,
                   double* rhs, double* u_true, double* xs, double* ys);

}  // namespace mi355cg
