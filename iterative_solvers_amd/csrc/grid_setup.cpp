// grid_setup.cpp -- see grid_setup.h.  Build with -ffp-contract=off: the reference is plain
// x86-64 -O2 code without FMA contraction (solver/CMakeLists.txt:68).
#include "grid_setup.h"

#include <algorithm>
#include <cmath>
#include <thread>
#include <vector>

namespace mi355cg {

bool grid_params_init(GridParams* gp, int n, int m, double a, double b, double c, double d) {
    if (n != m || n < 6 || (n & 1)) return false;
    GridParams g;
    g.n = n; g.m = m; g.a = a; g.b = b; g.c = c; g.d = d;
    g.x_step = (b - a) / (n);                                            // grid_system.cpp:314
    g.y_step = (d - c) / (m);                                            // :315
    g.A = -2 * (1 / (g.x_step * g.x_step) + 1 / (g.y_step * g.y_step));  // :316
    g.x_k = 1 / (g.x_step * g.x_step);                                   // :317
    g.y_k = 1 / (g.y_step * g.y_step);                                   // :318
    g.half = n / 2;
    g.bottom_size = (long long)(g.half - 1) * g.half;
    g.size = g.bottom_size + (long long)(n - 1) * (g.half - 1);
    *gp = g;
    return true;
}

namespace {

inline double f_rhs(double x, double y) { return 4 * (x * x + y * y) * std::exp(x * x - y * y); }  // :8-10
inline double u_exact(double x, double y) { return std::exp(x * x - y * y); }                      // :12-15

inline bool is_left(const GridParams& g, int x, int y) {          // :17-22
    return (x == 0 && (y >= g.m / 2 && y <= g.m)) || (x == g.n / 2 && (y >= 0 && y <= g.m / 2));
}
inline bool is_bottom(const GridParams& g, int x, int y) {        // :38-43
    return (y == 0 && (x >= g.n / 2 && x <= g.n)) || (y == g.m / 2 && (x >= 0 && x <= g.n / 2));
}

// interior columns of row y inside [xa, xb); outputs start at the first of them
void fill_row(const GridParams& g, int y, int xa, int xb, double* rhs, double* u, double* xs, double* ys) {
    const int x0 = std::max(xa, y <= g.half ? g.half + 1 : 1), x1 = std::min(xb, g.n);
    const double yp = g.c + y * g.y_step;                          // calculate_y :74-77
    for (int x = x0; x < x1; ++x) {
        const int i = x - x0;
        const double xp = g.a + x * g.x_step;                      // calculate_x :69-72
        if (xs) xs[i] = xp;
        if (ys) ys[i] = yp;
        if (u) u[i] = u_exact(xp, yp);
        if (rhs) {                                                 // calculate_value :45-67
            double value = f_rhs(xp, yp);
            if (is_left(g, x - 1, y)) value -= g.x_k * u_exact(g.a + (x - 1) * g.x_step, yp);
            if (x + 1 == g.n) value -= g.x_k * u_exact(g.a + (x + 1) * g.x_step, yp);
            if (y + 1 == g.m) value -= g.y_k * u_exact(xp, g.c + (y + 1) * g.y_step);
            if (is_bottom(g, x, y - 1)) value -= g.y_k * u_exact(xp, g.c + (y - 1) * g.y_step);
            rhs[i] = value;
        }
    }
}

}  // namespace

void grid_fill_rows(const GridParams& g, int y_begin, int y_end,
                    double* rhs, double* u_true, double* xs, double* ys) {
    grid_fill_box(g, y_begin, y_end, 0, g.n, rhs, u_true, xs, ys);
}

void grid_fill_box(const GridParams& g, int y_begin, int y_end, int x_begin, int x_end,
                   double* rhs, double* u_true, double* xs, double* ys) {
    if (y_end < y_begin) return;
    const int rows = y_end - y_begin + 1;
    // own columns per bottom-block / upper row, and where each row starts in the part's packed order
    const long long wb = std::max(0, std::min(x_end, g.n) - std::max(x_begin, g.half + 1));
    const long long wu = std::max(0, std::min(x_end, g.n) - std::max(x_begin, 1));
    const long long nb = std::max(0, std::min(y_end, g.half) - y_begin + 1);
    auto row_begin = [&](int y) { return y <= g.half ? (long long)(y - y_begin) * wb : nb * wb + (long long)(y - std::max(y_begin, g.half + 1)) * wu; };
    unsigned hw = std::thread::hardware_concurrency();
    int nthreads = (int)std::min<long long>(hw ? hw : 1, std::max<long long>(1, (long long)rows * g.n / 65536));
    nthreads = std::max(1, std::min(nthreads, 64));
    auto work = [&](int t) {
        for (int y = y_begin + t; y <= y_end; y += nthreads) {
            const long long off = row_begin(y);
            fill_row(g, y, x_begin, x_end, rhs ? rhs + off : nullptr, u_true ? u_true + off : nullptr,
                     xs ? xs + off : nullptr, ys ? ys + off : nullptr);
        }
    };
    if (nthreads == 1) { work(0); return; }
    std::vector<std::thread> pool;
    for (int t = 0; t < nthreads; ++t) pool.emplace_back(work, t);
    for (auto& th : pool) th.join();
}

}  // namespace mi355cg
