// mi355cg_compat.hpp -- C++17 host side of the drop-in: the reference's solver classes, re-implemented
// header-only on top of the C ABI (include/mi355cg.h, libmi355cg.so).  Code written against the
// reference's headers -- `#include "dirichlet_solver.hpp"`, `GridSystem grid(m, n, ...)`,
// `MSGSolver s(grid.get_matrix(), grid.get_rhs(), eps, maxIt); s.solve(u)`, the Qt worker's
// `DirichletSolver` calls -- compiles unchanged with `-I iterative_solvers_amd/compat` and links
// against libmi355cg.so instead of Kokkos/KokkosKernels.  All arithmetic runs on the MI355X.
//
// Interfaces mirrored (reference paths): solver/solver.hpp:17-66 (Solver), solver/msg_solver.hpp:9-120
// (StopCriterion, MSGSolver), solver/grid_system.h:16-88 (GridSystem), solver/matrix_free_system.hpp:12-127
// (MatrixFreeSystem, MatrixFreeSolver), solver/dirichlet_solver.hpp:11-184 (SolverResults, ResultsIO,
// DirichletSolver), plus the sliver of Kokkos the public signatures and the GUI mention
// (qt_gui/src/mainwindow.cpp:81-83,166-168).  Errors: MI355CG_ERR_INVALID -> std::invalid_argument,
// everything else -> std::runtime_error, as the reference throws (grid_system.cpp:86-93,277-279).
#pragma once

#include <atomic>
#include <cfloat>
#include <cmath>
#include <cstdlib>
#include <fstream>
#include <functional>
#include <iomanip>
#include <iostream>
#include <limits>
#include <memory>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/mi355cg.h"

// ---- the sliver of Kokkos the reference's signatures need ------------------------------------------
namespace Kokkos {
struct HostSpace {};
struct Serial {};
using DefaultExecutionSpace = Serial;
namespace detail { inline bool& live() { static bool v = false; return v; } }
inline void initialize() { detail::live() = true; }
inline void initialize(int&, char**) { detail::live() = true; }
inline void finalize() { detail::live() = false; }
inline bool is_initialized() { return detail::live(); }

// View<double*, HostSpace>: reference-counted host vector, zero-filled on construction.
template <class T, class Space = HostSpace>
class View;
template <class Space>
class View<double*, Space> {
    std::shared_ptr<std::vector<double>> d_;
    std::string label_;
public:
    View() : d_(std::make_shared<std::vector<double>>()) {}
    View(const std::string& label, size_t n) : d_(std::make_shared<std::vector<double>>(n, 0.0)), label_(label) {}
    size_t extent(int) const { return d_->size(); }
    size_t size() const { return d_->size(); }
    double& operator()(size_t i) const { return (*d_)[i]; }
    double& operator[](size_t i) const { return (*d_)[i]; }
    double* data() const { return d_->data(); }
    const std::string& label() const { return label_; }
};
template <class V> V create_mirror_view(const V& v) { return v; }
template <class V> void deep_copy(const V& dst, const V& src) {
    if (dst.extent(0) != src.extent(0)) throw std::runtime_error("Kokkos::deep_copy: extent mismatch");
    for (size_t i = 0; i < src.extent(0); ++i) dst(i) = src(i);
}
template <class V> void deep_copy(const V& dst, double value) { for (size_t i = 0; i < dst.extent(0); ++i) dst(i) = value; }
template <class... P> struct RangePolicy { long b, e; RangePolicy(long b_, long e_) : b(b_), e(e_) {} };
template <class... P, class F> void parallel_for(const RangePolicy<P...>& r, const F& f) { for (long i = r.b; i < r.e; ++i) f((int)i); }
}  // namespace Kokkos
#ifndef KOKKOS_LAMBDA
#define KOKKOS_LAMBDA [=]
#endif

using execution_space = Kokkos::DefaultExecutionSpace;
using memory_space = Kokkos::HostSpace;
using KokkosVector = Kokkos::View<double*, memory_space>;

namespace mi355cg_compat {

inline void check(int rc) {
    if (rc == MI355CG_OK) return;
    const std::string msg = mi355cg_last_error();
    if (rc == MI355CG_ERR_INVALID) throw std::invalid_argument(msg);
    throw std::runtime_error(msg);
}

// Owns one mi355cg context (one GPU).  Shared by the grid, its matrix proxy and the solvers.
struct Context {
    mi355cg_handle h = nullptr;
    int n = 0, m = 0;
    double a = 0, b = 0, c = 0, d = 0;
    Context(int n_, int m_, double a_, double b_, double c_, double d_, int device = 0)
        : n(n_), m(m_), a(a_), b(b_), c(c_), d(d_) {
        check(mi355cg_create(n, m, a, b, c, d, MI355CG_F64, device, &h));
    }
    // Optional: the same grid cut into parts that this process drives on one or several GPUs (LOCAL team, csrc/team.h).
    // MSGSolver::solve then runs on the team; results are bit-identical to the single-GPU solve.
    mi355cg_team team = nullptr;
    void distribute(const std::vector<int>& devices, int decomp) {
        if (csr) throw std::invalid_argument("a caller-supplied matrix cannot be distributed");
        if (team) { mi355cg_team_destroy(team); team = nullptr; }
        if (devices.size() > 1 || decomp != MI355CG_DECOMP_ROWS || !devices.empty())
            check(mi355cg_team_create_local(n, m, a, b, c, d, (int)std::max<size_t>(devices.size(), 1), devices.data(), (int)devices.size(), decomp, &team));
    }
    bool csr = false;
    // caller-supplied CSR matrix: the generic path of Solver(a, b, ...) (mi355cg_create_csr)
    Context(long long nrows, const int* row_map, const int* entries, const double* values, int device = 0) : csr(true) {
        check(mi355cg_create_csr(nrows, row_map, entries, values, device, &h));
    }
    ~Context() { if (team) mi355cg_team_destroy(team); mi355cg_destroy(h); }
    Context(const Context&) = delete;
    Context& operator=(const Context&) = delete;
    long long size() const { return mi355cg_size(h); }
};

}  // namespace mi355cg_compat

// ---- KokkosSparse: the system matrix is the stencil operator; CSR arrays are materialised lazily ----
namespace KokkosSparse {
template <class Scalar, class Ordinal, class Device, class Traits, class Offset>
class CrsMatrix {
public:
    struct Graph { std::vector<Offset> row_map; std::vector<Ordinal> entries; };
private:
    std::shared_ptr<mi355cg_compat::Context> ctx_;
    mutable bool built_ = false;
    void build() const;                                  // grid_system.cpp:157-274 order: diag, left, right, top, bottom
public:
    mutable Graph graph;
    mutable std::vector<Scalar> values;
    CrsMatrix() = default;
    explicit CrsMatrix(std::shared_ptr<mi355cg_compat::Context> c) : ctx_(std::move(c)) {}
    // KokkosSparse's raw-pointer constructor: a matrix of the caller's own (copied, then resident on the GPU)
    CrsMatrix(const std::string&, Ordinal nrows, Ordinal ncols, size_t annz, const Scalar* val, const Offset* rowmap, const Ordinal* cols) {
        if (nrows != ncols) throw std::invalid_argument("CrsMatrix: the CG path needs a square matrix");
        graph.row_map.assign(rowmap, rowmap + nrows + 1);
        graph.entries.assign(cols, cols + annz);
        values.assign(val, val + annz);
        built_ = true;
        ctx_ = std::make_shared<mi355cg_compat::Context>((long long)nrows, graph.row_map.data(), graph.entries.data(), values.data());
    }
    const std::shared_ptr<mi355cg_compat::Context>& context() const { return ctx_; }
    long long numRows() const { return ctx_ ? ctx_->size() : 0; }   // grid operator or caller-supplied matrix
    long long numCols() const { return numRows(); }
    long long nnz() const { materialize(); return (long long)values.size(); }
    void materialize() const { if (!built_ && ctx_) { build(); built_ = true; } }
};

// y = alpha * A x + beta * y   (msg_solver.cpp:93,236; dirichlet_solver.cpp:153).  A x runs on the GPU.
template <class M>
void spmv(const char* mode, double alpha, const M& A, const KokkosVector& x, double beta, const KokkosVector& y) {
    if (!mode || mode[0] != 'N') throw std::invalid_argument("spmv: only mode \"N\" is supported");
    if (!A.context()) throw std::runtime_error("spmv: matrix is not bound to a GridSystem");
    std::vector<double> ax(x.extent(0));
    mi355cg_compat::check(mi355cg_apply(A.context()->h, x.data(), ax.data()));
    for (size_t i = 0; i < ax.size(); ++i) y(i) = beta == 0.0 ? alpha * ax[i] : alpha * ax[i] + beta * y(i);
}
}  // namespace KokkosSparse

using KokkosCrsMatrix = KokkosSparse::CrsMatrix<double, int, execution_space, void, int>;

// ---- GridSystem (solver/grid_system.h) -------------------------------------------------------------
class GridSystem {
    std::shared_ptr<mi355cg_compat::Context> ctx_;
    KokkosCrsMatrix matrix_;
    KokkosVector rhs_;
    std::vector<double> xs_, ys_;
public:
    struct NodeCoordinates { double x; double y; };

    GridSystem(int m, int n, double a, double b, double c, double d)          // note the order: (m, n, ...)
        : ctx_(std::make_shared<mi355cg_compat::Context>(n, m, a, b, c, d)), matrix_(ctx_) {
        if (!Kokkos::is_initialized()) Kokkos::initialize();                  // grid_system.cpp:304-306
        const size_t U = (size_t)ctx_->size();
        rhs_ = KokkosVector("rhs", U);
        xs_.resize(U); ys_.resize(U);
        mi355cg_compat::check(mi355cg_get_rhs(ctx_->h, rhs_.data()));
        mi355cg_compat::check(mi355cg_get_node_coords(ctx_->h, xs_.data(), ys_.data()));
    }
    const KokkosCrsMatrix& get_matrix() const { return matrix_; }
    const KokkosVector& get_rhs() const { return rhs_; }
    KokkosVector get_true_solution_vector() {
        if (matrix_.numRows() == 0) throw std::runtime_error("Matrix not initialized, cannot determine size for true solution vector.");
        KokkosVector u("true_u", (size_t)ctx_->size());
        mi355cg_compat::check(mi355cg_get_true_solution(ctx_->h, u.data()));
        return u;
    }
    const std::vector<double>& get_x_coords() const { return xs_; }
    const std::vector<double>& get_y_coords() const { return ys_; }
    NodeCoordinates get_node_coordinates(int i) const {                       // O(1) instead of grid_system.cpp:332-397's search
        if (i < 0 || i >= (int)xs_.size()) return NodeCoordinates{0.0, 0.0};
        return NodeCoordinates{xs_[i], ys_[i]};
    }
    int get_n() const { return ctx_->n; }
    int get_m() const { return ctx_->m; }
    // Extension (no reference counterpart: the reference is single-device): cut the grid into devices.size() parts, part r on
    // GPU devices[r % size]; MI355CG_DECOMP_ROWS = row slabs, MI355CG_DECOMP_2D = (parts/2) x 2 blocks.  An empty list undoes it.
    void distribute(const std::vector<int>& devices, int decomp = MI355CG_DECOMP_ROWS) { ctx_->distribute(devices, decomp); }
    const std::shared_ptr<mi355cg_compat::Context>& context() const { return ctx_; }
    friend std::ostream& operator<<(std::ostream& os, const GridSystem& g) {
        os << "GridSystem Matrix Information:\n  Dimensions: " << g.ctx_->n << "x" << g.ctx_->m << "\n  Domain: [" << g.ctx_->a << ", "
           << g.ctx_->b << "] x [" << g.ctx_->c << ", " << g.ctx_->d << "]\n  Matrix size: " << g.matrix_.numRows() << " rows x "
           << g.matrix_.numCols() << " columns (matrix-free 5-point operator on the GPU)\n";
        return os;
    }
};

template <class S, class O, class D, class T, class F>
void KokkosSparse::CrsMatrix<S, O, D, T, F>::build() const {
    // CSR export for the save/inspect paths only; entry order of grid_system.cpp:193-218.
    const int n = ctx_->n, half = n / 2;
    const double xs = (ctx_->b - ctx_->a) / n, ys = (ctx_->d - ctx_->c) / ctx_->m;
    const double xk = 1 / (xs * xs), yk = 1 / (ys * ys), A = -2 * (1 / (xs * xs) + 1 / (ys * ys));
    const long long B = (long long)(half - 1) * half;
    auto interior = [&](int x, int y) { return y >= 1 && y <= n - 1 && x <= n - 1 && x >= (y <= half ? half + 1 : 1); };
    auto pos = [&](int x, int y) -> O { return (O)(y <= half ? (long long)(half - 1) * (y - 1) + (x - half - 1) : B + (long long)(y - half - 1) * (n - 1) + (x - 1)); };
    graph.row_map.assign(1, 0);
    for (int y = 1; y <= n - 1; ++y)
        for (int x = (y <= half ? half + 1 : 1); x <= n - 1; ++x) {
            graph.entries.push_back(pos(x, y)); values.push_back(A);
            if (interior(x - 1, y)) { graph.entries.push_back(pos(x - 1, y)); values.push_back(xk); }
            if (interior(x + 1, y)) { graph.entries.push_back(pos(x + 1, y)); values.push_back(xk); }
            if (interior(x, y + 1)) { graph.entries.push_back(pos(x, y + 1)); values.push_back(yk); }
            if (interior(x, y - 1)) { graph.entries.push_back(pos(x, y - 1)); values.push_back(yk); }
            graph.row_map.push_back((F)values.size());
        }
}

// ---- Solver base (solver/solver.hpp) ---------------------------------------------------------------
class Solver {
protected:
    const KokkosCrsMatrix& a;
    const KokkosVector& b;
    double eps;
    int maxIterations;
    int iterations = 0;
    std::string name;
    std::function<void(int, double, double, double)> iteration_callback;
    std::function<void(bool, const std::string&)> completion_callback;
public:
    Solver(const KokkosCrsMatrix& a_, const KokkosVector& b_, double eps_ = 1e-6, int maxIterations_ = 10000,
           const std::string& name_ = "Базовый решатель")
        : a(a_), b(b_), eps(eps_), maxIterations(maxIterations_), name(name_) {}
    virtual ~Solver() = default;
    virtual KokkosVector solve(const KokkosVector& true_solution) = 0;
    void setIterationCallback(std::function<void(int, double, double, double)> cb) { iteration_callback = std::move(cb); }
    void setCompletionCallback(std::function<void(bool, const std::string&)> cb) { completion_callback = std::move(cb); }
    int getIterations() const { return iterations; }
    std::string getName() const { return name; }
};

enum class StopCriterion { ITERATIONS, PRECISION, RESIDUAL, EXACT_ERROR, INTERRUPTED };   // msg_solver.hpp:9-15

namespace mi355cg_compat {
inline void iter_trampoline(void* user, int it, double p, double r, double e) {
    auto* f = static_cast<std::function<void(int, double, double, double)>*>(user);
    if (f && *f) (*f)(it, p, r, e);
}
}  // namespace mi355cg_compat

// ---- MSGSolver (solver/msg_solver.hpp, msg_solver.cpp:10-212) --------------------------------------
class MSGSolver : public Solver {
    double eps_precision, eps_residual, eps_exact_error;
    bool converged = false;
    StopCriterion stop_reason = StopCriterion::ITERATIONS;
    double final_residual_norm = 0.0, final_error_norm = 0.0, final_precision = 0.0;
    double initial_r_norm = 0.0, final_r_norm = 0.0, solve_ms = 0.0;
    std::function<void(int, double, double, double)> iteration_callback;     // shadows the base member, as in the reference
    std::atomic<int> stop_requested{0};
    bool verbose = true;
    int poll_interval = 0;
public:
    MSGSolver(const KokkosCrsMatrix& a_, const KokkosVector& b_, double eps_ = 1e-6, int maxIterations_ = 10000)
        : Solver(a_, b_, eps_, maxIterations_, "Метод серединных градиентов"),
          eps_precision(eps_), eps_residual(eps_), eps_exact_error(eps_) {}
    void setPrecisionEps(double e) { eps_precision = e; }
    void setResidualEps(double e) { eps_residual = e; }
    void setExactErrorEps(double e) { eps_exact_error = e; }
    bool hasConverged() const { return converged; }
    StopCriterion getStopReason() const { return stop_reason; }
    void requestStop() { stop_requested = 1; }
    void resetStop() { stop_requested = 0; }
    bool isStopRequested() const { return stop_requested != 0; }
    void setVerbose(bool v) { verbose = v; }              // the reference always prints (msg_solver.cpp:172-177,202-208)
    // Iterations queued on the GPU between two looks at the stop flag (and two deliveries of callbacks).  The reference
    // looks every iteration (msg_solver.cpp:82-87); here the default is the callback cadence (100), 1 is allowed and
    // costs one host round trip per iteration.  The first iteration of a solve is always on its own.
    void setPollInterval(int iterations) { poll_interval = iterations > 0 ? iterations : 0; }
    std::string getStopReasonText() const {
        switch (stop_reason) {
            case StopCriterion::ITERATIONS: return "Достигнуто максимальное число итераций";
            case StopCriterion::PRECISION: return "Достигнута требуемая точность по норме разности xn и xn-1";
            case StopCriterion::RESIDUAL: return "Достигнута требуемая точность по норме невязки";
            case StopCriterion::EXACT_ERROR: return "Достигнута требуемая точность по норме разности с истинным решением";
            case StopCriterion::INTERRUPTED: return "Прервано пользователем";
        }
        return "Неизвестная причина остановки";
    }
    double getFinalResidualNorm() const { return final_residual_norm; }
    double getFinalErrorNorm() const { return final_error_norm; }
    double getFinalPrecision() const { return final_precision; }
    void setIterationCallback(std::function<void(int, double, double, double)> cb) { iteration_callback = std::move(cb); }

    KokkosVector solve(const KokkosVector& true_solution) override {
        converged = false;
        stop_requested = 0;                                                   // msg_solver.cpp:12-13
        last_printed_ = -1;
        if (!a.context()) throw std::runtime_error("MSGSolver: the matrix is not a GridSystem operator");
        mi355cg_handle h = a.context()->h;
        mi355cg_compat::check(mi355cg_set_rhs(h, b.data()));
        if (true_solution.extent(0) > 0)       // the error norms use the vector that was passed in (msg_solver.cpp:64-72,132-139)
            mi355cg_compat::check(mi355cg_set_true_solution(h, true_solution.data()));
        mi355cg_params p;
        mi355cg_default_params(&p, MI355CG_RULE_MSG_MAXNORM);
        p.max_iterations = maxIterations;
        p.eps_precision = eps_precision; p.eps_residual = eps_residual; p.eps_exact_error = eps_exact_error;
        p.use_true_solution = true_solution.extent(0) > 0 ? 1 : 0;
        p.sync_every = poll_interval;
        p.callback_every = 100;
        std::function<void(int, double, double, double)> cb = [this](int it, double pr, double rs, double er) {
            // progress print of msg_solver.cpp:172-177 (cosmetic; a solve that converges exactly on a
            // multiple of 100 prints that iteration once, which the reference would not)
            if (verbose && it > 0 && (it % 100 == 0 || it == 1) && it != last_printed_) {
                last_printed_ = it;
                std::cout << "Итерация: " << it << "\nТочность ||x(n)-x(n-1)||: max-норма = " << std::scientific << pr
                          << "\nНевязка ||Ax-b||: max-норма = " << std::scientific << rs
                          << "\nОшибка ||u-x||: max-норма = " << std::scientific << er << "\n\n";
            }
            if (iteration_callback) iteration_callback(it, pr, rs, er);
        };
        mi355cg_results res;
        mi355cg_team team = a.context()->team;
        if (team) {
            // Solver(a, b, ...) takes any b (solver/solver.hpp:33-39) and solve() any true_solution: the parts take their entries
            mi355cg_compat::check(mi355cg_team_set_vector(team, 2, b.data()));
            if (true_solution.extent(0) > 0) mi355cg_compat::check(mi355cg_team_set_vector(team, 3, true_solution.data()));
            mi355cg_compat::check(mi355cg_team_solve(team, &p, &mi355cg_compat::iter_trampoline, &cb,
                                                     reinterpret_cast<const volatile int*>(&stop_requested), &res));
        } else
        mi355cg_compat::check(mi355cg_solve(h, &p, &mi355cg_compat::iter_trampoline, &cb,
                                            reinterpret_cast<const volatile int*>(&stop_requested), &res));
        iterations = res.iterations;                                          // msg_solver.cpp:187-190
        converged = res.converged != 0;
        stop_reason = static_cast<StopCriterion>(res.stop_reason);
        final_residual_norm = res.final_residual_norm; final_precision = res.final_precision; final_error_norm = res.final_error_norm;
        initial_r_norm = res.initial_r_norm2; final_r_norm = res.r_norm2; solve_ms = res.solve_seconds * 1e3;
        if (verbose)
            std::cout << "Метод серединных градиентов (MSG)\nИтераций: " << iterations << "\nВремя: " << (long long)solve_ms
                      << " мс\nНачальная невязка: " << initial_r_norm << "\nКонечная невязка: " << final_r_norm
                      << "\nСходимость: " << (converged ? "Да" : "Нет") << "\nПричина остановки: " << getStopReasonText() << std::endl;
        KokkosVector x("x", b.extent(0));
        if (team) mi355cg_compat::check(mi355cg_team_get_vector(team, 0, x.data()));
        else mi355cg_compat::check(mi355cg_get_solution(h, x.data()));
        return x;
    }
    // Same sections, labels and number formats as msg_solver.cpp:261-304 (the text is user-facing output).
    std::string generateReport(int n, int m, double a_, double b_, double c_, double d_) const {
        std::stringstream ss;
        ss << "ОТЧЕТ О РЕШЕНИИ ЗАДАЧИ ДИРИХЛЕ\n===========================\n\n";
        ss << "ПАРАМЕТРЫ ЗАДАЧИ:\n----------------\n";
        ss << "Размер сетки: " << n << "x" << m << " внутренних узлов\n";
        ss << "Область: [" << a_ << ", " << b_ << "] x [" << c_ << ", " << d_ << "]\n";
        ss << "Шаг по x: " << (b_ - a_) / (n + 1) << "\n";
        ss << "Шаг по y: " << (d_ - c_) / (m + 1) << "\n";
        ss << "Общее количество неизвестных: " << n * m << "\n\n";
        ss << "МЕТОД РЕШЕНИЯ:\n-------------\n";
        ss << "Название метода: " << name << "\n";
        ss << "Максимальное число итераций: " << maxIterations << "\n";
        ss << "Критерии остановки:\n";
        ss << "  - Точность ||xn-x(n-1)||: " << eps_precision << "\n";
        ss << "  - Норма невязки ||Ax-b||: " << eps_residual << "\n";
        ss << "  - Норма ошибки ||u-x||: " << eps_exact_error << "\n\n";
        ss << "РЕЗУЛЬТАТЫ РЕШЕНИЯ:\n-----------------\n";
        ss << "Выполнено итераций: " << iterations << "\n";
        ss << "Сходимость: " << (converged ? "Да" : "Нет") << "\n";
        ss << "Причина остановки: " << getStopReasonText() << "\n";
        ss << "Достигнутые величины:\n";
        ss << "  - Точность ||xn-x(n-1)||: " << std::scientific << final_precision << "\n";
        ss << "  - Норма невязки ||Ax-b||: " << std::scientific << final_residual_norm << "\n";
        ss << "  - Норма ошибки ||u-x||: " << std::scientific << final_error_norm << "\n\n";
        ss << "ПРИМЕЧАНИЯ:\n----------\n";
        ss << "- Все нормы вычислены как maximum-norm (максимальный модуль элемента)\n";
        ss << "- Для сравнения с истинным решением используется функция u(x,y) = exp(x^2 - y^2)\n";
        return ss.str();
    }
private:
    int last_printed_ = -1;
};

// ---- MatrixFreeSystem / MatrixFreeSolver (solver/matrix_free_system.hpp) ---------------------------
class MatrixFreeSystem {
    std::shared_ptr<mi355cg_compat::Context> ctx_;
    std::vector<double> rhs;
public:
    MatrixFreeSystem(int m, int n, double a, double b, double c, double d)
        : ctx_(std::make_shared<mi355cg_compat::Context>(n, m, a, b, c, d)), rhs((size_t)ctx_->size()) {
        mi355cg_compat::check(mi355cg_get_rhs(ctx_->h, rhs.data()));
    }
    const std::vector<double>& get_rhs() const { return rhs; }
    std::vector<double> get_true_solution_vector() {
        std::vector<double> u(rhs.size());
        mi355cg_compat::check(mi355cg_get_true_solution(ctx_->h, u.data()));
        return u;
    }
    void apply(const std::vector<double>& x, std::vector<double>& y) const {
        if (x.size() != rhs.size()) throw std::invalid_argument("apply: vector size does not match the system");
        y.resize(rhs.size());
        mi355cg_compat::check(mi355cg_apply(ctx_->h, x.data(), y.data()));
    }
    std::vector<double> operator*(const std::vector<double>& x) const { std::vector<double> y; apply(x, y); return y; }
    int size() const { return (int)rhs.size(); }
    const std::shared_ptr<mi355cg_compat::Context>& context() const { return ctx_; }
    friend std::ostream& operator<<(std::ostream& os, const MatrixFreeSystem& s) {
        return os << "MatrixFreeSystem Information:\n  Dimensions: " << s.ctx_->n << "x" << s.ctx_->m << "\n  System size: " << s.size() << "\n";
    }
};

class MatrixFreeSolver {
    const MatrixFreeSystem& system;
    const std::vector<double>& b;
    double eps;
    int maxIterations;
    int iterations = 0;
    std::string name;
    std::function<void(int, double, double, double)> iteration_callback;
    std::function<void(bool, const std::string&)> completion_callback;
public:
    MatrixFreeSolver(const MatrixFreeSystem& system_, const std::vector<double>& b_, double eps_ = 1e-6,
                     int maxIterations_ = 10000, const std::string& name_ = "Matrix-free solver")
        : system(system_), b(b_), eps(eps_), maxIterations(maxIterations_), name(name_) {}
    virtual ~MatrixFreeSolver() = default;
    void setIterationCallback(std::function<void(int, double, double, double)> cb) { iteration_callback = std::move(cb); }
    void setCompletionCallback(std::function<void(bool, const std::string&)> cb) { completion_callback = std::move(cb); }
    int getIterations() const { return iterations; }
    std::string getName() const { return name; }

    std::vector<double> solve(const std::vector<double>& true_solution) {        // matrix_free_system.cpp:383-482
        mi355cg_handle h = system.context()->h;
        mi355cg_compat::check(mi355cg_set_rhs(h, b.data()));
        if (true_solution.size() == b.size() && !true_solution.empty())           // :451-455 measures the error against the caller's vector
            mi355cg_compat::check(mi355cg_set_true_solution(h, true_solution.data()));
        mi355cg_params p;
        mi355cg_default_params(&p, MI355CG_RULE_REL_2NORM);
        p.max_iterations = maxIterations; p.eps_rel = eps;
        p.diagnostics = iteration_callback ? 1 : 0;       // per-iteration 2-norm report incl. the second apply (:444-468)
        mi355cg_results res;
        mi355cg_compat::check(mi355cg_solve(h, &p, &mi355cg_compat::iter_trampoline, &iteration_callback, nullptr, &res));
        iterations = res.iterations;
        if (completion_callback)                                                 // :472-479
            completion_callback(res.converged != 0, res.converged ? "Converged successfully" : "Failed to converge within maximum iterations");
        std::vector<double> x(b.size());
        mi355cg_compat::check(mi355cg_get_solution(h, x.data()));
        return x;
    }
};

// ---- SolverResults / ResultsIO / DirichletSolver (solver/dirichlet_solver.hpp) ---------------------
struct SolverResults {
    std::vector<double> solution, true_solution, residual, error, x_coords, y_coords;
    double residual_norm = 0.0, error_norm = 0.0;
    int iterations = 0;
    double precision = 0.0;          // never assigned by the reference either (dirichlet_solver.cpp:101-123)
    bool converged = false;
    std::string stop_reason;
};

// Text formats of solver/dirichlet_solver.cpp:255-457: section keywords, line order and number formats
// (default stream format for the header values, std::scientific for the vectors) as the reference writes
// them.  loadResults reads each section up to the next keyword instead of assuming n*m entries (the
// reference's reader cannot re-read its own writer's files, SURVEY section 5: save writes U entries).
class ResultsIO {
    static void put(std::ostream& os, const char* tag, const std::vector<double>& v) {
        os << tag << "\n";
        for (double x : v) os << std::scientific << x << "\n";
    }
    static bool get(std::istream& is, const std::string& tag, std::vector<double>& v, std::string& pending) {
        std::string line = pending;
        pending.clear();
        while (line.empty()) if (!std::getline(is, line)) return false;
        if (line != tag) { pending = line; return false; }
        v.clear();
        while (std::getline(is, line)) {
            if (line.empty()) continue;
            char* end = nullptr;
            const double x = std::strtod(line.c_str(), &end);
            if (end == line.c_str()) { pending = line; break; }        // next section keyword
            v.push_back(x);
        }
        return true;
    }
public:
    static bool saveResults(const std::string& filename, const SolverResults& r, int n, int m, double a, double b, double c,
                            double d, const std::string& solver_name) {
        std::ofstream file(filename);
        if (!file) return false;
        file << "PARAMETERS\n" << n << " " << m << "\n" << a << " " << b << " " << c << " " << d << "\n" << solver_name << "\n";
        file << "CONVERGENCE\n" << r.iterations << "\n" << (r.converged ? "1" : "0") << "\n" << r.stop_reason << "\n";
        file << std::scientific << r.residual_norm << " " << r.error_norm << "\n";
        put(file, "SOLUTION", r.solution); put(file, "TRUE_SOLUTION", r.true_solution); put(file, "RESIDUAL", r.residual);
        put(file, "ERROR", r.error); put(file, "X_COORDS", r.x_coords); put(file, "Y_COORDS", r.y_coords);
        return true;
    }
    static bool loadResults(const std::string& filename, SolverResults& r, int& n, int& m, double& a, double& b, double& c,
                            double& d, std::string& solver_name) {
        std::ifstream file(filename);
        if (!file) return false;
        std::string line;
        if (!std::getline(file, line) || line != "PARAMETERS") return false;
        file >> n >> m >> a >> b >> c >> d;
        file.ignore();
        std::getline(file, solver_name);
        if (!std::getline(file, line) || line != "CONVERGENCE") return false;
        int conv = 0;
        file >> r.iterations >> conv;
        r.converged = conv == 1;
        file.ignore();
        std::getline(file, r.stop_reason);
        file >> r.residual_norm >> r.error_norm;
        file.ignore();
        std::string pending;
        if (!get(file, "SOLUTION", r.solution, pending) || !get(file, "TRUE_SOLUTION", r.true_solution, pending) ||
            !get(file, "RESIDUAL", r.residual, pending) || !get(file, "ERROR", r.error, pending)) return false;
        get(file, "X_COORDS", r.x_coords, pending);                     // optional in the reference's reader too
        get(file, "Y_COORDS", r.y_coords, pending);
        return true;
    }
    static bool saveMatrixAndRhs(const std::string& filename, const KokkosCrsMatrix& A, const KokkosVector& b, int n, int m) {
        std::ofstream file(filename);
        if (!file) return false;
        A.materialize();
        const long long num_rows = A.numRows(), nnz = A.nnz();
        file << "MATRIX_INFO\n" << n << " " << m << "\n" << num_rows << " " << nnz << "\nMATRIX\n";
        for (long long i = 0; i <= num_rows; ++i) file << A.graph.row_map[(size_t)i] << "\n";
        for (long long i = 0; i < nnz; ++i) file << A.graph.entries[(size_t)i] << "\n";
        for (long long i = 0; i < nnz; ++i) file << std::scientific << A.values[(size_t)i] << "\n";
        file << "RHS\n";
        for (long long i = 0; i < num_rows; ++i) file << std::scientific << b((size_t)i) << "\n";
        return true;
    }
    static bool saveSolutionFor3D(const std::string& filename, const std::vector<std::vector<double>>& solution,
                                  double a_bound, double b_bound, double c_bound, double d_bound) {
        std::ofstream os(filename);                                               // gnuplot "x y z" (dirichlet_solver.hpp:44-76)
        if (!os.is_open() || solution.empty() || solution[0].empty()) return false;
        const int m = (int)solution.size(), n = (int)solution[0].size();
        const double hx = (b_bound - a_bound) / (n + 1), hy = (d_bound - c_bound) / (m + 1);
        for (int i = 0; i < m; ++i) {
            for (int j = 0; j < n; ++j) os << a_bound + (j + 1) * hx << " " << c_bound + (i + 1) * hy << " " << solution[i][j] << "\n";
            os << "\n";
        }
        return true;
    }
};

class DirichletSolver {
    int n_internal, m_internal;
    double a_bound, b_bound, c_bound, d_bound;
    double eps_precision = 1e-6, eps_residual = 1e-6, eps_exact_error = 1e-6;      // dirichlet_solver.cpp:14
    int max_iterations = 10000;
    bool use_precision_stopping = true, use_residual_stopping = true, use_error_stopping = false, use_max_iterations_stopping = true;
    std::atomic<bool> stop_requested{false};              // the reference's plain bool is a latent race
    std::function<void(int, double, double, double)> iteration_callback;
    std::function<void(const SolverResults&)> completion_callback;
    std::unique_ptr<GridSystem> grid;
    std::unique_ptr<MSGSolver> solver;
    KokkosVector solution, true_solution;

    static std::vector<double> to_std(const KokkosVector& v) { return std::vector<double>(v.data(), v.data() + v.extent(0)); }
public:
    DirichletSolver(int n = 10, int m = 10, double a = 0.0, double b = 1.0, double c = 0.0, double d = 1.0)
        : n_internal(n), m_internal(m), a_bound(a), b_bound(b), c_bound(c), d_bound(d) {
        if (!Kokkos::is_initialized()) Kokkos::initialize();
        grid = std::make_unique<GridSystem>(m_internal, n_internal, a_bound, b_bound, c_bound, d_bound);   // (m, n): .cpp:24
    }
    ~DirichletSolver() { solver.reset(); grid.reset(); }

    void setGridParameters(int n, int m, double a, double b, double c, double d) {
        n_internal = n; m_internal = m; a_bound = a; b_bound = b; c_bound = c; d_bound = d;
        solver.reset();
        grid = std::make_unique<GridSystem>(m_internal, n_internal, a_bound, b_bound, c_bound, d_bound);
        if (!devices_.empty()) grid->distribute(devices_, decomp_);
    }
    void setSolverParameters(double eps_p, double eps_r, double eps_e, int max_iter) {
        eps_precision = eps_p; eps_residual = eps_r; eps_exact_error = eps_e; max_iterations = max_iter;
    }
    void enablePrecisionStopping(bool e) { use_precision_stopping = e; }
    void enableResidualStopping(bool e) { use_residual_stopping = e; }
    void enableErrorStopping(bool e) { use_error_stopping = e; }
    void enableMaxIterationsStopping(bool e) { use_max_iterations_stopping = e; }   // never read, as in the reference
    void requestStop() { stop_requested = true; if (solver) solver->requestStop(); }
    std::string getMethodName() const { return solver ? solver->getName() : "МСГ"; }
    void setIterationCallback(std::function<void(int, double, double, double)> cb) { iteration_callback = std::move(cb); }
    void setCompletionCallback(std::function<void(const SolverResults&)> cb) { completion_callback = std::move(cb); }
    void setVerbose(bool v) { verbose_ = v; }
    void setPollInterval(int iterations) { poll_interval_ = iterations; }         // see MSGSolver::setPollInterval
    // Extension: solve on several GPUs of this process (see GridSystem::distribute).  Kept across setGridParameters.
    void setDevices(const std::vector<int>& devices, int decomp = MI355CG_DECOMP_ROWS) {
        devices_ = devices; decomp_ = decomp;
        if (grid) grid->distribute(devices_, decomp_);
    }

    SolverResults solve() {                                                       // dirichlet_solver.cpp:61-131
        if (!grid) throw std::runtime_error("Сетка не инициализирована");
        solver = std::make_unique<MSGSolver>(grid->get_matrix(), grid->get_rhs(),
                                             std::min({eps_precision, eps_residual, eps_exact_error}), max_iterations);
        solver->setVerbose(verbose_);
        solver->setPollInterval(poll_interval_);
        solver->setPrecisionEps(use_precision_stopping ? eps_precision : -1.0);
        solver->setResidualEps(use_residual_stopping ? eps_residual : -1.0);
        solver->setExactErrorEps(use_error_stopping ? eps_exact_error : -1.0);
        if (iteration_callback) solver->setIterationCallback(iteration_callback);
        true_solution = grid->get_true_solution_vector();
        solution = solver->solve(true_solution);
        SolverResults r;
        r.solution = to_std(solution);
        r.true_solution = to_std(true_solution);
        r.residual.resize(r.solution.size());                                     // A x - b, one more apply (.cpp:147-161)
        if (grid->context()->team) {                                              // x came from the parts: apply it on the whole-grid context
            mi355cg_compat::check(mi355cg_apply(grid->context()->h, r.solution.data(), r.residual.data()));
            for (size_t i = 0; i < r.residual.size(); ++i) r.residual[i] = r.residual[i] - grid->get_rhs()(i);
        } else
        mi355cg_compat::check(mi355cg_get_true_residual(grid->context()->h, r.residual.data()));
        r.error.resize(r.solution.size());                                        // x - u (.cpp:164-180)
        for (size_t i = 0; i < r.error.size(); ++i) r.error[i] = r.solution[i] - r.true_solution[i];
        r.x_coords = grid->get_x_coords();
        r.y_coords = grid->get_y_coords();
        r.iterations = solver->getIterations();
        r.converged = solver->hasConverged();
        r.stop_reason = solver->getStopReasonText();
        r.residual_norm = solver->getFinalResidualNorm();
        r.error_norm = solver->getFinalErrorNorm();
        last_ = r;
        if (completion_callback) completion_callback(r);
        return r;
    }
    std::vector<double> getSolution() const { return to_std(solution); }
    std::vector<double> getTrueSolution() const { return to_std(true_solution); }
    // The reference indexes sol[j*n + i] for j < m, i < n, which runs past the U packed unknowns
    // (dirichlet_solver.cpp:193-205); here out-of-range cells read 0 instead of invoking UB.
    std::vector<std::vector<double>> solutionToMatrix() const {
        std::vector<std::vector<double>> out(m_internal, std::vector<double>(n_internal, 0.0));
        const size_t U = solution.extent(0);
        for (int j = 0; j < m_internal; ++j)
            for (int i = 0; i < n_internal; ++i) { const size_t k = (size_t)j * n_internal + i; if (k < U) out[j][i] = solution(k); }
        return out;
    }
    std::string generateReport() const {
        return solver ? solver->generateReport(n_internal, m_internal, a_bound, b_bound, c_bound, d_bound) : std::string("Решение еще не выполнено");
    }
    bool saveResultsToFile(const std::string& filename) const {
        return solver && ResultsIO::saveResults(filename, last_, n_internal, m_internal, a_bound, b_bound, c_bound, d_bound, solver->getName());
    }
    bool saveMatrixAndRhsToFile(const std::string& filename) const {
        return grid && ResultsIO::saveMatrixAndRhs(filename, grid->get_matrix(), grid->get_rhs(), n_internal, m_internal);
    }
    const GridSystem* getGridSystem() const { return grid.get(); }
private:
    SolverResults last_;
    bool verbose_ = true;
    int poll_interval_ = 0;
    std::vector<int> devices_;
    int decomp_ = MI355CG_DECOMP_ROWS;
};
