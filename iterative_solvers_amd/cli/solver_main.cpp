// solver_main.cpp -- console driver with the flow of the reference's solver/main.cpp:596-712 (read the grid
// size from stdin, domain [1,2]^2, GridSystem -> MSGSolver -> residual and error summaries), written against
// the drop-in headers so every solve runs on the MI355X.  The reference's own file is stale (it calls a
// MSGSolver::getPrecision() that no longer exists); this one compiles and adds flags:
//
//   solver_cli [--n N] [--eps E] [--max-iter K] [--rule msg|rel2] [--quiet]
//   echo "256 256" | solver_cli                     # the reference's interactive prompts
//
// Build:  g++ -std=c++17 -O2 -I iterative_solvers_amd/compat iterative_solvers_amd/cli/solver_main.cpp
//             -L iterative_solvers_amd -lmi355cg -Wl,-rpath,$PWD/iterative_solvers_amd -o solver_cli
#include "dirichlet_solver.hpp"
#include "matrix_free_system.hpp"

#include <algorithm>
#include <chrono>
#include <cstring>

static void stats(const char* title, const std::vector<double>& v) {            // main.cpp:560-594 style summary
    double mn = 0, mx = 0, sum = 0, amax = 0, s2 = 0;
    if (!v.empty()) { mn = mx = v[0]; }
    for (double x : v) { mn = std::min(mn, x); mx = std::max(mx, x); sum += x; amax = std::max(amax, std::fabs(x)); s2 += x * x; }
    std::cout << title << ":\n  Минимальное значение: " << std::fixed << std::setprecision(6) << mn
              << "\n  Максимальное значение: " << mx << "\n  Среднее значение: " << (v.empty() ? 0.0 : sum / v.size())
              << "\n  max-норма: " << std::scientific << amax << "\n  2-норма: " << std::sqrt(s2) << "\n" << std::defaultfloat;
}

int main(int argc, char* argv[]) {
    Kokkos::initialize(argc, argv);
    double eps = 1e-9;            // main.cpp:601
    int max_iter = 2;             // main.cpp:602 (the reference's console default is a two-iteration trace)
    int n = 0, m = 0;
    std::string rule = "msg";
    bool quiet = false;
    for (int i = 1; i < argc; ++i) {
        auto val = [&](const char* flag) -> const char* { return (!std::strcmp(argv[i], flag) && i + 1 < argc) ? argv[++i] : nullptr; };
        if (const char* v = val("--n")) n = m = std::atoi(v);
        else if (const char* v = val("--eps")) eps = std::atof(v);
        else if (const char* v = val("--max-iter")) max_iter = std::atoi(v);
        else if (const char* v = val("--rule")) rule = v;
        else if (!std::strcmp(argv[i], "--quiet")) quiet = true;
        else { std::cerr << "unknown argument " << argv[i] << "\n"; return 2; }
    }
    if (n == 0) {                                                               // main.cpp:609-614
        std::cout << "Введите размерность сетки по оси X: "; std::cin >> n;
        std::cout << "Введите размерность сетки по оси Y: "; std::cin >> m;
    }
    const double a = 1.0, b = 2.0, c = 1.0, d = 2.0;                            // main.cpp:617-620
    try {
        std::cout << "\nСоздание сетки размером " << n << "x" << m << " для области [" << a << "," << b << "] x [" << c << "," << d << "]" << std::endl;
        const auto t0 = std::chrono::steady_clock::now();
        std::vector<double> x, residual, error;
        int iterations = 0;
        if (rule == "rel2") {                                                   // MatrixFreeSolver flow
            MatrixFreeSystem sys(m, n, a, b, c, d);
            std::cout << sys << std::endl;
            MatrixFreeSolver solver(sys, sys.get_rhs(), eps, max_iter);
            const std::vector<double> u = sys.get_true_solution_vector();
            x = solver.solve(u);
            iterations = solver.getIterations();
            std::vector<double> ax; sys.apply(x, ax);
            residual.resize(x.size()); error.resize(x.size());
            for (size_t i = 0; i < x.size(); ++i) { residual[i] = ax[i] - sys.get_rhs()[i]; error[i] = x[i] - u[i]; }
            std::cout << solver.getName() << ": " << iterations << " итераций\n";
        } else {                                                                // main.cpp:626-645
            GridSystem grid(m, n, a, b, c, d);
            std::cout << grid << std::endl;
            KokkosVector u = grid.get_true_solution_vector();
            MSGSolver solver(grid.get_matrix(), grid.get_rhs(), eps, max_iter);
            solver.setVerbose(!quiet);
            KokkosVector sol = solver.solve(u);
            iterations = solver.getIterations();
            KokkosVector Ax("Ax", sol.extent(0));
            KokkosSparse::spmv("N", 1.0, grid.get_matrix(), sol, 0.0, Ax);
            x.assign(sol.data(), sol.data() + sol.extent(0));
            residual.resize(x.size()); error.resize(x.size());
            for (size_t i = 0; i < x.size(); ++i) { residual[i] = Ax(i) - grid.get_rhs()(i); error[i] = x[i] - u(i); }
            if (quiet) std::cout << solver.getName() << ": " << iterations << " итераций, " << solver.getStopReasonText() << "\n";
        }
        const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        stats("Невязка Ax-b", residual);
        stats("Ошибка x-u", error);
        std::cout << "Неизвестных: " << x.size() << ", итераций: " << iterations << ", время (с сеткой и копированием): " << secs << " с\n";
    } catch (const std::exception& e) {
        std::cerr << "Ошибка: " << e.what() << std::endl;
        Kokkos::finalize();
        return 1;
    }
    Kokkos::finalize();
    return 0;
}
