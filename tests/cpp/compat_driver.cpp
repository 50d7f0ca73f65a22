// compat_driver.cpp -- exercises the C++ drop-in headers the way the reference's own callers do
// (solver/main.cpp:596-712 flow, the Qt worker's DirichletSolver flow qt_gui/src/mainwindow.cpp:55-68,
// 290-304, and the matrix-free pair).  Prints one JSON object; tests/test_gpu_cpp_compat.py compares it
// with the CPU oracle.  Build: g++ -std=c++17 -I iterative_solvers_amd/compat compat_driver.cpp -L... -lmi355cg
#include "dirichlet_solver.hpp"
#include "matrix_free_system.hpp"

#include <cstdio>
#include <cstdlib>
#include <unistd.h>

static void jvec(const char* key, const std::vector<double>& v, bool last = false) {
    std::printf("\"%s\": [", key);
    for (size_t i = 0; i < v.size(); ++i) std::printf("%s%.17g", i ? ", " : "", v[i]);
    std::printf("]%s\n", last ? "" : ",");
}

int main(int argc, char** argv) {
    const int N = argc > 1 ? std::atoi(argv[1]) : 16;
    const int max_it = argc > 2 ? std::atoi(argv[2]) : 10000;
    Kokkos::initialize();
    std::printf("{\n");
    {   // solver/main.cpp flow: GridSystem -> MSGSolver(matrix, rhs, eps, maxIt) -> solve(u) -> spmv residual
        GridSystem grid(N, N, 1.0, 2.0, 1.0, 2.0);
        KokkosVector u = grid.get_true_solution_vector();
        MSGSolver solver(grid.get_matrix(), grid.get_rhs(), 1e-9, max_it);
        solver.setVerbose(false);
        std::vector<double> cb_its;
        solver.setIterationCallback([&](int it, double, double, double) { cb_its.push_back(it); });
        KokkosVector x = solver.solve(u);
        const int n = (int)grid.get_rhs().extent(0);
        KokkosVector Ax("Ax", n);
        KokkosSparse::spmv("N", 1.0, grid.get_matrix(), x, 0.0, Ax);
        std::vector<double> xs(x.data(), x.data() + n), res(n);
        for (int i = 0; i < n; ++i) res[i] = Ax(i) - grid.get_rhs()(i);
        std::printf("\"msg_iterations\": %d, \"msg_converged\": %d, \"msg_reason\": %d, \"msg_rmax\": %.17g, \"msg_nnz\": %lld,\n",
                    solver.getIterations(), solver.hasConverged() ? 1 : 0, (int)solver.getStopReason(),
                    solver.getFinalResidualNorm(), grid.get_matrix().nnz());
        jvec("msg_x", xs); jvec("msg_residual", res); jvec("msg_cb_its", cb_its);
        {   // the abstract Solver contract with a matrix of the caller's own: the grid's CSR arrays re-wrapped
            const KokkosCrsMatrix& G = grid.get_matrix();
            G.materialize();
            KokkosCrsMatrix mine("A", (int)G.numRows(), (int)G.numCols(), G.values.size(), G.values.data(),
                                 G.graph.row_map.data(), G.graph.entries.data());
            MSGSolver s2(mine, grid.get_rhs(), 1e-9, max_it);
            s2.setVerbose(false);
            KokkosVector x2 = s2.solve(u);
            bool same = s2.getIterations() == solver.getIterations() && s2.getStopReason() == solver.getStopReason();
            for (int i = 0; i < n && same; ++i) same = x2(i) == x(i);
            std::printf("\"csr_same_as_grid\": %d, \"csr_error_norm_equal\": %d,\n", same ? 1 : 0,
                        s2.getFinalErrorNorm() == solver.getFinalErrorNorm() ? 1 : 0);
        }
        auto nc = grid.get_node_coordinates(0);
        std::printf("\"node0\": [%.17g, %.17g],\n", nc.x, nc.y);
    }
    {   // Qt worker flow
        DirichletSolver ds(N, N, 1.0, 2.0, 1.0, 2.0);
        ds.setVerbose(false);
        ds.setSolverParameters(1e-8, 1e-8, 1e-8, max_it);
        int completions = 0;
        ds.setCompletionCallback([&](const SolverResults&) { ++completions; });
        SolverResults r = ds.solve();
        std::printf("\"ds_iterations\": %d, \"ds_converged\": %d, \"ds_residual_norm\": %.17g, \"ds_error_norm\": %.17g, \"ds_completions\": %d,\n",
                    r.iterations, r.converged ? 1 : 0, r.residual_norm, r.error_norm, completions);
        jvec("ds_solution", r.solution); jvec("ds_residual", r.residual); jvec("ds_error", r.error);
        const std::string dir = argc > 3 ? argv[3] : "/tmp";
        const bool saved = ds.saveResultsToFile(dir + "/results.txt") && ds.saveMatrixAndRhsToFile(dir + "/matrix.txt");
        { std::ofstream rep(dir + "/report.txt"); rep << ds.generateReport(); }
        SolverResults back; int n2 = 0, m2 = 0; double a2, b2, c2, d2; std::string name2;
        const bool loaded = ResultsIO::loadResults(dir + "/results.txt", back, n2, m2, a2, b2, c2, d2, name2);
        const bool same = loaded && n2 == N && m2 == N && back.iterations == r.iterations && back.converged == r.converged &&
                          back.stop_reason == r.stop_reason && back.solution.size() == r.solution.size() &&
                          back.y_coords.size() == r.y_coords.size() && name2 == ds.getMethodName();
        std::printf("\"ds_saved\": %d, \"ds_roundtrip\": %d, \"ds_method\": \"%s\",\n", saved ? 1 : 0, same ? 1 : 0, ds.getMethodName().empty() ? "" : "set");
        jvec("ds_true_solution", r.true_solution); jvec("ds_x_coords", r.x_coords); jvec("ds_y_coords", r.y_coords);
        std::printf("\"ds_stop_reason\": \"%s\",\n", r.stop_reason.c_str());
        // the same facade on a distributed grid (extension: one process, several parts -- here 4 parts in a 2 x 2 split sharing GPU 0)
        DirichletSolver dd(N, N, 1.0, 2.0, 1.0, 2.0);
        dd.setVerbose(false);
        dd.setSolverParameters(1e-8, 1e-8, 1e-8, max_it);
        dd.setDevices({0, 0, 0, 0}, N >= 256 ? MI355CG_DECOMP_2D : MI355CG_DECOMP_ROWS);
        std::vector<double> cb_its;
        dd.setIterationCallback([&](int it, double, double, double) { cb_its.push_back(it); });
        SolverResults q = dd.solve();
        const bool same_dist = q.iterations == r.iterations && q.converged == r.converged && q.stop_reason == r.stop_reason &&
                               q.solution == r.solution && q.residual == r.residual && q.residual_norm == r.residual_norm && q.error_norm == r.error_norm;
        std::printf("\"dist_same_as_one_gpu\": %d, \"dist_iterations\": %d, \"dist_callbacks\": %d,\n", same_dist ? 1 : 0, q.iterations, (int)cb_its.size());
    }
    {   // the team surface of the C ABI from C++ (what an MPI-launched host would do per rank; here world = 1):
        // ncclGetUniqueId -> ncclCommInitRank inside the library, then the same solve call as on a single handle
        unsigned char id[128];
        mi355cg_team team = nullptr;
        mi355cg_params p;
        mi355cg_default_params(&p, MI355CG_RULE_REL_2NORM);
        p.eps_rel = 1e-8; p.max_iterations = 1000000;
        mi355cg_results tr{}, sr{};
        // RCCL prints a version banner on stdout when its first communicator comes up: keep this program's JSON clean
        std::fflush(stdout);
        const int saved_stdout = dup(1);
        dup2(2, 1);
        mi355cg_compat::check(mi355cg_team_unique_id(id));
        mi355cg_compat::check(mi355cg_team_create_rccl(N, N, 1.0, 2.0, 1.0, 2.0, 1, 0, 0, id, MI355CG_DECOMP_ROWS, &team));
        std::fflush(stdout);
        dup2(saved_stdout, 1);
        close(saved_stdout);
        mi355cg_compat::check(mi355cg_team_solve(team, &p, nullptr, nullptr, nullptr, &tr));
        MatrixFreeSystem one(N, N, 1.0, 2.0, 1.0, 2.0);
        mi355cg_compat::check(mi355cg_solve(one.context()->h, &p, nullptr, nullptr, nullptr, &sr));
        std::vector<double> xt(one.size(), -1.0), xs(one.size());
        mi355cg_compat::check(mi355cg_team_get_vector(team, 0, xt.data()));
        mi355cg_compat::check(mi355cg_get_solution(one.context()->h, xs.data()));
        std::printf("\"rccl_team_iterations\": %d, \"rccl_team_same_as_handle\": %d,\n", tr.iterations,
                    (tr.iterations == sr.iterations && tr.r_norm2 == sr.r_norm2 && xt == xs) ? 1 : 0);
        mi355cg_team_destroy(team);
    }
    {   // matrix-free pair
        MatrixFreeSystem sys(N, N, 1.0, 2.0, 1.0, 2.0);
        std::vector<double> ones(sys.size(), 1.0), y;
        sys.apply(ones, y);
        MatrixFreeSolver mf(sys, sys.get_rhs(), 1e-8, 1000000);
        int done_ok = -1;
        mf.setCompletionCallback([&](bool ok, const std::string&) { done_ok = ok ? 1 : 0; });
        std::vector<double> x = mf.solve(sys.get_true_solution_vector());
        std::printf("\"mf_iterations\": %d, \"mf_completed_ok\": %d,\n", mf.getIterations(), done_ok);
        jvec("mf_apply_ones", y); jvec("mf_x", x, true);
    }
    std::printf("}\n");
    bool threw = false;
    try { GridSystem bad(7, 7, 1.0, 2.0, 1.0, 2.0); } catch (const std::invalid_argument&) { threw = true; }
    Kokkos::finalize();
    return threw ? 0 : 3;
}
