"""The C++ drop-in headers (iterative_solvers_amd/compat/) driven the way the reference's own
callers drive its classes; output compared with the CPU oracle."""
import json
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cpp", "compat_driver.cpp")
EXE = os.path.join(ROOT, "tests", "cpp", "compat_driver")
HEADERS = [os.path.join(ROOT, "iterative_solvers_amd", "compat", "mi355cg_compat.hpp"), os.path.join(ROOT, "include", "mi355cg.h")]


def _stale(exe, src):
    """The binaries embed struct layouts of the C ABI: rebuild when a header is newer, not only the source."""
    return not os.path.exists(exe) or os.path.getmtime(exe) < max(os.path.getmtime(f) for f in [src] + HEADERS)


def build_driver():
    from iterative_solvers_amd import build as b
    b.build()
    if _stale(EXE, SRC):
        subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-Wextra", "-Werror",
                               "-I", os.path.join(ROOT, "iterative_solvers_amd", "compat"), SRC,
                               "-L", os.path.join(ROOT, "iterative_solvers_amd"), "-lmi355cg",
                               "-Wl,-rpath," + os.path.join(ROOT, "iterative_solvers_amd"), "-o", EXE])
    return EXE


def test_compat_headers_compile_warning_free():
    """Not a GPU test: reference-style user code compiles against the shim headers and links."""
    build_driver()
    assert os.path.exists(EXE)


@pytest.mark.gpu
@pytest.mark.parametrize("N", [16, 64, 258])
def test_cpp_dropin_matches_oracle(N):
    from oracle.oracle import OracleGrid
    exe = build_driver()
    import tempfile
    tmp = tempfile.mkdtemp(prefix="mi355cg_compat_")
    out = subprocess.run([exe, str(N), "10000", tmp], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    j = json.loads(out.stdout)
    og = OracleGrid(N, N)
    # solver/main.cpp flow
    ref = og.msg_solve(eps_precision=1e-9, eps_residual=1e-9, eps_exact_error=1e-9)
    assert (j["msg_iterations"], j["msg_reason"], bool(j["msg_converged"])) == (ref.iterations, ref.stop_reason, ref.converged)
    assert j["msg_cb_its"] == [c[0] for c in ref.callbacks]
    x = np.array(j["msg_x"])
    assert np.abs(x - ref.x).max() <= 1e-9 * np.abs(ref.x).max()
    assert np.array_equal(np.array(j["msg_residual"]), og.apply(x) - og.rhs())          # spmv shim = bit-exact operator
    assert abs(j["msg_rmax"] - ref.final_residual_norm) / ref.initial_r_norm2 <= 1e-12
    assert j["msg_nnz"] == len(og.csr()[2])
    assert (j["csr_same_as_grid"], j["csr_error_norm_equal"]) == (1, 1)      # generic CSR path == stencil path, bit for bit
    xs, ys = og.node_coords()
    assert j["node0"] == [xs[0], ys[0]]
    # DirichletSolver flow (error criterion off by default)
    ref = og.msg_solve(eps_precision=1e-8, eps_residual=1e-8, eps_exact_error=-1.0)
    assert (j["ds_iterations"], bool(j["ds_converged"]), j["ds_completions"], j["ds_saved"], j["ds_roundtrip"]) == (ref.iterations, True, 1, 1, 1)
    sol = np.array(j["ds_solution"])
    assert np.array_equal(np.array(j["ds_residual"]), og.apply(sol) - og.rhs())
    assert np.array_equal(np.array(j["ds_error"]), sol - og.true_solution())
    assert j["ds_error_norm"] == pytest.approx(ref.final_error_norm, rel=1e-9)
    # the facade on a distributed grid (4 parts driven by this process): bit-identical results
    assert (j["dist_same_as_one_gpu"], j["dist_iterations"]) == (1, ref.iterations) and j["dist_callbacks"] >= 3
    # matrix-free pair
    mf = og.mf_solve(eps=1e-8, max_iterations=10 ** 6)
    assert (j["mf_iterations"], j["mf_completed_ok"]) == (mf.iterations, 1)
    assert (j["rccl_team_iterations"], j["rccl_team_same_as_handle"]) == (mf.iterations, 1)      # team surface of the C ABI, called from C++
    assert np.array_equal(np.array(j["mf_apply_ones"]), og.apply(np.ones(og.size)))
    assert np.abs(np.array(j["mf_x"]) - mf.x).max() <= 1e-9 * np.abs(mf.x).max()
    # ---- file formats, restated here from solver/dirichlet_solver.cpp:255-457 and
    # solver/msg_solver.cpp:261-304: C++ default stream format = %g, std::scientific = %e
    sci = lambda v: "%e" % v
    name = "Метод серединных градиентов"
    exp = ["PARAMETERS", f"{N} {N}", "1 2 1 2", name, "CONVERGENCE", str(j["ds_iterations"]), "1", j["ds_stop_reason"],
           f"{sci(j['ds_residual_norm'])} {sci(j['ds_error_norm'])}"]
    for tag, key in (("SOLUTION", "ds_solution"), ("TRUE_SOLUTION", "ds_true_solution"), ("RESIDUAL", "ds_residual"),
                     ("ERROR", "ds_error"), ("X_COORDS", "ds_x_coords"), ("Y_COORDS", "ds_y_coords")):
        exp.append(tag)
        exp.extend(sci(v) for v in j[key])
    assert open(os.path.join(tmp, "results.txt"), encoding="utf-8").read() == "\n".join(exp) + "\n"
    row_map, entries, values = og.csr()
    expm = ["MATRIX_INFO", f"{N} {N}", f"{og.size} {len(values)}", "MATRIX"] + [str(v) for v in row_map] + \
           [str(v) for v in entries] + [sci(v) for v in values] + ["RHS"] + [sci(v) for v in og.rhs()]
    assert open(os.path.join(tmp, "matrix.txt")).read() == "\n".join(expm) + "\n"
    rep = open(os.path.join(tmp, "report.txt"), encoding="utf-8").read()
    assert rep.startswith("ОТЧЕТ О РЕШЕНИИ ЗАДАЧИ ДИРИХЛЕ\n===========================\n\nПАРАМЕТРЫ ЗАДАЧИ:\n")
    assert f"Размер сетки: {N}x{N} внутренних узлов\nОбласть: [1, 2] x [1, 2]\nШаг по x: {'%g' % (1.0 / (N + 1))}\n" in rep
    assert f"Общее количество неизвестных: {N * N}\n" in rep and f"Название метода: {name}\nМаксимальное число итераций: 10000\n" in rep
    assert f"Выполнено итераций: {j['ds_iterations']}\nСходимость: Да\nПричина остановки: {j['ds_stop_reason']}\n" in rep
    assert f"  - Норма невязки ||Ax-b||: {sci(j['ds_residual_norm'])}\n  - Норма ошибки ||u-x||: {sci(j['ds_error_norm'])}\n" in rep
    assert rep.endswith("- Для сравнения с истинным решением используется функция u(x,y) = exp(x^2 - y^2)\n")


def _build_cli():
    from iterative_solvers_amd import build as b
    b.build()
    src = os.path.join(ROOT, "iterative_solvers_amd", "cli", "solver_main.cpp")
    exe = os.path.join(ROOT, "iterative_solvers_amd", "cli", "solver_cli")
    if _stale(exe, src):
        subprocess.check_call(["g++", "-std=c++17", "-O2", "-Wall", "-Wextra", "-Werror", "-I",
                               os.path.join(ROOT, "iterative_solvers_amd", "compat"), src, "-L",
                               os.path.join(ROOT, "iterative_solvers_amd"), "-lmi355cg",
                               "-Wl,-rpath," + os.path.join(ROOT, "iterative_solvers_amd"), "-o", exe])
    return exe


def test_console_driver_compiles():
    assert os.path.exists(_build_cli())


@pytest.mark.gpu
def test_console_driver_config1_plumbing():
    """BASELINE config 1: 256 x 256 through the console flow of solver/main.cpp (stdin prompts, defaults
    eps 1e-9 / 2 iterations), then the same grid converged with flags."""
    exe = _build_cli()
    out = subprocess.run([exe], input="256 256\n", capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "Создание сетки размером 256x256 для области [1,2] x [1,2]" in out.stdout
    assert "Итераций: 2\n" in out.stdout and "Неизвестных: 48641, итераций: 2" in out.stdout
    out = subprocess.run([exe, "--n", "256", "--eps", "1e-8", "--max-iter", "100000", "--rule", "rel2"],
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "итераций: 701," in out.stdout            # the reference's own count
    out = subprocess.run([exe, "--n", "7"], capture_output=True, text=True, timeout=60)
    assert out.returncode == 1 and "rejected" in out.stderr


def _build_worker():
    from iterative_solvers_amd import build as b
    b.build()
    src = os.path.join(ROOT, "tests", "cpp", "worker_driver.cpp")
    exe = os.path.join(ROOT, "tests", "cpp", "worker_driver")
    if _stale(exe, src):
        subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-Wextra", "-Werror", "-pthread",
                               "-I", os.path.join(ROOT, "iterative_solvers_amd", "compat"), src,
                               "-L", os.path.join(ROOT, "iterative_solvers_amd"), "-lmi355cg",
                               "-Wl,-rpath," + os.path.join(ROOT, "iterative_solvers_amd"), "-o", exe])
    return exe


def test_worker_driver_compiles():
    assert os.path.exists(_build_worker())


@pytest.mark.gpu
@pytest.mark.parametrize("poll", [0, 1, 10])
def test_solver_worker_thread_and_stop_from_the_gui_thread(poll):
    """SURVEY 8f row f4, qt_gui/src/mainwindow.cpp:46-68,234-258: solve() on a worker thread, requestStop() from another
    thread while it runs, callbacks on the solving thread.  poll = iterations between two looks at the flag (0: default
    100; the reference looks every iteration, msg_solver.cpp:82-87)."""
    exe = _build_worker()
    out = subprocess.run([exe, "256", str(poll)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    j = json.loads(out.stdout.strip().splitlines()[-1])
    assert j["started"] == 1 and j["was_running"] == 1                       # the request arrived in the middle of the solve
    assert j["stop_reason"] == "Прервано пользователем" and j["converged"] == 0      # msg_solver.cpp:82-87, msg_solver.hpp:96-97
    assert j["iterations"] > 1 and j["last_callback_it"] == j["iterations"]  # final callback (msg_solver.cpp:193-195)
    assert j["callback_on_worker_thread"] == 1 and j["main_is_not_worker"] == 1
    assert j["stop_latency_ms"] < 1000
