"""GPU parity tests: the HIP path (through the C ABI, libmi355cg.so) against the CPU oracle on
the same inputs, against the committed golden fixture, and -- at BASELINE's full size -- through
size-independent properties.

Bars: bit-exact for everything the reference computes element-wise (RHS, exact solution,
stencil apply, packed index map); for CG, identical iteration counts / stop reasons and residual
norms within 1e-12 relative to ||b||_2 (north_star's tolerance; the only arithmetic that differs
from the oracle is the summation order of the inner products).
"""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

REL_TOL = 1e-12          # | ||r||_gpu - ||r||_oracle | / ||b||_2


@pytest.fixture(scope="module")
def isa():
    import torch
    assert torch.cuda.is_available(), "these tests need the MI355X"
    import iterative_solvers_amd as isa
    isa.load()
    return isa


@pytest.fixture(scope="module")
def oracle():
    from oracle import oracle as o
    return o


SIZES_SMALL = [6, 8, 10, 16, 30, 64, 66, 130, 256, 258]


@pytest.mark.parametrize("N", SIZES_SMALL)
def test_setup_vectors_bit_exact(isa, oracle, N):
    s = isa.GridSystem(N, N, 1.0, 2.0, 1.0, 2.0)
    og = oracle.OracleGrid(N, N, 1.0, 2.0, 1.0, 2.0)
    assert s.size() == og.size == (N // 2 - 1) * (3 * N // 2 - 1)
    assert np.array_equal(s.get_rhs(), og.rhs())
    assert np.array_equal(s.get_true_solution_vector(), og.true_solution())
    xs, ys = og.node_coords()
    assert np.array_equal(s.get_x_coords(), xs) and np.array_equal(s.get_y_coords(), ys)
    assert s.get_node_coordinates(0) == (xs[0], ys[0])
    assert s.get_node_coordinates(-1) == (0.0, 0.0) and s.get_node_coordinates(s.size()) == (0.0, 0.0)


@pytest.mark.parametrize("N", SIZES_SMALL + [1024])
def test_apply_bit_exact_seeded(isa, oracle, N):
    s = isa.MatrixFreeSystem(N, N, 1.0, 2.0, 1.0, 2.0)
    og = oracle.OracleGrid(N, N, 1.0, 2.0, 1.0, 2.0)
    rng = np.random.default_rng(12345)
    for _ in range(2):
        x = rng.uniform(-1.0, 1.0, s.size())
        assert np.array_equal(s.apply(x), og.apply(x))
    assert np.array_equal(s * np.ones(s.size()), og.apply(np.ones(s.size())))


def test_apply_other_domain(isa, oracle):
    s = isa.MatrixFreeSystem(32, 32, 0.0, 1.0, -1.0, 3.0)     # x_k != y_k
    og = oracle.OracleGrid(32, 32, 0.0, 1.0, -1.0, 3.0)
    x = np.random.default_rng(7).standard_normal(s.size())
    assert np.array_equal(s.apply(x), og.apply(x))
    assert np.array_equal(s.get_rhs(), og.rhs())


def test_operator_equals_check_py_matrix(isa, golden_n6):
    s = isa.MatrixFreeSystem(6, 6, 1.0, 2.0, 1.0, 2.0)
    A = np.array(golden_n6["A"])
    for j in range(16):
        e = np.zeros(16)
        e[j] = 1.0
        assert np.array_equal(s.apply(e), A[:, j])


def test_golden_two_iteration_trace(isa, golden_n6):
    """py_debug.txt through the GPU path: 8-decimal b of check_debug.py, two CG iterations."""
    s = isa.MatrixFreeSystem(6, 6, 1.0, 2.0, 1.0, 2.0)
    b = np.array(golden_n6["b_check_debug"])
    t = golden_n6["trace"]
    for k, key in ((1, "x1"), (2, "x2")):
        m = isa.MSGSolver(s, b, 1e-6, k)
        m.setPrecisionEps(-1.0); m.setResidualEps(-1.0); m.setExactErrorEps(-1.0)
        x = m.solve(None)
        assert m.getIterations() == k and not m.hasConverged() and m.getStopReason() == isa.StopCriterion.ITERATIONS
        assert np.allclose(x, t[key], rtol=1e-11, atol=1e-14)
    r = s._handle.recursive_residual()
    assert np.allclose(-r, t["r2"], rtol=1e-9, atol=1e-10)          # script keeps r = A x - b
    mf = isa.MatrixFreeSolver(s, b, 0.0, 2)
    assert np.allclose(mf.solve(), t["x2"], rtol=1e-11, atol=1e-14)


@pytest.mark.parametrize("N", [6, 16, 64, 256])
def test_cg_rel2norm_matches_oracle(isa, oracle, N):
    s = isa.MatrixFreeSystem(N, N, 1.0, 2.0, 1.0, 2.0)
    og = oracle.OracleGrid(N, N, 1.0, 2.0, 1.0, 2.0)
    ref = og.mf_solve(eps=1e-8, max_iterations=10 ** 6)
    sol = isa.MatrixFreeSolver(s, s.get_rhs(), 1e-8, 10 ** 6)
    done = []
    sol.setCompletionCallback(lambda ok, msg: done.append((ok, msg)))
    x = sol.solve()
    assert sol.getIterations() == ref.iterations
    assert done == [(True, "Converged successfully")]
    res = sol.last_results
    assert res.initial_r_norm2 == pytest.approx(ref.initial_r_norm, rel=1e-14)
    assert abs(res.r_norm2 - ref.r_norm) / ref.initial_r_norm <= REL_TOL
    assert np.abs(x - ref.x).max() <= 1e-9 * np.abs(ref.x).max()
    b = og.rhs()
    true_gpu = np.linalg.norm(b - og.apply(x))
    true_ref = np.linalg.norm(b - og.apply(ref.x))
    assert abs(true_gpu - true_ref) / ref.initial_r_norm <= REL_TOL


def test_n256_reference_run(isa, oracle):
    """the reference's own MatrixFreeSolver needs 701 iterations at N=256."""
    s = isa.MatrixFreeSystem(256, 256, 1.0, 2.0, 1.0, 2.0)
    sol = isa.MatrixFreeSolver(s, s.get_rhs(), 1e-8, 10 ** 6)
    x = sol.solve()
    assert sol.getIterations() == 701
    og = oracle.OracleGrid(256, 256)
    assert np.linalg.norm(og.rhs() - og.apply(x)) == pytest.approx(1.225611e-01, rel=1e-4)


@pytest.mark.parametrize("N", [16, 64])
def test_cg_diagnostics_callback_trace(isa, oracle, N):
    """Per-iteration (precision, TRUE residual, error) 2-norms of MatrixFreeSolver's callback."""
    s = isa.MatrixFreeSystem(N, N, 1.0, 2.0, 1.0, 2.0)
    og = oracle.OracleGrid(N, N, 1.0, 2.0, 1.0, 2.0)
    ref = og.mf_solve(eps=1e-8, max_iterations=10 ** 6, diagnostics=True)
    got = []
    sol = isa.MatrixFreeSolver(s, s.get_rhs(), 1e-8, 10 ** 6)
    sol.setIterationCallback(lambda it, p, r, e: got.append((it, p, r, e)))
    x = sol.solve(s.get_true_solution_vector())
    assert [g[0] for g in got] == [c[0] for c in ref.callbacks]
    g, c = np.array(got), np.array(ref.callbacks)
    assert np.abs(g[:, 2] - c[:, 2]).max() / ref.initial_r_norm <= REL_TOL       # true residual
    assert np.allclose(g[:, 1], c[:, 1], rtol=1e-8, atol=1e-13)                   # ||dx||_2
    assert np.allclose(g[:, 3], c[:, 3], rtol=1e-8, atol=1e-13)                   # ||x-u||_2
    assert np.abs(x - ref.x).max() <= 1e-9 * np.abs(ref.x).max()


def _msg_compare(isa, og, s, **eps):
    ref = og.msg_solve(**eps)
    m = isa.MSGSolver(s, s.get_rhs(), 1e-6, eps.get("max_iterations", 10000))
    m.setPrecisionEps(eps.get("eps_precision", 1e-6))
    m.setResidualEps(eps.get("eps_residual", 1e-6))
    m.setExactErrorEps(eps.get("eps_exact_error", -1.0))
    got = []
    m.setIterationCallback(lambda it, p, r, e: got.append((it, p, r, e)))
    x = m.solve(s.get_true_solution_vector())
    return ref, m, x, got


@pytest.mark.parametrize("N", [6, 16, 64, 256])
def test_msg_defaults_match_oracle(isa, oracle, N):
    s = isa.GridSystem(N, N, 1.0, 2.0, 1.0, 2.0)
    og = oracle.OracleGrid(N, N, 1.0, 2.0, 1.0, 2.0)
    ref, m, x, got = _msg_compare(isa, og, s)
    assert (m.getIterations(), int(m.getStopReason()), m.hasConverged()) == (ref.iterations, ref.stop_reason, ref.converged)
    assert [g[0] for g in got] == [c[0] for c in ref.callbacks]
    g, c = np.array(got), np.array(ref.callbacks)
    assert g[0, 1] == c[0, 1]                                   # DBL_MAX at iteration 0
    b2 = ref.initial_r_norm2
    assert np.abs(g[:, 2] - c[:, 2]).max() / b2 <= REL_TOL       # ||r||_inf history
    assert np.allclose(g[1:, 1], c[1:, 1], rtol=1e-7, atol=1e-14)
    assert np.allclose(g[:, 3], c[:, 3], rtol=1e-9, atol=1e-14)
    assert abs(m.getFinalResidualNorm() - ref.final_residual_norm) / b2 <= REL_TOL
    assert m.getFinalErrorNorm() == pytest.approx(ref.final_error_norm, rel=1e-9)
    assert np.abs(x - ref.x).max() <= 1e-9 * np.abs(ref.x).max()
    assert m.getStopReasonText() != ""


def test_msg_n256_known_counts(isa, oracle):
    s = isa.GridSystem(256, 256, 1.0, 2.0, 1.0, 2.0)
    og = oracle.OracleGrid(256, 256)
    ref, m, x, got = _msg_compare(isa, og, s, eps_precision=-1.0, eps_residual=1e-8)
    assert (m.getIterations(), m.getStopReason()) == (1004, isa.StopCriterion.RESIDUAL) == (ref.iterations, ref.stop_reason)
    assert m.getFinalResidualNorm() < 1e-8
    ref, m, x, got = _msg_compare(isa, og, s, eps_precision=1e-8, eps_residual=1e-8, eps_exact_error=1e-8)
    assert (m.getIterations(), m.getStopReason()) == (757, isa.StopCriterion.PRECISION)


def test_msg_error_criterion_and_no_true_solution(isa, oracle):
    N = 32
    s = isa.GridSystem(N, N, 1.0, 2.0, 1.0, 2.0)
    og = oracle.OracleGrid(N, N)
    ref, m, x, got = _msg_compare(isa, og, s, eps_precision=-1.0, eps_residual=-1.0, eps_exact_error=1e-2)
    assert (m.getIterations(), int(m.getStopReason())) == (ref.iterations, ref.stop_reason)
    assert m.getStopReason() == isa.StopCriterion.EXACT_ERROR
    # extent-0 true solution: error norm stays DBL_MAX, error criterion can never fire
    m = isa.MSGSolver(s, s.get_rhs(), 1e-6, 50)
    m.setPrecisionEps(-1.0); m.setResidualEps(-1.0); m.setExactErrorEps(1e30)
    got = []
    m.setIterationCallback(lambda it, p, r, e: got.append(e))
    m.solve(None)
    assert m.getIterations() == 50 and not m.hasConverged()
    assert all(e == np.finfo(np.float64).max for e in got)
    ref = og.msg_solve(true_solution=None, eps_precision=-1.0, eps_residual=-1.0, eps_exact_error=1e30, max_iterations=50)
    assert abs(m.getFinalResidualNorm() - ref.final_residual_norm) / ref.initial_r_norm2 <= REL_TOL


def test_iteration_caps_and_zero_iterations(isa):
    s = isa.GridSystem(16, 16, 1.0, 2.0, 1.0, 2.0)
    m = isa.MSGSolver(s, s.get_rhs(), 1e-30, 0)
    x = m.solve(s.get_true_solution_vector())
    assert m.getIterations() == 0 and not m.hasConverged() and np.all(x == 0.0)
    mf = isa.MatrixFreeSolver(s, s.get_rhs(), 1e-30, 3)
    msgs = []
    mf.setCompletionCallback(lambda ok, msg: msgs.append((ok, msg)))
    mf.solve()
    assert mf.getIterations() == 3 and msgs == [(False, "Failed to converge within maximum iterations")]


def test_stop_request_interrupts(isa):
    s = isa.GridSystem(128, 128, 1.0, 2.0, 1.0, 2.0)
    m = isa.MSGSolver(s, s.get_rhs(), 1e-30, 100000)
    m.setPrecisionEps(-1.0); m.setResidualEps(-1.0); m.setExactErrorEps(-1.0)
    m.setIterationCallback(lambda it, p, r, e: m.requestStop() if it == 200 else None)
    m.solve(s.get_true_solution_vector())
    assert m.getStopReason() == isa.StopCriterion.INTERRUPTED and not m.hasConverged()
    assert m.getIterations() == 200 and m.isStopRequested()


def test_stop_requested_from_the_first_callback_stops_at_iteration_one(isa, oracle):
    """msg_solver.cpp:82-87 polls the flag at the top of every iteration: a stop requested from the it == 1 callback
    leaves iterations == 1 in the reference.  (Here the first iteration of a watched solve is a chunk of its own.)"""
    s = isa.GridSystem(64, 64, 1.0, 2.0, 1.0, 2.0)
    m = isa.MSGSolver(s, s.get_rhs(), 1e-30, 100000)
    m.setPrecisionEps(-1.0); m.setResidualEps(-1.0); m.setExactErrorEps(-1.0)
    seen = []
    m.setIterationCallback(lambda it, p, r, e: (seen.append(it), m.requestStop() if it == 1 else None))
    x = m.solve(s.get_true_solution_vector())
    assert m.getStopReason() == isa.StopCriterion.INTERRUPTED and m.getIterations() == 1 and seen == [0, 1, 1]
    o = oracle.OracleGrid(64, 64).msg_solve(eps_precision=-1.0, eps_residual=-1.0, eps_exact_error=-1.0, max_iterations=1)
    np.testing.assert_allclose(x, o.x, rtol=1e-13)              # x after exactly one step (alpha from a differently summed dot)


def test_stop_request_from_another_thread_acts_at_the_next_iteration_not_at_the_next_poll(isa):
    """The reference tests its flag at the top of EVERY iteration (msg_solver.cpp:82-87).  Here the host queues chunks of up to 500
    iterations (65 ms at N = 4096), so the request has to be seen by the DEVICE: block 0 of every update launch samples a pinned word
    the solving thread keeps in step with the caller's flag, and the next stencil prologue turns it into INTERRUPTED.  No callback
    is involved: another thread raises the flag at an arbitrary moment and the solve is back a few milliseconds later (the rest of
    the chunk returns in its prologues), in the middle of a chunk."""
    import ctypes as C
    import threading
    import time
    s = isa.MatrixFreeSystem(4096, 4096, 1.0, 2.0, 1.0, 2.0)
    h = s._handle
    p = isa.default_params(isa.RULE_REL_2NORM)
    p.max_iterations, p.fixed_iterations, p.use_true_solution, p.callback_every, p.sync_every = 10 ** 6, 1, 0, 0, 500
    h.solve(isa.default_params(isa.RULE_REL_2NORM))                       # warm-up
    lags, its = [], []
    for delay in (0.21, 0.33, 0.27):
        stop = C.c_int(0)
        t_set = []

        def raiser():
            time.sleep(delay)
            t_set.append(time.perf_counter())
            stop.value = 1
        th = threading.Thread(target=raiser)
        th.start()
        res = h.solve(p, None, stop)
        t_back = time.perf_counter()
        th.join()
        assert res.stop_reason == isa.StopCriterion.INTERRUPTED and not res.converged
        assert 500 < res.iterations < 10 ** 5
        lags.append(t_back - t_set[0])
        its.append(res.iterations)
    assert max(lags) < 0.02, lags                                         # a poll-granular stop would lag 32 ms on average, up to 65
    assert any(i % 500 for i in its), its                                 # ... and end on a chunk boundary
    # the solver object's own flag (MSGSolver::requestStop from another thread), MSG rule
    g = isa.GridSystem(2048, 2048, 1.0, 2.0, 1.0, 2.0)
    m = isa.MSGSolver(g, g.get_rhs(), 1e-30, 10 ** 6)
    m.setPrecisionEps(-1.0); m.setResidualEps(-1.0); m.setExactErrorEps(-1.0)
    th = threading.Thread(target=lambda: (time.sleep(0.3), m.requestStop()))
    th.start()
    m.solve(g.get_true_solution_vector())
    th.join()
    assert m.getStopReason() == isa.StopCriterion.INTERRUPTED and not m.hasConverged() and 100 < m.getIterations() < 10 ** 5


def test_stop_reason_texts_are_the_references(isa):
    """solver/msg_solver.hpp:85-100, verbatim: the strings are part of SolverResults.stop_reason and of the report."""
    want = {0: "Достигнуто максимальное число итераций",
            1: "Достигнута требуемая точность по норме разности xn и xn-1",
            2: "Достигнута требуемая точность по норме невязки",
            3: "Достигнута требуемая точность по норме разности с истинным решением",
            4: "Прервано пользователем"}
    s = isa.GridSystem(16, 16, 1.0, 2.0, 1.0, 2.0)
    m = isa.MSGSolver(s, s.get_rhs(), 1e-6, 10)
    for k, text in want.items():
        m.stop_reason = isa.StopCriterion(k)
        assert m.getStopReasonText() == text
    m.stop_reason = 17
    assert m.getStopReasonText() == "Неизвестная причина остановки"


@pytest.mark.parametrize("rule", ["msg", "rel2"])
def test_apply_inside_an_iteration_callback_does_not_disturb_the_solve(isa, oracle, rule):
    """The reference's apply is const: a callback may call it (or spmv) in the middle of a solve.  Here it works on
    dedicated scratch vectors; the solve must still match the oracle exactly as without the call."""
    N = 64
    s = isa.GridSystem(N, N, 1.0, 2.0, 1.0, 2.0)
    og = oracle.OracleGrid(N, N)
    v = np.random.default_rng(5).uniform(-1, 1, s.size())
    inside = []
    if rule == "msg":
        m = isa.MSGSolver(s, s.get_rhs(), 1e-9, 10000)
        m.setExactErrorEps(-1.0)
        m.setIterationCallback(lambda it, p, r, e: inside.append(s.apply(v)) if it in (1, 100) else None)
        x = m.solve(s.get_true_solution_vector())
        ref = og.msg_solve(eps_precision=1e-9, eps_residual=1e-9, eps_exact_error=-1.0)
        assert (m.getIterations(), int(m.getStopReason())) == (ref.iterations, ref.stop_reason)
    else:
        m = isa.MatrixFreeSolver(s, s.get_rhs(), 1e-9, 10 ** 5)
        m.setIterationCallback(lambda it, p, r, e: inside.append(s.apply(v)) if it in (0, 7) else None)
        x = m.solve(s.get_true_solution_vector())
        ref = og.mf_solve(eps=1e-9, max_iterations=10 ** 5)
        assert m.getIterations() == ref.iterations
    assert len(inside) == 2 and all(np.array_equal(y, og.apply(v)) for y in inside)
    assert np.abs(x - ref.x).max() <= 1e-9 * np.abs(ref.x).max()
    s2 = isa.GridSystem(N, N, 1.0, 2.0, 1.0, 2.0)                # the same solve without the calls: same bits
    m2 = isa.MSGSolver(s2, s2.get_rhs(), 1e-9, 10000) if rule == "msg" else isa.MatrixFreeSolver(s2, s2.get_rhs(), 1e-9, 10 ** 5)
    if rule == "msg":
        m2.setExactErrorEps(-1.0)
    else:
        m2.setIterationCallback(lambda *a: None)
    assert np.array_equal(m2.solve(s2.get_true_solution_vector()), x)


def test_caller_supplied_true_solution_is_the_one_the_error_norm_uses(isa, oracle):
    """MSGSolver::solve(true_solution) measures x - true_solution (msg_solver.cpp:64-72,132-139), whatever vector that is."""
    N = 32
    s = isa.GridSystem(N, N, 1.0, 2.0, 1.0, 2.0)
    og = oracle.OracleGrid(N, N)
    u2 = og.true_solution() + 0.25
    m = isa.MSGSolver(s, s.get_rhs(), 1e-9, 10000)
    m.setExactErrorEps(-1.0)
    m.solve(u2)
    ref = og.msg_solve(true_solution=u2, eps_precision=1e-9, eps_residual=1e-9, eps_exact_error=-1.0)
    assert m.getIterations() == ref.iterations
    assert m.getFinalErrorNorm() == pytest.approx(ref.final_error_norm, rel=1e-12) and m.getFinalErrorNorm() > 0.2


def test_dirichlet_solver_facade(isa, oracle):
    N = 64
    d = isa.DirichletSolver(N, N, 1.0, 2.0, 1.0, 2.0)
    d.setSolverParameters(1e-8, 1e-8, 1e-8, 100000)
    its = []
    d.setIterationCallback(lambda it, p, r, e: its.append(it))
    fin = []
    d.setCompletionCallback(lambda res: fin.append(res.iterations))
    res = d.solve()
    og = oracle.OracleGrid(N, N)
    ref = og.msg_solve(eps_precision=1e-8, eps_residual=1e-8, eps_exact_error=-1.0, max_iterations=100000)
    assert res.iterations == ref.iterations and res.converged and fin == [res.iterations]
    assert its[0] == 0 and its[1] == 1 and its[-1] == res.iterations
    assert np.array_equal(res.residual, og.apply(res.solution) - og.rhs())      # A x - b with the bit-exact operator
    assert np.array_equal(res.error, res.solution - res.true_solution)
    assert res.error_norm == pytest.approx(np.abs(res.error).max(), rel=1e-12)
    assert res.stop_reason == d.solver.getStopReasonText()
    assert len(res.x_coords) == len(res.solution) == og.size


@pytest.mark.parametrize("n,m", [(7, 7), (8, 6), (6, 8), (4, 4), (9, 9)])
def test_invalid_grids_rejected(isa, n, m):
    with pytest.raises(ValueError):
        isa.GridSystem(m, n, 1.0, 2.0, 1.0, 2.0)


def test_errors_are_loud(isa):
    s = isa.MatrixFreeSystem(8, 8, 1.0, 2.0, 1.0, 2.0)
    with pytest.raises(ValueError):
        s.apply(np.zeros(5))
    with pytest.raises(isa.Mi355cgError):
        s._handle.solution()                                     # no solve yet


def test_full_size_4096_apply_and_cg_properties(isa, oracle):
    """BASELINE config 2 size.  Stencil bit-exact against the oracle; 25 CG iterations compared
    with the oracle's; recursive residual consistent with the true residual."""
    N = 4096
    s = isa.GridSystem(N, N, 1.0, 2.0, 1.0, 2.0)
    og = oracle.OracleGrid(N, N)
    assert s.size() == 12574721
    b = og.rhs()
    assert np.array_equal(s.get_rhs(), b)
    x = np.random.default_rng(12345).uniform(-1.0, 1.0, s.size())
    assert np.array_equal(s.apply(x), og.apply(x))
    k = 25
    m = isa.MSGSolver(s, b, 1e-30, k)
    m.setPrecisionEps(-1.0); m.setResidualEps(-1.0); m.setExactErrorEps(-1.0)
    xg = m.solve(s.get_true_solution_vector())
    ref = og.msg_solve(eps_precision=-1.0, eps_residual=-1.0, eps_exact_error=-1.0, max_iterations=k)
    assert m.getIterations() == k == ref.iterations
    # 25 iterations in, the residual is still ~1e8.  The oracle (like the reference) sums 12.6 M
    # products serially; that sum alone carries up to U*eps ~ 1e-9 relative rounding error, which CG
    # amplifies.  The GPU's fixed pairwise tree is the MORE accurate of the two (see
    # test_dot_products_are_more_accurate_than_serial), so mid-run agreement is bounded by the
    # reference's own summation error, not by 1e-12.
    assert m.getFinalResidualNorm() == pytest.approx(ref.final_residual_norm, rel=2e-8)
    assert m.last_results.r_norm2 == pytest.approx(ref.r_norm2, rel=2e-8)
    assert np.abs(xg - ref.x).max() <= 1e-9 * np.abs(ref.x).max()
    rg = s._handle.recursive_residual()
    assert np.abs(rg - (b - og.apply(xg))).max() <= 1e-9 * np.abs(b).max()


@pytest.mark.parametrize("N,k", [(64, 10 ** 5), (258, 10 ** 5), (514, 10 ** 5), (1026, 400), (4096, 40)])
def test_gpu_equals_the_oracle_with_exact_inner_products(isa, oracle, N, k):
    """What separates the GPU from the reference is the ORDER in which the inner products are summed, and nothing else: give the
    oracle inner products evaluated as if in twice the working precision (oracle.exact_dots: Dot2; every other operation of the
    loop is elementwise and untouched) and the GPU reproduces it BIT FOR BIT -- x, the residual, every norm, the iteration count
    and stop reason of full solves (N <= 514: to 1e-8 / 1e-9) and the first hundreds of iterations at the larger sizes."""
    s = isa.GridSystem(N, N, 1.0, 2.0, 1.0, 2.0)
    og = oracle.OracleGrid(N, N)
    b = og.rhs()
    sol = isa.MatrixFreeSolver(s, b, 1e-8, k)
    xg = sol.solve()
    with oracle.exact_dots():
        ref = og.mf_solve(eps=1e-8, max_iterations=k)
    assert (sol.getIterations(), bool(sol.last_results.converged)) == (ref.iterations, ref.converged)
    assert (sol.last_results.r_norm2, sol.last_results.initial_r_norm2) == (ref.r_norm, ref.initial_r_norm)
    assert np.array_equal(xg, ref.x)
    m = isa.MSGSolver(s, b, 1e-9, k)
    m.setPrecisionEps(1e-9); m.setResidualEps(1e-9); m.setExactErrorEps(1e-9)
    cbs = []
    m.setIterationCallback(lambda *a: cbs.append(a))
    xm = m.solve(s.get_true_solution_vector())
    with oracle.exact_dots():
        rm = og.msg_solve(eps_precision=1e-9, eps_residual=1e-9, eps_exact_error=1e-9, max_iterations=k)
    assert (m.getIterations(), int(m.getStopReason()), m.hasConverged()) == (rm.iterations, rm.stop_reason, rm.converged)
    assert (m.getFinalResidualNorm(), m.getFinalPrecision(), m.getFinalErrorNorm()) == (rm.final_residual_norm, rm.final_precision, rm.final_error_norm)
    assert [tuple(c) for c in cbs] == [tuple(c) for c in rm.callbacks]            # iteration numbers and all three norms of every callback
    assert np.array_equal(xm, rm.x) and np.array_equal(s._handle.recursive_residual(), rm.r)


def test_n8192_fp64_against_the_oracle(isa, oracle):
    """The first size that lives in HBM rather than in the Infinity Cache (50 M unknowns, 400 MB per vector; configs 3-5 run at
    such sizes): set-up vectors and the operator bit for bit; 10 CG iterations bit for bit against the oracle with exact inner
    products, and within the reach of its serial sums (~U * 2^-53 per inner product, amplified by CG) against the oracle proper."""
    import math
    N = 8192
    s = isa.MatrixFreeSystem(N, N, 1.0, 2.0, 1.0, 2.0)
    og = oracle.OracleGrid(N, N)
    assert s.size() == og.size == 50315265
    b = og.rhs()
    assert np.array_equal(s.get_rhs(), b) and np.array_equal(s.get_true_solution_vector(), og.true_solution())
    v = np.random.default_rng(12345).uniform(-1.0, 1.0, s.size())
    assert np.array_equal(s.apply(v), og.apply(v))
    del v
    k = 10
    sol = isa.MatrixFreeSolver(s, b, 1e-30, k)
    xg = sol.solve()
    with oracle.exact_dots():
        ex = og.mf_solve(eps=1e-30, max_iterations=k)
    assert sol.getIterations() == k == ex.iterations
    assert (sol.last_results.r_norm2, sol.last_results.initial_r_norm2) == (ex.r_norm, ex.initial_r_norm)
    assert np.array_equal(xg, ex.x)
    exact = math.sqrt(math.fsum(np.square(b)))
    assert abs(sol.last_results.initial_r_norm2 - exact) <= 2e-15 * exact
    ref = og.mf_solve(eps=1e-30, max_iterations=k)                                   # the reference's own serial sums
    assert sol.last_results.initial_r_norm2 == pytest.approx(ref.initial_r_norm, rel=1e-9)
    assert sol.last_results.r_norm2 == pytest.approx(ref.r_norm, rel=2e-8)
    assert np.abs(xg - ref.x).max() <= 1e-8 * np.abs(ref.x).max()


def test_config4_grid_against_the_oracle_with_exact_inner_products(isa, oracle):
    """N = 16384 (BASELINE config 4's grid, 201 M unknowns, 1.6 GB per vector) on one GPU: right-hand side and 4 CG iterations bit
    for bit against the oracle with exact inner products.  test_gpu_team.py ties the 2 x 2 and slab teams of this size to the same
    single-GPU bits."""
    N = 16384
    s = isa.MatrixFreeSystem(N, N, 1.0, 2.0, 1.0, 2.0)
    og = oracle.OracleGrid(N, N)
    assert s.size() == og.size == 201293825
    b = og.rhs()
    assert np.array_equal(s.get_rhs(), b)
    k = 4
    sol = isa.MatrixFreeSolver(s, b, 1e-30, k)
    xg = sol.solve()
    with oracle.exact_dots():
        ex = og.mf_solve(eps=1e-30, max_iterations=k)
    assert sol.getIterations() == k == ex.iterations
    assert (sol.last_results.r_norm2, sol.last_results.initial_r_norm2) == (ex.r_norm, ex.initial_r_norm)
    assert np.array_equal(xg, ex.x)


def test_config5_grid_against_the_oracle(isa, oracle):
    """N = 32768 (BASELINE config 5's grid: 805 240 833 unknowns, 6.4 GB per vector, 58 GB of vectors on the one GPU).  Against the
    oracle with exact inner products: right-hand side, ||r0||, ||r|| and ALL of x after 2 CG iterations, bit for bit; against the oracle
    proper (the reference's serial sums, matrix_free_system.cpp:364-366, 383-441): ||r|| within 1e-11 * ||b|| (see below).
    test_gpu_team.py ties the 8-part teams of this size to the same single-GPU bits.  Costs ~2.5 minutes (two serial CPU solves of
    two iterations each on 805 M unknowns, ~25 s per iteration) and ~70 GB of host memory."""
    N = 32768
    s = isa.MatrixFreeSystem(N, N, 1.0, 2.0, 1.0, 2.0)
    og = oracle.OracleGrid(N, N)
    assert s.size() == og.size == 805240833
    b = og.rhs()
    assert np.array_equal(s.get_rhs(), b)
    k = 2
    sol = isa.MatrixFreeSolver(s, b, 1e-30, k)
    xg = sol.solve()
    res = sol.last_results
    with oracle.exact_dots():
        ex = og.mf_solve(eps=1e-30, max_iterations=k)
    assert sol.getIterations() == k == ex.iterations
    assert (res.r_norm2, res.initial_r_norm2) == (ex.r_norm, ex.initial_r_norm)
    assert np.array_equal(xg, ex.x)
    del ex, xg
    # The reference's own serial sums.  A serial fp64 sum of U = 8e8 terms is itself only good to ~sqrt(U) * 2^-53 = 3e-12 relative
    # (U * 2^-53 = 9e-8 at worst), and two CG steps pass that on to ||r||: measured 2.7e-12 * ||b||.  So at THIS size north_star's
    # 1e-12 is inside the reference's own rounding noise; the bound asserted is 1e-11 * ||b||, and the statement with content is the
    # bit-for-bit one above.
    ref = og.mf_solve(eps=1e-30, max_iterations=k)
    assert abs(res.r_norm2 - ref.r_norm) <= 1e-11 * ref.initial_r_norm
    assert abs(res.initial_r_norm2 - ref.initial_r_norm) <= 1e-11 * ref.initial_r_norm


def test_fixed_iteration_mode_ignores_convergence(isa):
    s = isa.MatrixFreeSystem(16, 16, 1.0, 2.0, 1.0, 2.0)
    sol = isa.MatrixFreeSolver(s, s.get_rhs(), 1e-2, 40)
    sol.solve(fixed_iterations=True)
    assert sol.getIterations() == 40


def test_dot_products_are_more_accurate_than_serial(isa, oracle):
    """||b||_2 from the GPU's pairwise reduction vs the exactly rounded value (math.fsum) and vs
    the reference-style serial sum: the only arithmetic that differs from the oracle errs on the
    accurate side."""
    import math
    N = 2048
    s = isa.MatrixFreeSystem(N, N, 1.0, 2.0, 1.0, 2.0)
    b = s.get_rhs()
    sol = isa.MatrixFreeSolver(s, b, 1e-8, 1)
    sol.solve()
    exact = math.sqrt(math.fsum((b * b).tolist()))
    serial = math.sqrt(oracle.dot(b, b))
    gpu = sol.last_results.initial_r_norm2
    assert abs(gpu - exact) <= 4 * np.spacing(exact)
    assert abs(gpu - exact) <= abs(serial - exact) + np.spacing(exact)


@pytest.mark.parametrize("N,iters", [(512, 1371), (1024, 2673)])
def test_reference_iteration_counts_large(isa, N, iters):
    """SURVEY section 6 / BASELINE.md section 3: iteration counts of the reference's own
    MatrixFreeSolver (eps 1e-8) recorded from running the reference."""
    s = isa.MatrixFreeSystem(N, N, 1.0, 2.0, 1.0, 2.0)
    sol = isa.MatrixFreeSolver(s, s.get_rhs(), 1e-8, 10 ** 6)
    sol.solve()
    assert sol.getIterations() == iters and sol.last_results.converged


@pytest.mark.parametrize("N,max_it,sync", [(16, 10 ** 6, 200), (16, 10 ** 6, 7), (64, 33, 200), (64, 34, 8), (130, 10 ** 6, 50)])
def test_two_step_x_update_is_bit_identical_to_stepwise(isa, N, max_it, sync):
    """REL_2NORM default path (7.5 words/unknown: x touched every second iteration, two steps at once, the last one
    flushed after an odd count) against the path that updates x every iteration (the per-iteration diagnostics mode):
    same arithmetic in the same order, so the same bits -- whatever the chunking, the stopping iteration's parity
    or an iteration cap."""
    out = []
    for stepwise in (False, True):
        s = isa.MatrixFreeSystem(N, N, 1.0, 2.0, 1.0, 2.0)
        sol = isa.MatrixFreeSolver(s, s.get_rhs(), 1e-9, max_it)
        if stepwise:
            sol.setIterationCallback(lambda *a: None)
        x = sol.solve(sync_every=sync)
        out.append((x, sol.getIterations(), s._handle.recursive_residual(), sol.last_results.r_norm2))
    assert out[0][1] == out[1][1]
    assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][2], out[1][2]) and out[0][3] == out[1][3]


def test_caller_supplied_rhs_and_repeated_solves(isa, oracle):
    """Solver(a, b, ...) with a b that is not the grid's own (solver.hpp:33-39), twice on one handle."""
    N = 48
    s = isa.GridSystem(N, N, 1.0, 2.0, 1.0, 2.0)
    og = oracle.OracleGrid(N, N)
    rng = np.random.default_rng(3)
    for trial in range(2):
        b = rng.standard_normal(s.size()) * 10.0 ** trial
        ref = og.mf_solve(b=b, eps=1e-9, max_iterations=10 ** 5)
        sol = isa.MatrixFreeSolver(s, b, 1e-9, 10 ** 5)
        x = sol.solve()
        assert sol.getIterations() == ref.iterations
        assert np.abs(x - ref.x).max() <= 1e-9 * np.abs(ref.x).max()
        refm = og.msg_solve(b=b, eps_precision=1e-9, eps_residual=1e-9, eps_exact_error=-1.0)
        m = isa.MSGSolver(s, b, 1e-9, 10000)
        m.setExactErrorEps(-1.0)
        xm = m.solve(s.get_true_solution_vector())
        assert (m.getIterations(), int(m.getStopReason())) == (refm.iterations, refm.stop_reason)
        assert np.abs(xm - refm.x).max() <= 1e-9 * np.abs(refm.x).max()
        assert m.getFinalErrorNorm() == pytest.approx(refm.final_error_norm, rel=1e-9)   # error vs u although b is foreign
    assert np.array_equal(s.apply(b), og.apply(b))               # the handle still applies the operator bit-exactly


def test_zero_rhs_matches_the_reference_breakdown_behaviour(isa, oracle):
    """b = 0: MatrixFreeSolver never enters its loop; MSGSolver divides 0/0 in its first iteration
    (msg_solver.cpp:102, unchecked) and then stops on the precision test because std::max ignores NaN."""
    N = 16
    s = isa.GridSystem(N, N, 1.0, 2.0, 1.0, 2.0)
    og = oracle.OracleGrid(N, N)
    b = np.zeros(s.size())
    sol = isa.MatrixFreeSolver(s, b, 1e-8, 100)
    x = sol.solve()
    ref = og.mf_solve(b=b, eps=1e-8, max_iterations=100)
    assert (sol.getIterations(), bool(sol.last_results.converged)) == (ref.iterations, ref.converged) == (0, True)
    assert np.array_equal(x, ref.x)
    m = isa.MSGSolver(s, b, 1e-6, 100)
    xm = m.solve(s.get_true_solution_vector())
    refm = og.msg_solve(b=b)
    assert (m.getIterations(), int(m.getStopReason()), m.hasConverged()) == (refm.iterations, refm.stop_reason, refm.converged)
    assert np.array_equal(np.isnan(xm), np.isnan(refm.x))


def test_apply_device_pointer_entry(isa, oracle):
    """mi355cg_apply_device: packed vectors already resident in HBM (torch tensors as plain pointers)."""
    import ctypes as C
    import torch
    from iterative_solvers_amd import _capi
    N = 130
    s = isa.MatrixFreeSystem(N, N, 1.0, 2.0, 1.0, 2.0)
    og = oracle.OracleGrid(N, N)
    x = torch.rand(s.size(), dtype=torch.float64, device="cuda") * 2 - 1
    y = torch.empty_like(x)
    torch.cuda.synchronize()
    _capi.check(_capi.load().mi355cg_apply_device(s._handle._h, C.c_void_p(x.data_ptr()), C.c_void_p(y.data_ptr())))
    assert np.array_equal(y.cpu().numpy(), og.apply(x.cpu().numpy()))


def test_kernel_timing_hooks(isa):
    s = isa.MatrixFreeSystem(256, 256, 1.0, 2.0, 1.0, 2.0)
    h = s._handle
    p = isa.default_params(isa.RULE_REL_2NORM)
    p.max_iterations, p.fixed_iterations = 50, 1
    h.set_profiling(True)
    h.solve(p)
    h.set_profiling(False)
    (ms0, n0), (ms1, n1) = h.kernel_time(0), h.kernel_time(1)
    assert n0 == n1 == 50 and 0 < ms0 < 5 and 0 < ms1 < 5
    lay = h.layout()
    assert lay["pitch_upper"] % 32 == 0 and lay["pitch_bottom"] % 32 == 0 and lay["padded_len"] >= s.size()


def test_n2048_against_long_oracle_runs(isa):
    """N = 2048 (3.1 M unknowns): iteration counts, stop reason and final norms of the CPU oracle's 11- and 14-minute
    serial solves (tests/golden/oracle_n2048.json, made by tests/golden/make_oracle_n2048.py)."""
    import json, os
    with open(os.path.join(os.path.dirname(__file__), "golden", "oracle_n2048.json")) as f:
        ref = json.load(f)
    s = isa.GridSystem(2048, 2048, 1.0, 2.0, 1.0, 2.0)
    sol = isa.MatrixFreeSolver(s, s.get_rhs(), 1e-8, 10 ** 6)
    sol.solve()
    mf = ref["mf_2048"]
    assert sol.getIterations() == mf["iterations"]
    # ||b||_2 itself: the oracle's serial sum over 3.1 M squares is off by 9e-12 relative (the GPU's double-double value
    # agrees with math.fsum, see test_dot_products_are_more_accurate_than_serial)
    assert sol.last_results.initial_r_norm2 == pytest.approx(mf["initial_r_norm"], rel=1e-10)
    assert abs(sol.last_results.r_norm2 - mf["r_norm"]) / mf["initial_r_norm"] <= REL_TOL
    m = isa.MSGSolver(s, s.get_rhs(), 1e-8, 10 ** 6)
    m.setPrecisionEps(1e-8); m.setResidualEps(1e-8); m.setExactErrorEps(-1.0)
    m.solve(s.get_true_solution_vector())
    mr = ref["msg_2048"]
    assert (m.getIterations(), int(m.getStopReason())) == (mr["iterations"], mr["stop_reason"])
    assert abs(m.getFinalResidualNorm() - mr["final_residual_norm"]) / mf["initial_r_norm"] <= REL_TOL
    assert m.getFinalErrorNorm() == pytest.approx(mr["final_error_norm"], rel=1e-6)


def test_n4096_converged_against_the_oracle(isa):
    """BASELINE config 2 at full size, CONVERGED: the CPU oracle's serial solves of N = 4096 (12.6 M unknowns; 1.5 h per
    solve on one core, tests/golden/oracle_n4096.json made by tests/golden/make_oracle_n4096.py) against the GPU path:
    same iteration count and stop reason, final residual norms within north_star's 1e-12 * ||b||_2, the solution on a
    strided sample.  (matrix_free_system.cpp:409,472; msg_solver.cpp:144-162)"""
    import json, os
    with open(os.path.join(os.path.dirname(__file__), "golden", "oracle_n4096.json")) as f:
        ref = json.load(f)
    N = 4096
    s = isa.GridSystem(N, N, 1.0, 2.0, 1.0, 2.0)
    mf = ref["mf_4096"]
    sol = isa.MatrixFreeSolver(s, s.get_rhs(), 1e-8, 10 ** 6)
    x = sol.solve()
    assert sol.getIterations() == mf["iterations"] and sol.last_results.converged == mf["converged"]
    assert sol.last_results.initial_r_norm2 == pytest.approx(mf["initial_r_norm"], rel=1e-10)
    assert abs(sol.last_results.r_norm2 - mf["r_norm"]) / mf["initial_r_norm"] <= REL_TOL
    xs = np.array([float.fromhex(h) for h in mf["x"]["hex"]])
    assert np.abs(x[::mf["x"]["stride"]] - xs).max() <= 1e-9 * np.abs(xs).max()
    assert abs(np.abs(x).max() - mf["x"]["max_norm"]) <= 1e-9 * mf["x"]["max_norm"]
    if "mfexact_4096" in ref:         # the same solve by the oracle with exact inner products (oracle.exact_dots): bit for bit
        me = ref["mfexact_4096"]
        assert (sol.getIterations(), sol.last_results.converged) == (me["iterations"], me["converged"])
        assert (sol.last_results.r_norm2, sol.last_results.initial_r_norm2) == (me["r_norm"], me["initial_r_norm"])
        assert [float(v).hex() for v in x[::me["x"]["stride"]]] == me["x"]["hex"]
        assert np.abs(x).max() == me["x"]["max_norm"]
    if "mfdiag_4096" in ref:          # the reference's per-iteration report (2-norms, TRUE residual), sampled
        md = ref["mfdiag_4096"]
        got = {}
        sol2 = isa.MatrixFreeSolver(s, s.get_rhs(), 1e-8, 10 ** 6)
        sol2.setIterationCallback(lambda it, p, r, e: got.__setitem__(it, (p, r, e)))
        sol2.solve(s.get_true_solution_vector())
        assert sol2.getIterations() == md["iterations"] and len(got) == md["iterations"]
        for it, p, r, e in md["callbacks"]:
            gp, gr, ge = got[int(it)]
            assert abs(gr - r) / mf["initial_r_norm"] <= 1e-10 and gp == pytest.approx(p, rel=1e-5, abs=1e-9) and ge == pytest.approx(e, rel=1e-5)
    if "msg_4096" in ref:
        mr = ref["msg_4096"]
        m = isa.MSGSolver(s, s.get_rhs(), 1e-8, 10 ** 6)
        m.setPrecisionEps(1e-8); m.setResidualEps(1e-8); m.setExactErrorEps(-1.0)
        cbs = []
        m.setIterationCallback(lambda *a: cbs.append(a))
        xm = m.solve(s.get_true_solution_vector())
        assert (m.getIterations(), int(m.getStopReason()), m.hasConverged()) == (mr["iterations"], mr["stop_reason"], mr["converged"])
        assert abs(m.getFinalResidualNorm() - mr["final_residual_norm"]) / mf["initial_r_norm"] <= REL_TOL
        assert abs(m.last_results.r_norm2 - mr["r_norm2"]) / mf["initial_r_norm"] <= REL_TOL
        assert m.getFinalErrorNorm() == pytest.approx(mr["final_error_norm"], rel=1e-6)
        assert m.getFinalPrecision() == pytest.approx(mr["final_precision"], rel=1e-6)
        assert [c[0] for c in cbs] == [c[0] for c in mr["callbacks"]]
        assert np.abs(np.array([c[2] for c in cbs]) - np.array([c[2] for c in mr["callbacks"]])).max() / mf["initial_r_norm"] <= 1e-10
        xs = np.array([float.fromhex(h) for h in mr["x"]["hex"]])
        assert np.abs(xm[::mr["x"]["stride"]] - xs).max() <= 1e-9 * np.abs(xs).max()
        if "msgexact_4096" in ref:    # MSG rules by the oracle with exact inner products: every callback and the result bit for bit
            mx = ref["msgexact_4096"]
            assert (m.getIterations(), int(m.getStopReason()), m.hasConverged()) == (mx["iterations"], mx["stop_reason"], mx["converged"])
            assert (m.getFinalResidualNorm(), m.getFinalPrecision(), m.getFinalErrorNorm(), m.last_results.r_norm2) == \
                   (mx["final_residual_norm"], mx["final_precision"], mx["final_error_norm"], mx["r_norm2"])
            assert [list(c) for c in cbs] == [list(c) for c in mx["callbacks"]]
            assert [float(v).hex() for v in xm[::mx["x"]["stride"]]] == mx["x"]["hex"]
            assert [float(v).hex() for v in s._handle.recursive_residual()[::mx["r"]["stride"]]] == mx["r"]["hex"]


def test_a_solve_does_not_depend_on_what_the_previous_one_left_behind(isa):
    """The start of a solve rewrites only the owned range of x, r and the first direction (k_init_fresh): everything a launch
    reads beyond that has to be independent of the previous solve -- 7 iterations of one rule, then a full solve of the other,
    against a fresh context."""
    from iterative_solvers_amd import _capi
    n = 258
    fresh = isa.MatrixFreeSystem(n, n, 1.0, 2.0, 1.0, 2.0)
    used = isa.MatrixFreeSystem(n, n, 1.0, 2.0, 1.0, 2.0)
    for first, second in ((_capi.RULE_REL_2NORM, _capi.RULE_MSG_MAXNORM), (_capi.RULE_MSG_MAXNORM, _capi.RULE_REL_2NORM)):
        p = isa.default_params(first)
        p.max_iterations, p.eps_rel, p.eps_precision, p.eps_residual = 7, 1e-30, 1e-30, 1e-30
        used._handle.solve(p)
        q = isa.default_params(second)
        q.max_iterations, q.eps_rel, q.eps_precision, q.eps_residual, q.eps_exact_error = 10 ** 5, 1e-8, 1e-9, 1e-9, -1.0
        a, b = fresh._handle.solve(q), used._handle.solve(q)
        assert (a.iterations, a.stop_reason, a.r_norm2, a.initial_r_norm2, a.final_precision) == (b.iterations, b.stop_reason, b.r_norm2, b.initial_r_norm2, b.final_precision)
        assert np.array_equal(fresh._handle.solution(), used._handle.solution())
        assert np.array_equal(fresh._handle.recursive_residual(), used._handle.recursive_residual())
