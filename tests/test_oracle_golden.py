"""Pin the CPU oracle (oracle/cg_oracle.c) against every known-answer artefact the reference
holds for the hot path: check.py's operator, check_debug.py's RHS and py_debug.txt's
two-iteration CG trace (tests/golden/n6_pydebug.json, minted by tests/golden/make_golden.py),
plus the counts SURVEY.md (section 6 / Appendix B) recorded from the reference's own
MatrixFreeSolver."""
import numpy as np
import pytest

from oracle.oracle import OracleGrid, STOP_PRECISION, STOP_RESIDUAL, STOP_ITERATIONS, dot, max_norm


def test_sizes_match_closed_form():
    # U = (N/2-1)(3N/2-1), SURVEY 8.0 / Appendix B
    for N, U in ((6, 16), (8, 33), (16, 161), (256, 48641)):
        assert OracleGrid(N, N).size == U == (N // 2 - 1) * (3 * N // 2 - 1)


def test_operator_equals_check_py_matrix(golden_n6):
    g = OracleGrid(6, 6, 1, 2, 1, 2)
    assert np.array_equal(g.dense(), np.array(golden_n6["A"]))       # bit-exact: -144 / 36 / 0


def test_csr_assembly_matches_operator(golden_n6):
    g = OracleGrid(6, 6)
    row_map, entries, values = g.csr()
    A = np.zeros((g.size, g.size))
    for i in range(g.size):
        cols = entries[row_map[i]:row_map[i + 1]]
        assert cols[0] == i                                           # diagonal first
        A[i, cols] = values[row_map[i]:row_map[i + 1]]
    assert np.array_equal(A, np.array(golden_n6["A"]))


def test_rhs_matches_check_debug_to_8_decimals(golden_n6):
    g = OracleGrid(6, 6)
    b = np.array(golden_n6["b_check_debug"])
    assert np.abs(g.rhs() - b).max() <= 0.5e-8 + 1e-12                # file prints 8 decimals
    # entry 9 is the only one check.py's older RHS shares
    assert abs(golden_n6["b_check_old"][9] - g.rhs()[9]) <= 0.5e-6


def test_two_iteration_trace_msg(golden_n6):
    """MSGSolver restatement on the golden 8-decimal b reproduces py_debug.txt's x1, r1, x2, r2."""
    g = OracleGrid(6, 6)
    b = np.array(golden_n6["b_check_debug"])
    t = golden_n6["trace"]
    assert np.allclose(g.apply(-b), t["A_at_h0"], rtol=1e-13, atol=0)          # A @ h0, h0 = -b
    r1 = g.msg_solve(b=b, true_solution=None, eps_precision=-1, eps_residual=-1, max_iterations=1)
    assert np.allclose(r1.x, t["x1"], rtol=1e-12, atol=1e-15)
    assert np.allclose(-r1.r, t["r1"], rtol=1e-10, atol=1e-11)                 # r_py = A x - b
    r2 = g.msg_solve(b=b, true_solution=None, eps_precision=-1, eps_residual=-1, max_iterations=2)
    assert np.allclose(r2.x, t["x2"], rtol=1e-11, atol=1e-14)
    assert np.allclose(-r2.r, t["r2"], rtol=1e-9, atol=1e-10)
    assert r2.iterations == 2 and not r2.converged and r2.stop_reason == STOP_ITERATIONS
    # scalar step sizes: alpha0 = -(b.b)/(b.Ab) in the script's sign convention
    alpha0 = -dot(b, b) / dot(b, g.apply(b))
    assert alpha0 == pytest.approx(t["alpha0"], rel=1e-13)
    h1 = np.array(t["h1"])
    assert np.allclose(g.apply(h1), t["A_at_h1"], rtol=1e-12, atol=1e-9)


def test_two_iteration_trace_matrix_free_solver(golden_n6):
    g = OracleGrid(6, 6)
    b = np.array(golden_n6["b_check_debug"])
    m = g.mf_solve(b=b, eps=0.0, max_iterations=2)
    assert m.iterations == 2
    assert np.allclose(m.x, golden_n6["trace"]["x2"], rtol=1e-11, atol=1e-14)


def test_survey_recorded_reference_run_n256():
    """the reference's MatrixFreeSolver at N=256, eps 1e-8 -> 701 iterations,
    final ||b-Ax||_2 = 1.2256e-01, ||b||_2 = 1.246167e7, ||b||_inf = 2.601608e6."""
    g = OracleGrid(256, 256)
    b = g.rhs()
    assert np.sqrt(dot(b, b)) == pytest.approx(1.246167e7, rel=1e-6)
    assert max_norm(b) == pytest.approx(2.601608e6, rel=1e-6)
    m = g.mf_solve(eps=1e-8, max_iterations=10 ** 6)
    assert m.iterations == 701 and m.converged
    assert np.linalg.norm(b - g.apply(m.x)) == pytest.approx(1.225611e-01, rel=1e-5)


def test_mf_diagnostics_do_not_change_the_iterates():
    g = OracleGrid(16, 16)
    a = g.mf_solve(eps=1e-10, diagnostics=False)
    d = g.mf_solve(eps=1e-10, diagnostics=True)
    assert a.iterations == d.iterations and np.array_equal(a.x, d.x)
    assert len(d.callbacks) == d.iterations and d.callbacks[0][0] == 0
    # callback residual is the TRUE residual b - A x (matrix_free_system.cpp:457-463)
    assert d.callbacks[-1][2] == pytest.approx(np.linalg.norm(g.rhs() - g.apply(d.x)), rel=1e-12)


def test_msg_stop_rules_n256():
    """Indicative MSG behaviour listed in SURVEY Appendix A / BASELINE.md section 3."""
    g = OracleGrid(256, 256)
    r = g.msg_solve()                                                  # facade defaults 1e-6, error off
    assert (r.iterations, r.stop_reason, r.converged) == (631, STOP_PRECISION, True)
    assert [c[0] for c in r.callbacks] == [0, 1, 100, 200, 300, 400, 500, 600, 631]
    r = g.msg_solve(eps_precision=-1, eps_residual=1e-8)
    assert (r.iterations, r.stop_reason) == (1004, STOP_RESIDUAL)
    assert r.final_residual_norm < 1e-8


def test_invalid_positions_return_minus_one():
    g = OracleGrid(8, 8)
    assert g.position(1, 1) == -1 and g.position(0, 5) == -1 and g.position(8, 5) == -1
    assert g.position(5, 1) == 0 and g.position(1, 5) == (8 // 2 - 1) * (8 // 2)
    assert g.is_boundary(4, 2) and g.is_boundary(2, 4) and not g.is_boundary(5, 4)


def test_exact_dots_mode_of_the_oracle_is_what_it_says():
    """oracle.exact_dots() (Dot2: inner products as if in twice the working precision) is the tool the GPU parity tests use to
    separate the reference's summation ORDER from the rest of its arithmetic.  Here: it returns the exactly rounded value where
    the serial sum does not, it changes nothing else, and it switches off again."""
    import math
    import numpy as np
    from oracle import oracle as o
    g = o.OracleGrid(130, 130)
    b = g.rhs()
    serial = o.dot(b, b)
    with o.exact_dots():
        exact = o.dot(b, b)
        ex = g.mf_solve(eps=1e-8, max_iterations=10 ** 5)
    from fractions import Fraction
    true = sum(Fraction(float(v)) ** 2 for v in b)                       # exact rational arithmetic
    assert exact == float(true) and serial != exact and abs(serial - exact) <= 1e-12 * exact
    assert o.dot(b, b) == serial                                         # switched off again
    ref = g.mf_solve(eps=1e-8, max_iterations=10 ** 5)
    assert abs(ex.iterations - ref.iterations) <= 1 and np.abs(ex.x - ref.x).max() <= 1e-9 * np.abs(ref.x).max()
    assert not np.array_equal(ex.x, ref.x)                               # the summation order does show in the last bits


def test_mixed_precision_statement_of_the_oracle_converges_in_fp64_terms():
    """oracle.mixed_solve (CPU statement of the library's config-3 algorithm; not the reference, which has no fp32 path): its
    answer has to satisfy the fp64 system to 1e-8 and sit next to the fp64 CG solution."""
    import numpy as np
    from oracle import oracle as o
    g = o.OracleGrid(130, 130)
    b = g.rhs()
    x, its, outer, conv, rel = g.mixed_solve(b, eps=1e-8)
    assert conv and outer >= 2 and its > 100
    true_rel = np.linalg.norm(b - g.apply(x)) / np.linalg.norm(b)
    assert true_rel <= 1e-8 and abs(true_rel - rel) <= 1e-12
    ref = g.mf_solve(eps=1e-10, max_iterations=10 ** 5)
    assert np.abs(x - ref.x).max() <= 1e-6 * np.abs(ref.x).max()


def test_all_cores_timing_baseline_takes_the_same_steps_and_knows_its_cpu_share():
    """bench.py's "all host cores" figure: the OpenMP build of the same loop.  Its inner products are OpenMP reductions, so it is a
    timing baseline and never a checker -- but it has to run the same iteration: same count to a loose tolerance, x to rounding."""
    import os
    from oracle.oracle import OracleGrid, host_cpu_share, mf_solve_all_cores
    share = host_cpu_share()
    visible = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    assert 1 <= share <= visible
    og = OracleGrid(64, 64)
    b = og.rhs()
    ref = og.mf_solve(eps=1e-8, max_iterations=10 ** 4, diagnostics=False)
    for threads in (1, 2, min(4, share)):
        its, x, used = mf_solve_all_cores(64, b, 1e-8, 10 ** 4, threads)
        assert used == threads and abs(its - ref.iterations) <= 1
        assert np.abs(x - ref.x).max() <= 1e-9 * np.abs(ref.x).max()
