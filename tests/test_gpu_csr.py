"""Generic CSR path: Solver(a, b, ...) with a caller-supplied matrix.  The test matrices are
the grid operator's own CSR (oracle-assembled in the reference's entry order), so the CPU oracle remains the checker."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("N", [6, 16, 64, 130])
def test_csr_apply_bit_exact(N):
    import iterative_solvers_amd as isa
    from oracle.oracle import OracleGrid
    og = OracleGrid(N, N)
    A = isa.CrsMatrix(*og.csr())
    assert A.numRows() == og.size
    x = np.random.default_rng(5).uniform(-1, 1, og.size)
    assert np.array_equal(A.apply(x), og.apply(x))              # per-row sums in entry order = the serial CSR loop


@pytest.mark.parametrize("N", [16, 64])
def test_csr_cg_matches_oracle_both_rules(N):
    import iterative_solvers_amd as isa
    from oracle.oracle import OracleGrid
    og = OracleGrid(N, N)
    A = isa.CrsMatrix(*og.csr())
    b, u = og.rhs(), og.true_solution()
    ref = og.mf_solve(eps=1e-8, max_iterations=10 ** 5)
    sol = isa.MatrixFreeSolver(A, b, 1e-8, 10 ** 5)
    x = sol.solve()
    assert sol.getIterations() == ref.iterations
    assert np.abs(x - ref.x).max() <= 1e-9 * np.abs(ref.x).max()
    assert abs(sol.last_results.r_norm2 - ref.r_norm) / ref.initial_r_norm <= 1e-12
    refm = og.msg_solve(eps_precision=1e-9, eps_residual=1e-9, eps_exact_error=-1.0)
    m = isa.MSGSolver(A, b, 1e-9, 10000)
    m.setExactErrorEps(-1.0)
    got = []
    m.setIterationCallback(lambda *a: got.append(a))
    xm = m.solve(u)
    assert (m.getIterations(), int(m.getStopReason()), m.hasConverged()) == (refm.iterations, refm.stop_reason, refm.converged)
    assert [g[0] for g in got] == [c[0] for c in refm.callbacks]
    assert np.abs(xm - refm.x).max() <= 1e-9 * np.abs(refm.x).max()
    assert m.getFinalErrorNorm() == pytest.approx(refm.final_error_norm, rel=1e-9)
    # the stencil path and the CSR path walk the same iterates
    s = isa.GridSystem(N, N, 1.0, 2.0, 1.0, 2.0)
    ms = isa.MSGSolver(s, b, 1e-9, 10000)
    ms.setExactErrorEps(-1.0)
    assert np.array_equal(ms.solve(u), xm)


def test_csr_rejects_malformed_input_and_long_rows_work():
    import iterative_solvers_amd as isa
    with pytest.raises(ValueError):
        isa.CrsMatrix([0, 2, 1], [0, 1], [1.0, 1.0])            # row_map not monotone
    with pytest.raises(ValueError):
        isa.CrsMatrix([0, 1], [3], [1.0])                        # column out of range
    # a dense-ish SPD matrix: rows of 3000 entries cross the kernel's LDS chunk (2048 products)
    n = 3000
    rng = np.random.default_rng(11)
    M = rng.standard_normal((n, 40))
    S = M @ M.T + n * np.eye(n)
    row_map = np.arange(0, n * n + 1, n, dtype=np.int32)
    entries = np.tile(np.arange(n, dtype=np.int32), n)
    A = isa.CrsMatrix(row_map, entries, S.ravel())
    x = rng.standard_normal(n)
    y = A.apply(x)
    assert np.allclose(y, S @ x, rtol=1e-12, atol=1e-9)
    assert np.array_equal(y[:8], [float(np.cumsum(S[i] * x)[-1]) for i in range(8)])
    b = S @ np.ones(n)
    sol = isa.MatrixFreeSolver(A, b, 1e-12, 1000)
    xs = sol.solve()
    assert sol.last_results.converged and np.abs(xs - 1.0).max() < 1e-8
