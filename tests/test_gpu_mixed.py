"""Config 3: fp32-storage inner CG inside fp64 iterative refinement (dtype F32_MIXED).  The reference
has no fp32 path, so there is no oracle twin: the pin is the fp64 TRUE residual of the returned x,
evaluated with the CPU oracle's operator, and closeness to the fp64 solution."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("N,inner_eps", [(64, 0.0), (256, 0.0), (256, 1e-3), (1024, 0.0)])
def test_mixed_reaches_fp64_residual(N, inner_eps):
    import iterative_solvers_amd as isa
    from oracle.oracle import OracleGrid
    og = OracleGrid(N, N)
    b = og.rhs()
    s = isa.MatrixFreeSystem(N, N, 1.0, 2.0, 1.0, 2.0, dtype=isa.F32_MIXED)
    sol = isa.MatrixFreeSolver(s, b, 1e-8, 10 ** 6)
    outer = []
    sol.setIterationCallback(lambda it, p, r, e: outer.append((it, r)))
    x = sol.solve(inner_eps=inner_eps)
    res = sol.last_results
    bn = np.linalg.norm(b)
    true_rel = np.linalg.norm(b - og.apply(x)) / bn
    assert res.converged and true_rel <= 1e-8                      # north_star: residual checked against fp64
    assert res.refine_true_rel == pytest.approx(true_rel, rel=1e-6, abs=1e-13)
    assert res.refine_outer == len(outer) >= 2                      # fp32 alone cannot reach 1e-8
    assert [o[0] for o in outer] == sorted(o[0] for o in outer) and outer[-1][0] == res.iterations
    x64 = isa.MatrixFreeSolver(isa.MatrixFreeSystem(N, N, 1.0, 2.0, 1.0, 2.0), b, 1e-8, 10 ** 6).solve()
    assert np.abs(x - x64).max() <= 1e-6 * np.abs(x64).max()


def test_mixed_rejects_msg_rule_and_reports_stagnation_honestly():
    import iterative_solvers_amd as isa
    s = isa.GridSystem(32, 32, 1.0, 2.0, 1.0, 2.0, dtype=isa.F32_MIXED)
    m = isa.MSGSolver(s, s.get_rhs(), 1e-6, 100)
    with pytest.raises(ValueError):
        m.solve(None)
    # an unreachable target: the solver stops when refinement no longer helps and says "not converged"
    sol = isa.MatrixFreeSolver(s, s.get_rhs(), 1e-30, 10 ** 5)
    sol.solve()
    assert not sol.last_results.converged and sol.last_results.refine_true_rel < 1e-12


def test_config3_8192_mixed_full_solve():
    """BASELINE config 3: N = 8192 (50 315 265 unknowns), fp32 inner CG, fp64 residual <= 1e-8."""
    import iterative_solvers_amd as isa
    from oracle.oracle import OracleGrid
    N = 8192
    s = isa.MatrixFreeSystem(N, N, 1.0, 2.0, 1.0, 2.0, dtype=isa.F32_MIXED)
    assert s.size() == 50315265
    b = s.get_rhs()
    sol = isa.MatrixFreeSolver(s, b, 1e-8, 200000)
    x = sol.solve()
    res = sol.last_results
    assert res.converged and res.refine_true_rel <= 1e-8
    og = OracleGrid(N, N)                                            # fp64 reference operator on the CPU
    assert np.linalg.norm(b - og.apply(x)) / np.linalg.norm(b) <= 1e-8


FP32_SHAPES = [
    {},                                                         # default: 64-row items of 256-column strips, 3 rows in flight
    {"MI355CG_DEPTH": "2"},
    {"MI355CG_ITEM_ROWS": "5"},
    {"MI355CG_ITEM_ROWS": "1", "MI355CG_DEPTH": "2"},
    {"MI355CG_XSTEPS": "2"},
    {"MI355CG_XCD_CLASSES": "0"},
    {"MI355CG_BLOCKS": "40", "MI355CG_ITEM_ROWS": "23"},
]


@pytest.mark.parametrize("N", [514, 2050])
def test_fp32_launch_shapes_take_identical_steps(N, monkeypatch):
    """The fp32 kernels in every launch shape: same iteration counts, same x, same residual, bit for bit (the inner products are
    double-double sums, so the shape must not show).  N = 2050 has XCD-local item ranges and one item per wave: the shape in
    which a 128-bit store followed by a write to its data registers first went wrong (see buf_store in csrc/cg_kernels.h)."""
    import iterative_solvers_amd as isa
    ref = None
    for env in FP32_SHAPES:
        for k in ("MI355CG_DEPTH", "MI355CG_ITEM_ROWS", "MI355CG_XSTEPS", "MI355CG_XCD_CLASSES", "MI355CG_BLOCKS"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        s = isa.MatrixFreeSystem(N, N, 1.0, 2.0, 1.0, 2.0, dtype=isa.F32_MIXED)
        sol = isa.MatrixFreeSolver(s, s.get_rhs(), 1e-8, 10 ** 6)
        x = sol.solve()
        r = sol.last_results
        got = (r.iterations, r.refine_outer, r.refine_true_rel, r.converged)
        s._handle.close()
        if ref is None:
            ref = (got, x)
            assert r.converged
        else:
            assert got == ref[0], (env, got, ref[0])
            assert np.array_equal(x, ref[1]), env


@pytest.mark.parametrize("N", [34, 66, 258, 514])
def test_mixed_solve_equals_the_cpu_statement_of_the_algorithm(N):
    """The fp32 kernels value for value: oracle.mixed_solve is an independent CPU implementation of the SAME algorithm (float
    vectors and float arithmetic in the fp64 path's operation order, exact inner products rounded once, alpha / beta formed in
    double and rounded to float, fp64 refinement; oracle/cg_oracle.c og_mixed_solve -- the reference itself has no fp32 path).
    Same inner iteration total, same number of refinement steps, and the same x to the last bit."""
    import iterative_solvers_amd as isa
    from oracle.oracle import OracleGrid
    og = OracleGrid(N, N)
    b = og.rhs()
    s = isa.MatrixFreeSystem(N, N, 1.0, 2.0, 1.0, 2.0, dtype=isa.F32_MIXED)
    sol = isa.MatrixFreeSolver(s, b, 1e-8, 10 ** 6)
    x = sol.solve()
    r = sol.last_results
    xo, its, outer, conv, rel = og.mixed_solve(b, eps=1e-8)
    assert (r.iterations, r.refine_outer, bool(r.converged)) == (its, outer, conv)
    assert np.array_equal(x, xo)
    assert r.refine_true_rel == pytest.approx(rel, rel=1e-10)


def test_mixed_solve_n2050_against_the_cpu_fixture():
    """The same comparison at N = 2050 (XCD-local item ranges in the fp32 launch plan; 7 292 inner iterations), against the
    15-minute CPU run kept in tests/golden/oracle_mixed_n2050.json (tests/golden/make_oracle_mixed.py)."""
    import json
    import os
    import iterative_solvers_amd as isa
    path = os.path.join(os.path.dirname(__file__), "golden", "oracle_mixed_n2050.json")
    if not os.path.exists(path):
        pytest.skip("fixture not generated")
    ref = json.load(open(path))
    N = ref["n"]
    s = isa.MatrixFreeSystem(N, N, 1.0, 2.0, 1.0, 2.0, dtype=isa.F32_MIXED)
    sol = isa.MatrixFreeSolver(s, s.get_rhs(), 1e-8, 10 ** 6)
    x = sol.solve()
    r = sol.last_results
    assert (r.iterations, r.refine_outer, bool(r.converged)) == (ref["iterations"], ref["outer"], ref["converged"])
    assert [float(v).hex() for v in x[::ref["x"]["stride"]]] == ref["x"]["hex"]
    assert r.refine_true_rel == pytest.approx(ref["true_rel"], rel=1e-9)
