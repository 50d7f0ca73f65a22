"""The HIP path (through the C ABI) against the committed oracle fixtures (tests/golden/oracle_small.json):
data, not a live oracle build.  Bars as in test_gpu_parity.py: bit-exact for setup vectors and the operator, identical iteration
counts / stop reasons / callback iterations, norms within 1e-12 * ||b||_2."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
REL_TOL = 1e-12


@pytest.fixture(scope="module")
def fx():
    with open(os.path.join(HERE, "golden", "oracle_small.json")) as f:
        return json.load(f)


@pytest.fixture(scope="module")
def isa():
    import torch
    assert torch.cuda.is_available(), "these tests need the MI355X"
    import iterative_solvers_amd as isa
    isa.load()
    return isa


def unhex(v):
    return np.array([float.fromhex(s) for s in v])


def seeded(n):
    return np.random.Generator(np.random.PCG64(12345)).uniform(-1.0, 1.0, n)


@pytest.mark.parametrize("N", [6, 8, 16, 64])
def test_f2_setup_vectors_bit_exact(isa, fx, N):
    s, rec = isa.GridSystem(N, N, 1.0, 2.0, 1.0, 2.0), fx["F2"][str(N)]
    assert s.size() == rec["size"]
    assert np.array_equal(np.asarray(s.get_rhs()), unhex(rec["rhs"]))
    assert np.array_equal(np.asarray(s.get_true_solution_vector()), unhex(rec["u_true"]))


@pytest.mark.parametrize("N", [6, 8, 16, 64, 256])
def test_f3_apply_bit_exact(isa, fx, N):
    s, rec = isa.MatrixFreeSystem(N, N, 1.0, 2.0, 1.0, 2.0), fx["F3"][str(N)]
    y = s.apply(seeded(s.size()))
    if "y" in rec:
        assert np.array_equal(y, unhex(rec["y"]))
    else:
        assert np.array_equal(y[::97], unhex(rec["y_stride_97"]))
    assert float(np.sum(np.abs(y))) == pytest.approx(float.fromhex(rec["sum_abs_y"]), rel=1e-13)


@pytest.mark.parametrize("N", [6, 16, 64, 256])
def test_f4_cg_against_frozen_traces(isa, fx, N):
    rec = fx["F4"][str(N)]
    s = isa.MatrixFreeSystem(N, N, 1.0, 2.0, 1.0, 2.0)
    b = np.asarray(s.get_rhs())
    bnorm = float(np.linalg.norm(b))
    sol = isa.MatrixFreeSolver(s, b, 1e-8, 10 ** 5)
    x = np.asarray(sol.solve())
    assert sol.getIterations() == rec["iterations"]
    want_x = unhex(rec["x"]) if rec["x"] is not None else None
    if want_x is not None:
        assert np.abs(x - want_x).max() <= 1e-9 * np.abs(want_x).max()
    else:
        w = unhex(rec["x_stride_97"])
        assert np.abs(x[::97] - w).max() <= 1e-9 * np.abs(w).max()
    g = isa.GridSystem(N, N, 1.0, 2.0, 1.0, 2.0)
    cbs = []
    m = isa.MSGSolver(g, g.get_rhs(), 1e-8, 10 ** 5)
    m.setPrecisionEps(1e-8); m.setResidualEps(1e-8); m.setExactErrorEps(-1.0)
    m.setIterationCallback(lambda it, p, r, e: cbs.append((it, p, r, e)))
    m.solve(g.get_true_solution_vector())
    w = rec["msg"]
    assert (m.getIterations(), int(m.getStopReason()), m.hasConverged()) == (w["iterations"], w["stop_reason"], w["converged"])
    assert abs(m.getFinalResidualNorm() - float.fromhex(w["final_residual_norm"])) <= REL_TOL * bnorm
    assert [c[0] for c in cbs] == [c[0] for c in w["callbacks"]]
    for got, want in zip(cbs, w["callbacks"]):
        for a, bb in zip(got[1:3], want[1:3]):
            bb = float.fromhex(bb)
            assert a == bb or abs(a - bb) <= REL_TOL * bnorm


@pytest.mark.parametrize("N", [512, 1024])
def test_f5_iteration_counts(isa, fx, N):
    rec = fx["F5"][str(N)]
    s = isa.MatrixFreeSystem(N, N, 1.0, 2.0, 1.0, 2.0)
    b = np.asarray(s.get_rhs())
    sol = isa.MatrixFreeSolver(s, b, 1e-8, 10 ** 5)
    sol.solve()
    assert sol.getIterations() == rec["iterations"]
    assert abs(sol.last_results.r_norm2 - float.fromhex(rec["r_norm"])) <= REL_TOL * float(np.linalg.norm(b))
