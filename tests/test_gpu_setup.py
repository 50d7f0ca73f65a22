"""opt-in device-side problem setup (mi355cg_setup_on_device / MI355CG_DEVICE_SETUP=1).  The default host
path is bit-identical to the oracle (tests/test_gpu_parity.py::test_setup_vectors_bit_exact); the device path evaluates the
same expressions with the device library's exp(), so it is held to "<= 1 ulp per exp" instead."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("N", [6, 64, 258, 514])
def test_device_generated_rhs_and_exact_solution_are_within_an_ulp_of_exp(N):
    import iterative_solvers_amd as isa
    from oracle.oracle import OracleGrid
    og = OracleGrid(N, N)
    s = isa.GridSystem(N, N, 1.0, 2.0, 1.0, 2.0)
    assert np.array_equal(s.get_rhs(), og.rhs())                              # default: host libm, bit-exact
    s._handle.setup_on_device()
    u, b = s.get_true_solution_vector(), s.get_rhs()
    assert np.all(np.abs(u - og.true_solution()) <= np.spacing(og.true_solution()))          # exp within 1 ulp
    # b = f - x_k u(..) - y_k u(..): every term carries a 1-ulp exp; bound by the size of the terms, not of the (cancelled) sum
    A, xk, yk = og.coeffs
    scale = np.abs(og.rhs()) + 2 * (xk + yk) * np.abs(og.true_solution()).max()
    assert np.all(np.abs(b - og.rhs()) <= 4 * np.finfo(float).eps * scale)
    x_dev = isa.MatrixFreeSolver(s, b, 1e-8, 10 ** 6)
    xd = x_dev.solve()
    ref = og.mf_solve(eps=1e-8, max_iterations=10 ** 6)
    assert abs(x_dev.getIterations() - ref.iterations) <= 1
    assert np.abs(xd - ref.x).max() <= 1e-8 * np.abs(ref.x).max()


def test_device_setup_at_create_and_on_a_team(monkeypatch):
    import iterative_solvers_amd as isa
    from iterative_solvers_amd.distributed import Team
    N = 258
    host = isa.GridSystem(N, N, 1.0, 2.0, 1.0, 2.0)
    monkeypatch.setenv("MI355CG_DEVICE_SETUP", "1")
    dev = isa.GridSystem(N, N, 1.0, 2.0, 1.0, 2.0)
    t = Team.local(N, 4, 1)
    monkeypatch.delenv("MI355CG_DEVICE_SETUP")
    b_dev = dev.get_rhs()
    assert np.abs(b_dev - host.get_rhs()).max() <= 1e-15 * np.abs(host.get_rhs()).max() * 64
    assert np.array_equal(t.vector(2), b_dev) and np.array_equal(t.vector(3), dev.get_true_solution_vector())   # parts generate the same bits as the whole grid
    p = isa.default_params(isa.RULE_REL_2NORM)
    p.eps_rel, p.max_iterations = 1e-8, 10 ** 5
    r1, rt = dev._handle.solve(p), t.solve(p)
    assert (rt.iterations, rt.r_norm2) == (r1.iterations, r1.r_norm2)
    assert np.array_equal(t.vector(0), dev._handle.solution())
    t.close()
