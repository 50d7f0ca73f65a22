"""The oracle against its own frozen outputs (tests/golden/oracle_small.json, SURVEY 8c fixtures F2-F4): an accidental change
of oracle/cg_oracle.c shows up here as a diff against committed numbers.  (The oracle is pinned to the REFERENCE's artefacts in
test_oracle_golden.py; this file pins it to itself.)"""
import json
import os

import numpy as np
import pytest

from oracle.oracle import OracleGrid

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def fx():
    with open(os.path.join(HERE, "golden", "oracle_small.json")) as f:
        return json.load(f)


def unhex(v):
    return np.array([float.fromhex(s) for s in v])


def seeded(n):
    return np.random.Generator(np.random.PCG64(12345)).uniform(-1.0, 1.0, n)


@pytest.mark.parametrize("N", [6, 8, 16, 64])
def test_f2_setup_vectors(fx, N):
    g, rec = OracleGrid(N, N), fx["F2"][str(N)]
    assert g.size == rec["size"]
    assert np.array_equal(g.rhs(), unhex(rec["rhs"]))
    assert np.array_equal(g.true_solution(), unhex(rec["u_true"]))


@pytest.mark.parametrize("N", [6, 8, 16, 64, 256])
def test_f3_apply(fx, N):
    g, rec = OracleGrid(N, N), fx["F3"][str(N)]
    y = g.apply(seeded(g.size))
    if "y" in rec:
        assert np.array_equal(y, unhex(rec["y"]))
    else:
        assert np.array_equal(y[::97], unhex(rec["y_stride_97"]))
    assert float(np.sum(np.abs(y))) == pytest.approx(float.fromhex(rec["sum_abs_y"]), rel=1e-13)


@pytest.mark.parametrize("N", [6, 16, 64])
def test_f4_cg_traces(fx, N):
    g, rec = OracleGrid(N, N), fx["F4"][str(N)]
    m = g.mf_solve(eps=1e-8, max_iterations=10 ** 5, diagnostics=True)
    assert (m.iterations, bool(m.converged)) == (rec["iterations"], rec["converged"])
    assert m.r_norm == float.fromhex(rec["r_norm"])
    assert np.array_equal(m.x, unhex(rec["x"]))
    every = rec["callbacks_every"]
    for got, want in zip(m.callbacks[::every], rec["callbacks"]):
        assert got[0] == want[0] and tuple(got[1:]) == tuple(float.fromhex(s) for s in want[1:])
    r = g.msg_solve(eps_precision=1e-8, eps_residual=1e-8, eps_exact_error=-1.0, max_iterations=10 ** 5)
    w = rec["msg"]
    assert (r.iterations, r.stop_reason, bool(r.converged)) == (w["iterations"], w["stop_reason"], w["converged"])
    assert r.final_residual_norm == float.fromhex(w["final_residual_norm"])
    assert [c[0] for c in r.callbacks] == [c[0] for c in w["callbacks"]]
