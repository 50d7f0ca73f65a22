"""The iteration has several launch shapes selected by environment knobs read at context creation
(DESIGN.md section 4).  Every shape must take bit-identical CG steps: same iteration count, same recursive residual norm,
same x as the default path -- and the default path is pinned to the oracle by test_gpu_parity.py."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

VARIANTS = [
    {},                                                     # default: A p recomputed, x every second iteration (7.5 words)
    {"MI355CG_X2STEP": "0"},                                # x update folded into every stencil launch (8 words)
    {"MI355CG_RECOMPUTE": "0"},                             # A p stored and streamed back (9 words)
    {"MI355CG_RECOMPUTE": "0", "MI355CG_XFUSE": "0"},       # separate 4 + 6 word launches (10 words)
    {"MI355CG_SDEPTH": "4", "MI355CG_UDEPTH": "4", "MI355CG_UDEPTH_FULL": "4", "MI355CG_MSG_DEPTH": "4"},
    {"MI355CG_UPDATE_DESC": "0"},
    {"MI355CG_ROWS": "7"},                                  # odd item height, several rounds per wave
    {"MI355CG_STENCIL_BLOCKS": "37"},
]


def _solve(env, n, rule, max_it):
    import iterative_solvers_amd as isa
    from iterative_solvers_amd import _capi
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        s = isa.MatrixFreeSystem(n, n, 1.0, 2.0, 1.0, 2.0)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    p = isa.default_params(rule)
    p.max_iterations = max_it
    if rule == _capi.RULE_REL_2NORM:
        p.eps_rel = 1e-10
    else:
        p.eps_precision = p.eps_residual = 1e-9
    res = s._handle.solve(p)
    x = s._handle.solution()
    r = s._handle.recursive_residual()
    s._handle.close()
    return res, x, r


@pytest.mark.parametrize("n", [6, 64, 130, 1026])
@pytest.mark.parametrize("rule_name", ["rel2", "msg"])
def test_launch_shapes_take_identical_steps(n, rule_name):
    from iterative_solvers_amd import _capi
    rule = _capi.RULE_REL_2NORM if rule_name == "rel2" else _capi.RULE_MSG_MAXNORM
    ref = None
    for env in VARIANTS:
        res, x, r = _solve(env, n, rule, 6000)
        if ref is None:
            ref = (res, x, r)
            assert res.converged and res.iterations > 0, (res.iterations, res.converged, res.stop_reason, res.r_norm2, res.initial_r_norm2)
            continue
        assert res.iterations == ref[0].iterations, env
        assert res.r_norm2 == ref[0].r_norm2, env
        assert np.array_equal(r, ref[2]), env
        assert np.array_equal(x, ref[1]), env


@pytest.mark.parametrize("stop_at", [1, 2, 7, 8])
def test_two_step_x_update_is_flushed_for_odd_and_even_counts(stop_at):
    """x after exactly k iterations (k odd: one update still pending when the loop ends; k even: none)."""
    from iterative_solvers_amd import _capi
    from oracle.oracle import OracleGrid
    n = 64
    res, x, r = _solve({}, n, _capi.RULE_REL_2NORM, stop_at)
    assert res.iterations == stop_at
    og = OracleGrid(n, n)
    o = og.mf_solve(eps=1e-10, max_iterations=stop_at)
    assert o.iterations == stop_at
    np.testing.assert_allclose(x, o.x, rtol=1e-12, atol=1e-14)
