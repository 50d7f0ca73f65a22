"""The launch geometry (item height, waves, rows in flight) is selected by environment knobs read at context creation
(DESIGN.md section 4).  Every geometry must take bit-identical CG steps: same iteration count, same recursive residual norm,
same x as the default -- and the default is pinned to the oracle by test_gpu_parity.py."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

VARIANTS = [
    {},                                                     # default: one round of tall items over 2 048 waves, 3 rows in flight
    {"MI355CG_ITEM_ROWS": "7"},                             # odd item height, many rounds per wave
    {"MI355CG_ITEM_ROWS": "1"},                             # every item fetches 3 rows and computes 1: with 3 rows in flight the fetch cursor is a whole item ahead
    {"MI355CG_ITEM_ROWS": "1000000"},                       # one tall item per wave (round 1's shape)
    {"MI355CG_DEPTH": "2"},                                 # 2 rows in flight (default 3)
    {"MI355CG_DEPTH": "2", "MI355CG_ITEM_ROWS": "1"},
    {"MI355CG_BLOCKS": "37"},
    {"MI355CG_WAVES": "256", "MI355CG_ITEM_ROWS": "5"},
    {"MI355CG_XSTEPS": "8"},                                # x folded every 8th iteration (12-word launch, 247 VGPRs)
    {"MI355CG_XSTEPS": "2"},                                # x folded every 2nd iteration instead of every 4th (round 1's scheme)
    {"MI355CG_XCD_CLASSES": "0"},                           # items dealt to all workgroups alike (no per-XCD ranges)
    {"MI355CG_BLOCKS": "100", "MI355CG_ITEM_ROWS": "9"},    # grid not a multiple of 8 XCD classes -> rounded down to 96
    {"MI355CG_DYN_ROWS": "5"},                              # run-time item queues (experimental): every wave takes its next item from its group's counter
    {"MI355CG_DYN_ROWS": "3", "MI355CG_BLOCKS": "24"},      # ... with fewer workgroups per XCD class than sub-queues
    {"MI355CG_DYN_ROWS": "4", "MI355CG_DEPTH": "2"},
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


@pytest.mark.parametrize("stop_at", [1, 2, 3, 4, 5, 7, 8, 10])
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


def test_random_launch_shapes_take_identical_steps():
    """tools/shape_fuzz.py: 50 random combinations of the launch-geometry knobs, grid sizes, both precisions and both rules, a few
    iterations each, against the default shape of the same problem -- x and r bit for bit."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import shape_fuzz
    saved = {k: os.environ.get(k) for k in shape_fuzz.KNOBS}
    try:
        assert shape_fuzz.fuzz(50, 20261004, verbose=False) == []
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def test_headline_throughput_has_not_regressed():
    """Config 2 (N = 4096 fp64, REL_2NORM): 1 000 fixed iterations.  The slowest box seen this round ran 7 196 it/s, the build of
    round 1 6 700; the floor only catches a real regression (a lost overlap, a spilled kernel), not box-to-box spread."""
    import time
    import iterative_solvers_amd as isa
    from iterative_solvers_amd import _capi
    s = isa.MatrixFreeSystem(4096, 4096, 1.0, 2.0, 1.0, 2.0)
    p = isa.default_params(_capi.RULE_REL_2NORM)
    p.max_iterations, p.fixed_iterations, p.use_true_solution, p.callback_every, p.sync_every = 1000, 1, 0, 0, 500
    s._handle.solve(p)
    t0 = time.perf_counter()
    r = s._handle.solve(p)
    dt = time.perf_counter() - t0
    assert r.iterations == 1000
    assert 1000 / dt >= 6000, f"{1000 / dt:.0f} it/s"
    assert 1000 / r.loop_seconds >= 6000
