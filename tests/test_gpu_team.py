"""GPU tests of the native multi-GPU loop (csrc/team.h, mi355cg_team_*): a LOCAL team drives all parts of a decomposition
in one process on the one GPU of the test box -- the same kernels, records, halo messages, streams and events as the
one-process-per-GPU RCCL team, with device-to-device copies in place of ncclSend/ncclRecv.  Every decomposition must
reproduce the single-GPU solve BIT FOR BIT (the inner products travel as double-double pairs and are reduced in part
order), and that solve is pinned to the CPU oracle by test_gpu_parity.py.

BASELINE configs 4 (N = 16384, 2 x 2) and 5 (N = 32768, 8 parts) run here at full size for a bounded number of
iterations: vectors of that size stay on the device, so the comparison uses device-side double-double checksums, the
recursive-vs-true residual property, and the exact equality of the residual norms."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _params(isa, rule, **kw):
    p = isa.default_params(rule)
    for k, v in kw.items():
        setattr(p, k, v)
    return p


def _single(isa, n, rule, **kw):
    s = isa.MatrixFreeSystem(n, n, 1.0, 2.0, 1.0, 2.0)
    cbs = []
    res = s._handle.solve(_params(isa, rule, **kw), callback=(lambda *a: cbs.append(a)) if rule == 0 else None)
    return s, res, cbs


@pytest.mark.parametrize("n,world,decomp", [(64, 2, 0), (64, 3, 0), (130, 4, 0), (258, 4, 1), (258, 2, 1), (258, 8, 1), (514, 8, 1),
                                            (1026, 4, 1), (1026, 8, 0), (66, 16, 0)])
def test_local_team_rel2_is_bit_identical_to_one_gpu(n, world, decomp):
    import iterative_solvers_amd as isa
    from iterative_solvers_amd.distributed import Team
    from oracle.oracle import OracleGrid
    kw = dict(eps_rel=1e-8, max_iterations=10 ** 5)
    s1, r1, _ = _single(isa, n, 1, **kw)
    t = Team.local(n, world, decomp)
    rt = t.solve(_params(isa, 1, **kw))
    assert (rt.iterations, rt.converged, rt.stop_reason) == (r1.iterations, r1.converged, r1.stop_reason)
    assert rt.r_norm2 == r1.r_norm2 and rt.initial_r_norm2 == r1.initial_r_norm2
    x = t.vector(0)
    assert not np.isnan(x).any()                                   # every unknown is owned by exactly one part ...
    assert np.array_equal(x, s1._handle.solution())                # ... and holds the single-GPU bits
    assert np.array_equal(t.vector(1), s1._handle.recursive_residual())
    assert np.array_equal(t.vector(2), s1.get_rhs()) and np.array_equal(t.vector(3), s1.get_true_solution_vector())
    assert t.checksum(0) == tuple(_cs(s1, 0)) and t.checksum(1) == tuple(_cs(s1, 1))
    if n <= 514:                                                   # (the serial oracle needs 25 s at N = 1026; the single-GPU solve it
        og = OracleGrid(n, n)                                      #  is compared with there is pinned by test_gpu_parity.py)
        ref = og.mf_solve(eps=1e-8, max_iterations=10 ** 5)
        assert rt.iterations == ref.iterations
        assert abs(rt.r_norm2 - ref.r_norm) / ref.initial_r_norm <= 1e-12
    t.close()


def _cs(system, which):
    from iterative_solvers_amd import _capi
    o = (C.c_double * 2)()
    _capi.check(_capi.load().mi355cg_checksum(system._handle._h, which, o))
    return o[0], o[1]


@pytest.mark.parametrize("n,world,decomp", [(64, 2, 0), (258, 4, 1), (130, 3, 0)])
def test_local_team_msg_rule_callbacks_and_stop_reason(n, world, decomp):
    import iterative_solvers_amd as isa
    from iterative_solvers_amd.distributed import Team
    from oracle.oracle import OracleGrid
    kw = dict(eps_precision=1e-9, eps_residual=1e-9, eps_exact_error=-1.0, max_iterations=10 ** 5)
    s1, r1, cb1 = _single(isa, n, 0, **kw)
    t = Team.local(n, world, decomp)
    # a solve cut short first: the next one starts from vectors, ghost rows and ghost columns that hold an unrelated state
    t.solve(_params(isa, 1, eps_rel=1e-30, max_iterations=7))
    cbs = []
    rt = t.solve(_params(isa, 0, **kw), callback=lambda *a: cbs.append(a))
    assert (rt.iterations, rt.converged, rt.stop_reason) == (r1.iterations, r1.converged, r1.stop_reason)
    assert cbs == cb1                                              # iteration numbers and all three norms, exactly
    assert (rt.final_residual_norm, rt.final_precision, rt.final_error_norm) == (r1.final_residual_norm, r1.final_precision, r1.final_error_norm)
    assert np.array_equal(t.vector(0), s1._handle.solution())
    ref = OracleGrid(n, n).msg_solve(eps_precision=1e-9, eps_residual=1e-9, eps_exact_error=-1.0, max_iterations=10 ** 5)
    assert (rt.iterations, rt.stop_reason) == (ref.iterations, ref.stop_reason)
    assert [c[0] for c in cbs] == [c[0] for c in ref.callbacks]
    t.close()


@pytest.mark.parametrize("n,world,decomp,mail", [(258, 4, 1, False), (130, 3, 0, False), (66, 16, 0, False), (258, 4, 1, True)])
def test_local_team_with_one_thread_per_part(n, world, decomp, mail, monkeypatch):
    """The multi-GPU form of the LOCAL transport (a host thread per part, two barriers per iteration), forced on for parts that
    share the test box's one GPU: same bits as the one-thread loop and as the single context, both rules, 16 threads on few cores.
    mail: asking for mailboxes + pushed halo as well -- stream-level value waits between parts that share a device, which the threads
    must not enqueue in their own order (hardware queues are shared): the team falls back to the one-thread loop.  Same bits."""
    import iterative_solvers_amd as isa
    from iterative_solvers_amd.distributed import Team
    monkeypatch.setenv("MI355CG_TEAM_THREADS", "1")
    if mail:
        monkeypatch.setenv("MI355CG_TEAM_RECORDS", "mailbox")
        monkeypatch.setenv("MI355CG_TEAM_HALO", "push")
    kw = dict(eps_rel=1e-8, max_iterations=10 ** 5)
    s1, r1, _ = _single(isa, n, 1, **kw)
    t = Team.local(n, world, decomp)
    rt = t.solve(_params(isa, 1, **kw))
    assert (rt.iterations, rt.r_norm2, rt.initial_r_norm2) == (r1.iterations, r1.r_norm2, r1.initial_r_norm2)
    assert np.array_equal(t.vector(0), s1._handle.solution())
    kw = dict(eps_precision=1e-9, eps_residual=1e-9, eps_exact_error=-1.0, max_iterations=10 ** 5)
    s0, r0, cb0 = _single(isa, n, 0, **kw)
    cbs = []
    rm = t.solve(_params(isa, 0, **kw), callback=lambda *a: cbs.append(a))
    assert (rm.iterations, rm.stop_reason, rm.final_residual_norm, rm.final_precision) == (r0.iterations, r0.stop_reason, r0.final_residual_norm, r0.final_precision)
    assert cbs == cb0 and np.array_equal(t.vector(0), s0._handle.solution())
    stop = C.c_int(0)
    seen = []
    rs = t.solve(_params(isa, 0, **kw), callback=lambda it, *a: (seen.append(it), stop.__setattr__("value", 1 if it >= 1 else 0)), stop_flag=stop)
    assert rs.stop_reason == 4 and rs.iterations == 1                   # INTERRUPTED right after the it = 1 callback
    t.close()


@pytest.mark.parametrize("n,world,decomp,split", [(258, 4, 1, "1"), (130, 3, 0, "1"), (258, 4, 1, "2"), (514, 5, 0, "2")])
def test_team_with_interior_and_edge_launches(n, world, decomp, split, monkeypatch):
    """MI355CG_TEAM_SPLIT=1: the phases as interior + edge launches (the halo travels beside the interior items) instead of the
    default one launch per phase; = 2: only the update phase (the rows leave early, the stencil stays one launch).  Same bits."""
    import iterative_solvers_amd as isa
    from iterative_solvers_amd.distributed import Team
    monkeypatch.setenv("MI355CG_TEAM_SPLIT", split)
    for rule, kw in ((1, dict(eps_rel=1e-8, max_iterations=10 ** 5)),
                     (0, dict(eps_precision=1e-9, eps_residual=1e-9, eps_exact_error=-1.0, max_iterations=10 ** 5))):
        s1, r1, cb1 = _single(isa, n, rule, **kw)
        t = Team.local(n, world, decomp)
        cbs = []
        rt = t.solve(_params(isa, rule, **kw), callback=(lambda *a: cbs.append(a)) if rule == 0 else None)
        assert (rt.iterations, rt.stop_reason, rt.r_norm2) == (r1.iterations, r1.stop_reason, r1.r_norm2) and cbs == cb1
        assert np.array_equal(t.vector(0), s1._handle.solution()) and np.array_equal(t.vector(1), s1._handle.recursive_residual())
        t.close()


def test_team_stop_request_and_iteration_cap():
    import iterative_solvers_amd as isa
    from iterative_solvers_amd.distributed import Team
    t = Team.local(130, 4, 1)
    stop = C.c_int(0)

    def cb(it, *_):
        if it == 1:
            stop.value = 1
    res = t.solve(_params(isa, 0, eps_precision=1e-12, eps_residual=1e-12, eps_exact_error=-1.0, max_iterations=10 ** 5), callback=cb, stop_flag=stop)
    assert res.stop_reason == 4 and res.iterations == 1 and not res.converged     # msg_solver.cpp:82-87
    res = t.solve(_params(isa, 1, eps_rel=1e-30, max_iterations=37))
    assert res.iterations == 37 and not res.converged
    res = t.solve(_params(isa, 1, eps_rel=1e-8, max_iterations=10 ** 5))                              # the team is reusable
    assert res.converged
    t.close()


def test_rccl_team_of_one_rank_runs_the_collectives():
    """The RCCL transport (own communicator, ncclAllGather of the records on the comm stream, events) at world size 1 --
    all this box can offer -- must reproduce the single-GPU solve exactly."""
    import os
    import iterative_solvers_amd as isa
    from iterative_solvers_amd.distributed import Team
    n = 258
    kw = dict(eps_rel=1e-8, max_iterations=10 ** 5)
    s1, r1, _ = _single(isa, n, 1, **kw)
    os.environ["MI355CG_FORCE_COLLECTIVES"] = "1"
    try:
        t = Team.rccl(n, device=0)
        rt = t.solve(_params(isa, 1, **kw))
        os.environ["MI355CG_TEAM_HALO_INLINE"] = "1"                # the halo group on the compute stream (no events)
        rt2 = t.solve(_params(isa, 1, **kw))
    finally:
        os.environ.pop("MI355CG_FORCE_COLLECTIVES", None)
        os.environ.pop("MI355CG_TEAM_HALO_INLINE", None)
    for r in (rt, rt2):
        assert (r.iterations, r.r_norm2) == (r1.iterations, r1.r_norm2)
    assert np.array_equal(t.vector(0), s1._handle.solution())
    t.close()


def _big(n, world, decomp, iters):
    """Fixed-iteration run of a full-size config as a LOCAL team vs the single context, compared on the device."""
    import iterative_solvers_amd as isa
    from iterative_solvers_amd.distributed import Team
    kw = dict(max_iterations=iters, fixed_iterations=1, sync_every=iters)
    t = Team.local(n, world, decomp)
    rt = t.solve(_params(isa, 1, **kw))
    team = (rt.iterations, rt.r_norm2, rt.initial_r_norm2, t.checksum(0), t.checksum(1), t.checksum(2))
    t.close()
    s1, r1, _ = _single(isa, n, 1, **kw)
    one = (r1.iterations, r1.r_norm2, r1.initial_r_norm2, _cs(s1, 0), _cs(s1, 1), _cs(s1, 2))
    assert team == one, (team, one)
    assert rt.r_norm2 < rt.initial_r_norm2
    s1._handle.close()
    return rt


def test_config4_16384_as_2x2_team_matches_one_gpu():
    """BASELINE config 4: N = 16384 (201 M unknowns), 2 x 2 decomposition, 30 iterations."""
    _big(16384, 4, 1, 30)


def test_config4_16384_as_4_row_slabs_matches_one_gpu():
    _big(16384, 4, 0, 20)


def test_config5_32768_as_8_parts_matches_one_gpu():
    """BASELINE config 5: N = 32768 (805 M unknowns) over 8 parts (4 x 2), 12 iterations; 2 x 45 GB of vectors on the card."""
    _big(32768, 8, 1, 12)


@pytest.mark.parametrize("world,n,decomp", [(2, 5792, 0), (4, 8192, 0), (8, 11586, 0), (4, 8192, 1), (8, 11586, 1)])
def test_bench_weak_scaling_geometries_match_one_gpu(world, n, decomp):
    """The grids bench.py --gpus N runs (weak scaling: config-2-sized parts, N = 4096 sqrt(world) rounded to even; 11586 has an
    odd N/2), as LOCAL teams on one GPU, a fixed 24 iterations."""
    from iterative_solvers_amd.distributed import weak_scaling_n
    assert weak_scaling_n(4096, world) == n
    _big(n, world, decomp, 24)


def test_team_lifecycle_cycles_and_interleaved_handles():
    """Teams, single contexts and solves interleaved and torn down repeatedly: no stale state, no leaked events / streams."""
    import iterative_solvers_amd as isa
    from iterative_solvers_amd.distributed import Team
    ref = None
    for cycle in range(12):
        t = Team.local(130, 2 + cycle % 5, cycle % 2)
        s = isa.MatrixFreeSystem(130, 130, 1.0, 2.0, 1.0, 2.0)
        p = _params(isa, 1, eps_rel=1e-8, max_iterations=10 ** 5)
        r1 = s._handle.solve(p)
        rt = t.solve(p)
        rt2 = t.solve(p)                                           # a team is reusable; same answer again
        if ref is None:
            ref = (r1.iterations, r1.r_norm2, s._handle.solution())
        for r in (r1, rt, rt2):
            assert (r.iterations, r.r_norm2) == ref[:2]
        assert np.array_equal(t.vector(0), ref[2]) and np.array_equal(s._handle.solution(), ref[2])
        t.close(); s._handle.close()


def test_random_decompositions_match_one_gpu():
    """tools/team_fuzz.py: 60 random (grid, number of parts, rows / 2-D, rule, iteration count, launch knobs, interior+edge
    launches, one thread per part) LOCAL teams against the single context: x, r and the norms bit for bit."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import team_fuzz
    saved = {k: os.environ.get(k) for k in team_fuzz.TEAM_KNOBS}
    try:
        bad, skipped = team_fuzz.fuzz(60, 20261004, verbose=False)
        assert bad == [] and skipped < 30
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


@pytest.mark.parametrize("world", [1, 3])
def test_team_stop_request_from_another_thread_acts_inside_a_chunk(world):
    """As test_stop_request_from_another_thread_acts_at_the_next_iteration_not_at_the_next_poll, on a team: every part's update
    launch samples the pinned word, the sample travels in the part's record, and all parts end INTERRUPTED in the same iteration,
    in the middle of a queued chunk."""
    import threading
    import time
    import iterative_solvers_amd as isa
    from iterative_solvers_amd.distributed import Team
    t = Team.local(2050, world, 0)
    p = _params(isa, 1, max_iterations=10 ** 6, fixed_iterations=1, use_true_solution=0, callback_every=0, sync_every=500)
    t.solve(_params(isa, 1, max_iterations=50, fixed_iterations=1))
    stop = C.c_int(0)
    t_set = []

    def raiser():
        time.sleep(0.3)
        t_set.append(time.perf_counter())
        stop.value = 1
    th = threading.Thread(target=raiser)
    th.start()
    res = t.solve(p, stop_flag=stop)
    t_back = time.perf_counter()
    th.join()
    assert res.stop_reason == 4 and not res.converged and 100 < res.iterations < 10 ** 5
    # (parts that share one GPU order their streams with events: the launches left in the chunk return at once, the events and copies
    #  around them still run)
    assert t_back - t_set[0] < (0.05 if world == 1 else 0.4)
    r2 = t.solve(_params(isa, 1, eps_rel=1e-8, max_iterations=10 ** 5))       # and the team is as good as new
    s1, r1, _ = _single(isa, 2050, 1, eps_rel=1e-8, max_iterations=10 ** 5)
    assert (r2.iterations, r2.r_norm2) == (r1.iterations, r1.r_norm2)
    t.close()


@pytest.mark.parametrize("world,decomp", [(3, 0), (4, 1)])
def test_team_takes_a_foreign_right_hand_side_and_true_solution(world, decomp):
    """Solver(a, b, ...) takes ANY b (solver/solver.hpp:33-39) and MSGSolver::solve any true_solution (msg_solver.cpp:64-72): a team
    cuts the caller's global vectors up among its parts (mi355cg_team_set_vector) and solves as the single context does with them."""
    import iterative_solvers_amd as isa
    from iterative_solvers_amd.distributed import Team
    n = 258
    s1 = isa.MatrixFreeSystem(n, n, 1.0, 2.0, 1.0, 2.0)
    rng = np.random.default_rng(7)
    b = s1.get_rhs() * rng.uniform(0.5, 1.5, s1.size())
    u = s1.get_true_solution_vector() + rng.uniform(-1e-3, 1e-3, s1.size())
    kw = dict(eps_precision=1e-9, eps_residual=1e-9, eps_exact_error=1e-4, max_iterations=10 ** 5)
    s1._handle.set_rhs(b)
    s1._handle.set_true_solution(u)
    cb1 = []
    r1 = s1._handle.solve(_params(isa, 0, **kw), callback=lambda *a: cb1.append(a))
    t = Team.local(n, world, decomp)
    t.set_vector(2, b)
    t.set_vector(3, u)
    assert np.array_equal(t.vector(2), b) and np.array_equal(t.vector(3), u)
    cbs = []
    rt = t.solve(_params(isa, 0, **kw), callback=lambda *a: cbs.append(a))
    assert (rt.iterations, rt.stop_reason, rt.final_residual_norm, rt.final_precision, rt.final_error_norm) == \
           (r1.iterations, r1.stop_reason, r1.final_residual_norm, r1.final_precision, r1.final_error_norm)
    assert cbs == cb1 and np.array_equal(t.vector(0), s1._handle.solution())
    with pytest.raises(ValueError):
        t.set_vector(2, b[:-1])
    t.close()


@pytest.mark.parametrize("n,world,decomp", [(258, 4, 1), (130, 3, 0), (514, 8, 1)])
def test_local_team_with_mailboxes_and_pushed_halo(n, world, decomp, monkeypatch):
    """The rank-process transport's mechanisms inside ONE process: records through the parts' mailboxes with stream-level waits (parts that
    share a GPU must not poll for each other in kernels) and the halo pushed into the neighbours' receive buffers.  Same bits."""
    import iterative_solvers_amd as isa
    from iterative_solvers_amd.distributed import Team
    monkeypatch.setenv("MI355CG_TEAM_RECORDS", "mailbox")
    monkeypatch.setenv("MI355CG_TEAM_HALO", "push")
    kw = dict(eps_rel=1e-8, max_iterations=10 ** 5)
    s1, r1, _ = _single(isa, n, 1, **kw)
    t = Team.local(n, world, decomp)
    d = t.describe()
    assert (d["transport"], d["records"], d["wait"], d["halo"]) == ("local", "mailbox", "stream", "push")
    rt = t.solve(_params(isa, 1, **kw))
    assert (rt.iterations, rt.r_norm2, rt.initial_r_norm2) == (r1.iterations, r1.r_norm2, r1.initial_r_norm2)
    assert np.array_equal(t.vector(0), s1._handle.solution())
    kw = dict(eps_precision=1e-9, eps_residual=1e-9, eps_exact_error=-1.0, max_iterations=10 ** 5)
    s0, r0, cb0 = _single(isa, n, 0, **kw)
    cbs = []
    rm = t.solve(_params(isa, 0, **kw), callback=lambda *a: cbs.append(a))
    assert (rm.iterations, rm.stop_reason, rm.final_residual_norm, rm.final_precision) == (r0.iterations, r0.stop_reason, r0.final_residual_norm, r0.final_precision)
    assert cbs == cb0 and np.array_equal(t.vector(0), s0._handle.solution())
    t.close()


@pytest.mark.parametrize("n,world,env", [(256, 2, {}), (514, 3, {}), (1026, 4, {}), (258, 5, {"MI355CG_TEAM_RECORDS": "mailbox", "MI355CG_TEAM_HALO": "push"}),
                                         (514, 3, {"MI355CG_TEAM_THREADS": "1"})])
def test_team_mixed_precision_is_the_single_gpu_mixed_solve(n, world, env, monkeypatch):
    """BASELINE config 3's algorithm on a team of row slabs (mi355cg_team_set_dtype): the fp32 CG loop runs across the parts (halo rows
    of 4-byte elements in the residual vector's memory), the fp64 refinement steps exchange the halo of x.  Same inner iterations, same
    outer steps, the same x bit for bit as the single-GPU F32_MIXED solve; then the team is an fp64 team again."""
    import iterative_solvers_amd as isa
    from iterative_solvers_amd.distributed import Team
    from oracle.oracle import OracleGrid
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    kw = dict(eps_rel=1e-8, max_iterations=10 ** 6)
    s1 = isa.MatrixFreeSystem(n, n, 1.0, 2.0, 1.0, 2.0, dtype=isa.F32_MIXED)
    cb1 = []
    r1 = s1._handle.solve(_params(isa, 1, **kw), callback=lambda *a: cb1.append(a))
    t = Team.local(n, world, 0)
    t.set_dtype(isa.F32_MIXED)
    cbs = []
    rt = t.solve(_params(isa, 1, **kw), callback=lambda *a: cbs.append(a))
    assert (rt.iterations, rt.converged, rt.stop_reason, rt.refine_outer) == (r1.iterations, 1, r1.stop_reason, r1.refine_outer) and rt.refine_outer >= 2
    assert [c[0] for c in cbs] == [c[0] for c in cb1]                      # the inner solves end after the same iterations
    assert np.allclose([c[2] for c in cbs], [c[2] for c in cb1], rtol=1e-10, atol=0)
    x = t.vector(0)
    assert np.array_equal(x, s1._handle.solution())
    assert abs(rt.r_norm2 - r1.r_norm2) <= 1e-12 * r1.initial_r_norm2
    if n <= 514:
        og = OracleGrid(n, n)
        b = og.rhs()
        assert np.linalg.norm(b - og.apply(x)) <= 1e-8 * np.linalg.norm(b)            # the fp64 true residual by the oracle's operator
        r = t.vector(1)                                                             # ... and the team left b - A x in r
        assert np.abs(r - (b - og.apply(x))).max() <= 1e-12 * np.abs(b).max()
    with pytest.raises(ValueError):
        t.solve(_params(isa, 0, max_iterations=10))                                 # F32_MIXED: REL_2NORM only
    t.set_dtype(isa.F64)
    s64, r64, _ = _single(isa, n, 1, eps_rel=1e-8, max_iterations=10 ** 5)
    r2 = t.solve(_params(isa, 1, eps_rel=1e-8, max_iterations=10 ** 5))
    assert (r2.iterations, r2.r_norm2) == (r64.iterations, r64.r_norm2) and np.array_equal(t.vector(0), s64._handle.solution())
    t.close()
    t2 = Team.local(258, 4, 1)
    with pytest.raises(ValueError):
        t2.set_dtype(isa.F32_MIXED)                                                 # a 2 x 2 cut has no 256-column strips
    t2.close()


@pytest.mark.parametrize("n,world,decomp", [(1026, 2, 0), (1026, 4, 1), (2050, 8, 0)])
def test_local_team_across_several_gpus(n, world, decomp):
    """One process, one GPU per part (peer access over xGMI, one host thread per part, kernels polling their mailboxes): runs wherever
    the box has the GPUs; the one-GPU test box skips it."""
    import torch
    if torch.cuda.device_count() < world:
        pytest.skip(f"needs {world} GPUs, this box has {torch.cuda.device_count()}")
    import iterative_solvers_amd as isa
    from iterative_solvers_amd.distributed import Team
    kw = dict(eps_rel=1e-8, max_iterations=10 ** 5)
    s1, r1, _ = _single(isa, n, 1, **kw)
    t = Team.local(n, world, decomp, devices=list(range(world)))
    d = t.describe()
    assert (d["records"], d["wait"]) == ("mailbox", "kernel")
    rt = t.solve(_params(isa, 1, **kw))
    assert (rt.iterations, rt.r_norm2) == (r1.iterations, r1.r_norm2) and np.array_equal(t.vector(0), s1._handle.solution())
    kw = dict(eps_precision=1e-9, eps_residual=1e-9, eps_exact_error=-1.0, max_iterations=10 ** 5)
    s0, r0, cb0 = _single(isa, n, 0, **kw)
    cbs = []
    rm = t.solve(_params(isa, 0, **kw), callback=lambda *a: cbs.append(a))
    assert (rm.iterations, rm.stop_reason) == (r0.iterations, r0.stop_reason) and cbs == cb0 and np.array_equal(t.vector(0), s0._handle.solution())
    if decomp == 0:
        t.set_dtype(isa.F32_MIXED)
        s32 = isa.MatrixFreeSystem(n, n, 1.0, 2.0, 1.0, 2.0, dtype=isa.F32_MIXED)
        r32 = s32._handle.solve(_params(isa, 1, eps_rel=1e-8, max_iterations=10 ** 6))
        rt = t.solve(_params(isa, 1, eps_rel=1e-8, max_iterations=10 ** 6))
        assert (rt.iterations, rt.refine_outer) == (r32.iterations, r32.refine_outer) and np.array_equal(t.vector(0), s32._handle.solution())
    t.close()
