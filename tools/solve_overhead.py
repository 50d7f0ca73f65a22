#!/usr/bin/env python3
"""Fixed cost of one solve (initialisation pass, polls, final synchronisation): mean wall time of solves of K = 1 and K = 20 iterations
through mi355cg_solve and through mi355cg_team_solve (LOCAL team of one part), N = 4096.  The driver times 20-iteration solves, so
0.1 ms of fixed cost is 4 % of its window."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import iterative_solvers_amd as isa
from iterative_solvers_amd import _capi
from iterative_solvers_amd.distributed import Team

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096


def params(k):
    p = isa.default_params(_capi.RULE_REL_2NORM)
    p.max_iterations, p.fixed_iterations, p.use_true_solution, p.callback_every, p.sync_every = k, 1, 0, 0, 500
    return p


def timeit(solve, k, reps=100):
    solve(params(k)); solve(params(k))
    t0 = time.perf_counter()
    for _ in range(reps):
        solve(params(k))
    return (time.perf_counter() - t0) / reps * 1e3


s = isa.MatrixFreeSystem(N, N, 1.0, 2.0, 1.0, 2.0)
t = Team.local(N, 1)
for k in (1, 20, 100):
    a, b = timeit(s._handle.solve, k), timeit(t.solve, k)
    print(f"N={N} K={k:4d}: mi355cg_solve {a:7.3f} ms   mi355cg_team_solve (one part) {b:7.3f} ms   difference {1e3 * (b - a):7.1f} us", flush=True)
