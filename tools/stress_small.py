#!/usr/bin/env python3
"""Repeated create / solve / destroy of tiny grids between larger ones: flushes out stale-state and reuse bugs."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import iterative_solvers_amd as isa
from iterative_solvers_amd import _capi


def solve(n, rule=_capi.RULE_REL_2NORM, max_it=3000):
    s = isa.MatrixFreeSystem(n, n, 1.0, 2.0, 1.0, 2.0)
    p = isa.default_params(rule)
    p.max_iterations = max_it
    p.eps_rel = 1e-10
    res = s._handle.solve(p)
    s._handle.close()
    return res


expect = {}
bad = 0
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 60):
    for n in (6, 256, 8, 6, 64, 6):
        r = solve(n)
        key = n
        if key not in expect:
            expect[key] = r.iterations
            print("first", n, r.iterations, r.converged, r.r_norm2, r.initial_r_norm2, flush=True)
        elif r.iterations != expect[key] or not r.converged:
            bad += 1
            print("MISMATCH rep", rep, "n", n, "iterations", r.iterations, "expected", expect[key], "converged", r.converged,
                  "r_norm2", r.r_norm2, "initial", r.initial_r_norm2, flush=True)
print("bad =", bad)
