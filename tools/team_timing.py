#!/usr/bin/env python3
"""Per-iteration cost of the team loop on ONE GPU: LOCAL teams of 1..8 parts against the single-context solve.
Usage: python tools/team_timing.py N iters"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import iterative_solvers_amd as isa
from iterative_solvers_amd import _capi
from iterative_solvers_amd.distributed import Team

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
IT = int(sys.argv[2]) if len(sys.argv) > 2 else 300


def params():
    p = isa.default_params(_capi.RULE_REL_2NORM)
    p.max_iterations, p.fixed_iterations, p.use_true_solution, p.callback_every, p.sync_every = IT, 1, 0, 0, 500
    return p


s = isa.MatrixFreeSystem(N, N, 1.0, 2.0, 1.0, 2.0)
s._handle.solve(params())
r = s._handle.solve(params())
print(f"N={N} single context: {1e3 * r.loop_seconds / IT:.4f} ms/iteration (events), {1e3 * r.solve_seconds / IT:.4f} wall", flush=True)
s._handle.close()
for world, decomp in ((1, 0), (2, 0), (4, 0), (4, 1), (8, 0), (8, 1)):
    t = Team.local(N, world, decomp)
    t.solve(params())
    t.set_profiling(True)
    r = t.solve(params())
    ph = t.phase_times()
    print(f"N={N} LOCAL team world={world} decomp={'2d' if decomp else 'rows'}: {1e3 * r.solve_seconds / IT:.4f} ms/iteration wall | lead part: kernels {ph['kernels_ms']:.4f} comm {ph['comm_ms']:.4f}", flush=True)
    t.close()
