#!/usr/bin/env python3
"""BASELINE configs 4 and 5 as LOCAL teams on ONE GPU (all parts share the card, so the wall time is the sum of the parts'
kernels plus the single host thread's driving cost -- a rehearsal of the decomposition, not a scaling measurement):
  config 4: N = 16384, 2 x 2, REL_2NORM to 1e-8 (full solve) against the single-context solve of the same grid
  config 5: N = 32768, 8 parts (4 x 2) and 8 row slabs, a fixed number of iterations
  config 3 on a team: N = 8192, F32_MIXED, 4 row slabs, full solve against the single-GPU F32_MIXED solve
Usage: python tools/team_config_runs.py [3|4|5] [iters5]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ctypes as C
import iterative_solvers_amd as isa
from iterative_solvers_amd import _capi
from iterative_solvers_amd.distributed import Team

which = sys.argv[1] if len(sys.argv) > 1 else "4"


def cs(handle, k):
    o = (C.c_double * 2)()
    _capi.check(_capi.load().mi355cg_checksum(handle._h, k, o))
    return [o[0], o[1]]


def params(**kw):
    p = isa.default_params(isa.RULE_REL_2NORM)
    for k, v in kw.items():
        setattr(p, k, v)
    return p


if which == "3":
    n, kw = 8192, dict(eps_rel=1e-8, max_iterations=10 ** 6)
    t = Team.local(n, 4, _capi.DECOMP_ROWS)
    t.set_dtype(isa.F32_MIXED)
    t0 = time.perf_counter(); r = t.solve(params(**kw)); dt = time.perf_counter() - t0
    out = {"config": "3 on a team: 8192 F32_MIXED, 4 row slabs, LOCAL team on one GPU, rel2 1e-8", "iterations": r.iterations, "outer_steps": r.refine_outer, "converged": bool(r.converged),
           "true_rel_residual": r.refine_true_rel, "seconds": round(dt, 2), "checksum_x": list(t.checksum(0))}
    print(json.dumps(out), flush=True)
    t.close()
    s = isa.MatrixFreeSystem(n, n, 1.0, 2.0, 1.0, 2.0, dtype=isa.F32_MIXED)
    t0 = time.perf_counter(); r1 = s._handle.solve(params(**kw)); dt = time.perf_counter() - t0
    one = {"config": "3: the same grid on the single context", "iterations": r1.iterations, "outer_steps": r1.refine_outer, "converged": bool(r1.converged),
           "true_rel_residual": r1.refine_true_rel, "seconds": round(dt, 2), "iters_per_sec": round(r1.iterations / dt, 1), "checksum_x": cs(s._handle, 0)}
    print(json.dumps(one), flush=True)
    print(json.dumps({"config3_team_equals_single_context": {"iterations_and_outer_steps": (out["iterations"], out["outer_steps"]) == (one["iterations"], one["outer_steps"]),
                                                             "x_bit_for_bit": out["checksum_x"] == one["checksum_x"]}}), flush=True)
elif which == "4":
    n, kw = 16384, dict(eps_rel=1e-8, max_iterations=10 ** 6)
    t = Team.local(n, 4, _capi.DECOMP_2D)
    t0 = time.perf_counter(); r = t.solve(params(**kw)); dt = time.perf_counter() - t0
    out = {"config": "4: 16384 fp64, 2 x 2 LOCAL team on one GPU, rel2 1e-8", "iterations": r.iterations, "converged": bool(r.converged),
           "r_norm2": r.r_norm2, "initial_r_norm2": r.initial_r_norm2, "seconds": round(dt, 2), "checksum_x": list(t.checksum(0)), "checksum_r": list(t.checksum(1))}
    print(json.dumps(out), flush=True)
    t.close()
    s = isa.MatrixFreeSystem(n, n, 1.0, 2.0, 1.0, 2.0)
    t0 = time.perf_counter(); r1 = s._handle.solve(params(**kw)); dt = time.perf_counter() - t0
    one = {"config": "4': the same grid on the single context", "iterations": r1.iterations, "converged": bool(r1.converged), "r_norm2": r1.r_norm2,
           "initial_r_norm2": r1.initial_r_norm2, "seconds": round(dt, 2), "iters_per_sec": round(r1.iterations / dt, 1),
           "checksum_x": cs(s._handle, 0), "checksum_r": cs(s._handle, 1)}
    print(json.dumps(one), flush=True)
    same = all(out[k] == one[k] for k in ("iterations", "r_norm2", "initial_r_norm2", "checksum_x", "checksum_r"))
    print(json.dumps({"config4_team_equals_single_context_bit_for_bit": same}), flush=True)
else:
    n, iters = 32768, int(sys.argv[2]) if len(sys.argv) > 2 else 300
    kw = dict(max_iterations=iters, fixed_iterations=1, sync_every=500)
    res = {}
    for name, world, decomp in (("8 parts 4 x 2", 8, _capi.DECOMP_2D), ("8 row slabs", 8, _capi.DECOMP_ROWS)):
        t = Team.local(n, world, decomp)
        t0 = time.perf_counter(); r = t.solve(params(**kw)); dt = time.perf_counter() - t0
        res[name] = {"iterations": r.iterations, "r_norm2": r.r_norm2, "seconds": round(dt, 2), "ms_per_iteration": round(1e3 * dt / iters, 3),
                     "checksum_x": list(t.checksum(0)), "checksum_r": list(t.checksum(1))}
        print(json.dumps({"config": f"5: 32768 fp64, {name}, LOCAL team on one GPU, fixed {iters} iterations", **res[name]}), flush=True)
        t.close()
    s = isa.MatrixFreeSystem(n, n, 1.0, 2.0, 1.0, 2.0)
    t0 = time.perf_counter(); r1 = s._handle.solve(params(**kw)); dt = time.perf_counter() - t0
    one = {"iterations": r1.iterations, "r_norm2": r1.r_norm2, "seconds": round(dt, 2), "ms_per_iteration": round(1e3 * dt / iters, 3),
           "checksum_x": cs(s._handle, 0), "checksum_r": cs(s._handle, 1)}
    print(json.dumps({"config": "5': the same grid on the single context", **one}), flush=True)
    print(json.dumps({"config5_teams_equal_single_context_bit_for_bit": all(all(v[k] == one[k] for k in ("iterations", "r_norm2", "checksum_x", "checksum_r")) for v in res.values())}), flush=True)
