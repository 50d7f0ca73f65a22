#!/usr/bin/env python3
"""Run the BASELINE.json configurations that fit one MI355X to convergence and print one JSON line
each (iterations, stop reason, wall time, it/s, algorithmic GB/s, fp64 true residual)."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import iterative_solvers_amd as isa


def U(n): return (n // 2 - 1) * (3 * n // 2 - 1)


def run(name, n, dtype, rule, **kw):
    s = isa.GridSystem(n, n, 1.0, 2.0, 1.0, 2.0, dtype=dtype)
    h = s._handle
    p = isa.default_params(rule)
    for k, v in kw.items():
        setattr(p, k, v)
    h.solve(p) if n <= 1024 else None                       # warm-up on small grids only
    t0 = time.perf_counter()
    res = h.solve(p)
    dt = time.perf_counter() - t0
    r = h.true_residual()                                    # A x - b in fp64
    b = h.rhs()
    wb = 4.0 if dtype == isa.F32_MIXED else 8.0
    words = 8.0 if rule == 0 else (7.5 if os.environ.get('MI355CG_XSTEPS') == '2' else 7.25)     # words really moved per unknown and iteration
    out = {"config": name, "n": n, "unknowns": U(n), "dtype": "f32-mixed" if dtype == isa.F32_MIXED else "f64",
           "rule": "msg" if rule == 0 else "rel2", "iterations": res.iterations, "converged": bool(res.converged),
           "stop_reason": res.stop_reason, "seconds": round(dt, 4), "iters_per_sec": round(res.iterations / dt, 1),
           "moved_gbps": round(words * wb * U(n) * res.iterations / dt / 1e9, 1),
           "algorithmic_equivalent_gbps_88B": round(11 * wb * U(n) * res.iterations / dt / 1e9, 1),
           "true_residual_rel_2norm": float(np.linalg.norm(r) / np.linalg.norm(b)),
           "true_residual_maxnorm": float(np.abs(r).max()), "refine_outer": res.refine_outer}
    print(json.dumps(out), flush=True)
    h.close()


if __name__ == "__main__":
    R2, MSG = isa.RULE_REL_2NORM, isa.RULE_MSG_MAXNORM
    run("1: 256 fp64 rel2 1e-8 (reference run: 701)", 256, isa.F64, R2, eps_rel=1e-8, max_iterations=10 ** 6)
    run("1: 256 fp64 MSG facade defaults (631, PRECISION)", 256, isa.F64, MSG)
    run("2: 4096 fp64 rel2 1e-8", 4096, isa.F64, R2, eps_rel=1e-8, max_iterations=10 ** 6)
    run("2: 4096 fp64 MSG all-three 1e-8", 4096, isa.F64, MSG, eps_precision=1e-8, eps_residual=1e-8, eps_exact_error=1e-8, max_iterations=10 ** 6)
    run("3: 8192 f32-mixed rel2 1e-8", 8192, isa.F32_MIXED, R2, eps_rel=1e-8, max_iterations=10 ** 6)
    run("3': 8192 fp64 rel2 1e-8 (comparison)", 8192, isa.F64, R2, eps_rel=1e-8, max_iterations=10 ** 6)
    if os.environ.get("MI355CG_RUN_16384") == "1":
        run("4': 16384 fp64 rel2 1e-8 on ONE GPU", 16384, isa.F64, R2, eps_rel=1e-8, max_iterations=10 ** 6)
    run("4'': 16384 fp64 rel2, fixed 2000 iterations on ONE GPU", 16384, isa.F64, R2, max_iterations=2000, fixed_iterations=1)
