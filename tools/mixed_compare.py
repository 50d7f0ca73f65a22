#!/usr/bin/env python3
"""Config 3 (N=8192 by default): restarted refinement vs residual replacement vs fp64, one box."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import iterative_solvers_amd as isa
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8192


def run(tag, dtype, env, **kw):
    for k, v in env.items():
        os.environ[k] = str(v)
    s = isa.MatrixFreeSystem(N, N, 1.0, 2.0, 1.0, 2.0, dtype=dtype)
    sol = isa.MatrixFreeSolver(s, s.get_rhs(), 1e-8, 10 ** 6)
    t0 = time.perf_counter(); sol.solve(**kw); dt = time.perf_counter() - t0
    r = sol.last_results
    print(f"{tag:34s} iterations {r.iterations:6d}  outer {r.refine_outer}  {dt:7.3f} s  true rel {r.refine_true_rel:.2e}  conv {bool(r.converged)}", flush=True)
    s._handle.close()


run("fp64", isa.F64, {})
for ie in (1e-3, 1e-4, 1e-5):
    run(f"mixed restart inner={ie:g}", isa.F32_MIXED, {"MI355CG_MIXED_RESTART": 1}, inner_eps=ie)
    run(f"mixed replacement inner={ie:g}", isa.F32_MIXED, {"MI355CG_MIXED_RESTART": 0}, inner_eps=ie)
