"""Long CPU runs of the ORACLE (not the reference) at N = 2048, kept as a fixture so the GPU path is pinned against the
reference-faithful serial arithmetic at a size the test suite cannot afford to recompute (about 25 minutes on one core)."""
import sys, time, json
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__)))))
from oracle.oracle import OracleGrid
out = {}
for N in (2048,):
    g = OracleGrid(N, N)
    t = time.time(); m = g.mf_solve(eps=1e-8, max_iterations=10**6)
    out[f"mf_{N}"] = {"iterations": m.iterations, "r_norm": m.r_norm, "initial_r_norm": m.initial_r_norm, "seconds": time.time() - t}
    print(out, flush=True)
    t = time.time(); r = g.msg_solve(eps_precision=1e-8, eps_residual=1e-8, eps_exact_error=-1.0, max_iterations=10**6)
    out[f"msg_{N}"] = {"iterations": r.iterations, "stop_reason": r.stop_reason, "final_residual_norm": r.final_residual_norm,
                       "final_precision": r.final_precision, "final_error_norm": r.final_error_norm, "seconds": time.time() - t}
    print(out, flush=True)
json.dump(out, open('' + __import__('os').path.join(__import__('os').path.dirname(__import__('os').path.abspath(__file__)), 'oracle_n2048.json') + '', 'w'), indent=1)
