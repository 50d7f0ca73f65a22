"""Freeze small outputs of the ORACLE (oracle/cg_oracle.c, itself pinned to the reference's n = 6 artefacts by
tests/test_oracle_golden.py) as SURVEY 8c's fixtures F2-F5, so that (a) an accidental change of the oracle shows up as a
diff against committed numbers and (b) the GPU path can be checked against data, not only against a live oracle build.
Doubles are stored as C99 hex strings (float.hex()): bit-exact and platform independent.
Usage: python tests/golden/make_oracle_fixtures.py   (writes tests/golden/oracle_small.json)"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle.oracle import OracleGrid


def hx(v):
    return [float(x).hex() for x in np.asarray(v, dtype=np.float64).ravel()]


def seeded(n):
    return np.random.Generator(np.random.PCG64(12345)).uniform(-1.0, 1.0, n)         # PCG64 seed 12345, U(-1, 1)


out = {"doc": "oracle outputs; doubles as float.hex() strings", "F2": {}, "F3": {}, "F4": {}, "F5": {}}
for N in (6, 8, 16, 64):
    g = OracleGrid(N, N)
    out["F2"][str(N)] = {"size": g.size, "rhs": hx(g.rhs()), "u_true": hx(g.true_solution())}
for N in (6, 8, 16, 64, 256):
    g = OracleGrid(N, N)
    x = seeded(g.size)
    y = g.apply(x)
    rec = {"size": g.size, "sum_y": float(np.sum(y)).hex(), "sum_abs_y": float(np.sum(np.abs(y))).hex()}
    if N <= 64:
        rec["y"] = hx(y)
    else:
        rec["y_stride_97"] = hx(y[::97])
    out["F3"][str(N)] = rec
for N in (6, 16, 64, 256):
    g = OracleGrid(N, N)
    m = g.mf_solve(eps=1e-8, max_iterations=10 ** 5, diagnostics=True)
    cbs = np.array(m.callbacks, dtype=float)                     # (it, ||dx||_2, ||b - Ax||_2, ||x - u||_2) per iteration
    rec = {"iterations": m.iterations, "converged": bool(m.converged), "r_norm": float(m.r_norm).hex(),
           "initial_r_norm": float(m.initial_r_norm).hex(),
           "callbacks_every": 1 if N <= 16 else 25, "callbacks": [[int(c[0])] + hx(c[1:]) for c in cbs[::(1 if N <= 16 else 25)]],
           "x": hx(m.x) if N <= 64 else None, "x_stride_97": hx(m.x[::97]) if N > 64 else None}
    r = g.msg_solve(eps_precision=1e-8, eps_residual=1e-8, eps_exact_error=-1.0, max_iterations=10 ** 5)
    rec["msg"] = {"iterations": r.iterations, "stop_reason": r.stop_reason, "converged": bool(r.converged),
                  "final_residual_norm": float(r.final_residual_norm).hex(), "final_precision": float(r.final_precision).hex(),
                  "final_error_norm": float(r.final_error_norm).hex(),
                  "callbacks": [[int(c[0])] + hx(c[1:]) for c in r.callbacks]}
    out["F4"][str(N)] = rec
for N, it in ((512, 1371), (1024, 2673)):
    g = OracleGrid(N, N)
    m = g.mf_solve(eps=1e-8, max_iterations=10 ** 5)
    assert m.iterations == it, (N, m.iterations)
    out["F5"][str(N)] = {"iterations": m.iterations, "r_norm": float(m.r_norm).hex()}
with open(os.path.join(HERE, "oracle_small.json"), "w") as f:
    json.dump(out, f, indent=0)
print("written", os.path.getsize(os.path.join(HERE, "oracle_small.json")), "bytes")
