"""Converged CPU runs of the ORACLE (not the reference) at the headline size N = 4096 (config 2), kept as a fixture:
the serial reference-faithful arithmetic takes 30-90 minutes per solve on one core, which no test run can afford.

    python tests/golden/make_oracle_n4096.py mf        # MatrixFreeSolver rule, eps 1e-8 (no diagnostics)
    python tests/golden/make_oracle_n4096.py msg       # MSGSolver rules, precision+residual 1e-8
    python tests/golden/make_oracle_n4096.py mfdiag    # MatrixFreeSolver with the per-iteration diagnostics
    python tests/golden/make_oracle_n4096.py mfexact   # MatrixFreeSolver rule with oracle.exact_dots() (Dot2 inner products)
    python tests/golden/make_oracle_n4096.py msgexact  # MSGSolver rules with oracle.exact_dots()
    python tests/golden/make_oracle_n4096.py merge     # -> oracle_n4096.json

The three legs are independent processes (run them side by side).  Vectors are kept as a strided sample
(every STRIDE-th packed unknown, hex doubles) plus serial checksums.
"""
import json
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
N = 4096
STRIDE = 3071          # prime; 12 574 721 / 3071 -> 4095 samples


def sample(v):
    import numpy as np
    from oracle import oracle as og
    s = v[::STRIDE]
    return {"stride": STRIDE, "hex": [float(t).hex() for t in s],
            "serial_sum": og.dot(v, np.ones_like(v)), "max_norm": og.max_norm(v)}


def leg(which):
    from oracle.oracle import OracleGrid
    g = OracleGrid(N, N)
    t = time.time()
    if which == "mf":
        m = g.mf_solve(eps=1e-8, max_iterations=10**6)
        out = {"iterations": m.iterations, "converged": m.converged, "r_norm": m.r_norm,
               "initial_r_norm": m.initial_r_norm, "x": sample(m.x)}
    elif which == "mfdiag":
        m = g.mf_solve(eps=1e-8, max_iterations=10**6, diagnostics=True)
        cbs = [c for c in m.callbacks if c[0] <= 10 or c[0] % 250 == 0 or c[0] >= m.iterations - 10]
        out = {"iterations": m.iterations, "converged": m.converged, "r_norm": m.r_norm,
               "initial_r_norm": m.initial_r_norm, "callbacks": cbs}
    elif which == "mfexact":
        from oracle import oracle as og
        with og.exact_dots():
            m = g.mf_solve(eps=1e-8, max_iterations=10**6)
        out = {"iterations": m.iterations, "converged": m.converged, "r_norm": m.r_norm,
               "initial_r_norm": m.initial_r_norm, "x": sample(m.x)}
    elif which == "msgexact":
        from oracle import oracle as og
        with og.exact_dots():
            r = g.msg_solve(eps_precision=1e-8, eps_residual=1e-8, eps_exact_error=-1.0, max_iterations=10**6)
        out = {"iterations": r.iterations, "converged": r.converged, "stop_reason": r.stop_reason,
               "final_residual_norm": r.final_residual_norm, "final_precision": r.final_precision,
               "final_error_norm": r.final_error_norm, "r_norm2": r.r_norm2, "initial_r_norm2": r.initial_r_norm2,
               "callbacks": r.callbacks, "x": sample(r.x), "r": sample(r.r)}
    elif which == "msg":
        r = g.msg_solve(eps_precision=1e-8, eps_residual=1e-8, eps_exact_error=-1.0, max_iterations=10**6)
        out = {"iterations": r.iterations, "converged": r.converged, "stop_reason": r.stop_reason,
               "final_residual_norm": r.final_residual_norm, "final_precision": r.final_precision,
               "final_error_norm": r.final_error_norm, "r_norm2": r.r_norm2, "initial_r_norm2": r.initial_r_norm2,
               "callbacks": r.callbacks, "x": sample(r.x), "r": sample(r.r)}
    else:
        raise SystemExit(which)
    out["seconds"] = time.time() - t
    json.dump(out, open(os.path.join(HERE, f"_n4096_{which}.part.json"), "w"))
    print(which, {k: v for k, v in out.items() if k not in ("x", "r", "callbacks")}, flush=True)


if __name__ == "__main__":
    if sys.argv[1] == "merge":
        out = {}
        for which in ("mf", "msg", "mfdiag", "mfexact", "msgexact"):
            p = os.path.join(HERE, f"_n4096_{which}.part.json")
            if os.path.exists(p):
                out[f"{which}_{N}"] = json.load(open(p))
        json.dump(out, open(os.path.join(HERE, "oracle_n4096.json"), "w"), indent=0)
        print({k: v["iterations"] for k, v in out.items()})
    else:
        leg(sys.argv[1])
