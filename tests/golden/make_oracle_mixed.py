"""CPU statement of the mixed-precision algorithm (oracle.mixed_solve; NOT the reference, which has no fp32 path) at N = 2050 --
the smallest size with XCD-local item ranges in the fp32 launch plan, ~15 minutes on one core -- kept as a fixture for
tests/test_gpu_mixed.py.      python tests/golden/make_oracle_mixed.py"""
import json
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle.oracle import OracleGrid

N, STRIDE = 2050, 769
g = OracleGrid(N, N)
t = time.time()
x, its, outer, conv, rel = g.mixed_solve(eps=1e-8)
out = {"n": N, "iterations": its, "outer": outer, "converged": conv, "true_rel": rel, "seconds": time.time() - t,
       "x": {"stride": STRIDE, "hex": [float(v).hex() for v in x[::STRIDE]]}}
json.dump(out, open(os.path.join(HERE, "oracle_mixed_n2050.json"), "w"), indent=0)
print({k: v for k, v in out.items() if k != "x"})
