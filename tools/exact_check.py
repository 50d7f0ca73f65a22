#!/usr/bin/env python3
"""One-off: the first iterations of a large grid on the GPU against the CPU oracle with exact inner products (oracle.exact_dots),
bit for bit.  For sizes the test suite cannot afford (N = 32768: ~80 GB of host memory, minutes of serial CPU work).
Usage (GPU box): python tools/exact_check.py N iters"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import iterative_solvers_amd as isa
from oracle import oracle as o

N, K = int(sys.argv[1]), int(sys.argv[2])
t0 = time.time()
s = isa.MatrixFreeSystem(N, N, 1.0, 2.0, 1.0, 2.0)
og = o.OracleGrid(N, N)
b = og.rhs()
print(f"N={N} unknowns={og.size} set-up {time.time() - t0:.1f} s; rhs equal: {np.array_equal(s.get_rhs(), b)}", flush=True)
sol = isa.MatrixFreeSolver(s, b, 1e-30, K)
t0 = time.time()
xg = sol.solve()
tg = time.time() - t0
t0 = time.time()
with o.exact_dots():
    ex = og.mf_solve(eps=1e-30, max_iterations=K)
print(f"{K} iterations: GPU {tg:.2f} s (with the host round trips), oracle {time.time() - t0:.1f} s", flush=True)
print(f"||r|| gpu {sol.last_results.r_norm2!r} oracle {ex.r_norm!r} equal {sol.last_results.r_norm2 == ex.r_norm}; "
      f"||r0|| equal {sol.last_results.initial_r_norm2 == ex.initial_r_norm}; x bit-identical: {np.array_equal(xg, ex.x)}", flush=True)
