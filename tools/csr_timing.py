#!/usr/bin/env python3
"""Timing of the generic CSR path (mi355cg_create_csr): the Poisson matrix of an N x N grid handed over as CSR,
fixed-iteration CG.  Usage: python tools/csr_timing.py [N] [iters]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import iterative_solvers_amd as isa
from iterative_solvers_amd import _capi
from iterative_solvers_amd.solver import _Handle
from oracle.oracle import OracleGrid        # test infrastructure: only used to assemble the matrix for this tool

N = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 500
og = OracleGrid(N, N)
row_map, entries, values = og.csr()
b = og.rhs()
h = _Handle.from_csr(row_map, entries, values)
h.set_rhs(b)
p = isa.default_params(_capi.RULE_REL_2NORM)
p.max_iterations, p.fixed_iterations, p.use_true_solution, p.callback_every = iters, 1, 0, 0
h.solve(p)
t0 = time.perf_counter(); res = h.solve(p); dt = time.perf_counter() - t0
U, nnz = len(b), len(values)
bytes_it = nnz * 12 + (U + 1) * 4 + 11 * 8 * U          # matrix stream + row map + 11 vector words (x gather counted once)
print(f"N {N} unknowns {U} nnz {nnz}: {res.iterations / dt:.1f} it/s, {dt / res.iterations * 1e6:.1f} us/iteration, "
      f"~{bytes_it * res.iterations / dt / 1e9:.0f} GB/s of compulsory traffic ({bytes_it / 1e6:.0f} MB/iteration)")
