#!/usr/bin/env python3
"""Generic CSR path (Solver(const KokkosCrsMatrix&, ...), KokkosSparse::spmv) timed beside the stencil path
on the same operator.  Usage (GPU box): python tools/csr_timing.py [N] [iters]
The CSR of the L-shaped 5-point operator is assembled here with numpy (packed order, entries bottom/left/centre/right/top)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import iterative_solvers_amd as isa
from iterative_solvers_amd import _capi

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
ITERS = int(sys.argv[2]) if len(sys.argv) > 2 else 300
half = N // 2
h = 1.0 / N
idx = -np.ones((N + 1, N + 1), dtype=np.int64)               # [y, x]
k = 0
for y0, y1, x0 in ((1, half, half + 1), (half + 1, N - 1, 1)):
    ny, nx = y1 - y0 + 1, N - x0
    idx[y0:y1 + 1, x0:N] = k + np.arange(ny * nx).reshape(ny, nx)
    k += ny * nx
U = k
ys, xs = np.nonzero(idx >= 0)
order = np.argsort(idx[ys, xs])
ys, xs = ys[order], xs[order]
cols = np.stack([idx[ys - 1, xs], idx[ys, xs - 1], idx[ys, xs], idx[ys, xs + 1], idx[ys + 1, xs]], axis=1)
vals = np.broadcast_to(np.array([-1.0, -1.0, 4.0, -1.0, -1.0]) / (h * h), cols.shape)
mask = cols >= 0
row_map = np.zeros(U + 1, dtype=np.int32)
row_map[1:] = np.cumsum(mask.sum(axis=1))
entries = cols[mask].astype(np.int32)
values = np.ascontiguousarray(vals[mask])
nnz = len(values)
rng = np.random.default_rng(12345)
b = rng.uniform(-1.0, 1.0, U)


def timed(handle, rule):
    handle.set_rhs(b)
    p = isa.default_params(rule)
    p.max_iterations, p.fixed_iterations, p.use_true_solution, p.callback_every, p.sync_every = ITERS, 1, 0, 0, 500
    handle.solve(p)
    t0 = time.perf_counter()
    r = handle.solve(p)
    dt = time.perf_counter() - t0
    assert r.iterations == ITERS
    return ITERS / dt


A = isa.CrsMatrix(row_map, entries, values)
S = isa.MatrixFreeSystem(N, N, 1.0, 2.0, 1.0, 2.0)
# bytes per iteration of the CSR loop: SpMV = 12 B per non-zero + row_map + x gathered (>= once) + y written; xpay 3 words; update 6 words
csr_bytes = 12.0 * nnz + 4.0 * U + 8.0 * U * (2 + 3 + 6)
for name, rule in (("REL_2NORM", _capi.RULE_REL_2NORM), ("MSG", _capi.RULE_MSG_MAXNORM)):
    ic = timed(A._handle, rule)
    im = timed(S._handle, rule)
    print(f"N={N} U={U} nnz={nnz} rule={name}: CSR path {ic:8.1f} it/s ({csr_bytes * ic / 1e9:6.0f} GB/s of {csr_bytes / 1e6:.0f} MB compulsory per iteration)"
          f" | stencil path {im:8.1f} it/s | ratio {im / ic:.2f}", flush=True)
