#!/usr/bin/env python3
"""Run-to-run determinism soak: the same converged solve many times, every result compared bit for bit with the first
(iteration count, norms, double-double checksums of x and r on the device).  Usage (GPU box): python tools/soak.py [N] [runs] [f32]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import iterative_solvers_amd as isa
from iterative_solvers_amd import _capi

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
RUNS = int(sys.argv[2]) if len(sys.argv) > 2 else 20
F32 = len(sys.argv) > 3 and sys.argv[3] == "f32"
lib = _capi.load()


def cs(h, which):
    o = (C.c_double * 2)()
    _capi.check(lib.mi355cg_checksum(h._h, which, o))
    return o[0], o[1]


bad = 0
for rule in ((_capi.RULE_REL_2NORM,) if F32 else (_capi.RULE_REL_2NORM, _capi.RULE_MSG_MAXNORM)):
    first = None
    for k in range(RUNS):
        s = isa.MatrixFreeSystem(N, N, 1.0, 2.0, 1.0, 2.0, dtype=isa.F32_MIXED if F32 else isa.F64) if k % 5 == 0 else s   # a fresh context every 5th run
        p = isa.default_params(rule)
        p.max_iterations, p.eps_rel, p.eps_precision, p.eps_residual, p.eps_exact_error = 10 ** 6, 1e-8, 1e-8, 1e-8, -1.0
        r = s._handle.solve(p)
        got = (r.iterations, r.stop_reason, r.r_norm2, r.final_residual_norm, r.final_precision) + (() if F32 else (cs(s._handle, 0), cs(s._handle, 1)))
        if F32:
            import numpy as np
            got = got + (float(np.abs(s._handle.solution()).sum()),)
        if first is None:
            first = got
        same = got == first
        bad += 0 if same else 1
        print(f"rule {rule} run {k}: {'same' if same else 'DIFFERENT'} {got[:3]}", flush=True)
print(f"{bad} runs differ")
sys.exit(1 if bad else 0)
