#!/usr/bin/env python3
"""In-process sweep of the launch-geometry knobs (env vars read at mi355cg_create): item height, waves, rows in flight.
Usage: python tools/tune.py N iters [f32] -- "K=V K=V" "K=V" ...      (each quoted group is one configuration; "" = defaults)
Prints iterations/s and, per launch kind, the mean in-loop duration (HIP events) and the bandwidth its compulsory
bytes correspond to (stencil 3 words; update 3 words on three iterations of four and 8 on the fourth = 4.25 on average;
MI355CG_XSTEPS=2: 3 / 6 alternating = 4.5)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import iterative_solvers_amd as isa
from iterative_solvers_amd import _capi

args = sys.argv[1:]
cfgs = args[args.index("--") + 1:] if "--" in args else [""]
pos = args[:args.index("--")] if "--" in args else args
N = int(pos[0]) if pos else 4096
ITERS = int(pos[1]) if len(pos) > 1 else 400
F32 = len(pos) > 2 and pos[2] == "f32"
U = (N // 2 - 1) * (3 * N // 2 - 1)
WB = 4 if F32 else 8
KNOBS = ("MI355CG_DYN_ROWS", "MI355CG_XSTEPS", "MI355CG_ITEM_ROWS", "MI355CG_WAVES", "MI355CG_BLOCKS", "MI355CG_DEPTH", "MI355CG_GRAPH", "MI355CG_XCD_CLASSES")


def measure(cfg):
    for k in KNOBS:
        os.environ.pop(k, None)
    for kv in cfg.split():
        k, v = kv.split("=")
        os.environ[k] = v
    s = isa.MatrixFreeSystem(N, N, 1.0, 2.0, 1.0, 2.0, dtype=isa.F32_MIXED if F32 else isa.F64)
    h = s._handle
    p = isa.default_params(_capi.RULE_REL_2NORM)
    p.max_iterations, p.fixed_iterations, p.use_true_solution, p.callback_every, p.sync_every = ITERS, 1, 0, 0, 500
    h.solve(p)
    t0 = time.perf_counter(); h.solve(p); dt = time.perf_counter() - t0
    h.set_profiling(True); h.solve(p); h.set_profiling(False)
    ts, tu = h.kernel_time(0)[0], h.kernel_time(1)[0]
    lay = h.layout()
    h.close()
    return ITERS / dt, ts, tu, lay


for rep in range(2):
    for cfg in cfgs:
        its, ts, tu, lay = measure(cfg)
        wu = 4.5 if "MI355CG_XSTEPS=2" in cfg else 4.25            # words per unknown of the average update launch
        print(f"N={N} {'f32' if F32 else 'f64'} [{cfg:40s}] {its:8.1f} it/s = {(3+wu)*WB*U*its/1e9:6.0f} GB/s moved | stencil {ts*1e3:8.1f} us ({3*WB*U/ts/1e6:5.0f} GB/s) "
              f"update {tu*1e3:8.1f} us ({wu*WB*U/tu/1e6:5.0f} GB/s) grid={lay['grid_stencil']} ty={lay['rows_per_item']}", flush=True)
