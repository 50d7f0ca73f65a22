#!/usr/bin/env python3
"""In-process sweep of the launch-geometry knobs (env vars read at mi355cg_create).
Usage: python tools/tune.py [N] [iters]"""
import itertools
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import iterative_solvers_amd as isa
from iterative_solvers_amd import _capi

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
ITERS = int(sys.argv[2]) if len(sys.argv) > 2 else 400
U = (N // 2 - 1) * (3 * N // 2 - 1)


def measure(env):
    for k, v in env.items():
        os.environ[k] = str(v)
    s = isa.MatrixFreeSystem(N, N, 1.0, 2.0, 1.0, 2.0)
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


def run(env):
    its, ts, tu, lay = measure(env)
    xf = os.environ.get('MI355CG_XFUSE', '1') != '0'
    ws, wu = (48, 24) if xf else (32, 48)
    print(f"{env} its/s={its:8.1f} GB/s(88B)={88*U*its/1e9:7.1f} stencil={ts*1e3:7.1f}us ({ws*U/ts/1e6:6.0f} GB/s) "
          f"update={tu*1e3:7.1f}us ({wu*U/tu/1e6:6.0f} GB/s) grid={lay['grid_stencil']}/{lay['grid_update']} ty={lay['rows_per_item']}", flush=True)


def ab(configs, rounds=5):
    """Interleaved rounds in one process (the boxes drift by several % within seconds): median and best per config."""
    res = {i: [] for i in range(len(configs))}
    for _ in range(rounds):
        for i, env in enumerate(configs):
            res[i].append(measure(env)[0])
    for i, env in enumerate(configs):
        v = sorted(res[i])
        print(f"{env} median {v[len(v)//2]:8.1f} it/s   best {v[-1]:8.1f}   worst {v[0]:8.1f}", flush=True)


if __name__ == "__main__":
    ab([{"MI355CG_MAX_ROWS": 24}, {"MI355CG_MAX_ROWS": 16}, {"MI355CG_MAX_ROWS": 32}, {"MI355CG_MAX_ROWS": 48}, {"MI355CG_MAX_ROWS": 512}], rounds=3)
