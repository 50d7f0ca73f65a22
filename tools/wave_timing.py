#!/usr/bin/env python3
"""Per-wave timeline of the last stencil / update launch (diagnostic build, -DMI355CG_WAVE_TIMING).
Usage (GPU box): MI355CG_LIB=iterative_solvers_amd/libmi355cg_wt.so python tools/wave_timing.py [N] [iters]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
ITERS = int(sys.argv[2]) if len(sys.argv) > 2 else 51
OUT = "/tmp/wave_timing.bin"
os.environ["MI355CG_WAVE_TIMING_OUT"] = OUT
import iterative_solvers_amd as isa
from iterative_solvers_amd import _capi

s = isa.MatrixFreeSystem(N, N, 1.0, 2.0, 1.0, 2.0)
p = isa.default_params(_capi.RULE_REL_2NORM)
p.max_iterations, p.fixed_iterations, p.use_true_solution, p.callback_every, p.sync_every = ITERS, 1, 0, 0, 500
s._handle.solve(p)
s._handle.solve(p)
d = np.fromfile(OUT, dtype=np.uint64).reshape(2, -1, 6).astype(np.int64)
for k, name in enumerate(("stencil", "update")):
    w = d[k]
    w = w[w[:, 5] > 0]
    t0 = w[:, 0].min()
    start, mid, end = (w[:, 0] - t0) / 100.0, (w[:, 4] - t0) / 100.0, (w[:, 5] - t0) / 100.0     # microseconds
    q = lambda v: " ".join(f"{np.percentile(v, x):7.1f}" for x in (0, 5, 25, 50, 75, 95, 100))
    print(f"{name}: {len(w)} waves, launch span {end.max():.1f} us")
    print(f"   percentiles          min      5     25     50     75     95    max")
    print(f"   wave start      {q(start)}")
    print(f"   prologue length {q(mid - start)}")
    if w[:, 1].max() > 0:
        print(f"     state loaded  {q((w[:, 1] - w[:, 0]) / 100.0)}")
        print(f"     reduce+decide {q((w[:, 2] - w[:, 1]) / 100.0)}")
        print(f"     rest          {q((w[:, 4] - w[:, 2]) / 100.0)}")
    print(f"   main loop length{q(end - mid)}")
    print(f"   wave end        {q(end)}")
    hist, edges = np.histogram(end, bins=12)
    print("   end-time histogram:", " ".join(f"{e:.0f}:{h}" for h, e in zip(hist, edges[:-1])))
    # where are the slow waves?  one item per wave at N = 4096: wave index = item index
    idx = np.nonzero(d[k][:, 5] > 0)[0]
    dur = end - mid
    nb = 16
    edges = np.linspace(0, idx.max() + 1, nb + 1)
    print("   main loop (us) by item-index bin :", " ".join(f"{dur[(idx >= edges[i]) & (idx < edges[i + 1])].mean():5.1f}" for i in range(nb)))
    print("   main loop (us) by XCD (block % 8):", " ".join(f"{dur[(idx // 4) % 8 == x].mean():5.1f}" for x in range(8)))
    print("   main loop (us) by wave in block  :", " ".join(f"{dur[idx % 4 == x].mean():5.1f}" for x in range(4)))
    print("   main loop (us) by item % 32      :", " ".join(f"{dur[idx % 32 == x].mean():5.1f}" for x in range(32)))

# Are the slow waves the same ones from launch to launch?  Correlate per-workgroup main-loop durations between the two kernels
# of this dump and with a second, independent solve (another launch of each kernel).
def per_block(dd):
    out = []
    for k in range(2):
        w = dd[k]
        ok = w[:, 5] > 0
        dur = np.where(ok, (w[:, 5] - w[:, 4]) / 100.0, np.nan)
        nb = len(dur) // 4
        out.append(np.nanmean(dur[:nb * 4].reshape(nb, 4), axis=1))
    return out


first = per_block(d)
s._handle.solve(p)
d2 = np.fromfile(OUT, dtype=np.uint64).reshape(2, -1, 6).astype(np.int64)
second = per_block(d2)
nb = int(np.sum(~np.isnan(first[0])))
for name, a, b in (("stencil launch vs update launch (same solve)", first[0][:nb], first[1][:nb]),
                   ("stencil launch vs stencil launch of another solve", first[0][:nb], second[0][:nb]),
                   ("update launch vs update launch of another solve", first[1][:nb], second[1][:nb])):
    m = ~np.isnan(a) & ~np.isnan(b)
    # remove the XCD-parity component first: it is known and handled by the class split
    par = np.arange(len(a)) % 8
    a2, b2 = a.copy(), b.copy()
    for x in range(8):
        a2[par == x] -= np.nanmean(a[par == x]); b2[par == x] -= np.nanmean(b[par == x])
    print(f"per-workgroup duration correlation, {name}: raw {np.corrcoef(a[m], b[m])[0, 1]:.3f}, XCD means removed {np.corrcoef(a2[m], b2[m])[0, 1]:.3f} "
          f"(std {np.nanstd(a2):.2f} / {np.nanstd(b2):.2f} us)")


# Where on the grid are the slow items?  (one item per wave: N <= 4096 with the default plan)
def item_map(dd, k, reverse):
    import ctypes as C
    lib = _capi.load()
    np_, grid, nitems = C.c_int(), C.c_int(), C.c_int()
    panels = (C.c_int * 64)()
    cls = (C.c_int * 10)()
    _capi.check(lib.mi355cg_debug_plan(N, 1, 0, 0, 0, C.byref(np_), panels, C.byref(grid), C.byref(nitems), cls))
    P = [tuple(panels[8 * i:8 * i + 8]) for i in range(np_.value)]
    if cls[0] != 8 or nitems.value > grid.value * 4:
        return
    w = dd[k]
    dur = np.where(w[:, 5] > 0, (w[:, 5] - w[:, 4]) / 100.0, np.nan)
    print(f"--- main loop (us) of launch kind {k} by grid position (rows: chunk rows bottom -> top; columns: 128-column strips), N = {N}")
    maps = {i: np.full((p[5], p[3]), np.nan) for i, p in enumerate(P)}
    for wave in range(min(len(dur), grid.value * 4)):
        b, wl = divmod(wave, 4)
        c = b % 8
        begin, end = cls[1 + c], cls[2 + c]
        idx = begin + (b // 8) * 4 + wl
        if idx >= end or np.isnan(dur[wave]):
            continue
        item = end - 1 - (idx - begin) if reverse else idx
        pi = max(i for i, p in enumerate(P) if item >= p[6])
        local = item - P[pi][6]
        maps[pi][local // P[pi][3], local % P[pi][3]] = dur[wave]
    for i, p in enumerate(P):
        print(f"panel {i}: rows {p[0]}..{p[1]}, strips {p[2]}..{p[2] + p[3] - 1}, {p[4]} rows per item; column means:",
              " ".join(f"{v:4.0f}" for v in np.nanmean(maps[i], axis=0)))
        for r in range(maps[i].shape[0]):
            print(f"  chunk {r:3d} (mean {np.nanmean(maps[i][r]):5.1f}):", " ".join(f"{v:4.0f}" for v in maps[i][r]))


item_map(d2, 0, False)
item_map(d2, 1, True)
