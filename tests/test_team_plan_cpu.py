"""CPU tests of the team layer's host logic (csrc/team.h through the C ABI, no GPU needed):
  * mi355cg_decompose: row slabs and the 2-D split (BASELINE config 4's "2 x 2": y-cut where the slabs hold equal
    unknowns, every slab cut in x where ITS unknowns halve, on 128-column strip boundaries) tile the L-shaped grid
    exactly once and balance the unknowns;
  * mi355cg_halo_plan: every message has its counterpart on the peer, and -- world_size-4 and -8 `gloo` runs -- exchanging
    exactly those messages gives every rank the neighbour values its 5-point operator needs: each rank holds ONLY its own
    cells (everything else NaN), receives its plan's messages over torch.distributed, applies the operator to its cells
    and must reproduce the CPU oracle's apply bit for bit.  A missing, misplaced or mis-sized message leaves a NaN or a
    wrong value in somebody's result.
The kernels that consume these halos are tested on the GPU (tests/test_gpu_team.py)."""
import os
import sys
import tempfile

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _unknowns(n, box):
    y_lo, y_hi, x_lo, x_hi = box
    half = n // 2
    cnt = 0
    for y in range(y_lo, y_hi + 1):
        lo = half + 1 if y <= half else 1
        cnt += max(0, min(x_hi, n) - max(x_lo, lo))
    return cnt


@pytest.mark.parametrize("n,world,decomp", [(4096, 4, 1), (4096, 8, 1), (16384, 4, 1), (32768, 8, 1), (32768, 8, 0), (258, 4, 1), (258, 8, 1),
                                            (514, 6, 1), (64, 4, 1), (1026, 3, 1)])
def test_decomposition_tiles_the_grid_and_balances(n, world, decomp):
    from iterative_solvers_amd.distributed import decompose
    boxes = decompose(n, world, decomp)
    assert len(boxes) == world
    half, U = n // 2, (n // 2 - 1) * (3 * n // 2 - 1)
    assert sum(_unknowns(n, b) for b in boxes) == U
    # disjoint: row ranges of different y-slabs do not overlap; inside a slab the x-ranges abut
    slabs = {}
    for (y_lo, y_hi, x_lo, x_hi) in boxes:
        assert 1 <= y_lo <= y_hi <= n - 1 and 0 <= x_lo < x_hi <= n
        assert x_lo % 128 == 0 and (x_hi % 128 == 0 or x_hi == n)            # x-cuts on strip boundaries
        slabs.setdefault((y_lo, y_hi), []).append((x_lo, x_hi))
    rows = sorted(slabs)
    assert rows[0][0] == 1 and rows[-1][1] == n - 1
    assert all(rows[k][1] + 1 == rows[k + 1][0] for k in range(len(rows) - 1))
    for xs in slabs.values():
        xs.sort()
        assert xs[0][0] == 0 and xs[-1][1] == n and all(xs[k][1] == xs[k + 1][0] for k in range(len(xs) - 1))
    if n >= 4096:
        cnt = [_unknowns(n, b) for b in boxes]
        # the slowest part sets the pace: largest part vs the mean.  x-cuts fall on 128-column strips, so a 4096 grid
        # (32 strips) balances to ~4 %, BASELINE config 4's 16384 grid to ~1 %, row slabs to 0.01 %
        assert max(cnt) / (sum(cnt) / world) < (1.05 if n == 4096 else 1.02), cnt
    if (n, world, decomp) == (4096, 4, 1):                                   # y-cut N/8 rows into the upper block,
        assert boxes[0][1] == 5 * n // 8                                     # lower slab cut near 0.7 N, upper slab at N / 2
        assert abs(boxes[0][3] - 0.7 * n) <= 128 and boxes[2][3] == n // 2


@pytest.mark.parametrize("n,world,decomp", [(258, 4, 1), (514, 8, 1), (1026, 4, 1), (130, 3, 0), (4096, 8, 1), (66, 16, 0)])
def test_halo_plan_is_symmetric(n, world, decomp):
    from iterative_solvers_amd.distributed import decompose, halo_plan
    boxes = decompose(n, world, decomp)
    plans = [halo_plan(n, world, decomp, r) for r in range(world)]
    for r, plan in enumerate(plans):
        for m in plan:
            twin = [t for t in plans[m["peer"]] if t["id"] == m["id"]]
            assert len(twin) == 1
            t = twin[0]
            assert t["peer"] == r and t["send"] == 1 - m["send"] and t["count"] == m["count"]
            assert (t["kind"], t["y0"], t["y1"], t["x0"], t["x1"]) == (m["kind"], m["y0"], m["y1"], m["x0"], m["x1"])
            y_lo, y_hi, x_lo, x_hi = boxes[r if m["send"] else m["peer"]]      # the cells belong to the sender
            assert y_lo <= m["y0"] and m["y1"] <= y_hi
            if m["kind"] == 1:
                assert x_lo <= m["x0"] < x_hi and m["count"] == m["y1"] - m["y0"] + 1
    if world > 1:
        assert all(len(p) > 0 for p in plans)


def _worker(rank, world, port, n, decomp, outdir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from iterative_solvers_amd.distributed import decompose, halo_plan
        from oracle.oracle import OracleGrid
        og = OracleGrid(n, n)
        A, xk, yk = og.coeffs
        half = n // 2
        rng = np.random.default_rng(777)
        v = rng.uniform(-1, 1, og.size)                                       # the same global vector on every rank
        y_lo, y_hi, x_lo, x_hi = decompose(n, world, decomp)[rank]
        # node grid (y, x), rows padded to the storage pitch (a row message covers stored columns, pads included): this rank
        # knows its OWN cells and the (zero) boundary nodes and pads; everything else is NaN
        W = (n + 1 + 31) // 32 * 32
        V = np.full((n + 1, W), np.nan)
        interior = np.zeros((n + 1, W), dtype=bool)
        idx = np.full((n + 1, W), -1, dtype=np.int64)
        for y in range(1, n):
            lo = half + 1 if y <= half else 1
            interior[y, lo:n] = True
            base = (half - 1) * (y - 1) - (half + 1) if y <= half else (half - 1) * half + (y - half - 1) * (n - 1) - 1
            idx[y, lo:n] = base + np.arange(lo, n)
        V[~interior] = 0.0
        own = np.zeros_like(interior)
        own[y_lo:y_hi + 1, x_lo:min(x_hi, n)] = True
        own &= interior
        V[own] = v[idx[own]]
        reqs, post = [], []
        for m in sorted(halo_plan(n, world, decomp, rank), key=lambda m: (m["peer"], m["id"])):
            sl = (m["y0"], slice(m["x0"], m["x1"])) if m["kind"] == 0 else (slice(m["y0"], m["y1"] + 1), m["x0"])
            if m["send"]:
                buf = torch.from_numpy(np.ascontiguousarray(V[sl]).copy())
                assert buf.numel() == m["count"] and not torch.isnan(buf).any()       # a rank only sends what it has
                reqs.append(dist.isend(buf, m["peer"], tag=m["id"]))
            else:
                buf = torch.empty(m["count"], dtype=torch.float64)
                reqs.append(dist.irecv(buf, m["peer"], tag=m["id"]))
                post.append((sl, buf))
        for r in reqs:
            r.wait()
        for sl, buf in post:
            V[sl] = buf.numpy()
        # the reference's operation order: ((((A c + xk L) + xk R) + yk T) + yk B)   matrix_free_system.cpp:217-261
        C = V[1:n, 1:n]
        Y = A * C
        Y = Y + xk * V[1:n, 0:n - 1]
        Y = Y + xk * V[1:n, 2:n + 1]
        Y = Y + yk * V[2:n + 1, 1:n]
        Y = Y + yk * V[0:n - 1, 1:n]
        Yfull = np.full((n + 1, W), np.nan)
        Yfull[1:n, 1:n] = Y
        mine = Yfull[own]
        np.savez(os.path.join(outdir, f"r{rank}.npz"), y=mine, idx=idx[own], ref=og.apply(v)[idx[own]])
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n,decomp", [(4, 258, 1), (4, 514, 1), (8, 514, 1), (4, 130, 0), (2, 258, 1)])
def test_halo_plan_delivers_what_the_operator_needs_over_gloo(world, n, decomp):
    port = 29100 + (os.getpid() * 11 + world * 37 + n + decomp) % 1800
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_worker, args=(world, port, n, decomp, d), nprocs=world, join=True)
        seen = []
        for r in range(world):
            with np.load(os.path.join(d, f"r{r}.npz")) as f:
                assert not np.isnan(f["y"]).any(), f"rank {r}: a neighbour value never arrived"
                assert np.array_equal(f["y"], f["ref"]), f"rank {r}: operator on the exchanged halo differs from the oracle"
                seen.append(f["idx"])
        allidx = np.sort(np.concatenate(seen))
        U = (n // 2 - 1) * (3 * n // 2 - 1)
        assert np.array_equal(allidx, np.arange(U))                          # every unknown on exactly one rank


def _plan(n, world, decomp, rank, which):
    import ctypes as C
    from iterative_solvers_amd import _capi
    lib = _capi.load()
    np_, grid, nitems = C.c_int(), C.c_int(), C.c_int()
    panels = (C.c_int * 64)()
    cls = (C.c_int * 10)()
    _capi.check(lib.mi355cg_debug_plan(n, world, decomp, rank, which, C.byref(np_), panels, C.byref(grid), C.byref(nitems), cls))
    P = [tuple(panels[8 * k:8 * k + 8]) for k in range(np_.value)]
    return P, grid.value, nitems.value, list(cls)


def _cover(panels):
    """{(y, strip): count} of a panel list, and per-item consistency of the chunking."""
    cov = {}
    items = 0
    for (y0, y1, s0, ns, ty, nch, item0, gc) in panels:
        assert item0 == items and ty >= 1 and nch == -(-(y1 - y0 + 1) // ty)
        items += ns * nch
        for y in range(y0, y1 + 1):
            for s in range(s0, s0 + ns):
                cov[(y, s)] = cov.get((y, s), 0) + 1
    return cov, items


@pytest.mark.parametrize("n,world,decomp", [(64, 1, 0), (258, 1, 0), (258, 4, 1), (514, 8, 1), (130, 3, 0), (1026, 4, 1), (4096, 1, 0), (4096, 4, 1), (66, 16, 0)])
def test_launch_plans_tile_every_part_exactly_once(n, world, decomp):
    """Work items of the whole-part launch cover each (row, 128-column strip) of the part once; interior + edge launches are a
    partition of the same set; XCD class boundaries are monotone and end at the item count."""
    from iterative_solvers_amd.distributed import decompose
    half = n // 2
    ns_all = (n - 1) // 128 + 1
    s0b = (half + 1) // 128
    seen_rows = set()
    for rank, (y_lo, y_hi, x_lo, x_hi) in enumerate(decompose(n, world, decomp)):
        want = set()
        s_lo, s_hi = x_lo // 128, (ns_all if x_hi >= n else x_hi // 128)
        for y in range(y_lo, y_hi + 1):
            for s in range(max(s_lo, s0b) if y <= half else s_lo, s_hi):
                want.add((y, s))
        whole, grid, nitems, cls = _plan(n, world, decomp, rank, 0)
        cov, items = _cover(whole)
        assert items == nitems and set(cov) == want and set(cov.values()) <= {1}
        assert 1 <= grid <= 512
        if cls[0] == 8:
            assert cls[1] == 0 and cls[9] == nitems and all(cls[k] <= cls[k + 1] for k in range(1, 9)) and grid % 8 == 0
        inner, _, ni, _ = _plan(n, world, decomp, rank, 1)
        edge, _, ne, _ = _plan(n, world, decomp, rank, 2)
        ci, _ = _cover(inner)
        ce, _ = _cover(edge)
        assert set(ci) | set(ce) == want and not (set(ci) & set(ce))
        assert set(ci.values()) <= {1} and set(ce.values()) <= {1}
        # everything that touches a ghost row or a ghost column is an edge item
        for (y, s) in ci:
            assert y_lo < y < y_hi
        for (y0, y1, s0, ns, ty, nch, item0, gc) in edge:
            assert gc in (0, 1, 2, 3)
        seen_rows |= {y for (y, s) in want}
    assert seen_rows == set(range(1, n))
