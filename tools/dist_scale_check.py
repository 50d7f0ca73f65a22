#!/usr/bin/env python3
"""One-GPU rehearsal of the multi-GPU bench geometry: `world` slabs of an N x N grid as separate processes sharing the GPU
(gloo staging), fixed iterations, compared bit for bit with the single-context solve.  Usage: dist_scale_check.py N world iters"""
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def worker(rank, world, port, n, iters, halo, outdir):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import iterative_solvers_amd as isa
    from iterative_solvers_amd.distributed import DistributedCG, SlabEngine, slab_rows
    torch.cuda.set_device(0)
    y_lo, y_hi = slab_rows(n, world, rank)
    eng = SlabEngine(n, y_lo, y_hi, device=0)
    p = isa.default_params(isa.RULE_REL_2NORM)
    p.max_iterations, p.fixed_iterations, p.sync_every = iters, 1, 50
    res = DistributedCG(eng, halo=halo).solve(p)
    np.save(os.path.join(outdir, f"x{rank}.npy"), eng.solution())
    if rank == 0:
        print(f"world {world} halo {halo}: rows {[slab_rows(n, world, r) for r in range(world)]} iterations {res.iterations} r_norm2 {res.r_norm2!r}", flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    import torch.multiprocessing as mp
    n, world, iters = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    import iterative_solvers_amd as isa
    s = isa.MatrixFreeSystem(n, n, 1.0, 2.0, 1.0, 2.0)
    p = isa.default_params(isa.RULE_REL_2NORM)
    p.max_iterations, p.fixed_iterations, p.sync_every = iters, 1, 50
    res = s._handle.solve(p)
    x1 = s._handle.solution()
    print(f"single context: N {n} unknowns {s.size()} iterations {res.iterations} r_norm2 {res.r_norm2!r}", flush=True)
    s._handle.close()
    for halo in ("gather", "p2p"):
        with tempfile.TemporaryDirectory() as d:
            mp.spawn(worker, args=(world, 29650 + world, n, iters, halo, d), nprocs=world, join=True)
            x = np.concatenate([np.load(os.path.join(d, f"x{r}.npy")) for r in range(world)])
        print(f"  {halo}: bit-identical to the single context: {np.array_equal(x, x1)}  max |dx| {np.abs(x - x1).max():.3e}", flush=True)
