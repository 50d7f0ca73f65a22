"""Debug helper: two rank processes on ONE GPU (nccl stand-in), each holding half of a large grid; where does the time go?
   python tools/dbg_big_parts.py N"""
import faulthandler
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def worker(rank, world, port, n):
    faulthandler.dump_traceback_later(90, repeat=True, file=sys.stderr)
    from tests import nccl_shim
    os.environ["MI355CG_RCCL_LIB"] = nccl_shim.build()
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    os.environ.setdefault("MI355CG_TEAM_TIMEOUT_MS", "8000")
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import iterative_solvers_amd as isa
    from iterative_solvers_amd import _capi
    from iterative_solvers_amd.distributed import Team
    torch.cuda.set_device(0)
    t0 = time.time()

    def say(what):
        print(f"[rank {rank} +{time.time() - t0:6.1f} s] {what}", flush=True)
    say("creating the team")
    t = Team.rccl(n, 0, device=0)
    say(f"team: {t.describe()}")
    p = isa.default_params(_capi.RULE_REL_2NORM)
    p.max_iterations, p.fixed_iterations, p.use_true_solution, p.callback_every, p.sync_every = 5, 1, 0, 0, 500
    r = t.solve(p)
    say(f"5 iterations: {r.solve_seconds:.3f} s, r_norm2 {r.r_norm2}")
    p.max_iterations = 20
    r = t.solve(p)
    say(f"20 iterations: {r.solve_seconds:.3f} s ({1e3 * r.loop_seconds / 20:.3f} ms per iteration)")
    t.close()
    say("closed")
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    import torch.multiprocessing as mp
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
    mp.spawn(worker, args=(2, 29611, n), nprocs=2, join=True)
