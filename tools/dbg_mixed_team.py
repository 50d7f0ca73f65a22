"""Debug helper: F32_MIXED on a LOCAL team against the single-GPU mixed solve (outer steps, inner iterations)."""
import sys
import numpy as np
sys.path.insert(0, ".")
import iterative_solvers_amd as isa
from iterative_solvers_amd.distributed import Team

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
p = isa.default_params(1)
p.eps_rel, p.max_iterations = 1e-8, 10 ** 6
s1 = isa.MatrixFreeSystem(n, n, 1.0, 2.0, 1.0, 2.0, dtype=isa.F32_MIXED)
cb1 = []
r1 = s1._handle.solve(p, callback=lambda *a: cb1.append(a))
print("single", r1.iterations, r1.converged, r1.refine_outer, [(c[0], c[2]) for c in cb1], flush=True)
for world in [int(w) for w in (sys.argv[2] if len(sys.argv) > 2 else "1,2,3").split(",")]:
    t = Team.local(n, world, 0)
    t.set_dtype(isa.F32_MIXED)
    cbs = []
    rt = t.solve(p, callback=lambda *a: cbs.append(a))
    x = t.vector(0)
    print("team", world, rt.iterations, rt.converged, rt.refine_outer, [(c[0], c[2]) for c in cbs], "x equal:", np.array_equal(x, s1._handle.solution()), flush=True)
    t.close()
import time
for world in (5,):
    t = Team.local(130, world, 0)
    t.set_dtype(isa.F32_MIXED)
    t0 = time.time(); rt = t.solve(p); print("local world", world, "N=130 mixed:", rt.iterations, rt.refine_outer, f"{time.time() - t0:.3f} s", flush=True)
    t.set_dtype(isa.F64)
    t0 = time.time(); rt = t.solve(p); print("local world", world, "N=130 fp64:", rt.iterations, f"{time.time() - t0:.3f} s", flush=True)
    t.close()
