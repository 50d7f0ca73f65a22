import os, sys, gc
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import iterative_solvers_amd as isa
from iterative_solvers_amd import _capi

def small(tag):
    s = isa.MatrixFreeSystem(6, 6, 1.0, 2.0, 1.0, 2.0)
    p = isa.default_params(_capi.RULE_REL_2NORM); p.max_iterations = 3000; p.eps_rel = 1e-10
    res = s._handle.solve(p)
    print(tag, "it", res.iterations, "conv", res.converged, "rn", res.r_norm2, "r0", res.initial_r_norm2,
          "|b|", np.linalg.norm(s._handle.rhs()), "layout", s._handle.layout(), flush=True)
    s._handle.close()

small("fresh")
mode = sys.argv[1] if len(sys.argv) > 1 else "big"
if mode == "big":
    s = isa.GridSystem(2048, 2048, 1.0, 2.0, 1.0, 2.0)
    sol = isa.MatrixFreeSolver(s, s.get_rhs(), 1e-8, 10 ** 6)
    sol.solve()
    print("big done", sol.getIterations(), flush=True)
    m = isa.MSGSolver(s, s.get_rhs(), 1e-8, 10 ** 6)
    m.solve(s.get_true_solution_vector())
    print("msg done", m.getIterations(), flush=True)
small("after big (big alive)")
del s, sol, m
gc.collect()
small("after big freed")
small("again")
