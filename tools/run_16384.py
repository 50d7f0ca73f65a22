import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import importlib
import iterative_solvers_amd as isa
import config_runs
config_runs.run("4': 16384 fp64 rel2 1e-8 on ONE GPU", 16384, isa.F64, isa.RULE_REL_2NORM, eps_rel=1e-8, max_iterations=10 ** 6)
