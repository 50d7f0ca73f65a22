"""iterative_solvers_amd -- MI355X-native matrix-free CG for the 2-D Dirichlet Poisson problem
on the reference's L-shaped grid.  HIP kernels + C ABI in csrc/ (libmi355cg.so); this package
is the host-side mirror of the reference's operator / solver interface."""
from ._capi import (F64, F32_MIXED, RULE_MSG_MAXNORM, RULE_REL_2NORM, Mi355cgError, lib_path, load)
from .solver import (CrsMatrix, DirichletSolver, GridSystem, MatrixFreeSolver, MatrixFreeSystem, MSGSolver,
                     SolverResults, StopCriterion, default_params)

__all__ = ["CrsMatrix", "DirichletSolver", "GridSystem", "MatrixFreeSolver", "MatrixFreeSystem", "MSGSolver",
           "SolverResults", "StopCriterion", "default_params", "Mi355cgError", "lib_path", "load",
           "F64", "F32_MIXED", "RULE_MSG_MAXNORM", "RULE_REL_2NORM"]
