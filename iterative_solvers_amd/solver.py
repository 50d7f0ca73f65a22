"""Host-side mirror of the reference's operator / solver interface for the hot path, over the
C ABI (include/mi355cg.h).  Same class and method names, argument meaning and error behaviour
as the reference's C++ classes, so the parity tests read like the reference's own usage:

  MatrixFreeSystem / MatrixFreeSolver   solver/matrix_free_system.hpp:12-127
  GridSystem                            solver/grid_system.h:16-88
  Solver / MSGSolver / StopCriterion    solver/solver.hpp:17-66, solver/msg_solver.hpp:9-120
  DirichletSolver / SolverResults       solver/dirichlet_solver.hpp:11-24,79-184

All compute runs in libmi355cg.so on the GPU; nothing here falls back to the CPU.
"""
from __future__ import annotations

import ctypes as C
import enum
import sys
from dataclasses import dataclass, field
from typing import Callable, Optional

import numpy as np

from . import _capi

DBL_MAX = sys.float_info.max


class StopCriterion(enum.IntEnum):          # solver/msg_solver.hpp:9-15
    ITERATIONS = 0
    PRECISION = 1
    RESIDUAL = 2
    EXACT_ERROR = 3
    INTERRUPTED = 4


_STOP_TEXT = {                              # solver/msg_solver.hpp:85-100
    StopCriterion.ITERATIONS: "Достигнуто максимальное число итераций",
    StopCriterion.PRECISION: "Достигнута требуемая точность по норме разности xn и xn-1",
    StopCriterion.RESIDUAL: "Достигнута требуемая точность по норме невязки",
    StopCriterion.EXACT_ERROR: "Достигнута требуемая точность по норме разности с истинным решением",
    StopCriterion.INTERRUPTED: "Прервано пользователем",
}


class _Handle:
    """Owns one mi355cg context (one GPU)."""

    def __init__(self, n, m, a, b, c, d, dtype=_capi.F64, device=0):
        self._lib = _capi.load()
        self._h = C.c_void_p()
        rc = self._lib.mi355cg_create(int(n), int(m), float(a), float(b), float(c), float(d),
                                      int(dtype), int(device), C.byref(self._h))
        if rc == _capi.ERR_INVALID:
            raise ValueError(self._lib.mi355cg_last_error().decode())   # std::invalid_argument
        _capi.check(rc)
        self.size = int(self._lib.mi355cg_size(self._h))

    @classmethod
    def from_csr(cls, row_map, entries, values, device=0):
        """Handle for a caller-supplied CSR matrix (mi355cg_create_csr)."""
        self = cls.__new__(cls)
        self._lib = _capi.load()
        self._h = C.c_void_p()
        row_map = np.ascontiguousarray(row_map, dtype=np.int32)
        entries = np.ascontiguousarray(entries, dtype=np.int32)
        values = np.ascontiguousarray(values, dtype=np.float64)
        rc = self._lib.mi355cg_create_csr(len(row_map) - 1, row_map, entries, values, int(device), C.byref(self._h))
        if rc == _capi.ERR_INVALID:
            raise ValueError(self._lib.mi355cg_last_error().decode())
        _capi.check(rc)
        self.size = int(self._lib.mi355cg_size(self._h))
        return self

    def set_true_solution(self, u):
        u = np.ascontiguousarray(u, dtype=np.float64)
        if u.shape != (self.size,):
            raise ValueError(f"true solution has shape {u.shape}, expected ({self.size},)")
        _capi.check(self._lib.mi355cg_set_true_solution(self._h, u))

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            self._lib.mi355cg_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # thin wrappers -------------------------------------------------------------------------
    def _vec_out(self, fn) -> np.ndarray:
        out = np.empty(self.size)
        _capi.check(fn(self._h, out))
        return out

    def rhs(self): return self._vec_out(self._lib.mi355cg_get_rhs)
    def true_solution(self): return self._vec_out(self._lib.mi355cg_get_true_solution)
    def solution(self): return self._vec_out(self._lib.mi355cg_get_solution)
    def recursive_residual(self): return self._vec_out(self._lib.mi355cg_get_recursive_residual)
    def true_residual(self): return self._vec_out(self._lib.mi355cg_get_true_residual)

    def node_coords(self):
        xs, ys = np.empty(self.size), np.empty(self.size)
        _capi.check(self._lib.mi355cg_get_node_coords(self._h, xs, ys))
        return xs, ys

    def set_rhs(self, b):
        b = np.ascontiguousarray(b, dtype=np.float64)
        if b.shape != (self.size,):
            raise ValueError(f"rhs has shape {b.shape}, expected ({self.size},)")
        _capi.check(self._lib.mi355cg_set_rhs(self._h, b))

    def apply(self, x):
        x = np.ascontiguousarray(x, dtype=np.float64)
        if x.shape != (self.size,):
            raise ValueError(f"vector has shape {x.shape}, expected ({self.size},)")
        y = np.empty(self.size)
        _capi.check(self._lib.mi355cg_apply(self._h, x, y))
        return y

    def solve(self, params: _capi.Params, callback=None, stop_flag: Optional[C.c_int] = None) -> _capi.Results:
        res = _capi.Results()
        cb = _capi.ITER_CB(lambda user, it, p, r, e: callback(it, p, r, e)) if callback else _capi.ITER_CB()
        sp = C.cast(C.pointer(stop_flag), C.c_void_p) if stop_flag is not None else None
        rc = self._lib.mi355cg_solve(self._h, C.byref(params), cb, None, sp, C.byref(res))
        if rc == _capi.ERR_INVALID:
            raise ValueError(self._lib.mi355cg_last_error().decode())     # std::invalid_argument
        _capi.check(rc)
        return res

    def setup_on_device(self):
        """Opt-in: regenerate b and u on the GPU (<= 1 ulp from the host values)."""
        _capi.check(self._lib.mi355cg_setup_on_device(self._h))

    def set_profiling(self, on: bool):
        _capi.check(self._lib.mi355cg_set_profiling(self._h, 1 if on else 0))

    def kernel_time(self, kernel: int):
        ms, n = C.c_double(), C.c_longlong()
        _capi.check(self._lib.mi355cg_get_kernel_time(self._h, kernel, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def layout(self):
        L = C.c_longlong()
        v = [C.c_int() for _ in range(5)]
        _capi.check(self._lib.mi355cg_get_layout(self._h, C.byref(L), *[C.byref(i) for i in v]))
        return {"padded_len": L.value, "pitch_bottom": v[0].value, "pitch_upper": v[1].value,
                "grid_stencil": v[2].value, "grid_update": v[3].value, "rows_per_item": v[4].value}


def default_params(rule: int) -> _capi.Params:
    p = _capi.Params()
    _capi.load().mi355cg_default_params(C.byref(p), rule)
    return p


# ---------------------------------------------------------------------------------------------
class MatrixFreeSystem:
    """solver/matrix_free_system.hpp:12-70.  Constructor order is (m, n, a, b, c, d)."""

    def __init__(self, m, n, a, b, c, d, device: int = 0, dtype: int = _capi.F64):
        self.n, self.m = n, m
        self.domain = (a, b, c, d)
        self._handle = _Handle(n, m, a, b, c, d, dtype=dtype, device=device)

    def get_rhs(self) -> np.ndarray: return self._handle.rhs()
    def get_true_solution_vector(self) -> np.ndarray: return self._handle.true_solution()
    def size(self) -> int: return self._handle.size

    def apply(self, x, y=None) -> np.ndarray:
        out = self._handle.apply(x)
        if y is not None:
            y[...] = out
            return y
        return out

    def __mul__(self, x): return self.apply(x)             # operator* (matrix_free_system.hpp:59-63)


class GridSystem(MatrixFreeSystem):
    """solver/grid_system.h:16-88: same geometry; the CSR matrix is never materialised on the
    hot path -- get_matrix() hands back the operator itself."""

    def get_matrix(self): return self
    def get_x_coords(self): return self._handle.node_coords()[0]
    def get_y_coords(self): return self._handle.node_coords()[1]

    def get_node_coordinates(self, solution_index: int):
        if solution_index < 0 or solution_index >= self.size():
            return (0.0, 0.0)                                   # grid_system.cpp:339-341
        xs, ys = self._handle.node_coords()
        return (float(xs[solution_index]), float(ys[solution_index]))


class CrsMatrix:
    """A caller-supplied sparse matrix (KokkosCrsMatrix, solver/solver.hpp:15) as the operator of
    `MSGSolver(a, b, ...)` / `MatrixFreeSolver(a, b, ...)`: the generic path of the abstract `Solver` contract."""

    def __init__(self, row_map, entries, values, device: int = 0):
        self._handle = _Handle.from_csr(row_map, entries, values, device=device)

    def numRows(self): return self._handle.size
    def numCols(self): return self._handle.size
    def size(self): return self._handle.size
    def apply(self, x): return self._handle.apply(x)
    def __mul__(self, x): return self.apply(x)


class MatrixFreeSolver:
    """solver/matrix_free_system.hpp:73-127, MatrixFreeSolver::solve (.cpp:383-482):
    textbook CG, relative 2-norm stop rule."""

    def __init__(self, system: MatrixFreeSystem, b, eps: float = 1e-6, maxIterations: int = 10000,
                 name: str = "Matrix-free solver"):
        self.system, self.b, self.eps, self.maxIterations, self.name = system, b, eps, maxIterations, name
        self.iterations = 0
        self.iteration_callback = None
        self.completion_callback = None
        self.last_results = None

    def setIterationCallback(self, cb): self.iteration_callback = cb
    def setCompletionCallback(self, cb): self.completion_callback = cb
    def getIterations(self): return self.iterations
    def getName(self): return self.name

    def solve(self, true_solution=None, fixed_iterations: bool = False, sync_every: int = 0,
              inner_eps: float = 0.0) -> np.ndarray:
        """inner_eps only matters for a MatrixFreeSystem created with dtype=F32_MIXED (config 3)."""
        h = self.system._handle
        h.set_rhs(self.b)
        if true_solution is not None and len(true_solution) > 0:
            h.set_true_solution(true_solution)                   # matrix_free_system.cpp:451-455 uses the caller's vector
        p = default_params(_capi.RULE_REL_2NORM)
        p.eps_rel, p.max_iterations = self.eps, self.maxIterations
        p.diagnostics = 1 if self.iteration_callback else 0
        p.fixed_iterations = 1 if fixed_iterations else 0
        p.sync_every = sync_every
        p.inner_eps = inner_eps
        res = h.solve(p, self.iteration_callback)
        self.iterations, self.last_results = res.iterations, res
        if self.completion_callback:                              # matrix_free_system.cpp:472-479
            ok = bool(res.converged)
            self.completion_callback(ok, "Converged successfully" if ok else
                                     "Failed to converge within maximum iterations")
        return h.solution()


class MSGSolver:
    """solver/msg_solver.hpp:17-120, MSGSolver::solve (msg_solver.cpp:10-212)."""

    def __init__(self, a: MatrixFreeSystem, b, eps: float = 1e-6, maxIterations: int = 10000):
        self.a, self.b = a, b
        self.eps, self.maxIterations = eps, maxIterations
        self.name = "Метод серединных градиентов"
        self.eps_precision = self.eps_residual = self.eps_exact_error = eps
        self.converged = False
        self.stop_reason = StopCriterion.ITERATIONS
        self.final_residual_norm = self.final_error_norm = self.final_precision = 0.0
        self.iterations = 0
        self.iteration_callback = None
        self.completion_callback = None
        self._stop = C.c_int(0)
        self.last_results = None

    def setPrecisionEps(self, eps): self.eps_precision = eps
    def setResidualEps(self, eps): self.eps_residual = eps
    def setExactErrorEps(self, eps): self.eps_exact_error = eps
    def hasConverged(self): return self.converged
    def getStopReason(self): return self.stop_reason
    def getStopReasonText(self): return _STOP_TEXT.get(self.stop_reason, "Неизвестная причина остановки")
    def requestStop(self): self._stop.value = 1
    def resetStop(self): self._stop.value = 0
    def isStopRequested(self): return bool(self._stop.value)
    def getFinalResidualNorm(self): return self.final_residual_norm
    def getFinalErrorNorm(self): return self.final_error_norm
    def getFinalPrecision(self): return self.final_precision
    def setIterationCallback(self, cb): self.iteration_callback = cb
    def setCompletionCallback(self, cb): self.completion_callback = cb
    def getIterations(self): return self.iterations
    def getName(self): return self.name

    def solve(self, true_solution=None, callback_every: int = 100) -> np.ndarray:
        """true_solution: None / empty = extent 0 (error criterion and norm off)."""
        self.converged = False
        self._stop.value = 0                                     # msg_solver.cpp:12-13
        h = self.a._handle
        h.set_rhs(self.b)
        if true_solution is not None and len(true_solution) > 0:
            h.set_true_solution(true_solution)                   # the error norms use the vector that was passed in (msg_solver.cpp:64-72,132-139)
        p = default_params(_capi.RULE_MSG_MAXNORM)
        p.max_iterations = self.maxIterations
        p.eps_precision, p.eps_residual, p.eps_exact_error = self.eps_precision, self.eps_residual, self.eps_exact_error
        p.use_true_solution = 0 if true_solution is None or len(true_solution) == 0 else 1
        p.callback_every = callback_every
        res = h.solve(p, self.iteration_callback, self._stop)
        self.last_results = res
        self.iterations = res.iterations                         # msg_solver.cpp:187-190
        self.converged = bool(res.converged)
        self.stop_reason = StopCriterion(res.stop_reason)
        self.final_residual_norm, self.final_precision, self.final_error_norm = \
            res.final_residual_norm, res.final_precision, res.final_error_norm
        return h.solution()


@dataclass
class SolverResults:                        # solver/dirichlet_solver.hpp:11-24
    solution: np.ndarray = field(default_factory=lambda: np.empty(0))
    true_solution: np.ndarray = field(default_factory=lambda: np.empty(0))
    residual: np.ndarray = field(default_factory=lambda: np.empty(0))      # A x - b (true)
    error: np.ndarray = field(default_factory=lambda: np.empty(0))         # x - u
    x_coords: np.ndarray = field(default_factory=lambda: np.empty(0))
    y_coords: np.ndarray = field(default_factory=lambda: np.empty(0))
    residual_norm: float = 0.0              # recursive max-norm (dirichlet_solver.cpp:122)
    error_norm: float = 0.0
    iterations: int = 0
    precision: float = 0.0                  # never assigned by the reference; kept at 0
    converged: bool = False
    stop_reason: str = ""


class DirichletSolver:
    """Facade, solver/dirichlet_solver.hpp:79-184 / .cpp:11-131."""

    def __init__(self, n: int = 10, m: int = 10, a: float = 0.0, b: float = 1.0, c: float = 0.0,
                 d: float = 1.0, device: int = 0):
        self._device = device
        self.eps_precision = self.eps_residual = self.eps_exact_error = 1e-6     # .cpp:14
        self.max_iterations = 10000
        self.use_precision, self.use_residual, self.use_error, self.use_max_iterations = True, True, False, True  # .cpp:15-16
        self.iteration_callback = None
        self.completion_callback = None
        self.solver: Optional[MSGSolver] = None
        self.solution = np.empty(0)
        self.true_solution = np.empty(0)
        self.setGridParameters(n, m, a, b, c, d)

    def setGridParameters(self, n, m, a, b, c, d):
        self.n, self.m, self.a, self.b, self.c, self.d = n, m, a, b, c, d
        self.grid = GridSystem(m, n, a, b, c, d, device=self._device)     # note (m, n): .cpp:24

    def setSolverParameters(self, eps_p, eps_r, eps_e, max_iter):
        self.eps_precision, self.eps_residual, self.eps_exact_error, self.max_iterations = eps_p, eps_r, eps_e, max_iter

    def enablePrecisionStopping(self, on): self.use_precision = on
    def enableResidualStopping(self, on): self.use_residual = on
    def enableErrorStopping(self, on): self.use_error = on
    def enableMaxIterationsStopping(self, on): self.use_max_iterations = on      # never read (as in the reference)
    def setIterationCallback(self, cb): self.iteration_callback = cb
    def setCompletionCallback(self, cb): self.completion_callback = cb
    def getMethodName(self): return self.solver.getName() if self.solver is not None else "МСГ"    # dirichlet_solver.hpp:159-161
    def getGridSystem(self): return self.grid
    def getSolution(self): return self.solution
    def getTrueSolution(self): return self.true_solution

    def requestStop(self):
        if self.solver is not None:
            self.solver.requestStop()

    def solve(self) -> SolverResults:
        if self.grid is None:
            raise RuntimeError("Grid system not initialized")                     # .cpp:62-64
        eps = min(self.eps_precision, self.eps_residual, self.eps_exact_error)    # .cpp:67-68
        s = MSGSolver(self.grid.get_matrix(), self.grid.get_rhs(), eps, self.max_iterations)
        s.setPrecisionEps(self.eps_precision if self.use_precision else -1.0)     # .cpp:71-87
        s.setResidualEps(self.eps_residual if self.use_residual else -1.0)
        s.setExactErrorEps(self.eps_exact_error if self.use_error else -1.0)
        if self.iteration_callback:
            s.setIterationCallback(self.iteration_callback)
        self.solver = s
        u = self.grid.get_true_solution_vector()                                   # .cpp:95
        x = s.solve(u)                                                             # .cpp:98
        r = SolverResults()
        r.solution, r.true_solution = x, u
        r.residual = self.grid._handle.true_residual()                             # A x - b, .cpp:106,147-161
        r.error = x - u                                                            # .cpp:110,164-180
        r.x_coords, r.y_coords = self.grid.get_x_coords(), self.grid.get_y_coords()
        r.iterations, r.converged = s.getIterations(), s.hasConverged()
        r.stop_reason = s.getStopReasonText()
        r.residual_norm, r.error_norm = s.getFinalResidualNorm(), s.getFinalErrorNorm()
        self.solution, self.true_solution = x, u
        if self.completion_callback:
            self.completion_callback(r)
        return r
