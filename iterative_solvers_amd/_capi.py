"""ctypes binding of include/mi355cg.h (libmi355cg.so).  Fails loudly when the HIP library is
missing or unusable: there is no CPU path in this package."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import build as _build

OK, ERR_INVALID, ERR_HIP, ERR_STATE = 0, 1, 2, 3
F64, F32_MIXED = 0, 1
RULE_MSG_MAXNORM, RULE_REL_2NORM = 0, 1
STOP_ITERATIONS, STOP_PRECISION, STOP_RESIDUAL, STOP_EXACT_ERROR, STOP_INTERRUPTED = range(5)

EXPORTS = [
    "mi355cg_create", "mi355cg_create_csr", "mi355cg_set_true_solution", "mi355cg_destroy", "mi355cg_last_error", "mi355cg_version", "mi355cg_size",
    "mi355cg_get_rhs", "mi355cg_get_true_solution", "mi355cg_get_node_coords", "mi355cg_set_rhs",
    "mi355cg_apply", "mi355cg_apply_device", "mi355cg_default_params", "mi355cg_solve",
    "mi355cg_get_solution", "mi355cg_get_recursive_residual", "mi355cg_get_true_residual",
    "mi355cg_set_profiling", "mi355cg_get_kernel_time", "mi355cg_get_layout",
    "mi355cg_slab_rows", "mi355cg_create_slab", "mi355cg_owned_range", "mi355cg_dist_begin",
    "mi355cg_dist_reduce", "mi355cg_dist_sums_ptr", "mi355cg_dist_record_layout", "mi355cg_dist_scatter_ghosts", "mi355cg_dist_stencil", "mi355cg_dist_flip",
    "mi355cg_dist_update", "mi355cg_dist_update_reads_ghosts", "mi355cg_dist_check", "mi355cg_dist_summary", "mi355cg_dist_finish", "mi355cg_dist_history",
    "mi355cg_dist_halo", "mi355cg_dist_halo_recv_counts",
    "mi355cg_create_part", "mi355cg_checksum", "mi355cg_decompose", "mi355cg_halo_plan",
    "mi355cg_team_create_local", "mi355cg_team_unique_id", "mi355cg_team_create_rccl", "mi355cg_team_destroy",
    "mi355cg_team_solve", "mi355cg_team_info", "mi355cg_team_part", "mi355cg_team_get_vector", "mi355cg_team_set_vector", "mi355cg_team_checksum",
    "mi355cg_team_set_profiling", "mi355cg_team_phase_times", "mi355cg_team_describe", "mi355cg_setup_on_device", "mi355cg_team_setup_on_device", "mi355cg_debug_plan",
    "mi355cg_team_set_dtype",
]
DECOMP_ROWS, DECOMP_2D = 0, 1


class Params(C.Structure):
    _fields_ = [("rule", C.c_int), ("max_iterations", C.c_int),
                ("eps_precision", C.c_double), ("eps_residual", C.c_double),
                ("eps_exact_error", C.c_double), ("eps_rel", C.c_double),
                ("use_true_solution", C.c_int), ("callback_every", C.c_int),
                ("diagnostics", C.c_int), ("sync_every", C.c_int), ("fixed_iterations", C.c_int),
                ("inner_eps", C.c_double)]


class Results(C.Structure):
    _fields_ = [("iterations", C.c_int), ("converged", C.c_int), ("stop_reason", C.c_int),
                ("final_residual_norm", C.c_double), ("final_precision", C.c_double),
                ("final_error_norm", C.c_double), ("r_norm2", C.c_double),
                ("initial_r_norm2", C.c_double), ("solve_seconds", C.c_double),
                ("refine_true_rel", C.c_double), ("refine_outer", C.c_int), ("loop_seconds", C.c_double)]


class HaloMsg(C.Structure):
    _fields_ = [("id", C.c_int), ("peer", C.c_int), ("send", C.c_int), ("kind", C.c_int),
                ("y0", C.c_int), ("y1", C.c_int), ("x0", C.c_int), ("x1", C.c_int), ("count", C.c_longlong)]


ITER_CB = C.CFUNCTYPE(None, C.c_void_p, C.c_int, C.c_double, C.c_double, C.c_double)
_DP = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")

_lib = None


class Mi355cgError(RuntimeError):
    def __init__(self, code: int, text: str):
        super().__init__(f"mi355cg error {code}: {text}")
        self.code = code


def lib_path() -> str:
    return _build.LIB_PATH


def load():
    """Load libmi355cg.so (building it with hipcc first when the sources are newer)."""
    global _lib
    if _lib is not None:
        return _lib
    path = os.environ.get("MI355CG_LIB") or _build.LIB_PATH        # MI355CG_LIB: A/B an alternative build of the same ABI
    if path == _build.LIB_PATH and _build.needs_build():
        try:
            _build.build()
        except Exception as e:  # pre-built .so shipped to a box without hipcc is fine
            if not os.path.exists(path):
                raise RuntimeError(f"libmi355cg.so is missing and could not be built: {e}") from e
    # PyTorch-ROCm wheels bundle their own libamdhip64 / libhsa-runtime64.  If this library pulled /opt/rocm's copies
    # into the process first, torch would come up without a GPU (torch.cuda.is_available() == False) as soon as it is
    # imported later for device tensors or RCCL.  Importing torch first makes both sides share one runtime.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(path)
    H = C.c_void_p
    L.mi355cg_create.argtypes = [C.c_int, C.c_int] + [C.c_double] * 4 + [C.c_int, C.c_int, C.POINTER(H)]
    _IPn = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")
    L.mi355cg_create_csr.argtypes = [C.c_longlong, _IPn, _IPn, _DP, C.c_int, C.POINTER(H)]
    L.mi355cg_set_true_solution.argtypes = [H, _DP]
    L.mi355cg_destroy.argtypes = [H]
    L.mi355cg_destroy.restype = None
    L.mi355cg_last_error.restype = C.c_char_p
    L.mi355cg_version.restype = C.c_char_p
    L.mi355cg_size.argtypes = [H]
    L.mi355cg_size.restype = C.c_longlong
    for name in ("mi355cg_get_rhs", "mi355cg_get_true_solution", "mi355cg_set_rhs",
                 "mi355cg_get_solution", "mi355cg_get_recursive_residual", "mi355cg_get_true_residual"):
        getattr(L, name).argtypes = [H, _DP]
    L.mi355cg_get_node_coords.argtypes = [H, _DP, _DP]
    L.mi355cg_apply.argtypes = [H, _DP, _DP]
    L.mi355cg_apply_device.argtypes = [H, C.c_void_p, C.c_void_p]
    L.mi355cg_default_params.argtypes = [C.POINTER(Params), C.c_int]
    L.mi355cg_default_params.restype = None
    L.mi355cg_solve.argtypes = [H, C.POINTER(Params), ITER_CB, C.c_void_p, C.c_void_p, C.POINTER(Results)]
    L.mi355cg_set_profiling.argtypes = [H, C.c_int]
    L.mi355cg_get_kernel_time.argtypes = [H, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_longlong)]
    L.mi355cg_get_layout.argtypes = [H, C.POINTER(C.c_longlong)] + [C.POINTER(C.c_int)] * 5
    LLP, IP, VPP = C.POINTER(C.c_longlong), C.POINTER(C.c_int), C.POINTER(C.c_void_p)
    L.mi355cg_slab_rows.argtypes = [C.c_int, C.c_int, C.c_int, IP, IP]
    L.mi355cg_create_slab.argtypes = [C.c_int, C.c_int] + [C.c_double] * 4 + [C.c_int] * 4 + [C.POINTER(H)]
    L.mi355cg_owned_range.argtypes = [H, LLP, LLP, IP, IP]
    L.mi355cg_dist_begin.argtypes = [H, C.POINTER(Params), C.c_void_p]
    L.mi355cg_dist_reduce.argtypes = [H, C.c_int, C.c_int, C.c_void_p]
    L.mi355cg_dist_sums_ptr.argtypes = [H, C.c_int, VPP, IP]
    L.mi355cg_dist_record_layout.argtypes = [H, IP, IP, IP]
    L.mi355cg_dist_scatter_ghosts.argtypes = [H, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    L.mi355cg_dist_stencil.argtypes = [H, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
    L.mi355cg_dist_flip.argtypes = [H]
    L.mi355cg_dist_update.argtypes = [H, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
    L.mi355cg_dist_update_reads_ghosts.argtypes = [H]
    L.mi355cg_dist_check.argtypes = [H, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    L.mi355cg_dist_summary.argtypes = [H, C.POINTER(Results), IP]
    L.mi355cg_dist_finish.argtypes = [H, C.c_void_p]
    DBP = C.POINTER(C.c_double)
    L.mi355cg_dist_history.argtypes = [H, C.c_int, DBP, DBP, DBP]
    L.mi355cg_dist_halo.argtypes = [H, C.c_int, VPP, VPP, LLP, VPP, VPP, LLP]
    L.mi355cg_dist_halo_recv_counts.argtypes = [H, LLP, LLP]
    L.mi355cg_create_part.argtypes = [C.c_int, C.c_int] + [C.c_double] * 4 + [C.c_int] * 6 + [C.POINTER(H)]
    L.mi355cg_checksum.argtypes = [H, C.c_int, DBP]
    L.mi355cg_decompose.argtypes = [C.c_int] * 4 + [IP] * 4
    L.mi355cg_halo_plan.argtypes = [C.c_int] * 5 + [IP, C.POINTER(HaloMsg)]
    L.mi355cg_team_create_local.argtypes = [C.c_int, C.c_int] + [C.c_double] * 4 + [C.c_int, IP, C.c_int, C.c_int, C.POINTER(H)]
    L.mi355cg_team_unique_id.argtypes = [C.c_void_p]
    L.mi355cg_team_create_rccl.argtypes = [C.c_int, C.c_int] + [C.c_double] * 4 + [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.POINTER(H)]
    L.mi355cg_team_destroy.argtypes = [H]
    L.mi355cg_team_destroy.restype = None
    L.mi355cg_team_solve.argtypes = [H, C.POINTER(Params), ITER_CB, C.c_void_p, C.c_void_p, C.POINTER(Results)]
    L.mi355cg_team_info.argtypes = [H, IP, IP, IP, LLP]
    L.mi355cg_team_part.argtypes = [H, C.c_int, C.POINTER(H), IP]
    L.mi355cg_team_get_vector.argtypes = [H, C.c_int, _DP]
    L.mi355cg_team_set_vector.argtypes = [H, C.c_int, _DP]
    L.mi355cg_team_checksum.argtypes = [H, C.c_int, DBP]
    L.mi355cg_debug_plan.argtypes = [C.c_int] * 5 + [IP, IP, IP, IP, IP]
    L.mi355cg_setup_on_device.argtypes = [H]
    L.mi355cg_team_setup_on_device.argtypes = [H]
    L.mi355cg_team_set_dtype.argtypes = [H, C.c_int]
    L.mi355cg_team_set_profiling.argtypes = [H, C.c_int]
    L.mi355cg_team_phase_times.argtypes = [H, DBP, DBP, DBP]
    L.mi355cg_team_describe.argtypes = [H, C.c_char_p, C.c_int]
    _lib = L
    return L


def check(rc: int):
    if rc != OK:
        raise Mi355cgError(rc, load().mi355cg_last_error().decode("utf-8", "replace"))
