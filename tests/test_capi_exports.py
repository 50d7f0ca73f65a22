"""CPU-side checks of the drop-in boundary: libmi355cg.so builds/loads, exports every symbol
include/mi355cg.h declares, validates arguments, and refuses to compute without a GPU."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "mi355cg.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mi355cg_[a-z0-9_]+)\s*\(", text)) - {"mi355cg_iter_cb"})


def test_library_exports_every_declared_symbol():
    from iterative_solvers_amd import _capi
    lib = _capi.load()
    declared = _declared_symbols()
    assert len(declared) >= 15
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/mi355cg.h but not exported"
    assert set(declared) == set(_capi.EXPORTS)
    assert b"gfx950" in lib.mi355cg_version()


def test_struct_layouts_match_header():
    from iterative_solvers_amd import _capi
    p = _capi.Params()
    _capi.load().mi355cg_default_params(C.byref(p), _capi.RULE_MSG_MAXNORM)
    assert (p.rule, p.max_iterations, p.callback_every) == (0, 10000, 100)
    assert p.eps_precision == p.eps_residual == p.eps_exact_error == 1e-6
    _capi.load().mi355cg_default_params(C.byref(p), _capi.RULE_REL_2NORM)
    assert (p.rule, p.callback_every, p.eps_rel) == (1, 1, 1e-6)


def test_invalid_grid_is_rejected_before_touching_the_gpu():
    from iterative_solvers_amd import _capi
    lib = _capi.load()
    h = C.c_void_p()
    for n, m in ((7, 7), (8, 6), (4, 4)):
        assert lib.mi355cg_create(n, m, 1.0, 2.0, 1.0, 2.0, 0, 0, C.byref(h)) == _capi.ERR_INVALID
        assert b"rejected" in lib.mi355cg_last_error()
    assert lib.mi355cg_create(8, 8, 1.0, 2.0, 1.0, 2.0, 7, 0, C.byref(h)) == _capi.ERR_INVALID


def test_no_cpu_fallback_without_a_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import iterative_solvers_amd as isa
    with pytest.raises(isa.Mi355cgError) as e:
        isa.MatrixFreeSystem(8, 8, 1.0, 2.0, 1.0, 2.0)
    assert e.value.code == 2 and "no CPU fallback" in str(e.value)


def test_product_package_never_imports_the_oracle():
    """The oracle is test infrastructure: nothing shipped under iterative_solvers_amd/ may
    import, include, link or dlopen it."""
    pkg = os.path.join(ROOT, "iterative_solvers_amd")
    pat = re.compile(r"(import\s+oracle|from\s+oracle|cg_oracle|libcg_oracle|oracle[/.]oracle|#include\s*\"[^\"]*oracle)")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp", ".hpp")):
                src = open(os.path.join(dirpath, f), encoding="utf-8").read()
                assert not pat.search(src), f"{f} references the oracle"
