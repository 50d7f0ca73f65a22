"""Build libmi355cg.so (HIP kernels + C ABI) in-tree with hipcc for gfx950.

The shared library is the product: it is git-ignored (history stays source-only) but travels
to the GPU box with the tree.  hipcc cross-compiles without a GPU.
"""
from __future__ import annotations

import os
import shutil
import subprocess

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG_DIR, "csrc")
LIB_PATH = os.path.join(PKG_DIR, "libmi355cg.so")
SOURCES = ["mi355cg.hip", "grid_setup.cpp"]
HEADERS = ["cg_kernels.h", "csr_kernels.h", "team.h", "grid_setup.h", os.path.join("..", "..", "include", "mi355cg.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fPIC", "-shared", "-std=c++17",
         "-Wno-unused-value", "-Wno-unused-result", "-pthread",
         # a one-lane atomic stays a one-lane atomic: the optimizer's wave-aggregated form reads its result back at once, which drains
         # the load pipeline of the row-marching kernels at every item of the dynamic queues (cg_kernels.h: QueueSpec)
         "-mllvm", "-amdgpu-atomic-optimizer-strategy=None"]


def _hipcc() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: libmi355cg.so cannot be built (there is no CPU fallback)")


STAMP_PATH = LIB_PATH + ".srchash"


def _source_hash() -> str:
    """Contents of everything the library is built from (not modification times: a copy of the tree, e.g. the snapshot
    sent to a GPU box, need not keep them in order)."""
    import hashlib
    h = hashlib.sha256(" ".join(FLAGS).encode())
    for f in SOURCES + HEADERS:
        with open(os.path.join(CSRC, f), "rb") as fh:
            h.update(f.encode() + b"\0" + fh.read())
    return h.hexdigest()


def needs_build() -> bool:
    if not os.path.exists(LIB_PATH) or not os.path.exists(STAMP_PATH):
        return True
    with open(STAMP_PATH) as fh:
        return fh.read().strip() != _source_hash()


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not needs_build():
        return LIB_PATH
    # Compile to a private file and rename it into place: several processes (torchrun ranks, spawned test workers) may
    # decide to build at the same moment, and a reader must never see a half-written library.
    tmp = f"{LIB_PATH}.tmp.{os.getpid()}"
    cmd = [_hipcc()] + FLAGS + ["-o", tmp] + [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        print(" ".join(cmd))
    try:
        stamp = _source_hash()                      # of what the compiler is about to read
        subprocess.check_call(cmd, cwd=CSRC)
        os.replace(tmp, LIB_PATH)
        with open(f"{STAMP_PATH}.tmp.{os.getpid()}", "w") as fh:
            fh.write(stamp + "\n")
        os.replace(f"{STAMP_PATH}.tmp.{os.getpid()}", STAMP_PATH)
    finally:
        if os.path.exists(tmp):
            os.remove(tmp)
    return LIB_PATH


if __name__ == "__main__":
    print(build(force=True, verbose=True))
