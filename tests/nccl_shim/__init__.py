"""Builds tests/nccl_shim/libnccl_shim.so (test infrastructure: see nccl_shim.cpp)."""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libnccl_shim.so")
SRC = os.path.join(HERE, "nccl_shim.cpp")


def build(force: bool = False) -> str:
    if not force and os.path.exists(LIB_PATH) and os.path.getmtime(LIB_PATH) >= os.path.getmtime(SRC):
        return LIB_PATH
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    tmp = f"{LIB_PATH}.tmp.{os.getpid()}"
    try:
        subprocess.check_call([hipcc, "-x", "hip", "--offload-arch=gfx950", "-O2", "-fPIC", "-shared", "-std=c++17", "-pthread", "-o", tmp, SRC, "-lrt"])
        os.replace(tmp, LIB_PATH)
    finally:
        if os.path.exists(tmp):
            os.remove(tmp)
    return LIB_PATH
