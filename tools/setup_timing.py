#!/usr/bin/env python3
"""Time-to-first-iteration of a handle: host setup (std::exp on all host cores + upload) vs MI355CG_DEVICE_SETUP=1.
Usage: python tools/setup_timing.py N"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: F401
import iterative_solvers_amd as isa

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
isa.MatrixFreeSystem(64, 64, 1.0, 2.0, 1.0, 2.0)          # runtime warm-up
for mode in ("host", "device"):
    if mode == "device":
        os.environ["MI355CG_DEVICE_SETUP"] = "1"
    t0 = time.perf_counter()
    s = isa.MatrixFreeSystem(N, N, 1.0, 2.0, 1.0, 2.0)
    dt = time.perf_counter() - t0
    print(f"N={N} create with {mode} setup: {dt:.3f} s ({os.cpu_count()} host cores)", flush=True)
    s._handle.close()
