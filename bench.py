#!/usr/bin/env python3
"""bench.py -- CG iterations/second and achieved HBM GB/s of the matrix-free CG hot path.

A "step" is one CG iteration (fused stencil kernel A' + fused update kernel B) on the
BASELINE config-2 workload: N x N = 4096 x 4096 intervals on the L-shaped domain, fp64,
U = 12 574 721 unknowns per GPU, deterministic synthetic RHS (the reference's f and Dirichlet
data), x0 = 0, convergence tests disabled so that exactly K iterations are timed.

  python bench.py --gpus 1 --steps 2000 --warmup 200
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...   (weak scaling:
      every rank owns a 4096-interval-class slab; the global grid grows with N)

Prints ONE JSON line (rank 0).  value = total CG iterations/s of the job; hbm_gbps uses the
ALGORITHMIC bytes of SURVEY 8d (88 B per unknown per iteration, fp64).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md); 6290 measured-achievable
ALG_BYTES_PER_UNKNOWN = 88.0    # 11 words fp64 per unknown per iteration
# Compulsory words per unknown and launch of THIS implementation -- what roofline.achieved counts.
# Default REL_2NORM (A p recomputed in the update launch, never stored; x updated every second iteration, two steps at
# once): stencil launch = read r,p / write p = 3; update launch = read p,r / write r = 3 on odd iterations and
# read p,p_prev,r,x / write r,x = 6 on even ones, 4.5 on average (7.5 per iteration).
# MSG: stencil read r,p / write p = 3, update read p,r,x / write r,x = 5.
# MI355CG_X2STEP=0: REL_2NORM 5 + 3.  MI355CG_RECOMPUTE=0: REL_2NORM 6 + 3 (A p stored and streamed back), MSG or MI355CG_XFUSE=0: 4 + 6.
KERNEL_ALG_WORDS_X2 = {"stencil": 3, "update": 4.5}
KERNEL_ALG_WORDS_RECOMP = {"stencil": 5, "update": 3}
KERNEL_ALG_WORDS_RECOMP_MSG = {"stencil": 3, "update": 5}
KERNEL_ALG_WORDS_XFUSE = {"stencil": 6, "update": 3}
KERNEL_ALG_WORDS_PLAIN = {"stencil": 4, "update": 6}


def unknowns(n: int) -> int:
    return (n // 2 - 1) * (3 * n // 2 - 1)


def cpu_baseline(n: int, iters: int):
    """Time the CPU oracle (a port of the reference's MatrixFreeSolver loop, 1 thread) on a bounded
    sample of the same workload."""
    from oracle.oracle import OracleGrid
    g = OracleGrid(n, n, 1.0, 2.0, 1.0, 2.0)
    b = g.rhs()
    u = g.true_solution()
    t0 = time.perf_counter()
    r = g.mf_solve(b=b, true_solution=u, eps=0.0, max_iterations=iters, diagnostics=False)
    dt = time.perf_counter() - t0
    assert r.iterations == iters
    out = {"value": iters / dt, "unit": "iters/s", "cores": 1, "kind": "port",
           "sample": f"{iters} CG iterations of oracle/cg_oracle.c (MatrixFreeSolver loop without the "
                     f"diagnostic second apply) at N={n}, {dt:.1f} s on 1 of {os.cpu_count()} host cores"}
    try:        # the same loop on every host core (OpenMP build of the same source; BASELINE.md section 4 "ref-omp")
        from oracle.oracle import mf_solve_all_cores
        t0 = time.perf_counter()
        mf_solve_all_cores(n, b, 0.0, 3)                                # calibration: keep this leg to about 10 s
        per_it = (time.perf_counter() - t0) / 3
        k = max(3, min(10 * iters, int(10.0 / max(per_it, 1e-4))))
        t0 = time.perf_counter()
        its, _, threads = mf_solve_all_cores(n, b, 0.0, k)
        dt2 = time.perf_counter() - t0
        out["all_cores"] = {"value": its / dt2, "unit": "iters/s", "cores": threads, "kind": "port-openmp",
                            "sample": f"{its} iterations, {dt2:.1f} s, {threads} OpenMP threads"}
    except Exception as e:                                           # never fail the bench on the baseline leg
        out["all_cores"] = {"error": repr(e)[:200]}
    return out


def read_traffic():
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC summary."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        with open(path) as f:
            return json.load(f)
    except Exception:
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--grid", dest="n", type=int, default=4096, help="grid intervals per side on one GPU")
    ap.add_argument("--rule", choices=["rel2", "msg"], default="rel2")
    ap.add_argument("--dtype", choices=["f64", "f32"], default="f64",
                    help="f32 = BASELINE config 3: the fp32-storage inner CG of the mixed-precision path (use --grid 8192)")
    ap.add_argument("--cpu-iters", type=int, default=20, help="oracle iterations for cpu_baseline (0 = skip)")
    ap.add_argument("--no-roofline-pass", action="store_true")
    args = ap.parse_args()

    import torch
    import iterative_solvers_amd as isa
    from iterative_solvers_amd import _capi

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    rule = _capi.RULE_REL_2NORM if args.rule == "rel2" else _capi.RULE_MSG_MAXNORM

    if world > 1 or args.gpus > 1 or os.environ.get("MI355CG_BENCH_DIST") == "1":
        from iterative_solvers_amd import distributed as dist_cg
        out = dist_cg.bench(args, rule)
        if rank == 0:
            print(json.dumps(out))
        return

    n = args.n
    U = unknowns(n)
    f32 = args.dtype == "f32"
    wbytes = 4.0 if f32 else 8.0
    sysm = isa.MatrixFreeSystem(n, n, 1.0, 2.0, 1.0, 2.0, device=local_rank, dtype=isa.F32_MIXED if f32 else isa.F64)
    h = sysm._handle

    def run(iters: int, profile: bool):
        p = isa.default_params(rule)
        p.max_iterations = iters
        p.fixed_iterations = 1
        p.use_true_solution = 0
        p.callback_every = 0
        p.sync_every = 500
        h.set_profiling(profile)
        return h.solve(p)

    run(args.warmup, False)                                   # untimed warm-up iterations
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    res = run(args.steps, False)                              # exactly K iterations (plus the init pass)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    assert res.iterations == args.steps, (res.iterations, args.steps)
    its = args.steps / dt

    roofline = None
    if not args.no_roofline_pass:
        # same loop again with a HIP-event pair around every launch on the solve stream
        k = min(args.steps, 500)
        run(k, True)
        xfuse = args.rule == "rel2" and os.environ.get("MI355CG_XFUSE", "1") != "0"
        recomp = os.environ.get("MI355CG_RECOMPUTE", "1") != "0" and (xfuse or args.rule == "msg")
        x2 = recomp and xfuse and os.environ.get("MI355CG_X2STEP", "1") != "0"
        if x2:
            KERNEL_ALG_WORDS = KERNEL_ALG_WORDS_X2
        elif recomp:
            KERNEL_ALG_WORDS = KERNEL_ALG_WORDS_RECOMP if xfuse else KERNEL_ALG_WORDS_RECOMP_MSG
        else:
            KERNEL_ALG_WORDS = KERNEL_ALG_WORDS_XFUSE if xfuse else KERNEL_ALG_WORDS_PLAIN
        words_iter = sum(KERNEL_ALG_WORDS.values())
        t = {name: h.kernel_time(i) for i, name in enumerate(("stencil", "update"))}
        dom = max(t, key=lambda name: t[name][0] * t[name][1])
        ms, launches = t[dom]
        alg = KERNEL_ALG_WORDS[dom] * wbytes * U
        achieved = alg / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
        tr = read_traffic()
        kname = {"stencil": "k_stencil", "update": "k_update_st" if recomp else "k_update"}[dom]
        roofline = {"bound": "hbm", "kernel": kname, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS,
                    "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 4), "frac_of_measured_6290": round(achieved / 6290.0, 4),
                    "traffic": (tr or {}).get(dom), "avg_ms": round(ms, 5), "launches": launches,
                    "alg_bytes_per_launch": alg, "alg_words_per_unknown": KERNEL_ALG_WORDS[dom],
                    # what the whole iteration really has to move in this implementation, and the rate that is
                    "words_per_unknown_per_iteration": words_iter,
                    "alg_words": KERNEL_ALG_WORDS,
                    "moved_gbps_per_iteration": round(words_iter * wbytes * U * its / 1e9, 1),
                    "other": {name: {"avg_ms": round(t[name][0], 5),
                                     "achieved": round(KERNEL_ALG_WORDS[name] * wbytes * U / (t[name][0] * 1e-3) / 1e9, 1) if t[name][0] > 0 else 0}
                              for name in t}}

    out = {
        "metric": "cg_iters_per_sec", "value": round(its, 2), "unit": "iters/s",
        "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 5),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": f"{n}x{n} L-shaped Dirichlet Poisson {'fp32 inner CG of the mixed-precision path' if f32 else 'fp64'}, matrix-free CG, fixed {args.steps} iterations",
                   "n": n, "unknowns": U, "rule": args.rule, "layout": h.layout()},
        # SURVEY 8d's convention: 11 words (88 B fp64) per unknown and iteration, the compulsory traffic of textbook
        # three-phase CG.  This implementation moves fewer words (roofline.words_per_unknown_per_iteration), so the
        # figure can exceed the 8 TB/s pin rate: it is an algorithmic equivalent, not bytes on the bus -- those are
        # roofline.moved_gbps_per_iteration and the per-kernel roofline.achieved.
        "hbm_gbps": round(ALG_BYTES_PER_UNKNOWN * (wbytes / 8.0) * U * its / 1e9, 1),
        "hbm_gbps_convention": "88 B (fp64) / 44 B (fp32) per unknown per iteration (SURVEY 8d); see roofline.moved_gbps_per_iteration for bytes really moved",
        "roofline": roofline,
    }
    if args.cpu_iters > 0 and not f32:
        out["cpu_baseline"] = cpu_baseline(n, args.cpu_iters)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
