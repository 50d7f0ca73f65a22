#!/usr/bin/env python3
"""bench.py -- CG iterations/second and achieved HBM GB/s of the matrix-free CG hot path.

A "step" is one CG iteration (fused stencil launch A' + fused update launch B).  Default workload = BASELINE
config 2: N x N = 4096 x 4096 intervals on the L-shaped domain, fp64, U = 12 574 721 unknowns, deterministic synthetic
RHS (the reference's f and Dirichlet data), x0 = 0, convergence tests disabled so that exactly K iterations are timed.

  python bench.py --gpus 1 --steps 2000 --warmup 200
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...
      weak scaling (default): every rank owns a config-2-sized part, the global grid grows with N
      --scaling strong --grid 32768: BASELINE config 5's fixed grid cut into N parts
      --decomp rows | 2d: row slabs, or (N/2) x 2 blocks (config 4's "2 x 2" at N = 4)

Prints ONE JSON line (rank 0).  `value` = CG iterations/s of the whole job (weak scaling: in units of config-2-sized
parts advanced per second), timed on the host around K iterations between device synchronisations; `hbm_gbps` = bytes the
iteration really moves (58 B per unknown for the REL_2NORM loop) per second; the per-kernel roofline comes from HIP events.
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

HBM_PEAK_GBPS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md); 6290 measured-achievable (float4 copy, guide)
SURVEY_BYTES_PER_UNKNOWN = 88.0  # SURVEY 8d's convention: 11 words per unknown per iteration (textbook three-phase CG)
# Compulsory words per unknown and launch of THIS implementation -- what roofline.achieved counts.
# REL_2NORM: stencil launch reads r, p and writes p = 3; update launch reads p, r and writes r = 3 on three iterations out of M = 4
# and reads p, r, x, three older directions and writes r, x = 8 on the fourth: 4.25 on average, 7.25 per iteration
# (MI355CG_XSTEPS=2: 3 and 6 alternating = 4.5, 7.5 per iteration).  MSG: 3 + 5 (x every iteration).
_M = 2 if os.environ.get("MI355CG_XSTEPS") == "2" else 4
WORDS = {"rel2": {"stencil": 3, "update": (3 * (_M - 1) + (4 + _M)) / _M}, "msg": {"stencil": 3, "update": 5}}


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
    """HBM bytes per launch of the kernels from the committed rocprofv3 PMC summary (N = 4096 fp64 only)."""
    try:
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
            return json.load(f)
    except Exception:
        return None


def bench_team(args, rule):
    """N > 1: one process per GPU, the native RCCL team (csrc/team.h)."""
    import torch
    import torch.distributed as dist
    import iterative_solvers_amd as isa
    from iterative_solvers_amd import _capi
    from iterative_solvers_amd import distributed as D

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if "RANK" not in os.environ:                            # started without a launcher: one rank, still through RCCL
        os.environ.update({"RANK": "0", "WORLD_SIZE": "1", "LOCAL_RANK": "0"})
        os.environ.setdefault("MASTER_PORT", "29531")
    torch.cuda.set_device(local_rank)
    if not dist.is_initialized():
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    strong = args.scaling == "strong"
    n = args.n if strong else D.weak_scaling_n(args.n, world)
    U1, U = unknowns(args.n), unknowns(n)
    decomp = _capi.DECOMP_2D if args.decomp == "2d" else _capi.DECOMP_ROWS
    boxes = D.decompose(n, world, decomp)

    path, note = "native RCCL team (csrc/team.h: ncclAllGather of 16-double records + neighbour ncclSend/ncclRecv on a second stream)", None
    team = cg = None
    try:
        team = D.Team.rccl(n, decomp, device=local_rank)
    except Exception as e:                                  # keep the scaling run alive: the torch.distributed driver of round 1
        note = f"native team unavailable ({repr(e)[:160]}); fell back to the torch.distributed driver over row slabs"
        path = "torch.distributed driver (iterative_solvers_amd/distributed.py DistributedCG, halo=p2p)"
        y_lo, y_hi = D.slab_rows(n, world, rank)
        cg = D.DistributedCG(D.SlabEngine(n, y_lo, y_hi, device=local_rank), halo="p2p")

    def run(iters):
        p = isa.default_params(rule)
        p.max_iterations, p.fixed_iterations, p.use_true_solution, p.callback_every, p.sync_every = iters, 1, 0, 0, 500
        return team.solve(p) if team else cg.solve(p)

    run(args.warmup)
    torch.cuda.synchronize(); dist.barrier()
    t0 = time.perf_counter()
    res = run(args.steps)
    torch.cuda.synchronize(); dist.barrier()
    dt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=torch.device("cuda", local_rank))
    dist.all_reduce(dt, op=dist.ReduceOp.MAX)
    dt = float(dt.item())
    assert res.iterations == args.steps
    its = args.steps / dt
    phases = None
    if team:
        team.set_profiling(True)
        run(min(args.steps, 200))
        team.set_profiling(False)
        phases = team.phase_times()
        phases["driver_and_wait_ms"] = max(0.0, phases["wall_ms"] - phases["kernels_ms"])
        phases["note"] = "rank 0, per iteration: device time of its kernels; device time of the collectives + halo messages on the comm stream (they overlap the kernels); wall"
    # Cross-check of the distributed loop (untimed): V iterations on the team against the SAME global problem solved by ONE context
    # on rank 0's GPU.  The team's reductions are summed part by part in a fixed order, so the residual norm has to agree to
    # rounding of the last bit or two; a halo row that arrived late or in the wrong place shows up in the leading digits.
    verify = None
    if (world > 1 or os.environ.get("MI355CG_BENCH_DIST") == "1") and args.verify > 0:
        if U <= args.verify_max_unknowns:
            rv = run(args.verify)
            torch.cuda.synchronize()
            if rank == 0:
                one = isa.MatrixFreeSystem(n, n, 1.0, 2.0, 1.0, 2.0, device=local_rank)
                p = isa.default_params(rule)
                p.max_iterations, p.fixed_iterations, p.use_true_solution, p.callback_every, p.sync_every = args.verify, 1, 0, 0, 500
                r1 = one._handle.solve(p)
                one._handle.close()
                rel = abs(rv.r_norm2 - r1.r_norm2) / max(abs(r1.r_norm2), 1e-300)
                verify = {"iterations": args.verify, "team_r_norm2": rv.r_norm2, "single_gpu_r_norm2": r1.r_norm2,
                          "rel_diff": rel, "ok": bool(rel <= 1e-12 and rv.iterations == r1.iterations)}
            dist.barrier()
        else:
            verify = {"skipped": f"{U} unknowns > --verify-max-unknowns {args.verify_max_unknowns}"}
    bytes_it = 8.0 * sum(WORDS[args.rule].values())
    units = 1.0 if strong else U / U1
    moved = bytes_it * U * its / 1e9 / world
    out = {
        "metric": "cg_iters_per_sec", "value": round(its * units, 2), "unit": "iters/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 5),
        "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": (f"{n}x{n} L-shaped Dirichlet Poisson fp64 cut into {world} parts, matrix-free CG, fixed {args.steps} iterations"
                                if strong else
                                f"{n}x{n} L-shaped Dirichlet Poisson fp64 over {world} parts of ~{U1} unknowns (config-2 size per GPU), matrix-free CG, fixed {args.steps} iterations"),
                   "n": n, "unknowns": U, "unknowns_per_gpu": U / world, "rule": args.rule,
                   "value_is": "global CG iterations/s" if strong else "global CG iterations/s x (unknowns / config-2 unknowns) = config-2-sized part iterations/s",
                   "decomposition": {"kind": "2d" if decomp else "rows", "parts": boxes},
                   "parallelism": f"{'(N/2) x 2 blocks' if decomp else 'row slabs'} x{world}, one process per GPU, {path}"},
        "global_iters_per_sec": round(its, 2),
        "hbm_gbps": round(bytes_it * U * its / 1e9, 1),
        "hbm_gbps_is": f"bytes really moved, summed over GPUs: {bytes_it:.0f} B per unknown per iteration",
        "algorithmic_equivalent_gbps_88B": round(SURVEY_BYTES_PER_UNKNOWN * U * its / 1e9, 1),
        "phases_ms": phases,
        "verify_against_one_gpu": verify,
        # whole-iteration roofline per GPU (kernels + collectives + driver); the per-kernel figures are in the 1-GPU bench line
        "roofline": {"bound": "hbm", "achieved": round(moved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(moved / HBM_PEAK_GBPS, 4),
                     "traffic": None, "scope": "bytes one GPU has to move per iteration / wall time per iteration, collectives included"},
    }
    if note:
        out["note"] = note
    dist.barrier()
    if team:
        team.close()
    if "TORCHELASTIC_RUN_ID" not in os.environ and world == 1:
        dist.destroy_process_group()                          # started without a launcher: leave nothing behind
    return out


def main():
    # ONE JSON line on stdout: libraries chat there too (RCCL prints a version banner when its first communicator comes up),
    # so everything else this process writes to fd 1 goes to stderr until the line is printed.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    def emit(obj):
        sys.stdout.flush()
        os.dup2(real_stdout, 1)
        print(json.dumps(obj), flush=True)

    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--grid", dest="n", type=int, default=4096, help="grid intervals per side (per GPU for weak scaling)")
    ap.add_argument("--rule", choices=["rel2", "msg"], default="rel2")
    ap.add_argument("--dtype", choices=["f64", "f32"], default="f64",
                    help="f32 = BASELINE config 3: the fp32-storage inner CG of the mixed-precision path (use --grid 8192)")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak", help="N > 1: weak (part size fixed) or strong (--grid fixed)")
    ap.add_argument("--decomp", choices=["rows", "2d"], default="rows", help="N > 1: row slabs or (N/2) x 2 blocks")
    ap.add_argument("--cpu-iters", type=int, default=20, help="oracle iterations for cpu_baseline (0 = skip)")
    ap.add_argument("--no-roofline-pass", action="store_true")
    ap.add_argument("--verify", type=int, default=30, help="N > 1: iterations of the untimed cross-check against one GPU (0 = skip)")
    ap.add_argument("--verify-max-unknowns", type=float, default=2.6e8, help="skip that cross-check above this size (host set-up time)")
    ap.add_argument("--watchdog", type=float, default=1500.0, help="seconds after which a stuck run reports an error line and exits")
    args = ap.parse_args()

    # A collective that never completes must not hang the caller for ever: report and leave.
    import threading
    def _expired():
        if int(os.environ.get("RANK", "0")) == 0:
            emit({"metric": "cg_iters_per_sec", "value": None, "unit": "iters/s", "n_gpus": int(os.environ.get("WORLD_SIZE", "1")),
                  "error": f"no result after {args.watchdog:.0f} s (watchdog): the run was stuck, most likely in a collective"})
        os._exit(3)
    wd = threading.Timer(args.watchdog, _expired)
    wd.daemon = True
    wd.start()

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
        out = bench_team(args, rule)
        wd.cancel()
        if rank == 0:
            emit(out)
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
    res = run(args.steps, False)                              # exactly K iterations (plus the initialisation pass of a solve)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    assert res.iterations == args.steps, (res.iterations, args.steps)
    its = args.steps / dt
    its_loop = args.steps / res.loop_seconds if res.loop_seconds > 0 else None     # HIP events around the K iterations alone

    words = WORDS[args.rule]
    words_iter = sum(words.values())
    roofline = None
    if not args.no_roofline_pass:
        # same loop again with a HIP-event pair around every launch on the solve stream
        k = min(args.steps, 500)
        run(k, True)
        t = {name: h.kernel_time(i) for i, name in enumerate(("stencil", "update"))}
        dom = max(t, key=lambda name: t[name][0] * t[name][1])
        ms, launches = t[dom]
        alg = words[dom] * wbytes * U
        achieved = alg / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
        tr = read_traffic() if (n == 4096 and not f32 and args.rule == "rel2") else None
        kname = {"stencil": "k_stencil", "update": "k_update_st"}[dom]
        roofline = {"bound": "hbm", "kernel": kname, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS,
                    "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 4), "frac_of_measured_6290": round(achieved / 6290.0, 4),
                    "traffic": (tr or {}).get(dom), "avg_ms": round(ms, 5), "launches": launches,
                    "alg_bytes_per_launch": alg, "alg_words_per_unknown": words[dom],
                    "words_per_unknown_per_iteration": words_iter, "alg_words": words,
                    "other": {name: {"avg_ms": round(t[name][0], 5),
                                     "achieved": round(words[name] * wbytes * U / (t[name][0] * 1e-3) / 1e9, 1) if t[name][0] > 0 else 0}
                              for name in t}}

    out = {
        "metric": "cg_iters_per_sec", "value": round(its, 2), "unit": "iters/s",
        "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 5),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": f"{n}x{n} L-shaped Dirichlet Poisson {'fp32 inner CG of the mixed-precision path' if f32 else 'fp64'}, matrix-free CG, fixed {args.steps} iterations",
                   "n": n, "unknowns": U, "rule": args.rule, "layout": h.layout()},
        "timing": "host clock around one solve of K iterations between device synchronisations (includes the solve's initialisation pass: "
                  "one pass x = 0, r = b, z = 0, ||r0||, and the last poll ~ 0.2 ms); loop_only_* = HIP events around the K iterations on the solve stream",
        "loop_only_iters_per_sec": round(its_loop, 2) if its_loop else None,
        "loop_only_ms_per_step": round(1e3 * res.loop_seconds / args.steps, 5) if its_loop else None,
        # bytes the iteration really moves in this implementation: never above the HBM pin rate
        "hbm_gbps": round(words_iter * wbytes * U * its / 1e9, 1),
        "hbm_gbps_is": f"bytes really moved: {words_iter} words = {words_iter * wbytes:.0f} B per unknown per iteration",
        # SURVEY 8d's convention (11 words = 88 B fp64 per unknown and iteration, textbook three-phase CG): an algorithmic
        # equivalent, NOT bus bytes -- 3.5 of those 11 words are no longer moved here, so it may exceed the pin rate
        "algorithmic_equivalent_gbps_88B": round(SURVEY_BYTES_PER_UNKNOWN * (wbytes / 8.0) * U * its / 1e9, 1),
        "roofline": roofline,
    }
    if args.cpu_iters > 0 and not f32:
        out["cpu_baseline"] = cpu_baseline(n, args.cpu_iters)
    wd.cancel()
    emit(out)


if __name__ == "__main__":
    main()
