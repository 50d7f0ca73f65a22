#!/usr/bin/env python3
"""bench.py -- CG iterations/second and achieved HBM GB/s of the matrix-free CG hot path.

A "step" is one CG iteration (fused stencil launch A' + fused update launch B).  Default workload = BASELINE
config 2: N x N = 4096 x 4096 intervals on the L-shaped domain, fp64, U = 12 574 721 unknowns, deterministic synthetic
RHS (the reference's f and Dirichlet data), x0 = 0, convergence tests disabled so that exactly K iterations are timed.

  python bench.py --gpus 1 --steps 2000 --warmup 200
  python bench.py --gpus N ...                               (no launcher needed: rank processes are started from here)
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...   (the driver's form: same result)
      weak scaling (default): every rank owns a config-2-sized part, the global grid grows with N
      --scaling strong --grid 32768: BASELINE config 5's fixed grid cut into N parts
      --decomp rows | 2d: row slabs, or (N/2) x 2 blocks (config 4's "2 x 2" at N = 4)

N > 1 (and MI355CG_BENCH_DIST=1 at N = 1): the measurement is a sequence of LEGS, each a fresh set of N rank processes (one per
GPU) started by a coordinator that never touches a GPU itself: the order-safe RCCL schedule first, then the mailbox + push
transport without and with the interior / edge split, then sub-records (config 5 strong, config 4's 2 x 2 at N = 4, two-stream
RCCL halo, one process driving N devices).
A leg that stalls is killed at its own time limit and noted; the legs that finished are never lost.  `value` = the best leg of
the requested configuration that passed its cross-check against one GPU; every leg is in `legs`.

Prints ONE JSON line (rank 0).  `value` = CG iterations/s of the whole job (weak scaling: in units of config-2-sized
parts advanced per second), the MEDIAN of R timed solves of exactly K iterations each (host clock between device
synchronisations, max over ranks); `hbm_gbps` = bytes the iteration really moves (58 B per unknown for the REL_2NORM loop) per
second; the per-kernel roofline comes from HIP events.
"""
from __future__ import annotations

import argparse
import json
import os
import signal
import subprocess
import sys
import tempfile
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


def weak_scaling_n(n1: int, world: int) -> int:
    """Grid size whose unknown count is ~world x that of n1 (even).  (= iterative_solvers_amd.distributed.weak_scaling_n, repeated
    here because the coordinator must not import anything that could touch a GPU.)"""
    return max(int(round(n1 * (world ** 0.5) / 2.0)) * 2, 6)


def median(v):
    s = sorted(v)
    return s[len(s) // 2] if len(s) % 2 else 0.5 * (s[len(s) // 2 - 1] + s[len(s) // 2])


def repeats(args, ms_per_step_guess: float) -> int:
    """R timed solves of K iterations each: 11 unless that would take more than ~3 s of device time."""
    if args.repeats > 0:
        return args.repeats
    per = max(1e-6, args.steps * ms_per_step_guess * 1e-3)
    return max(1, min(11, int(3.0 / per)))


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
        from oracle.oracle import host_cpu_share, mf_solve_all_cores
        # How many threads?  A GPU box shows all of the host's logical CPUs and grants a share of them (16 per GPU): one thread per
        # visible CPU is throttled to a crawl.  Candidates: the cgroup's quota where it can be read, and a few fixed counts; two
        # iterations each, the fastest one is then timed for about 10 s.
        visible = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
        cands = sorted({min(visible, c) for c in (host_cpu_share(), 8, 16, 32, 64, visible)})
        trials = {}
        mf_solve_all_cores(n, b, 0.0, 1, cands[0])                      # (library load, page faults of the first call)
        for th in cands:
            t0 = time.perf_counter()
            mf_solve_all_cores(n, b, 0.0, 2, th)
            trials[th] = (time.perf_counter() - t0) / 2
        best = min(trials, key=trials.get)
        per_it = trials[best]
        k = max(3, min(10 * iters, int(10.0 / max(per_it, 1e-4))))
        t0 = time.perf_counter()
        its, _, threads = mf_solve_all_cores(n, b, 0.0, k, best)
        dt2 = time.perf_counter() - t0
        out["all_cores"] = {"value": its / dt2, "unit": "iters/s", "cores": threads, "cpu_quota": host_cpu_share(), "kind": "port-openmp",
                            "sample": f"{its} iterations, {dt2:.1f} s, {threads} OpenMP threads ({visible} logical CPUs visible; seconds per iteration by thread count: "
                                      + ", ".join(f"{th}: {trials[th]:.3f}" for th in cands) + ")"}
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


# ---------------------------------------------------------------------------------------------------------------------
# N > 1: legs.  The coordinator (this process, or every torchrun worker) only starts and watches child processes.
# ---------------------------------------------------------------------------------------------------------------------
TRANSPORTS = {
    # name -> (environment of csrc/team.h's transport selection, what it is)
    "rccl-inline": ({"MI355CG_TEAM_RECORDS": "rccl", "MI355CG_TEAM_HALO": "inline", "MI355CG_TEAM_IPC": "0"},
                    "records: ncclAllGather, halo: one ncclSend/ncclRecv group, both on the compute stream of ONE communicator (order-safe)"),
    "mailbox+push": ({"MI355CG_TEAM_RECORDS": "auto", "MI355CG_TEAM_HALO": "auto"},
                     "records: a one-workgroup reducer launch beside the producer stores them straight into every OTHER rank's IPC-mapped mailbox; the consumer launch reduces its own partials and polls only for the others; "
                     "halo: pushed into the neighbours' ghost cells by one small launch + a stream-ordered flag (RCCL only bootstraps)"),
    "mailbox+push+split": ({"MI355CG_TEAM_RECORDS": "auto", "MI355CG_TEAM_HALO": "auto", "MI355CG_TEAM_SPLIT": "1"},
                           "as mailbox+push, with north_star's overlap: every phase is an interior launch and an edge launch, the halo travels (and is waited for) between them"),
    "mailbox+push+split-update": ({"MI355CG_TEAM_RECORDS": "auto", "MI355CG_TEAM_HALO": "auto", "MI355CG_TEAM_SPLIT": "2"},
                                  "as mailbox+push, with only the update phase in two launches: the edge rows are updated and pushed first, so the neighbours' rows are there when the next stencil launch -- ONE launch -- is due"),
    "rccl-stream": ({"MI355CG_TEAM_RECORDS": "rccl", "MI355CG_TEAM_HALO": "stream", "MI355CG_TEAM_IPC": "0"},
                    "records: ncclAllGather on the compute stream, halo: ncclSend/ncclRecv on a second stream + second communicator"),
    "mailbox+rccl-halo": ({"MI355CG_TEAM_RECORDS": "auto", "MI355CG_TEAM_HALO": "inline"},
                          "records through the mailboxes, halo as one RCCL group on the compute stream"),
}


def plan_legs(args):
    """The legs of one invocation, in the order they run.  `headline` legs measure the requested configuration."""
    N = args.gpus
    base = {"scaling": args.scaling, "decomp": args.decomp, "grid": args.n, "driver": "ranks"}
    legs = [dict(base, name="rccl-inline", transport="rccl-inline", headline=True),
            dict(base, name="mailbox+push", transport="mailbox+push", headline=True),
            dict(base, name="mailbox+push+split-update", transport="mailbox+push+split-update", headline=True),
            dict(base, name="mailbox+push+split", transport="mailbox+push+split", headline=True)]
    # sub-records: BASELINE's own configurations first, then the other transports
    if not (args.scaling == "strong" and args.n == 32768):
        legs.append(dict(base, name="config5-strong-32768", transport="mailbox+push", headline=False, scaling="strong", grid=32768, decomp="rows", verify=0))
    if N == 4 and not (args.decomp == "2d" and args.scaling == "strong" and args.n == 16384):
        legs.append(dict(base, name="config4-2x2-16384", transport="mailbox+push", headline=False, scaling="strong", grid=16384, decomp="2d"))
    legs.append(dict(base, name="rccl-stream", transport="rccl-stream", headline=False))
    if N > 1:          # last: in the torchrun form only rank 0's coordinator runs it, while the others are already done
        legs.append(dict(base, name="local-one-process", transport="local", headline=False, driver="local"))
    if args.legs == "default":
        legs = legs[:2]
    elif args.legs != "all":
        want = args.legs.split(",")
        legs = [l for l in legs if l["name"] in want] + [dict(base, name=w, transport=w, headline=True) for w in want if w in TRANSPORTS and w not in [l["name"] for l in legs]]
    return legs


def run_leg_child(spec, args):
    """One rank of one leg (a fresh process: RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* in the environment)."""
    import datetime
    import torch
    import torch.distributed as dist
    import iterative_solvers_amd as isa
    from iterative_solvers_amd import _capi
    from iterative_solvers_amd import distributed as D

    world, rank, local_rank = int(os.environ["WORLD_SIZE"]), int(os.environ["RANK"]), int(os.environ["LOCAL_RANK"])
    local = spec["driver"] == "local"
    nparts = args.gpus if local else world
    # Rehearsal on a one-GPU box (MI355CG_BENCH_ONE_GPU=1 + MI355CG_RCCL_LIB = the tests' nccl stand-in, because RCCL refuses two
    # ranks on one device): every rank uses device 0; everything else -- coordinators, legs, clocks, cross-check -- is the real thing.
    dev = 0 if os.environ.get("MI355CG_BENCH_ONE_GPU") == "1" else local_rank
    torch.cuda.set_device(dev)
    open(f"{args.child_out}.r{rank}.ready", "w").close()             # imports done: the leg's clock starts when every rank is here
    if not local:
        dist.init_process_group("gloo", timeout=datetime.timedelta(seconds=120))       # control plane only: the id, barriers, max over ranks
    rule = _capi.RULE_REL_2NORM if args.rule == "rel2" else _capi.RULE_MSG_MAXNORM
    strong = spec["scaling"] == "strong"
    n = spec["grid"] if strong else weak_scaling_n(spec["grid"], nparts)
    U1, U = unknowns(spec["grid"]), unknowns(n)                      # weak: a part is as large as the --grid problem (config 2 by default)
    decomp = _capi.DECOMP_2D if spec["decomp"] == "2d" else _capi.DECOMP_ROWS
    boxes = D.decompose(n, nparts, decomp)
    ndev = torch.cuda.device_count()
    if local:
        team = D.Team.local(n, nparts, decomp, devices=list(range(min(ndev, nparts))))
    else:
        team = D.Team.rccl(n, decomp, device=dev)
    desc = team.describe()
    if not local and desc["rccl_nranks"] != args.gpus:
        raise RuntimeError(f"RCCL reports {desc['rccl_nranks']} ranks, --gpus asked for {args.gpus}")

    def barrier():
        torch.cuda.synchronize()
        if not local:
            dist.barrier()

    def run(iters):
        p = isa.default_params(rule)
        p.max_iterations, p.fixed_iterations, p.use_true_solution, p.callback_every, p.sync_every = iters, 1, 0, 0, 500
        return team.solve(p)

    run(args.warmup)
    times = []
    for _ in range(repeats(args, 0.14 * max(1.0, U / nparts / 12.6e6))):
        # K iterations between barrier + device synchronisation on both sides.  Every rank reads its clock when ITS device is done
        # and before it enters the closing barrier (a gloo barrier over TCP costs 0.2 - 1 ms, a tenth of a 20-iteration window and
        # no part of the solve); the job's time is the maximum over the ranks.
        barrier()
        if not local and world > 1:
            # (untimed) a gloo barrier over TCP lets the ranks go up to a millisecond apart -- a third of a 20-iteration window, which
            # the ranks that left first would spend waiting for the last one's records.  A one-iteration solve aligns them: its last
            # records are a barrier on the devices, and every host sees its own copy within microseconds.
            run(1)
            torch.cuda.synchronize()
        t0 = time.perf_counter()
        res = run(args.steps)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        if not local:
            dist.barrier()
        dt = torch.tensor([t1 - t0], dtype=torch.float64)
        if not local:
            dist.all_reduce(dt, op=dist.ReduceOp.MAX)
        times.append(float(dt.item()))
        assert res.iterations == args.steps
    dt = median(times)
    its = args.steps / dt
    team.set_profiling(True)
    run(min(args.steps, 200))
    team.set_profiling(False)
    phases = team.phase_times()
    phases["driver_and_wait_ms"] = max(0.0, phases["wall_ms"] - phases["kernels_ms"])
    # Cross-check of the distributed loop (untimed): V iterations on the team against the SAME global problem solved by ONE context
    # on rank 0's GPU.  The team's reductions are summed part by part in a fixed order, so the residual norm has to agree to
    # rounding of the last bit or two; a halo row that arrived late or in the wrong place shows up in the leading digits.
    verify = None
    nverify = spec.get("verify", args.verify)
    if nverify > 0 and U <= args.verify_max_unknowns:
        rv = run(nverify)
        torch.cuda.synchronize()
        if rank == 0:
            one = isa.MatrixFreeSystem(n, n, 1.0, 2.0, 1.0, 2.0, device=dev)
            p = isa.default_params(rule)
            p.max_iterations, p.fixed_iterations, p.use_true_solution, p.callback_every, p.sync_every = nverify, 1, 0, 0, 500
            r1 = one._handle.solve(p)
            one._handle.close()
            rel = abs(rv.r_norm2 - r1.r_norm2) / max(abs(r1.r_norm2), 1e-300)
            verify = {"iterations": nverify, "team_r_norm2": rv.r_norm2, "single_gpu_r_norm2": r1.r_norm2,
                      "rel_diff": rel, "bit_identical": bool(rv.r_norm2 == r1.r_norm2), "ok": bool(rel <= 1e-12 and rv.iterations == r1.iterations)}
        barrier()
    elif nverify > 0:
        verify = {"skipped": f"{U} unknowns > --verify-max-unknowns {args.verify_max_unknowns}"}
    bytes_it = 8.0 * sum(WORDS[args.rule].values())
    units = 1.0 if strong else U / U1
    moved = bytes_it * U * its / 1e9 / nparts
    out = {
        "leg": spec["name"], "value": round(its * units, 2), "n_gpus": nparts, "global_iters_per_sec": round(its, 2),
        "ms_per_step": round(1e3 * dt / args.steps, 5), "repeats": len(times),
        "ms_per_step_min_max": [round(1e3 * min(times) / args.steps, 5), round(1e3 * max(times) / args.steps, 5)],
        "scaling": "strong" if strong else "weak", "n": n, "unknowns": U, "unknowns_per_gpu": U / nparts,
        "decomposition": {"kind": spec["decomp"], "parts": boxes},
        "transport": desc, "processes": 1 if local else world, "devices_used": min(ndev, nparts) if local else world,
        "hbm_gbps": round(bytes_it * U * its / 1e9, 1), "per_gpu_gbps": round(moved, 1), "per_gpu_frac_of_8000": round(moved / HBM_PEAK_GBPS, 4),
        "phases_ms": phases, "verify_against_one_gpu": verify,
        # the lead part's two launches alone (HIP events of the profiled pass): what the GPU does while it is not waiting for anybody
        "kernels_only_gbps_per_gpu": round(bytes_it * U / nparts / (phases["kernels_ms"] * 1e-3) / 1e9, 1) if phases.get("kernels_ms", 0) > 0 else None,
    }
    team.close()
    if not local:
        dist.barrier()
        dist.destroy_process_group()
    return out


def child_main(args):
    spec = json.loads(args.child_leg)
    rank = int(os.environ.get("RANK", "0"))
    try:
        out = run_leg_child(spec, args)
    except BaseException as e:                                       # noqa: BLE001 -- the coordinator wants the reason, whatever it was
        out = {"leg": spec["name"], "error": f"rank {rank}: {type(e).__name__}: {str(e)[:300]}"}
        with open(args.child_out + f".r{rank}", "w") as f:
            json.dump(out, f)
        raise
    if rank == 0:
        with open(args.child_out + ".r0", "w") as f:
            json.dump(out, f)


def leg_command(args, spec, out_path):
    cmd = [sys.executable, os.path.abspath(__file__), "--gpus", str(args.gpus), "--steps", str(args.steps), "--warmup", str(args.warmup),
           "--grid", str(args.n), "--rule", args.rule, "--verify", str(args.verify), "--verify-max-unknowns", str(args.verify_max_unknowns),
           "--repeats", str(args.repeats), "--leg-timeout", str(args.leg_timeout), "--import-allowance", str(args.import_allowance),
           "--child-leg", json.dumps(spec), "--child-out", out_path]
    return cmd


def leg_env(base_env, spec, rank, world, local_rank, port):
    env = dict(base_env)
    rehearsal_wait = env.get("MI355CG_TEAM_WAIT") if env.get("MI355CG_BENCH_ONE_GPU") == "1" else None
    for k in ("MI355CG_TEAM_RECORDS", "MI355CG_TEAM_HALO", "MI355CG_TEAM_IPC", "MI355CG_TEAM_WAIT", "MI355CG_TEAM_SPLIT", "MI355CG_TEAM_HALO_INLINE"):
        env.pop(k, None)
    if rehearsal_wait:                                  # one-GPU rehearsal of what ranks on GPUs of their own do by default: kernels poll their mailboxes
        env["MI355CG_TEAM_WAIT"] = rehearsal_wait
    if spec["transport"] in TRANSPORTS:
        env.update(TRANSPORTS[spec["transport"]][0])
    env.update({"RANK": str(rank), "WORLD_SIZE": str(world), "LOCAL_RANK": str(local_rank), "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port),
                "MI355CG_BENCH_CHILD": "1", "GLOO_SOCKET_IFNAME": env.get("GLOO_SOCKET_IFNAME", "lo"), "HSA_ENABLE_IPC_MODE_LEGACY": env.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")})
    env.setdefault("MI355CG_TEAM_TIMEOUT_MS", "8000")   # a kernel gives a missing record 8 s, the host the stream behind it 24 s: a leg that cannot work costs half a minute, not two
    env.setdefault("GPU_MAX_HW_QUEUES", "8")           # compute, comm and side stream (+ torch's, + RCCL's) each on a hardware queue of its own: the reducer launch runs BESIDE the producer
    for k in list(env):
        if k.startswith("TORCHELASTIC") or k in ("GROUP_RANK", "ROLE_RANK", "LOCAL_WORLD_SIZE", "GROUP_WORLD_SIZE", "ROLE_WORLD_SIZE", "TORCH_NCCL_ASYNC_ERROR_HANDLING"):
            env.pop(k)
    return env


class _Terminated(Exception):
    pass


def _write_json(path, obj):
    tmp = f"{path}.tmp{os.getpid()}"
    with open(tmp, "w") as f:
        json.dump(obj, f)
    os.replace(tmp, path)                                          # readers see the whole file or none of it


def _read_json(path):
    try:
        with open(path) as f:
            return json.load(f)
    except (OSError, ValueError):
        return None


def meeting_dir(launched, world, now):
    """Where the coordinators of one invocation meet.  Without a launcher there is one coordinator and the directory is its own.
    Under torchrun every worker is the coordinator of ONE rank; all of them run on this node (--nnodes=1), so they agree on a
    directory named after what the launcher gave all of them alike, and rank 0's coordinator (the lead) opens the session there.
    A session file left by an earlier invocation with the same port is told by its age."""
    if not launched:
        return tempfile.mkdtemp(prefix="mi355cg_bench_")
    d = os.path.join(tempfile.gettempdir(), "mi355cg_bench_%s_%s_w%d" % (os.environ.get("TORCHELASTIC_RUN_ID", "none"), os.environ.get("MASTER_PORT", "0"), world))
    os.makedirs(d, exist_ok=True)
    return d


def coordinate(args):
    """Start the rank processes of every leg, watch them, collect rank 0's record.  Touches no GPU (imports no torch).

    Time: the driver kills a bench run after 600 s, and a killed run leaves no line.  So everything here ends by `--budget` seconds
    (420) after the start, whatever the legs do: a leg gets `--leg-timeout` seconds from the moment all its ranks have finished
    importing (the first import of torch on a fresh box can take two minutes), never more than what is left of the budget; a leg is
    not started with less than 45 s left.  SIGTERM ends the current leg and still prints the line.

    Under torchrun the coordinators act in step through files in a directory they share (same node): the lead says for every leg
    whether it is run and until when, every coordinator says when its rank is ready or has failed, the lead says when the leg is
    over -- so no coordinator waits for ranks another one never started, or keeps a rank alive that the lead has given up on."""
    t_start = time.time()
    launched = int(os.environ.get("WORLD_SIZE", "1")) > 1          # torchrun (the driver's N > 1 form): this process is ONE rank's coordinator
    world = args.gpus
    if launched and int(os.environ["WORLD_SIZE"]) != args.gpus:
        raise SystemExit(f"bench.py: the launcher started {os.environ['WORLD_SIZE']} ranks but --gpus is {args.gpus}")
    my_ranks = [int(os.environ["RANK"])] if launched else list(range(world))
    local_of = {r: (int(os.environ["LOCAL_RANK"]) if launched else r) for r in my_ranks}
    lead = 0 in my_ranks
    legs = plan_legs(args)
    results, notes = {}, []
    t_end = t_start + args.budget
    meet = meeting_dir(launched, world, t_start)
    session = os.path.join(meet, "session")
    if lead:
        for f in os.listdir(meet):                                  # whatever an earlier invocation left here
            try:
                os.unlink(os.path.join(meet, f))
            except OSError:
                pass
        _write_json(session, {"t_start": t_start, "pid": os.getpid()})
        lead_start = t_start
    else:
        while time.time() < min(t_end, t_start + 90.0):
            got = _read_json(session)
            if got and abs(got["t_start"] - t_start) < 60.0:
                t_end = got["t_start"] + args.budget                # one clock for all: the lead's
                lead_start = got["t_start"]
                break
            time.sleep(0.05)
        else:
            return None                                             # no lead (it failed before it got here): nothing to take part in

    # The legs' rendezvous ports: above the launcher's own (MASTER_PORT stays its store's), and not the ones an invocation a few seconds
    # ago used -- the driver runs N = 1, 2, 4, 8 back to back with one --master-port, and a port that has only just been let go of made a
    # rendezvous take a minute in the tests.  Every coordinator derives the same numbers from the lead's start time.
    base_port = int(os.environ.get("MASTER_PORT", "29531")) + 101 + (int(lead_start) % 97) * 8

    def on_term(signum, frame):
        raise _Terminated()
    import threading
    hook = threading.current_thread() is threading.main_thread()     # every coordinator: its rank process runs in a session of its own and has to be ended by hand
    old_term = signal.signal(signal.SIGTERM, on_term) if hook else None

    def kill(procs):
        for r, p, log in procs:                                     # exactly the process groups started here
            if p.poll() is None:
                try:
                    os.killpg(p.pid, signal.SIGKILL)
                except ProcessLookupError:
                    pass
        for r, p, log in procs:
            try:
                p.wait(timeout=10)
            except subprocess.TimeoutExpired:
                pass
            if not log.closed:
                log.close()

    procs = []
    mailbox_broken = None                                           # the first leg of the mailbox transport that stalled or failed
    try:
        for li, spec in enumerate(legs):
            local = spec["driver"] == "local"
            if local and not lead:
                continue
            tag = os.path.join(meet, f"leg{li}")
            # ---- go or skip: the lead's call, the same for every coordinator
            if lead:
                left = t_end - time.time()
                go = left >= min(45.0, args.leg_timeout) or not results
                if not go:
                    notes.append(f"leg {spec['name']} not started: {left:.0f} s left of the {args.budget:.0f} s budget")
                # Once the mailbox transport has failed on this node (a leg that stalled or ended in an error), its variants would only
                # repeat that at a leg's time limit each: the other headline variants are skipped, and the sub-records (BASELINE's own
                # configurations) run on the order-safe RCCL schedule instead.
                swap = None
                if go and mailbox_broken and spec["transport"].startswith("mailbox+push") and not local:
                    if spec["headline"]:
                        go = False
                        notes.append(f"leg {spec['name']} not started: the mailbox transport failed in leg {mailbox_broken}")
                    else:
                        swap = "rccl-inline"
                        notes.append(f"leg {spec['name']} runs on rccl-inline: the mailbox transport failed in leg {mailbox_broken}")
                _write_json(tag + ".go", {"go": go, "transport": swap})
            else:
                got = None
                while got is None and time.time() < t_end:
                    got = _read_json(tag + ".go")
                    if got is None:
                        time.sleep(0.02)
                go = bool(got and got["go"])
                if got is None:
                    return None
                swap = got.get("transport")
            if not go:
                continue
            if swap:
                spec = dict(spec, transport=swap)
            ranks = [0] if local else my_ranks
            need = [0] if local else list(range(world))
            out_path = tag
            procs = []
            for r in ranks:
                env = leg_env(os.environ, spec, 0 if local else r, 1 if local else world, local_of.get(r, 0), base_port + li)
                log = open(f"{tag}.r{r}.log", "w")
                procs.append((r, subprocess.Popen(leg_command(args, spec, out_path), env=env, stdout=log, stderr=subprocess.STDOUT, start_new_session=True), log))
            # ---- watch: until all of this coordinator's ranks are gone, the deadline passes, or the lead declares the leg over
            deadline = min(t_end, time.time() + args.import_allowance + args.leg_timeout)
            armed = failed = False
            over = None
            while time.time() < deadline:
                alive = [p for p in procs if p[1].poll() is None]
                for r, p, log in procs:
                    if p.returncode not in (None, 0) and not os.path.exists(f"{tag}.fail.r{r}"):
                        _write_json(f"{tag}.fail.r{r}", {"rc": p.returncode})
                if not alive:
                    break
                if not failed and any(os.path.exists(f"{tag}.fail.r{r}") for r in need):
                    failed = True
                    deadline = min(deadline, time.time() + 15.0)     # a rank died: the others are waiting for it in vain
                if not armed:
                    if lead and all(os.path.exists(f"{out_path}.r{r}.ready") for r in need):
                        _write_json(tag + ".armed", {"deadline": min(t_end, time.time() + args.leg_timeout)})
                    got = _read_json(tag + ".armed")
                    if got:
                        armed = True
                        deadline = min(deadline, got["deadline"]) if failed else got["deadline"]
                if not lead and over is None and os.path.exists(tag + ".over"):
                    over = time.time()
                    deadline = min(deadline, over + 10.0)             # the lead is done with this leg: a rank still running is stuck
                time.sleep(0.05)
            stalled = any(p[1].poll() is None for p in procs)
            kill(procs)
            if lead:
                _write_json(tag + ".over", {"stalled": stalled})
            rec = None
            for r in need:
                got = _read_json(f"{out_path}.r{r}")
                if got is not None and (r == 0 or "error" in got):
                    rec = got if rec is None or "error" in got else rec
            if lead:
                if rec is None:
                    tail = ""
                    try:
                        with open(f"{tag}.r0.log") as fh:
                            tail = fh.read()[-400:]
                    except OSError:
                        pass
                    why = "the imports" if not armed else "%.0f s" % args.leg_timeout
                    rec = {"leg": spec["name"], "error": (f"stalled: killed after {why} ({time.time() - t_start:.0f} s into the run)") if stalled and not failed
                           else f"no record (rank exit codes {[p[1].returncode for p in procs]}): {tail}"}
                elif stalled and "error" not in rec:
                    rec["note"] = "some ranks had to be killed after the record was written"
                rec["what"] = TRANSPORTS.get(spec["transport"], (None, "one process drives all parts (LOCAL transport: peer access, one host thread per device)"))[1]
                rec["headline_candidate"] = bool(spec["headline"])
                results[spec["name"]] = rec
                if "error" in rec and spec["transport"].startswith("mailbox+push") and not local and not mailbox_broken:
                    mailbox_broken = spec["name"]
            procs = []
    except _Terminated:
        kill(procs)
        for lj in range(len(legs) if lead else 0):                   # the other coordinators: this leg is over, no later one is run
            for suffix, obj in ((".over", {"stalled": True}), (".go", {"go": False})):
                if not os.path.exists(os.path.join(meet, f"leg{lj}{suffix}")):
                    _write_json(os.path.join(meet, f"leg{lj}{suffix}"), obj)
        notes.append(f"SIGTERM {time.time() - t_start:.0f} s into the run: the legs not listed were not run")
    finally:
        if hook:
            signal.signal(signal.SIGTERM, old_term)
    if not lead:
        return None
    coordinate.elapsed_s = round(time.time() - t_start, 1)
    return legs, results, notes


def compose(args, legs, results, notes):
    cands = [results[l["name"]] for l in legs if l["headline"] and l["name"] in results and "error" not in results[l["name"]]]
    verified = [r for r in cands if not r.get("verify_against_one_gpu") or r["verify_against_one_gpu"].get("ok", True)]
    best = max(verified or cands, key=lambda r: r["value"]) if (verified or cands) else None
    strong = args.scaling == "strong"
    out = {"metric": "cg_iters_per_sec", "value": best["value"] if best else None, "unit": "iters/s", "n_gpus": args.gpus,
           "steps": args.steps, "warmup": args.warmup, "ms_per_step": best["ms_per_step"] if best else None, "higher_is_better": True,
           "scaling": "strong" if strong else "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic"}
    if best:
        n, U = best["n"], best["unknowns"]
        U1 = unknowns(args.n)
        out["config"] = {
            "workload": (f"{n}x{n} L-shaped Dirichlet Poisson fp64 cut into {args.gpus} parts, matrix-free CG, fixed {args.steps} iterations" if strong else
                         f"{n}x{n} L-shaped Dirichlet Poisson fp64 over {args.gpus} parts of ~{U1} unknowns (config-2 size per GPU), matrix-free CG, fixed {args.steps} iterations"),
            "n": n, "unknowns": U, "unknowns_per_gpu": U / args.gpus, "rule": args.rule,
            "value_is": ("global CG iterations/s" if strong else "global CG iterations/s x (unknowns / config-2 unknowns) = config-2-sized part iterations/s") +
                        f"; median of {best['repeats']} timed solves of {args.steps} iterations, max over ranks",
            "decomposition": best["decomposition"],
            "parallelism": f"{'(N/2) x 2 blocks' if args.decomp == '2d' else 'row slabs'} x{args.gpus}, one process per GPU, leg '{best['leg']}': {best['what']}"}
        out["headline_leg"] = best["leg"]
        out["rccl_nranks"] = best["transport"].get("rccl_nranks")
        out["global_iters_per_sec"] = best["global_iters_per_sec"]
        out["ms_per_step_min_max"] = best["ms_per_step_min_max"]
        out["hbm_gbps"] = best["hbm_gbps"]
        out["hbm_gbps_is"] = f"bytes really moved, summed over GPUs: {8.0 * sum(WORDS[args.rule].values()):.0f} B per unknown per iteration"
        out["algorithmic_equivalent_gbps_88B"] = round(SURVEY_BYTES_PER_UNKNOWN * U * best["global_iters_per_sec"] / 1e9, 1)
        out["phases_ms"] = best["phases_ms"]
        out["verify_against_one_gpu"] = best["verify_against_one_gpu"]
        # whole-iteration roofline per GPU (kernels + record hops + halo + driver); the per-kernel figures are in the 1-GPU bench line
        out["roofline"] = {"bound": "hbm", "achieved": best["per_gpu_gbps"], "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": best["per_gpu_frac_of_8000"],
                           "traffic": None, "scope": "bytes one GPU has to move per iteration / wall time per iteration, record hops and halo included"}
    else:
        out["error"] = "no leg of the requested configuration produced a record"
    out["legs"] = results
    out["coordinator"] = {"seconds": getattr(coordinate, "elapsed_s", None), "budget_s": args.budget, "leg_timeout_s": args.leg_timeout, "import_allowance_s": args.import_allowance}
    if notes:
        out["notes"] = notes
    return out


# ---------------------------------------------------------------------------------------------------------------------
def bench_one_gpu(args, emit):
    import torch
    import iterative_solvers_amd as isa
    from iterative_solvers_amd import _capi

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    rule = _capi.RULE_REL_2NORM if args.rule == "rel2" else _capi.RULE_MSG_MAXNORM
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
    # R timed solves of exactly K iterations each (plus the initialisation pass of a solve), back to back; the median is the value
    times, loops = [], []
    for _ in range(repeats(args, 0.14 * max(1.0, U / 12.6e6) * (0.5 if f32 else 1.0))):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        res = run(args.steps, False)
        torch.cuda.synchronize()
        times.append(time.perf_counter() - t0)
        loops.append(res.loop_seconds)
        assert res.iterations == args.steps, (res.iterations, args.steps)
    dt = median(times)
    its = args.steps / dt
    loop = median(loops)
    its_loop = args.steps / loop if loop > 0 else None        # HIP events around the K iterations alone

    words = WORDS[args.rule]
    words_iter = sum(words.values())
    roofline = None
    if not args.no_roofline_pass:
        # same loop again with a HIP-event pair around every launch on the solve stream
        k = min(max(args.steps, 200), 500)
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
                    "traffic": (tr or {}).get(dom),
                    "traffic_source": "profiles/traffic.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of an earlier run of this command (gfx950 correction applied), not measured in this run" if tr else None,
                    "avg_ms": round(ms, 5), "launches": launches,
                    "avg_ms_is": "mean of HIP-event pairs around each launch on the solve stream (the pairs add a little: their sum exceeds loop_only_ms_per_step)",
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
        "timing": f"median of {len(times)} back-to-back solves of K iterations each, host clock between device synchronisations (each includes the solve's "
                  "initialisation pass: one pass x = 0, r = b, z = 0, ||r0||, and the last poll ~ 0.2 ms); loop_only_* = HIP events around the K iterations on the solve stream",
        "repeats": len(times),
        "value_min_max": [round(args.steps / max(times), 2), round(args.steps / min(times), 2)],
        "loop_only_iters_per_sec": round(its_loop, 2) if its_loop else None,
        "loop_only_ms_per_step": round(1e3 * loop / args.steps, 5) if its_loop else None,
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
    emit(out)


def main():
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
    ap.add_argument("--repeats", type=int, default=0, help="timed solves of K iterations each, the median is reported (0 = 11, fewer for long solves)")
    ap.add_argument("--verify", type=int, default=30, help="N > 1: iterations of the untimed cross-check against one GPU (0 = skip)")
    ap.add_argument("--verify-max-unknowns", type=float, default=2.6e8, help="skip that cross-check above this size (host set-up time)")
    ap.add_argument("--legs", default="all", help="N > 1: all | default (the two headline transports only) | comma-separated leg names")
    ap.add_argument("--leg-timeout", type=float, default=110.0, help="seconds a leg gets once all its ranks have finished importing; then its processes are killed")
    ap.add_argument("--import-allowance", type=float, default=150.0, help="seconds the ranks of a leg may take to start up (first import of torch on a fresh box)")
    ap.add_argument("--budget", type=float, default=420.0, help="seconds after which the coordinator has ended every leg and prints (the driver kills a run after 600 s)")
    ap.add_argument("--child-leg", default=None, help=argparse.SUPPRESS)
    ap.add_argument("--child-out", default=None, help=argparse.SUPPRESS)
    args = ap.parse_args()

    if args.child_leg:                                            # a rank process of one leg
        import threading
        # Die with the coordinator, however it dies: no rank process may outlive an invocation and sit on a GPU (it runs in a session of
        # its own, so signals to the launcher's process group pass it by).  The parent-death signal where the kernel delivers it, and a
        # look at the parent every two seconds where it does not (it did not in the build container).
        ppid0 = os.getppid()
        try:
            import ctypes
            ctypes.CDLL(None).prctl(1, int(signal.SIGKILL))       # PR_SET_PDEATHSIG
        except Exception:
            pass

        def orphan_watch():
            while True:
                time.sleep(2.0)
                if os.getppid() != ppid0:
                    os._exit(3)
        threading.Thread(target=orphan_watch, daemon=True).start()
        wd = threading.Timer(args.import_allowance + args.leg_timeout + 30.0, lambda: os._exit(3))     # last resort: never outlive the coordinator's patience
        wd.daemon = True
        wd.start()
        child_main(args)
        return

    # ONE JSON line on stdout: libraries chat there too (RCCL prints a version banner when its first communicator comes up),
    # so everything else this process writes to fd 1 goes to stderr until the line is printed.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    def emit(obj):
        sys.stdout.flush()
        os.dup2(real_stdout, 1)
        print(json.dumps(obj), flush=True)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1 or args.gpus > 1 or os.environ.get("MI355CG_BENCH_DIST") == "1":
        got = coordinate(args)                                     # before anything could touch a GPU in this process
        if got is None:
            return                                                 # a non-zero rank's coordinator: rank 0 prints
        out = compose(args, *got)
        assert out["n_gpus"] == args.gpus
        emit(out)
        if out.get("value") is None:
            sys.exit(4)
        return
    bench_one_gpu(args, emit)


if __name__ == "__main__":
    main()
