"""GPU tests of the ONE-PROCESS-PER-RANK transport of csrc/team.h (mi355cg_team_create_rccl) at world 2 .. 5 on the one GPU
of the test box.  RCCL refuses several ranks on one device, so the rank processes talk through tests/nccl_shim (a host-staged
stand-in for the eleven nccl* entry points team.h resolves; MI355CG_RCCL_LIB) -- everything else is the product path:
communicator bootstrap, the exchange of IPC handles and the mapping of the other ranks' mailboxes / residual vectors / column
buffers (real hipIpc* between real processes), records through mailboxes or ncclAllGather, the halo as a push into the
neighbours' memory or as grouped ncclSend / ncclRecv (on the compute stream, or on a second stream + second communicator),
interior / edge launches, a stop request raised by ONE rank, a callback passed on ONE rank.  Every combination has to reproduce
the single-context solve bit for bit.  (A box allows six processes on its GPU: the pytest process + at most five ranks, so
world = 8 cannot be rehearsed this way; LOCAL teams of 8 and 16 parts are in test_gpu_team.py.)"""
import ctypes as C
import os
import sys
import tempfile
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# name -> environment of the transport (team.h: team_pick_modes)
MODES = {
    "rccl+inline": {"MI355CG_TEAM_RECORDS": "rccl", "MI355CG_TEAM_HALO": "inline"},                 # the order-safe schedule: one communicator, one stream
    "rccl+stream": {"MI355CG_TEAM_RECORDS": "rccl", "MI355CG_TEAM_HALO": "stream"},                 # halo on the second stream + second communicator
    "rccl+stream+split": {"MI355CG_TEAM_RECORDS": "rccl", "MI355CG_TEAM_HALO": "stream", "MI355CG_TEAM_SPLIT": "1"},
    "mailbox+push": {},                                                                              # the default (auto): IPC mailboxes, pushed halo
    "mailbox+push+split": {"MI355CG_TEAM_SPLIT": "1"},
    "mailbox+push+split-update": {"MI355CG_TEAM_SPLIT": "2"},                                        # only the update phase in two launches: the rows leave early
    "mailbox+inline": {"MI355CG_TEAM_HALO": "inline"},
    "rccl+push": {"MI355CG_TEAM_RECORDS": "rccl", "MI355CG_TEAM_HALO": "push"},
    "no-ipc": {"MI355CG_TEAM_IPC": "0"},
    "too-large-to-map": {"MI355CG_TEAM_IPC_MAX_GIB": "0"},                                           # vectors above the limit (2 GiB; here: any) are not IPC-mapped -- mapping 3 GiB never returned: RCCL for both                                                             # the ranks cannot map each other: falls back to RCCL for both
    # what ranks on GPUs of their OWN do (the default there): no stream-level waits for records, the consumer launches poll their mailboxes
    "mailbox+push+kernel-wait": {"MI355CG_TEAM_WAIT": "kernel"},
}
EXPECT = {
    "rccl+inline": ("rccl", "rccl-inline", 0), "rccl+stream": ("rccl", "rccl-stream", 0), "rccl+stream+split": ("rccl", "rccl-stream", 1),
    "mailbox+push": ("mailbox", "push", 0), "mailbox+push+split": ("mailbox", "push", 1), "mailbox+push+split-update": ("mailbox", "push", 2), "mailbox+inline": ("mailbox", "rccl-inline", 0),
    "rccl+push": ("rccl", "push", 0), "no-ipc": ("rccl", "rccl-inline", 0), "too-large-to-map": ("rccl", "rccl-inline", 0), "mailbox+push+kernel-wait": ("mailbox", "push", 0),
}
for _name in filter(None, os.environ.get("MI355CG_TEST_EXTRA_MODES", "").split(",")):      # tools/dbg_modes.py: "mailbox+push#3" = a further run of that mode
    MODES[_name], EXPECT[_name] = MODES[_name.split("#")[0]], EXPECT[_name.split("#")[0]]
KEYS = ("MI355CG_TEAM_RECORDS", "MI355CG_TEAM_HALO", "MI355CG_TEAM_SPLIT", "MI355CG_TEAM_IPC", "MI355CG_TEAM_WAIT", "MI355CG_TEAM_IPC_MAX_GIB")


def _params(isa, rule, **kw):
    p = isa.default_params(rule)
    for k, v in kw.items():
        setattr(p, k, v)
    return p


REL = dict(eps_rel=1e-8, max_iterations=10 ** 5)
MSG = dict(eps_precision=1e-9, eps_residual=1e-9, eps_exact_error=-1.0, max_iterations=10 ** 5)


def _worker(rank, world, port, n, decomp, modes, outdir):
    sys.path.insert(0, ROOT)
    real = os.environ.get("MI355CG_TEST_REAL_RCCL") == "1"             # one GPU per rank and librccl itself (boxes with several GPUs)
    if not real:
        from tests import nccl_shim
        os.environ["MI355CG_RCCL_LIB"] = nccl_shim.build()
    os.environ["NCCL_SHIM_TIMEOUT_MS"] = "60000"
    os.environ["MI355CG_TEAM_TIMEOUT_MS"] = "20000"
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    out = {}
    try:
        import iterative_solvers_amd as isa
        from iterative_solvers_amd.distributed import Team
        dev = rank if real else 0
        torch.cuda.set_device(dev)
        step = "start"
        for mode in modes:
            try:
                for k in KEYS:
                    os.environ.pop(k, None)
                os.environ.update(MODES[mode])
                step = "create"
                t_mode = time.time()
                t = Team.rccl(n, decomp, device=dev)
                d = t.describe()
                out[f"{mode}/desc"] = np.array([d["records"], d["wait"], d["halo"], str(d["split"]), str(d["ipc"]), str(d["shared_device"]), str(d["rccl_nranks"]), d["rccl_lib"], d["ipc_note"]])
                step = "(1) REL_2NORM to convergence"
                r = t.solve(_params(isa, 1, **REL))
                out[f"{mode}/rel"] = np.array([r.iterations, r.converged, r.stop_reason, r.r_norm2, r.initial_r_norm2])
                out[f"{mode}/x"] = t.vector(0)
                out[f"{mode}/cs"] = np.array(t.checksum(1))
                step = "(2) MSG rule; only rank 0 passes a callback (the chunk schedule must not depend on it)"
                cbs = []
                r = t.solve(_params(isa, 0, **MSG), callback=(lambda *a: cbs.append(a)) if rank == 0 else None)
                out[f"{mode}/msg"] = np.array([r.iterations, r.converged, r.stop_reason, r.final_residual_norm, r.final_precision])
                out[f"{mode}/cbs"] = np.array(cbs, dtype=float).reshape(-1, 4)
                out[f"{mode}/xm"] = t.vector(0)
                step = "(3) a stop request raised by the LAST rank only, before the solve: every rank ends INTERRUPTED after the same iteration"
                stop = C.c_int(1 if rank == world - 1 else 0)
                r = t.solve(_params(isa, 0, **MSG), stop_flag=stop if rank == world - 1 else None)
                out[f"{mode}/stop0"] = np.array([r.iterations, r.converged, r.stop_reason])
                step = "(4) ... and raised by rank 1 from ITS it = 1 callback, in the middle of the solve"
                stop = C.c_int(0)
                if rank == 1:
                    r = t.solve(_params(isa, 0, **MSG), callback=lambda it, *a: stop.__setattr__("value", 1 if it >= 1 else 0), stop_flag=stop)
                else:
                    r = t.solve(_params(isa, 0, **MSG))
                out[f"{mode}/stop1"] = np.array([r.iterations, r.converged, r.stop_reason])
                step = "(5) the team still solves after all that"
                r = t.solve(_params(isa, 1, eps_rel=1e-8, max_iterations=37, fixed_iterations=1))
                out[f"{mode}/fixed"] = np.array([r.iterations, r.r_norm2])
                t_mixed = time.time()
                if decomp == 0 and (world <= 3 or mode == "mailbox+push"):       # (five ranks through the host-staged stand-in: 8 s per RCCL mode)
                    step = "(6) F32_MIXED on the team (row slabs): fp64 refinement around the fp32 CG loop, halo messages of 4-byte elements"
                    t.set_dtype(isa.F32_MIXED)
                    r = t.solve(_params(isa, 1, eps_rel=1e-8, max_iterations=10 ** 6))
                    out[f"{mode}/mixed"] = np.array([r.iterations, r.converged, r.stop_reason, r.refine_outer, r.r_norm2, r.initial_r_norm2])
                    out[f"{mode}/xmixed"] = t.vector(0)
                    t.set_dtype(isa.F64)
                    r = t.solve(_params(isa, 1, eps_rel=1e-8, max_iterations=37, fixed_iterations=1))
                    out[f"{mode}/fixed2"] = np.array([r.iterations, r.r_norm2])
                step = "close"
                t.close()
                dist.barrier()
                if rank == 0 and os.environ.get("MI355CG_TEST_TIMES") == "1":
                    print(f"[times] world {world} N {n} mode {mode}: create..(5) {t_mixed - t_mode:.2f} s, (6) {time.time() - t_mixed:.2f} s", flush=True)
            except Exception as e:
                raise RuntimeError(f"rank {rank}, mode {mode}, step {step}: {e}") from e
        np.savez(os.path.join(outdir, f"r{rank}.npz"), **out)
    finally:
        dist.destroy_process_group()


def _run(world, n, decomp, modes):
    import torch.multiprocessing as mp
    port = 29500 + (os.getpid() * 7 + world * 31 + n) % 1500
    with tempfile.TemporaryDirectory() as d:
        ctx = mp.spawn(_worker, args=(world, port, n, decomp, modes, d), nprocs=world, join=False)
        deadline = time.time() + 300.0                                   # every wait inside is bounded; this is the belt to those braces
        while not ctx.join(timeout=1.0):
            if time.time() > deadline:
                for proc in ctx.processes:                               # exactly the rank processes started above
                    if proc.is_alive():
                        proc.kill()
                pytest.fail(f"rank processes still running after 300 s (world {world}, N {n}, modes {modes})")
        parts = []
        for r in range(world):
            with np.load(os.path.join(d, f"r{r}.npz")) as f:
                parts.append({k: f[k] for k in f.files})
    return parts


def _reference(n):
    import iterative_solvers_amd as isa
    s = isa.MatrixFreeSystem(n, n, 1.0, 2.0, 1.0, 2.0)
    rel = s._handle.solve(_params(isa, 1, **REL))
    x = s._handle.solution()
    o = (C.c_double * 2)()
    from iterative_solvers_amd import _capi
    _capi.check(_capi.load().mi355cg_checksum(s._handle._h, 1, o))
    cs = (o[0], o[1])
    cbs = []
    msg = s._handle.solve(_params(isa, 0, **MSG), callback=lambda *a: cbs.append(a))
    xm = s._handle.solution()
    fixed = s._handle.solve(_params(isa, 1, eps_rel=1e-8, max_iterations=37, fixed_iterations=1))
    s32 = isa.MatrixFreeSystem(n, n, 1.0, 2.0, 1.0, 2.0, dtype=isa.F32_MIXED)
    mixed = s32._handle.solve(_params(isa, 1, eps_rel=1e-8, max_iterations=10 ** 6))
    return dict(rel=rel, x=x, cs=cs, msg=msg, cbs=np.array(cbs, dtype=float).reshape(-1, 4), xm=xm, fixed=fixed, mixed=mixed, xmixed=s32._handle.solution())


def _check(parts, ref, world, modes, real=False):
    for mode in modes:
        rec, halo, split = EXPECT[mode]
        x = np.full_like(ref["x"], np.nan)
        xm = np.full_like(ref["x"], np.nan)
        for rank, p in enumerate(parts):
            d = p[f"{mode}/desc"]
            assert (d[0], d[2], int(d[3])) == (rec, halo, split), (mode, d)
            if real:
                assert d[1] == "kernel" and int(d[5]) == 0 and int(d[6]) == world and d[7].startswith("librccl")     # a GPU each: kernels poll, RCCL is RCCL
            else:
                assert d[1] == ("kernel" if "kernel-wait" in mode else "stream") and int(d[5]) == 1     # the ranks found out that they share one GPU: no polling kernels unless asked for
                assert int(d[4]) == (0 if mode in ("no-ipc", "too-large-to-map") else 1) and int(d[6]) == world and d[7].endswith("libnccl_shim.so")
                if mode == "too-large-to-map":
                    assert "not_mapped" in d[8]
            it, conv, reason, rn, r0 = p[f"{mode}/rel"]
            assert (it, conv, reason) == (ref["rel"].iterations, ref["rel"].converged, ref["rel"].stop_reason), (mode, rank)
            assert rn == ref["rel"].r_norm2 and r0 == ref["rel"].initial_r_norm2, (mode, rank)
            own = ~np.isnan(p[f"{mode}/x"])
            assert not (own & ~np.isnan(x)).any()                               # no unknown has two owners
            x[own] = p[f"{mode}/x"][own]
            xm[own] = p[f"{mode}/xm"][own]
            it, conv, reason, rmax, prec = p[f"{mode}/msg"]
            assert (it, conv, reason, rmax, prec) == (ref["msg"].iterations, ref["msg"].converged, ref["msg"].stop_reason,
                                                      ref["msg"].final_residual_norm, ref["msg"].final_precision), (mode, rank)
            if rank == 0:
                assert np.array_equal(p[f"{mode}/cbs"], ref["cbs"]), mode       # iteration numbers and all three norms of every callback
            else:
                assert p[f"{mode}/cbs"].size == 0
            assert tuple(p[f"{mode}/stop0"]) == (1, 0, 4), (mode, rank, p[f"{mode}/stop0"])     # the flag was up before iteration 1: its record carries it
            assert tuple(p[f"{mode}/stop1"]) == tuple(parts[0][f"{mode}/stop1"]) and p[f"{mode}/stop1"][2] == 4 and 1 <= p[f"{mode}/stop1"][0] <= 3, (mode, rank, p[f"{mode}/stop1"])
            assert tuple(p[f"{mode}/fixed"]) == (37, ref["fixed"].r_norm2), (mode, rank)
        if f"{mode}/mixed" in parts[0]:
            xx = np.full_like(ref["x"], np.nan)
            for rank, p in enumerate(parts):
                it, conv, reason, outer, rn, r0 = p[f"{mode}/mixed"]
                m = ref["mixed"]
                assert (it, conv, reason, outer) == (m.iterations, m.converged, m.stop_reason, m.refine_outer) and conv == 1, (mode, rank, p[f"{mode}/mixed"])
                assert abs(rn - m.r_norm2) <= 1e-12 * m.initial_r_norm2 and abs(r0 - m.initial_r_norm2) <= 1e-13 * m.initial_r_norm2      # (block sums added part by part)
                own = ~np.isnan(p[f"{mode}/xmixed"])
                xx[own] = p[f"{mode}/xmixed"][own]
                assert tuple(p[f"{mode}/fixed2"]) == (37, ref["fixed"].r_norm2), (mode, rank)           # and fp64 again afterwards
            assert np.array_equal(xx, ref["xmixed"]), mode                       # the single-GPU mixed solve's x, bit for bit
        assert np.array_equal(x, ref["x"]), mode                                # every unknown, the single-context bits
        assert np.array_equal(xm, ref["xm"]), mode
        cs = np.sum([p[f"{mode}/cs"] for p in parts], axis=0)
        assert abs(cs[1] - ref["cs"][1]) <= 1e-12 * abs(ref["cs"][1])          # (the per-rank checksums are rounded before they are added here)


@pytest.mark.parametrize("world,n,decomp,modes", [
    (2, 258, 0, list(MODES)),                                                   # (kernel-wait included: two small launches can share the GPU)
    (4, 258, 1, list(MODES)),                                                   # 2 x 2: column messages, packed and unpacked
    (5, 130, 0, ["rccl+inline", "mailbox+push", "rccl+stream+split"]),
    (3, 1026, 0, ["mailbox+push", "rccl+inline", "mailbox+push+kernel-wait"]),      # (polling kernels of three ranks on one GPU: the reducer launches fit beside them)
    (4, 1026, 1, ["mailbox+push", "rccl+stream"]),
])
def test_rank_processes_reproduce_the_single_context(world, n, decomp, modes):
    ref = _reference(n)
    parts = _run(world, n, decomp, modes)
    _check(parts, ref, world, modes)


def _gpus():
    import torch
    return torch.cuda.device_count()


@pytest.mark.parametrize("world,n,decomp", [(2, 1026, 0), (4, 1026, 1), (8, 2050, 0)])
def test_rank_processes_on_gpus_of_their_own_over_real_rccl(world, n, decomp, monkeypatch):
    """The same scenarios with one GPU per rank and librccl itself: runs wherever the box has the GPUs (the one-GPU test box skips it).
    Every transport combination, kernels polling their mailboxes (the default when no two ranks share a device)."""
    if _gpus() < world:
        pytest.skip(f"needs {world} GPUs, this box has {_gpus()}")
    monkeypatch.setenv("MI355CG_TEST_REAL_RCCL", "1")
    modes = [m for m in MODES if "kernel-wait" not in m]
    ref = _reference(n)
    parts = _run(world, n, decomp, modes)
    _check(parts, ref, world, modes, real=True)
