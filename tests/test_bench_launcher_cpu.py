"""CPU tests of bench.py's multi-GPU launcher: the legs it plans, the environment and command line every rank process gets,
and the coordinator itself -- started without a launcher (it spawns all N ranks) and as one of a launcher's N workers (it
spawns its own rank only) -- with stand-in rank processes: records collected from rank 0, a stalled leg killed at its time
limit without losing the legs before or after it, a failing rank reported.  No GPU is touched: the coordinator never imports torch."""
import json
import os
import sys
import types

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def _args(**kw):
    a = types.SimpleNamespace(gpus=8, steps=20, warmup=5, n=4096, rule="rel2", dtype="f64", scaling="weak", decomp="rows", cpu_iters=0,
                              no_roofline_pass=False, repeats=0, verify=30, verify_max_unknowns=2.6e8, legs="all", leg_timeout=20.0, import_allowance=20.0, budget=420.0,
                              child_leg=None, child_out=None)
    for k, v in kw.items():
        setattr(a, k, v)
    return a


def test_the_coordinator_cannot_touch_a_gpu():
    assert "torch" not in [m.split(".")[0] for m in bench.coordinate.__code__.co_names]
    src = open(os.path.join(ROOT, "bench.py")).read()
    head = src[:src.index("def run_leg_child")]
    assert "import torch" not in head and "iterative_solvers_amd" not in head.replace("iterative_solvers_amd.distributed.weak_scaling_n", "")


def test_legs_of_an_eight_gpu_run():
    legs = bench.plan_legs(_args())
    names = [l["name"] for l in legs]
    assert names[:2] == ["rccl-inline", "mailbox+push"]                 # the order-safe schedule is timed first
    assert [l["name"] for l in legs if l["headline"]] == ["rccl-inline", "mailbox+push", "mailbox+push+split-update", "mailbox+push+split"]
    assert names.index("config5-strong-32768") < names.index("rccl-stream")          # BASELINE's own configurations before the other transports
    assert "config5-strong-32768" in names and names[-1] == "local-one-process" and "config4-2x2-16384" not in names
    strong = next(l for l in legs if l["name"] == "config5-strong-32768")
    assert (strong["scaling"], strong["grid"], strong["decomp"]) == ("strong", 32768, "rows")
    four = bench.plan_legs(_args(gpus=4))
    c4 = next(l for l in four if l["name"] == "config4-2x2-16384")
    assert (c4["scaling"], c4["grid"], c4["decomp"]) == ("strong", 16384, "2d")
    assert [l["name"] for l in bench.plan_legs(_args(legs="default"))] == ["rccl-inline", "mailbox+push"]
    assert [l["name"] for l in bench.plan_legs(_args(gpus=1))][-1] != "local-one-process"


def test_rank_environment_and_command_line():
    base = {"PATH": os.environ["PATH"], "TORCHELASTIC_RUN_ID": "x", "MI355CG_TEAM_HALO": "stream", "LOCAL_WORLD_SIZE": "8", "MASTER_PORT": "29400"}
    spec = bench.plan_legs(_args())[0]
    env = bench.leg_env(base, spec, rank=3, world=8, local_rank=3, port=29999)
    assert (env["RANK"], env["WORLD_SIZE"], env["LOCAL_RANK"], env["MASTER_ADDR"], env["MASTER_PORT"]) == ("3", "8", "3", "127.0.0.1", "29999")
    assert env["MI355CG_TEAM_RECORDS"] == "rccl" and env["MI355CG_TEAM_HALO"] == "inline" and env["MI355CG_BENCH_CHILD"] == "1"
    assert "TORCHELASTIC_RUN_ID" not in env and "LOCAL_WORLD_SIZE" not in env and env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    env2 = bench.leg_env(base, bench.plan_legs(_args())[1], 0, 8, 0, 30000)
    assert env2["MI355CG_TEAM_RECORDS"] == "auto" and env2["MI355CG_TEAM_HALO"] == "auto" and "MI355CG_TEAM_IPC" not in env2
    cmd = bench.leg_command(_args(), spec, "/tmp/x")
    assert cmd[0] == sys.executable and cmd[1].endswith("bench.py")
    assert cmd[cmd.index("--gpus") + 1] == "8" and cmd[cmd.index("--steps") + 1] == "20" and json.loads(cmd[cmd.index("--child-leg") + 1])["name"] == "rccl-inline"


# a stand-in for a rank process: behaves as the leg's name says and writes rank 0's record
STUB = r"""
import json, os, sys, time
spec = json.loads(sys.argv[sys.argv.index("--child-leg") + 1]); out = sys.argv[sys.argv.index("--child-out") + 1]
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
if os.environ.get("STUB_NEVER_READY") == "1":
    time.sleep(600)
open(out + ".r%d.ready" % rank, "w").close()
time.sleep(float(os.environ.get("STUB_WORK_S", "0")))
if spec["name"] == "mailbox+push" and os.environ.get("STUB_PUSH_FAIL") == "1":
    json.dump({"leg": spec["name"], "error": "rank %d: the mapping never came back" % rank}, open(out + ".r%d" % rank, "w")); sys.exit(1)
if spec["name"] == "rccl-stream":
    time.sleep(600)                                    # a collective that never completes
if spec["name"] == "config5-strong-32768" and rank == world - 1:
    json.dump({"leg": spec["name"], "error": "rank %d: boom" % rank}, open(out + ".r%d" % rank, "w")); sys.exit(1)
if spec["name"] == "config5-strong-32768":
    time.sleep(600)                                    # the other ranks wait for the dead one
if rank == 0:
    json.dump({"leg": spec["name"], "value": {"rccl-inline": 50000.0, "mailbox+push": 56000.0}.get(spec["name"], 1.0), "n_gpus": world, "global_iters_per_sec": 7000.0,
               "ms_per_step": 0.143, "repeats": 11, "ms_per_step_min_max": [0.14, 0.15], "n": 11586, "unknowns": 100629441, "unknowns_per_gpu": 1.0,
               "decomposition": {"kind": "rows", "parts": []}, "transport": {"rccl_nranks": world, "port": os.environ["MASTER_PORT"], "records": os.environ.get("MI355CG_TEAM_RECORDS")},
               "hbm_gbps": 1.0, "per_gpu_gbps": 5000.0, "per_gpu_frac_of_8000": 0.625, "phases_ms": {}, "verify_against_one_gpu": {"ok": spec["name"] != "mailbox+push" or os.environ.get("STUB_PUSH_OK", "1") == "1"}},
              open(out + ".r0", "w"))
"""


@pytest.fixture
def stub(tmp_path, monkeypatch):
    path = tmp_path / "stub_rank.py"
    path.write_text(STUB)
    real = bench.leg_command
    monkeypatch.setattr(bench, "leg_command", lambda args, spec, out: [sys.executable, str(path)] + real(args, spec, out)[2:])
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setenv("MASTER_PORT", "29600")
    return path


def test_coordinator_without_a_launcher_runs_every_leg_and_survives_a_stall(stub):
    args = _args(gpus=4, leg_timeout=3.0)
    legs, results, notes = bench.coordinate(args)
    assert list(results) == [l["name"] for l in legs]
    assert results["rccl-inline"]["value"] == 50000.0 and results["mailbox+push"]["n_gpus"] == 4
    assert "stalled" in results["rccl-stream"]["error"]                  # killed at its limit ...
    assert "boom" in results["config5-strong-32768"]["error"]            # ... a dead rank is reported, its peers are not waited for
    assert results["config4-2x2-16384"]["value"] == 1.0 and results["local-one-process"]["n_gpus"] == 1      # ... and the legs behind them still ran
    ports = [results[n]["transport"]["port"] for n in ("rccl-inline", "mailbox+push")]
    assert len(set(ports)) == 2 and all(int(p) > 29600 for p in ports)   # every leg meets on a port of its own, none on the launcher's
    out = bench.compose(args, legs, results, notes)
    assert out["n_gpus"] == 4 and out["value"] == 56000.0 and out["headline_leg"] == "mailbox+push" and out["rccl_nranks"] == 4
    assert out["roofline"]["frac"] == 0.625 and set(out["legs"]) == set(results) and out["scaling"] == "weak"


def test_an_unverified_leg_is_not_the_headline(stub, monkeypatch):
    monkeypatch.setenv("STUB_PUSH_OK", "0")
    args = _args(gpus=2, legs="default", leg_timeout=5.0)
    out = bench.compose(args, *bench.coordinate(args))
    assert out["headline_leg"] == "rccl-inline" and out["value"] == 50000.0
    assert out["legs"]["mailbox+push"]["verify_against_one_gpu"]["ok"] is False


DRIVER = r"""
import json, sys, types
sys.path.insert(0, sys.argv[1])
import bench
stub, kw = sys.argv[2], json.loads(sys.argv[3])
real = bench.leg_command
bench.leg_command = lambda args, spec, out: [sys.executable, stub] + real(args, spec, out)[2:]
args = types.SimpleNamespace(**kw)
got = bench.coordinate(args)
if got is not None:
    print(json.dumps(bench.compose(args, *got)), flush=True)
"""


def _launch_workers(tmp_path, stub, world, port, extra_env=None, **kw):
    """What torchrun does: `world` workers, each with its RANK -- every one of them a coordinator."""
    import subprocess
    drv = tmp_path / "driver.py"
    drv.write_text(DRIVER)
    a = vars(_args(gpus=world, **kw))
    procs = []
    for r in range(world):
        env = dict(os.environ, WORLD_SIZE=str(world), RANK=str(r), LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), **(extra_env or {}))
        procs.append(subprocess.Popen([sys.executable, str(drv), ROOT, str(stub), json.dumps(a)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    return procs


def _collect(procs, timeout=120):
    outs = [p.communicate(timeout=timeout) for p in procs]
    assert [p.returncode for p in procs] == [0] * len(procs), [o[1][-600:] for o in outs]
    assert all(o[0] == "" for o in outs[1:])                             # only rank 0's coordinator prints
    return json.loads(outs[0][0])


def test_coordinators_under_a_launcher_act_in_step(stub, tmp_path):
    import time
    t0 = time.time()
    out = _collect(_launch_workers(tmp_path, stub, 3, 29700, legs="all", leg_timeout=4.0, import_allowance=5.0))
    legs = out["legs"]
    assert legs["rccl-inline"]["n_gpus"] == 3 and legs["rccl-inline"]["transport"]["records"] == "rccl" and out["rccl_nranks"] == 3
    assert "stalled" in legs["rccl-stream"]["error"]                     # every coordinator gave the leg up at the lead's deadline ...
    assert "boom" in legs["config5-strong-32768"]["error"]               # ... rank 2's failure reached the lead through the meeting directory ...
    assert legs["local-one-process"]["n_gpus"] == 1 and out["value"] == 56000.0      # ... and the later legs ran
    assert time.time() - t0 < 60.0                                       # the failed leg was not waited out: 15 s, not import allowance + limit per coordinator


def test_a_world_size_that_disagrees_with_gpus_is_refused(stub, monkeypatch):
    monkeypatch.setenv("WORLD_SIZE", "4")
    monkeypatch.setenv("RANK", "0")
    monkeypatch.setenv("LOCAL_RANK", "0")
    with pytest.raises(SystemExit):
        bench.coordinate(_args(gpus=2, legs="default", leg_timeout=5.0))


def test_the_budget_ends_the_run_whatever_the_legs_do(stub, tmp_path):
    import time
    t0 = time.time()
    # every leg takes 6 s; 16 s of budget: two legs fit, the third would have < min(45, limit) s left and is not started
    out = _collect(_launch_workers(tmp_path, stub, 2, 29800, extra_env={"STUB_WORK_S": "6"}, legs="rccl-inline,mailbox+push,mailbox+rccl-halo", leg_timeout=8.0, import_allowance=5.0, budget=16.0))
    assert time.time() - t0 < 16.0 + 8.0
    assert set(out["legs"]) == {"rccl-inline", "mailbox+push"} and out["value"] == 56000.0
    assert any("mailbox+rccl-halo not started" in n for n in out["notes"])
    # a leg whose ranks never finish importing is cut at what is left of the budget, and the line is still printed
    t0 = time.time()
    out = _collect(_launch_workers(tmp_path, stub, 2, 29810, extra_env={"STUB_NEVER_READY": "1"}, legs="default", leg_timeout=50.0, import_allowance=50.0, budget=6.0))
    assert time.time() - t0 < 6.0 + 8.0 and out["value"] is None and "imports" in out["legs"]["rccl-inline"]["error"] and "mailbox+push" not in out["legs"]


def test_sigterm_still_prints_the_line(stub, tmp_path):
    import signal
    import time
    procs = _launch_workers(tmp_path, stub, 2, 29820, extra_env={"STUB_WORK_S": "4"}, legs="default", leg_timeout=30.0, import_allowance=30.0)
    time.sleep(6.5)                                                      # leg 1 is done, leg 2 is under way
    procs[0].send_signal(signal.SIGTERM)
    out = _collect(procs, timeout=90)
    assert out["value"] == 50000.0 and out["headline_leg"] == "rccl-inline" and any("SIGTERM" in n for n in out["notes"])


def test_a_failed_mailbox_leg_spares_its_variants_and_reroutes_the_sub_records(stub, monkeypatch):
    """The mailbox transport fails on this node (first leg that uses it): its headline variants are not started (each would repeat the
    failure at a leg's time limit), BASELINE's configurations run on the order-safe RCCL schedule instead, the line carries the notes."""
    monkeypatch.setenv("STUB_PUSH_FAIL", "1")
    args = _args(gpus=4, leg_timeout=3.0, legs="rccl-inline,mailbox+push,mailbox+push+split-update,mailbox+push+split,config5-strong-32768,config4-2x2-16384,local-one-process")
    legs, results, notes = bench.coordinate(args)
    assert "never came back" in results["mailbox+push"]["error"]
    assert "mailbox+push+split-update" not in results and "mailbox+push+split" not in results
    assert results["config4-2x2-16384"]["transport"]["records"] == "rccl"                 # the rank processes got the RCCL schedule's environment
    assert "ncclAllGather" in results["config5-strong-32768"]["what"] and "ncclAllGather" in results["config4-2x2-16384"]["what"]
    assert results["local-one-process"]["n_gpus"] == 1
    assert sum("the mailbox transport failed in leg mailbox+push" in n for n in notes) == 4
    out = bench.compose(args, legs, results, notes)
    assert out["headline_leg"] == "rccl-inline" and out["value"] == 50000.0


def test_the_reroute_reaches_every_coordinator(stub, tmp_path):
    out = _collect(_launch_workers(tmp_path, stub, 4, 29840, extra_env={"STUB_PUSH_FAIL": "1"}, legs="rccl-inline,mailbox+push,mailbox+push+split,config4-2x2-16384", leg_timeout=4.0, import_allowance=5.0))
    assert "never came back" in out["legs"]["mailbox+push"]["error"] and "mailbox+push+split" not in out["legs"]
    assert out["legs"]["config4-2x2-16384"]["transport"]["records"] == "rccl" and out["legs"]["config4-2x2-16384"]["n_gpus"] == 4    # all four rank processes ran the RCCL schedule
    assert out["headline_leg"] == "rccl-inline" and len(out["notes"]) == 2
