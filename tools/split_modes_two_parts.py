#!/usr/bin/env python3
"""Two parts on ONE GPU with polling kernels (what two GPUs would run): iteration time of the halo schedules against each other.
The two parts' launches share the card (each gets about half of it), so the figures are not a 2-GPU prediction; the push -> wait ->
stencil dependency between the parts is the real one, which is what the three schedules differ in.
Usage: python tools/split_modes_two_parts.py [N] [iters]"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
IT = int(sys.argv[2]) if len(sys.argv) > 2 else 2000

CHILD = r"""
import sys, time
sys.path.insert(0, %r)
import iterative_solvers_amd as isa
from iterative_solvers_amd import _capi
from iterative_solvers_amd.distributed import Team
N, IT = %d, %d
p = isa.default_params(_capi.RULE_REL_2NORM)
p.max_iterations, p.fixed_iterations, p.use_true_solution, p.callback_every, p.sync_every = IT, 1, 0, 0, 500
t = Team.local(N, 2, 0)
d = t.describe()
t.solve(p)
best = 1e9
for _ in range(5):
    r = t.solve(p)
    best = min(best, r.loop_seconds / IT)
print("%%s/%%s/%%s split=%%s: %%.4f ms per iteration" %% (d["records"], d["wait"], d["halo"], d["split"], 1e3 * best), flush=True)
t.close()
""" % (ROOT, N, IT)

base = {"MI355CG_TEAM_RECORDS": "mailbox", "MI355CG_TEAM_WAIT": "kernel", "MI355CG_TEAM_HALO": "push", "MI355CG_TEAM_TIMEOUT_MS": "5000"}
for name, extra in (("events + copies (the default of parts that share a GPU)", None), ("one launch per phase", {}), ("update phase split", {"MI355CG_TEAM_SPLIT": "2"}), ("both phases split", {"MI355CG_TEAM_SPLIT": "1"})):
    env = dict(os.environ)
    if extra is not None:
        env.update(base)
        env.update(extra)
    out = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, timeout=300)
    print(f"N={N} world=2 on one GPU, {name}: {out.stdout.strip() or out.stderr.strip()[-300:]}", flush=True)
