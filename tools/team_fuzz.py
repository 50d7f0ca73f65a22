#!/usr/bin/env python3
"""Random decompositions (LOCAL teams on the one GPU) against the single context: bit-identical x, r and norms.
Usage (GPU box): python tools/team_fuzz.py [n_cases] [seed]"""
import os
import random
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import iterative_solvers_amd as isa
from iterative_solvers_amd import _capi
from iterative_solvers_amd.distributed import Team

TEAM_KNOBS = {"MI355CG_TEAM_SPLIT": ["0", "1"], "MI355CG_TEAM_THREADS": ["0", "1"], "MI355CG_ITEM_ROWS": ["1", "3", "16", "100000"],
              "MI355CG_DEPTH": ["2", "3"], "MI355CG_DYN_ROWS": ["0", "1", "2", "5", "16"], "MI355CG_XSTEPS": ["2", "4"]}


def params(rule, iters):
    p = isa.default_params(rule)
    p.max_iterations, p.fixed_iterations, p.use_true_solution, p.callback_every, p.sync_every = iters, 1, 1 if rule == 0 else 0, 0, 7
    return p


def fuzz(cases, seed, verbose=True):
    rng = random.Random(seed)
    bad, skipped = [], 0
    for c in range(cases):
        n = rng.choice([34, 66, 130, 258, 514, 1026])
        world = rng.choice([2, 3, 4, 5, 6, 8, 12, 16])
        decomp = rng.choice([0, 1])
        rule = rng.choice([_capi.RULE_REL_2NORM, _capi.RULE_MSG_MAXNORM])
        iters = rng.choice([1, 2, 3, 4, 5, 9, 17])
        env = {k: rng.choice(v) for k, v in TEAM_KNOBS.items() if rng.random() < 0.5}
        for k in TEAM_KNOBS:
            os.environ.pop(k, None)
        os.environ.update(env)
        try:
            t = Team.local(n, world, decomp)
        except ValueError as e:                       # more parts than rows, odd world for the 2-D split, ...
            skipped += 1
            if verbose:
                print(f"skip n={n} world={world} decomp={decomp}: {str(e)[:80]}", flush=True)
            continue
        rt = t.solve(params(rule, iters))
        xt, rvt = t.vector(0), t.vector(1)
        t.close()
        for k in TEAM_KNOBS:
            if k.startswith("MI355CG_TEAM"):
                os.environ.pop(k, None)
        s = isa.MatrixFreeSystem(n, n, 1.0, 2.0, 1.0, 2.0)
        r1 = s._handle.solve(params(rule, iters))
        same = (rt.iterations, rt.r_norm2, rt.final_residual_norm, rt.final_precision) == (r1.iterations, r1.r_norm2, r1.final_residual_norm, r1.final_precision) \
            and np.array_equal(xt, s._handle.solution()) and np.array_equal(rvt, s._handle.recursive_residual())
        s._handle.close()
        if not same:
            bad.append((n, world, decomp, rule, iters, env))
        if verbose:
            print(f"{'ok ' if same else 'BAD'} n={n} world={world} decomp={decomp} rule={rule} it={iters} {env}", flush=True)
    for k in TEAM_KNOBS:
        os.environ.pop(k, None)
    return bad, skipped


if __name__ == "__main__":
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    bad, skipped = fuzz(cases, int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    print(f"{cases} cases, {skipped} skipped (invalid decompositions), {len(bad)} mismatches")
    sys.exit(1 if bad else 0)
