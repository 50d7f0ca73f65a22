#!/usr/bin/env python3
"""Random launch shapes against the default one: every shape has to take bit-identical steps (the inner products are double-double
sums; nothing else may depend on the shape).  Usage (GPU box): python tools/shape_fuzz.py [n_cases] [seed]"""
import os
import random
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import iterative_solvers_amd as isa
from iterative_solvers_amd import _capi

KNOBS = {
    "MI355CG_ITEM_ROWS": ["1", "2", "3", "5", "7", "13", "32", "64", "200", "100000"],
    "MI355CG_WAVES": ["256", "512", "1024", "2048", "3072"],
    "MI355CG_BLOCKS": ["8", "24", "37", "100", "256", "512"],
    "MI355CG_DEPTH": ["2", "3"], "MI355CG_DYN_ROWS": ["0", "1", "2", "5", "16"],
    "MI355CG_XSTEPS": ["2", "4", "8"],
    "MI355CG_XCD_CLASSES": ["0", "1"],
}


def run(n, f32, rule, iters, env):
    for k in KNOBS:
        os.environ.pop(k, None)
    os.environ.update(env)
    s = isa.MatrixFreeSystem(n, n, 1.0, 2.0, 1.0, 2.0, dtype=isa.F32_MIXED if f32 else isa.F64)
    p = isa.default_params(rule)
    p.max_iterations, p.fixed_iterations, p.use_true_solution, p.callback_every, p.sync_every = iters, 1, 1 if rule == 0 else 0, 0, 500
    r = s._handle.solve(p)
    out = (r.iterations, r.r_norm2, r.final_residual_norm, r.final_precision, s._handle.solution(), s._handle.recursive_residual())
    s._handle.close()
    return out


def fuzz(cases, seed, verbose=True):
    rng = random.Random(seed)
    bad = []
    refs = {}
    for c in range(cases):
        n = rng.choice([66, 130, 258, 514, 1026, 2050, 3074])
        f32 = rng.random() < 0.35
        rule = _capi.RULE_REL_2NORM if f32 else rng.choice([_capi.RULE_REL_2NORM, _capi.RULE_MSG_MAXNORM])
        iters = rng.choice([1, 2, 3, 4, 5, 7, 8, 9, 13, 21])
        env = {k: rng.choice(v) for k, v in KNOBS.items() if rng.random() < 0.6}
        key = (n, f32, rule, iters)
        if key not in refs:
            refs[key] = run(n, f32, rule, iters, {})
        got, ref = run(n, f32, rule, iters, env), refs[key]
        same = got[:4] == ref[:4] and np.array_equal(got[4], ref[4]) and np.array_equal(got[5], ref[5])
        if not same:
            bad.append((n, f32, rule, iters, env))
        if verbose:
            print(f"{'ok ' if same else 'BAD'} n={n} {'f32' if f32 else 'f64'} rule={rule} it={iters} {env}", flush=True)
    for k in KNOBS:
        os.environ.pop(k, None)
    return bad


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    bad = fuzz(cases, int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    print(f"{cases} cases, {len(bad)} mismatches")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
