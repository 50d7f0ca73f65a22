#!/usr/bin/env python3
"""A/B of the recomputing update (8-word iteration) against the flat one (9 words), interleaved in one process.
Usage: python tools/ab_recompute.py [N] [iters]"""
import os
import sys

sys.argv = [sys.argv[0]] + sys.argv[1:]
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import tune

if __name__ == "__main__":
    cfgs = [{"MI355CG_RECOMPUTE": 0}, {"MI355CG_RECOMPUTE": 1, "MI355CG_UDEPTH": 4}, {"MI355CG_RECOMPUTE": 1, "MI355CG_UDEPTH": 2}]
    for env in cfgs:
        its, ts, tu, lay = tune.measure(env)
        print(f"{env} its/s={its:8.1f} stencil={ts*1e3:7.1f}us update={tu*1e3:7.1f}us", flush=True)
    tune.ab(cfgs, rounds=4)
