#!/usr/bin/env python3
"""Copy the results of the round's standard measurement run (see the gpurun command in profiles/r01_tune_notes.md:
tools/profile_gpu.sh, bench.py x4, tools/config_runs.py, tools/wave_timing.py x2, tools/pmc_sq.sh) from gpurun_out/
into the committed profiles/ set.  Usage: python tools/refresh_profiles.py [tag]"""
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
G = lambda f: os.path.join(ROOT, "gpurun_out", f)
P = lambda f: os.path.join(ROOT, "profiles", f)

subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "make_profiles.py"), tag, G("bench_n4096.json")])
for src, dst in (("bench_f64_8192.json", f"{tag}_bench_n8192_f64.json"), ("bench_f32_8192.json", f"{tag}_bench_n8192_f32.json"),
                 ("bench_n4096_msg.json", f"{tag}_bench_n4096_msg.json"), ("config_runs.jsonl", f"{tag}_config_runs.jsonl")):
    if os.path.exists(G(src)) and os.path.getsize(G(src)) > 0:
        shutil.copy(G(src), P(dst))


def clean(f):
    return "".join(l for l in open(G(f)) if "by item %" not in l and "amdgpu.ids" not in l)


wt = open(P(f"{tag}_wave_timing.txt")).read()
head = wt[:wt.index("===== AFTER")]
open(P(f"{tag}_wave_timing.txt"), "w").write(
    head + "===== AFTER (7.5-word build: scalar state loads, buffer-resource addressing, first rows requested before the prologue,\n"
           "      2 016 waves = one 49-row item per wave, 2 workgroups per CU) =====\n"
           "--- 51 iterations (last update launch: odd iteration, 3 words)\n" + clean("wt_odd.log") +
           "--- 52 iterations (last update launch: even iteration, 6 words)\n" + clean("wt_even.log"))
sq = open(P(f"{tag}_sq_counters.txt")).read()
head = sq[:sq.index("AFTER (7.5-word build")]
rows = "".join(l for l in open(G("prof_sq_summary.txt")) if l.startswith(("k_stencil", "k_update_st")))
open(P(f"{tag}_sq_counters.txt"), "w").write(
    head + "AFTER (7.5-word build, buffer-resource addressing, ~120 issued instructions per row, 2 016 waves of 49 rows):\n" + rows)
print("profiles refreshed")
