#!/usr/bin/env python3
"""Turn the rocprofv3 output of tools/profile_gpu.sh (gpurun_out/prof_*) + a bench JSON into the committed
profiles/ summary files.  Usage: python tools/make_profiles.py <round-tag> <bench.json>"""
import glob
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, bench = sys.argv[1], sys.argv[2]
U = 12574721


def summ(d):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "prof_summary.py"), os.path.join(ROOT, "gpurun_out", d),
                          "--json", f"/tmp/{d}.json"], capture_output=True, text=True).stdout
    return out, json.load(open(f"/tmp/{d}.json"))


kt, k = summ("prof_kt")
fe, f = summ("prof_fetch")
wr, w = summ("prof_write")
j = json.load(open(bench))
words = j["roofline"]["alg_words"]
traffic, lines = {}, []
for name, key in (("stencil", "k_stencil<fused>"), ("update", "k_update_st")):
    rd = f[key]["FETCH_SIZE"] * 1024 * 2          # gfx950: FETCH_SIZE counts 64 B per 128-B request on 16-B/lane streams
    wb = w[key]["WRITE_SIZE"] * 1024
    traffic[name] = rd + wb
    alg = words[name] * 8 * U
    lines.append(f"{key:18s} avg {k[key]['avg_us']:8.2f} us under rocprofv3 | FETCH_SIZE {f[key]['FETCH_SIZE']:10.1f} KiB x2 = {rd/1e6:7.1f} MB read, "
                 f"WRITE_SIZE {w[key]['WRITE_SIZE']:10.1f} KiB = {wb/1e6:7.1f} MB written, HBM traffic {traffic[name]/1e6:7.1f} MB/launch "
                 f"vs algorithmic {alg/1e6:7.1f} MB ({traffic[name]/alg:.3f}x)")
json.dump(traffic, open(os.path.join(ROOT, "profiles", "traffic.json"), "w"), indent=1)
shutil.copy(bench, os.path.join(ROOT, "profiles", f"{tag}_bench_n4096.json"))
stats = glob.glob(os.path.join(ROOT, "gpurun_out", "prof_kt", "**", "*kernel_stats.csv"), recursive=True)
if stats:
    shutil.copy(sorted(stats)[-1], os.path.join(ROOT, "profiles", f"{tag}_kernel_stats.csv"))
o = j["roofline"]["other"]
hdr = f"""Round-1 rocprofv3 summary, default path (REL_2NORM; A p recomputed in the update launch, x updated every second
iteration: {j['roofline']['words_per_unknown_per_iteration']} words/unknown/iteration; double-double inner products; buffer-resource addressing).
Commands (on the MI355X box, from /tmp with TMPDIR=/tmp, see tools/profile_gpu.sh):
  rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --steps 300 --warmup 50 --cpu-iters 0 --no-roofline-pass
  rocprofv3 --pmc FETCH_SIZE --output-format csv -- (same)      rocprofv3 --pmc WRITE_SIZE --output-format csv -- (same)
Workload: N=4096 (U=12 574 721 unknowns), fp64, fixed-iteration CG; 350 iterations per pass.
Un-profiled bench of the same build on the same box: {j['value']} it/s, {j['ms_per_step']} ms/iteration (profiles/{tag}_bench_n4096.json);
HIP-event per-launch means in that bench: k_stencil {o['stencil']['avg_ms']} ms, k_update_st {o['update']['avg_ms']} ms
(k_update_st alternates between 3-word launches on odd iterations and 6-word launches on even ones; the figures are means over both).
Earlier iterations of the same loop, kept for comparison: r01a_* (10 words: 4 + 6), r01b_* (9 words: 6 + 3, flat update).

== kernel trace (--kernel-trace --stats) ==
"""
open(os.path.join(ROOT, "profiles", f"{tag}_rocprof_summary.txt"), "w").write(
    hdr + kt + "\n== PMC passes (per-launch means) ==\n" + fe + wr +
    "\n== HBM traffic per launch (FETCH_SIZE corrected x2 per MI355X_MICROARCH.md section HBM) ==\n" + "\n".join(lines) + "\n")
print("\n".join(lines))
