#!/usr/bin/env python3
"""Turn the rocprofv3 output of tools/profile_gpu.sh (gpurun_out/prof_<GRID>_*) into the committed profiles/ summary files.
Usage: python tools/make_profiles.py <round-tag> <GRID>"""
import glob
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, grid = sys.argv[1], int(sys.argv[2])
P = os.path.join(ROOT, "gpurun_out", f"prof_{grid}")
U = (grid // 2 - 1) * (3 * grid // 2 - 1)


def summ(kind):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "prof_summary.py"), f"{P}_{kind}", "--json", f"/tmp/prof_{kind}.json"],
                         capture_output=True, text=True).stdout
    return out, json.load(open(f"/tmp/prof_{kind}.json"))


kt, k = summ("kt")
fe, f = summ("fetch")
wr, w = summ("write")
j = json.load(open(f"{P}_bench.json"))
words = j["roofline"]["alg_words"]
traffic, lines = {}, []
for name, key in (("stencil", "k_stencil<fused>"), ("update", "k_update_st")):
    rd = f[key]["FETCH_SIZE"] * 1024 * 2          # gfx950: FETCH_SIZE counts 64 B per 128-B request on 16-B/lane streams
    wb = w[key]["WRITE_SIZE"] * 1024
    traffic[name] = rd + wb
    alg = words[name] * 8 * U
    lines.append(f"{key:18s} avg {k[key]['avg_us']:9.2f} us under rocprofv3 = {alg / k[key]['avg_us'] / 1e3:6.0f} GB/s algorithmic | FETCH_SIZE {f[key]['FETCH_SIZE']:12.1f} KiB x2 = {rd/1e6:9.1f} MB read, "
                 f"WRITE_SIZE {w[key]['WRITE_SIZE']:12.1f} KiB = {wb/1e6:9.1f} MB written, HBM traffic {traffic[name]/1e6:9.1f} MB/launch "
                 f"vs algorithmic {alg/1e6:9.1f} MB ({traffic[name]/alg:.3f}x)")
if grid == 4096:
    json.dump(traffic, open(os.path.join(ROOT, "profiles", "traffic.json"), "w"), indent=1)
shutil.copy(f"{P}_bench.json", os.path.join(ROOT, "profiles", f"{tag}_bench_n{grid}.json"))
stats = glob.glob(f"{P}_kt/**/*kernel_stats.csv", recursive=True)
if stats:
    shutil.copy(sorted(stats)[-1], os.path.join(ROOT, "profiles", f"{tag}_kernel_stats_n{grid}.csv"))
o = j["roofline"]["other"]
hdr = f"""rocprofv3 summary ({tag}), default path: REL_2NORM, A p recomputed in the update launch, x updated every fourth iteration
({j['roofline']['words_per_unknown_per_iteration']} words/unknown/iteration); double-double inner products; buffer-resource addressing; fetch cursor running
ahead of the compute cursor across work items; launch geometry {j['config']['layout']}.
Commands (on the MI355X box, from /tmp with TMPDIR=/tmp, see tools/profile_gpu.sh {grid}):
  rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --grid {grid} --steps N --warmup 50 --cpu-iters 0 --no-roofline-pass
  rocprofv3 --pmc FETCH_SIZE --output-format csv -- (same)      rocprofv3 --pmc WRITE_SIZE --output-format csv -- (same)
Workload: N={grid} (U={U} unknowns), fp64, fixed-iteration CG.
Un-profiled bench of the same build on the same box: {j['value']} it/s wall ({j.get('loop_only_iters_per_sec')} it/s by HIP events around the iterations alone),
{j['hbm_gbps']} GB/s really moved (profiles/{tag}_bench_n{grid}.json); HIP-event per-launch means in that bench:
k_stencil {o['stencil']['avg_ms']} ms = {o['stencil']['achieved']} GB/s, k_update_st {o['update']['avg_ms']} ms = {o['update']['achieved']} GB/s
(k_update_st: 3-word launches on three iterations of four, an 8-word launch on the fourth; the figures are means over all).

== kernel trace (--kernel-trace --stats) ==
"""
open(os.path.join(ROOT, "profiles", f"{tag}_rocprof_summary_n{grid}.txt"), "w").write(
    hdr + kt + "\n== PMC passes (per-launch means) ==\n" + fe + wr +
    "\n== HBM traffic per launch (FETCH_SIZE corrected x2 per MI355X_MICROARCH.md section HBM) ==\n" + "\n".join(lines) + "\n")
print("\n".join(lines))
