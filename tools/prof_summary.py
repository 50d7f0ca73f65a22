#!/usr/bin/env python3
"""Summarise rocprofv3 CSV output (kernel trace and/or PMC counter collection) per kernel.
Usage: python tools/prof_summary.py <rocprof_output_dir> [--json out.json]
Prints a table: kernel, launches, avg/min/max duration (us); for counter runs the per-launch mean
of each counter.  FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KiB; on gfx950 FETCH_SIZE
counts 64 B per 128-B request for 16-B/lane streaming reads, so HBM read bytes = 2 x FETCH_SIZE
(MI355X_MICROARCH.md, section HBM)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def short(name: str) -> str:
    for key in ("k_stencil", "k_update_st", "k_update", "k_check", "k_flush_x", "k_team_record", "k_team_stop", "k_cols", "k_pack", "k_unpack", "k_sub", "k_resid2", "k_checksum", "k_setup"):
        if key in name:
            tag = key
            if key == "k_stencil":
                rest = name.split("k_stencil")[1]
                fused = ("Lb1E" in rest[:12]) or rest.replace(" ", "").startswith("<double,2,true") or rest.replace(" ", "").startswith("<float,4,true")
                tag += "<fused>" if fused else "<plain>"
            return tag
    return name[:60]


def main():
    d = sys.argv[1]
    out_json = sys.argv[sys.argv.index("--json") + 1] if "--json" in sys.argv else None
    files = glob.glob(os.path.join(d, "**", "*.csv"), recursive=True)
    dur = defaultdict(list)
    ctr = defaultdict(lambda: defaultdict(list))
    for f in files:
        with open(f, newline="") as fh:
            rd = csv.DictReader(fh)
            cols = rd.fieldnames or []
            if "Start_Timestamp" in cols and "Kernel_Name" in cols and "Counter_Name" not in cols:
                for row in rd:
                    dur[short(row["Kernel_Name"])].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3)
            elif "Counter_Name" in cols:
                for row in rd:
                    ctr[short(row["Kernel_Name"])][row["Counter_Name"]].append(float(row["Counter_Value"]))
    result = {}
    if dur:
        print(f"{'kernel':28s} {'launches':>8s} {'avg_us':>10s} {'min_us':>10s} {'max_us':>10s} {'total_ms':>10s}")
        for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
            print(f"{k:28s} {len(v):8d} {sum(v)/len(v):10.2f} {min(v):10.2f} {max(v):10.2f} {sum(v)/1e3:10.2f}")
            result.setdefault(k, {})["avg_us"] = sum(v) / len(v)
            result[k]["launches"] = len(v)
    if ctr:
        print(f"{'kernel':28s} {'counter':>16s} {'launches':>8s} {'mean':>16s}")
        for k, cs in sorted(ctr.items()):
            for c, v in sorted(cs.items()):
                # rocprofv3 emits one row per dispatch (and per XCD/dimension on some builds): sum rows of a dispatch is
                # not recoverable here, so report the mean per row and the row count.
                print(f"{k:28s} {c:>16s} {len(v):8d} {sum(v)/len(v):16.1f}")
                result.setdefault(k, {})[c] = sum(v) / len(v)
                result[k][c + "_rows"] = len(v)
    if out_json:
        with open(out_json, "w") as fh:
            json.dump(result, fh, indent=1)


if __name__ == "__main__":
    main()
