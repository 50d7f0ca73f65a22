#!/usr/bin/env python3
"""Instruction mix of the loops of one kernel in a hipcc -S dump.
Usage: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -std=c++17 -w --cuda-device-only -S -o /tmp/dev.s iterative_solvers_amd/csrc/mi355cg.hip
       python tools/isa_loop_stats.py /tmp/dev.s <mangled-name-substring> [min_instructions]
Prints, for every backward branch (a loop), the number of instructions by class between its target label and the branch."""
import collections
import re
import sys

path, pat = sys.argv[1], sys.argv[2]
minlen = int(sys.argv[3]) if len(sys.argv) > 3 else 40
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\S*:", l) and pat in l)
end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
body = lines[start:end + 1]
labels = {m.group(1): i for i, l in enumerate(body) if (m := re.match(r"^(\.LBB\d+_\d+):", l))}


def klass(op):
    if op.startswith("buffer_load") or op.startswith("global_load"): return "vmem_load"
    if op.startswith("buffer_store") or op.startswith("global_store"): return "vmem_store"
    if op.startswith("s_load") or op.startswith("s_buffer_load"): return "smem"
    if op.startswith("ds_"): return "lds"
    if op.startswith("s_waitcnt"): return "waitcnt"
    if op.startswith("s_"): return "salu"
    if op.startswith("v_") and ("_f64" in op): return "valu_f64"
    if op.startswith("v_") and "dpp" in op: return "valu_dpp"
    if op.startswith("v_"): return "valu_other"
    return "other"


print(f"{pat}: {end - start} lines")
for i, l in enumerate(body):
    m = re.match(r"^\s+(s_cbranch_\w+|s_branch)\s+(\.LBB\d+_\d+)", l)
    if not m or m.group(2) not in labels or labels[m.group(2)] > i:
        continue
    lo = labels[m.group(2)]
    cnt, ops = collections.Counter(), collections.Counter()
    n = 0
    for k in range(lo, i + 1):
        t = body[k].strip()
        if not t or t.startswith(";") or t.startswith(".") or t.endswith(":"):
            continue
        op = t.split()[0]
        if "dpp" in t and op.startswith("v_"):
            op += "_dpp"
        cnt[klass(op)] += 1
        ops[op] += 1
        n += 1
    if n < minlen:
        continue
    f64 = {o: c for o, c in ops.items() if "_f64" in o}
    print(f"loop {m.group(2)} lines {lo}-{i}: {n} instructions: " + ", ".join(f"{k} {v}" for k, v in sorted(cnt.items())))
    print("     f64 ops: " + ", ".join(f"{o} {c}" for o, c in sorted(f64.items(), key=lambda kv: -kv[1])))
    print("     waits: " + " | ".join(body[k].strip() for k in range(lo, i + 1) if body[k].strip().startswith("s_waitcnt"))[:400])
