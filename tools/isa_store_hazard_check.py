#!/usr/bin/env python3
"""Static check of the device code for the store-data hazard described at buf_store (csrc/cg_kernels.h): a VALU or VMEM-load
write to a data register of a 128-bit (or 96-bit) buffer store in one of the two slots after it.
Usage: python tools/isa_store_hazard_check.py [dump.s]      (without an argument: compiles csrc/mi355cg.hip to assembly first)
Exit code 1 and one line per finding if any."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def compile_to_asm(out):
    src = os.path.join(ROOT, "iterative_solvers_amd", "csrc")
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-std=c++17", "-w", "--cuda-device-only", "-S",
                           "-mllvm", "-amdgpu-atomic-optimizer-strategy=None", "-o", out, "mi355cg.hip"], cwd=src)


def regs(tok):
    tok = tok.rstrip(",")
    m = re.match(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", tok)
    return {int(m.group(1))} if m else set()


def findings(path, window=2):
    code, cur = [], None
    for line in open(path):
        m = re.match(r"^(_Z\S+):", line)
        if m:
            cur = m.group(1)
        t = line.strip()
        if not t or t[0] in ";." or t.endswith(":"):
            continue
        code.append((cur, t))
    out = []
    for i, (kern, t) in enumerate(code):
        if not re.match(r"buffer_store_dwordx[34]\b", t):
            continue
        data = regs(t.split()[1])
        slots = 0
        for kern2, t2 in code[i + 1:i + 8]:
            m = re.match(r"s_nop (\d+)", t2)
            if m:
                slots += int(m.group(1)) + 1
            else:
                if (t2.startswith("v_") or t2.startswith("buffer_load") or t2.startswith("global_load") or t2.startswith("ds_read")) \
                        and regs(t2.split()[1]) & data:
                    out.append((kern, t, t2, slots))
                    break
                slots += 1
            if slots >= window:
                break
    return out


if __name__ == "__main__":
    if len(sys.argv) > 1:
        path = sys.argv[1]
    else:
        path = os.path.join(tempfile.mkdtemp(prefix="mi355cg_isa_"), "dev.s")
        compile_to_asm(path)
    bad = findings(path)
    for kern, st, wr, slots in bad:
        print(f"{kern}: `{st}` then `{wr}` after {slots} wait state(s)")
    print(f"{len(bad)} store-data hazard(s)")
    sys.exit(1 if bad else 0)
