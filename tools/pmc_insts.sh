#!/bin/bash
# Dynamic instruction mix of the CG kernels (two PMC passes).  Usage (on the GPU box): tools/pmc_insts.sh [bench.py args]
set -u
REPO="${GRAFT_REPO_ROOT:-$(pwd)}"
OUT="$REPO/gpurun_out"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAVES --output-format csv -d "$OUT/prof_insts" -- python3 "$REPO/bench.py" --steps 100 --warmup 20 --cpu-iters 0 --no-roofline-pass "$@" > "$OUT/prof_insts.log" 2>&1
echo "pmc insts rc=$?"
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY --output-format csv -d "$OUT/prof_active" -- python3 "$REPO/bench.py" --steps 100 --warmup 20 --cpu-iters 0 --no-roofline-pass "$@" > "$OUT/prof_active.log" 2>&1
echo "pmc active rc=$?"
cd "$REPO"
python3 tools/prof_summary.py "$OUT/prof_insts" > "$OUT/prof_insts_summary.txt" 2>&1
python3 tools/prof_summary.py "$OUT/prof_active" > "$OUT/prof_active_summary.txt" 2>&1
find "$OUT/prof_insts" "$OUT/prof_active" -name "*counter_collection.csv" -size +2M -delete
cat "$OUT/prof_insts_summary.txt" "$OUT/prof_active_summary.txt"
