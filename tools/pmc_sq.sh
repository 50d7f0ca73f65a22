#!/bin/bash
# SQ occupancy / stall counters of the CG kernels (one PMC pass).  Usage (on the GPU box): tools/pmc_sq.sh [bench.py args]
set -u
REPO="${GRAFT_REPO_ROOT:-$(pwd)}"
OUT="$REPO/gpurun_out"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d "$OUT/prof_sq" -- python3 "$REPO/bench.py" --steps 100 --warmup 20 --cpu-iters 0 --no-roofline-pass "$@" > "$OUT/prof_sq.log" 2>&1
echo "pmc rc=$?"
cd "$REPO"
python3 tools/prof_summary.py "$OUT/prof_sq" > "$OUT/prof_sq_summary.txt" 2>&1
find "$OUT/prof_sq" -name "*counter_collection.csv" -size +2M -delete
cat "$OUT/prof_sq_summary.txt"
