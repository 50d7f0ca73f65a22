#!/bin/bash
# Run on the GPU box (gpurun): rocprofv3 kernel trace + two PMC passes of the bench loop.
#   tools/profile_gpu.sh [GRID] [STEPS]        (defaults 4096 300)
# Outputs land under gpurun_out/prof_<GRID>_*; tools/make_profiles.py turns them into the committed profiles/ summaries.
set -u
REPO="${GRAFT_REPO_ROOT:-$(pwd)}"
OUT="$REPO/gpurun_out"
GRID="${1:-4096}"; STEPS="${2:-300}"
P="$OUT/prof_${GRID}"
cd /tmp && export TMPDIR=/tmp
ARGS="$REPO/bench.py --grid $GRID --steps $STEPS --warmup 50 --cpu-iters 0 --no-roofline-pass"
rocprofv3 --kernel-trace --stats --output-format csv -d "${P}_kt" -- python3 $ARGS > "${P}_kt.log" 2>&1
echo "kernel-trace rc=$?"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "${P}_fetch" -- python3 $ARGS > "${P}_fetch.log" 2>&1
echo "pmc FETCH_SIZE rc=$?"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "${P}_write" -- python3 $ARGS > "${P}_write.log" 2>&1
echo "pmc WRITE_SIZE rc=$?"
cd "$REPO"
python3 bench.py --grid $GRID --steps $STEPS --warmup 50 --cpu-iters 0 2>/dev/null | tail -1 > "${P}_bench.json"
for k in kt fetch write; do python3 tools/prof_summary.py "${P}_$k" --json "${P}_$k.json" > "${P}_${k}_summary.txt" 2>&1; done
cat "${P}_kt_summary.txt" "${P}_fetch_summary.txt" "${P}_write_summary.txt"
# keep the merged-back payload small: drop the per-dispatch CSVs, keep stats
find "$OUT" -name "*kernel_trace.csv" -size +2M -delete
find "$OUT" -name "*counter_collection.csv" -size +2M -delete
