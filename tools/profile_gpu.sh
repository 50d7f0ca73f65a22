#!/bin/bash
# Run on the GPU box (gpurun): rocprofv3 kernel trace + two PMC passes of the bench loop.
# Outputs land under gpurun_out/prof_*; copy the summaries into profiles/ afterwards.
set -u
REPO="${GRAFT_REPO_ROOT:-$(pwd)}"
OUT="$REPO/gpurun_out"
cd /tmp && export TMPDIR=/tmp
ARGS="$REPO/bench.py --steps 300 --warmup 50 --cpu-iters 0 --no-roofline-pass"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_kt" -- python3 $ARGS > "$OUT/prof_kt.log" 2>&1
echo "kernel-trace rc=$?"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/prof_fetch" -- python3 $ARGS > "$OUT/prof_fetch.log" 2>&1
echo "pmc FETCH_SIZE rc=$?"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/prof_write" -- python3 $ARGS > "$OUT/prof_write.log" 2>&1
echo "pmc WRITE_SIZE rc=$?"
cd "$REPO"
python3 tools/prof_summary.py "$OUT/prof_kt" --json "$OUT/prof_kt.json" > "$OUT/prof_kt_summary.txt" 2>&1
python3 tools/prof_summary.py "$OUT/prof_fetch" --json "$OUT/prof_fetch.json" > "$OUT/prof_fetch_summary.txt" 2>&1
python3 tools/prof_summary.py "$OUT/prof_write" --json "$OUT/prof_write.json" > "$OUT/prof_write_summary.txt" 2>&1
cat "$OUT/prof_kt_summary.txt" "$OUT/prof_fetch_summary.txt" "$OUT/prof_write_summary.txt"
find "$OUT/prof_kt" -name "*stats*" | head
# keep the merged-back payload small: drop the per-dispatch CSVs, keep stats
find "$OUT" -name "*kernel_trace.csv" -size +2M -delete
find "$OUT" -name "*counter_collection.csv" -size +2M -delete
