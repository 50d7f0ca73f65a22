#!/bin/bash
# A/B the 9-word (XFUSE) and 10-word iteration in one gpurun call
for xf in 0 1 0 1; do
  MI355CG_XFUSE=$xf python bench.py --steps 1000 --warmup 100 --cpu-iters 0 > /tmp/b.json 2>/dev/null
  python - "$xf" <<'PY'
import json, sys
j = json.load(open('/tmp/b.json'))
print("xfuse", sys.argv[1], j["value"], j["ms_per_step"], j["roofline"]["other"])
PY
done
