#!/bin/bash
# A/B two builds of libmi355cg.so on one box, interleaved: $1 = alternative library
for rep in 1 2 3 4; do
  for lib in "" "$1"; do
    MI355CG_LIB=$lib python bench.py --steps 1500 --warmup 100 --cpu-iters 0 > /tmp/b.json 2>/dev/null
    python - "${lib:-default}" <<'PY'
import json, sys
j = json.load(open('/tmp/b.json'))
print(f"{sys.argv[1]:40s} {j['value']:9.1f} it/s  stencil {j['roofline']['other']['stencil']['avg_ms']*1e3:6.1f} us  update {j['roofline']['other']['update']['avg_ms']*1e3:6.1f} us")
PY
  done
done
