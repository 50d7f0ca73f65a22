#!/bin/bash
# Interleaved A/B of environment settings for one bench.py configuration.
# Usage: tools/ab_env.sh "ENV1=a ENV2=b" "ENV1=c" -- <bench.py args>      (each quoted group is one configuration; "" = defaults)
cfgs=()
while [ "$#" -gt 0 ] && [ "$1" != "--" ]; do cfgs+=("$1"); shift; done
shift
for rep in 1 2 3; do
  for cfg in "${cfgs[@]}"; do
    env $cfg python bench.py --cpu-iters 0 "$@" > /tmp/b.json 2>/tmp/b.err || { cat /tmp/b.err; exit 1; }
    python - "$cfg" <<'PY'
import json, sys
j = json.load(open('/tmp/b.json'))
o = j["roofline"]["other"]
print(f"{sys.argv[1]:60s} {j['value']:9.1f} it/s  stencil {o['stencil']['avg_ms']*1e3:6.1f} us  update {o['update']['avg_ms']*1e3:6.1f} us", flush=True)
PY
  done
done
