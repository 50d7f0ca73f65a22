#!/bin/bash
# A/B the recomputing update (MI355CG_RECOMPUTE=1, default) against the flat one for a bench.py configuration.
# Usage: tools/ab_recompute.sh <bench.py args...>     e.g.  tools/ab_recompute.sh --rule msg --steps 1000
for rc in 0 1 0 1; do
  MI355CG_RECOMPUTE=$rc python bench.py --cpu-iters 0 "$@" > /tmp/b.json 2>/tmp/b.err || { cat /tmp/b.err; exit 1; }
  python - "$rc" "$*" <<'PY'
import json, sys
j = json.load(open('/tmp/b.json'))
print("recompute", sys.argv[1], sys.argv[2], j["value"], "it/s", j["roofline"]["other"], flush=True)
PY
done
