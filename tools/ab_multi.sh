#!/bin/bash
# A/B two environment settings over several bench.py configurations.  Usage: tools/ab_multi.sh "ENV_A" "ENV_B"
A="$1"; B="$2"
for args in "--steps 1500 --warmup 100" "--steps 1000 --warmup 100 --rule msg" "--steps 600 --warmup 50 --grid 8192" "--steps 800 --warmup 50 --grid 8192 --dtype f32" "--steps 200 --warmup 20 --grid 16384" "--steps 3000 --warmup 200 --grid 1024" "--steps 2000 --warmup 100 --grid 2048"; do
  echo "== $args"
  tools/ab_env.sh "$A" "$B" -- $args 2>&1 | sort | awk '{print}'
done
