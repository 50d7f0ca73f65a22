#!/bin/bash
# Build libmi355cg.so plus the diagnostic variants (wave-timing probe, Dot2 accumulation) next to it.
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
cd "$ROOT"
python -c "from iterative_solvers_amd import build; build.build(force=True)"
FLAGS="--offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -shared -std=c++17 -Wno-unused-value -Wno-unused-result -w -mllvm -amdgpu-atomic-optimizer-strategy=None"
cd "$ROOT/iterative_solvers_amd/csrc"
hipcc $FLAGS -DMI355CG_WAVE_TIMING -o ../libmi355cg_wt.so mi355cg.hip grid_setup.cpp
[ "${1:-}" = "all" ] && hipcc $FLAGS -DMI355CG_DOT2 -o ../libmi355cg_dot2.so mi355cg.hip grid_setup.cpp
cd "$ROOT"; ls -la iterative_solvers_amd/*.so
