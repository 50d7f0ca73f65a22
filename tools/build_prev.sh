#!/bin/bash
# Build the library of git revision $1 (default HEAD) as iterative_solvers_amd/libmi355cg_prev.so for a same-box A/B (MI355CG_LIB=...).
set -e
REV="${1:-HEAD}"
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
T=$(mktemp -d)
mkdir -p "$T/a/b/csrc" "$T/a/include"
for f in cg_kernels.h csr_kernels.h grid_setup.cpp grid_setup.h mi355cg.hip; do git -C "$ROOT" show "$REV:iterative_solvers_amd/csrc/$f" > "$T/a/b/csrc/$f"; done
git -C "$ROOT" show "$REV:include/mi355cg.h" > "$T/a/include/mi355cg.h"
cd "$T/a/b/csrc"
hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -shared -std=c++17 -w -o "$ROOT/iterative_solvers_amd/libmi355cg_prev.so" mi355cg.hip grid_setup.cpp
rm -rf "$T"; ls -la "$ROOT/iterative_solvers_amd/libmi355cg_prev.so"
