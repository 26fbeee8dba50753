#!/bin/bash
# Same-box A/B material: checks out an earlier commit of THIS repository into ab/<name>/ (sources + its own
# bench.py / pylbm, no tests, no history) and builds its liblbm_hip.so there.  ab/ is git-ignored and travels to the GPU box.
#   scripts/r04_ab_setup.sh <commit> <name>
set -euo pipefail
commit=$1; name=$2
root=$(cd "$(dirname "$0")/.." && pwd)
dst=$root/ab/$name
rm -rf "$dst"; mkdir -p "$dst"
git -C "$root" archive "$commit" bench.py __graft_entry__.py include lattice-boltzmann-method_amd oracle/pyoracle.py oracle/lbm_oracle.cpp oracle/lbm_oracle.h oracle/Makefile profiles/traffic.json 2>/dev/null | tar -x -C "$dst"
make -C "$dst/lattice-boltzmann-method_amd/csrc" -j8 > "$dst/build.log" 2>&1
rm -rf "$dst/lattice-boltzmann-method_amd/lib/obj"
ls -la "$dst/lattice-boltzmann-method_amd/lib/liblbm_hip.so"
