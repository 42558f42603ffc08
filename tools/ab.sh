#!/bin/bash
# Developer tool (GPU box): A/B of library builds on the C3 (or C5) render, scene built per process, six renders, median of the last three.
#   bash tools/ab.sh C3 128 main unsigned ...     ("main" = the in-tree library, other names = csrc/variants/lib_<name>.so)
R=${GRAFT_REPO_ROOT:-$PWD}
which=$1; spp=$2; shift 2
for round in 1 2; do
for v in "$@"; do
  if [ "$v" = "main" ]; then unset PYRITE_GPU_LIB; else export PYRITE_GPU_LIB=$R/pyrite_amd/csrc/variants/lib_$v.so; fi
  printf "%-14s " $v; timeout -k 10 300 python3 $R/tools/sweep_c3.py $which $spp 2>/dev/null | tail -n 1
done
done
