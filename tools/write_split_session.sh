#!/bin/bash
# Developer tool (GPU box): where the render kernel's written bytes come from. One `--pmc WRITE_SIZE` (and
# FETCH_SIZE) pass over a C3 32-spp launch for each build: the product, -DPYR_TAPE_NOSTORE (no tape records written: what is
# left is the film's atomics), -DPYR_REPLAY_NOEXPOSE (no film atomics: what is left is the tape) and, if built, the blocked tape
# layout (-DPYR_TAPE_BLOCK=4).       bash tools/write_split_session.sh OUTDIR
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$1
mkdir -p $OUT
for v in main nostore noexpose tb4; do
  if [ $v = main ]; then unset PYRITE_GPU_LIB; else
    [ -f $R/pyrite_amd/csrc/variants/lib_$v.so ] || continue
    export PYRITE_GPU_LIB=$R/pyrite_amd/csrc/variants/lib_$v.so
  fi
  PMC_BENCH_ARGS="--workload C3 --spp 32 --steps 1 --warmup 0 --no-cpu-baseline --no-c2 --no-c5 --no-traversal --no-c1" bash $R/tools/pmc_passes.sh $OUT/$v "WRITE_SIZE" "FETCH_SIZE" > $OUT/$v.log 2>&1
  echo "== $v"; grep -A3 "render_kernel_sm<false" $OUT/$v.log | grep -E "WRITE_SIZE|FETCH_SIZE"
done
