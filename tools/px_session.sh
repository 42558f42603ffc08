#!/bin/bash
# Developer tool (GPU box): parity test of the path-exchange scheduler, A/B against the stage scheduler, phase profile.
#   bash tools/px_session.sh OUTDIR [spp]
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$1; SPP=${2:-128}
mkdir -p $OUT
timeout -k 10 300 python -m pytest $R/tests/test_gpu_parity.py -m gpu -x -q -k "path_exchange" > $OUT/px_tests.log 2>&1; echo "px tests rc=$?"; tail -n 3 $OUT/px_tests.log
(echo sm; bash $R/tools/ab.sh C3 $SPP main; echo px; PYRITE_SCHEDULER=px bash $R/tools/ab.sh C3 $SPP main) > $OUT/ab_px.log 2>&1; cat $OUT/ab_px.log
if [ -f $R/pyrite_amd/csrc/variants/lib_prof.so ]; then
  export PYRITE_GPU_LIB=$R/pyrite_amd/csrc/variants/lib_prof.so
  (echo "== px"; PYRITE_SCHEDULER=px timeout -k 10 300 python $R/tools/phase_profile.py C3 1920 1080 16) > $OUT/phase_px.log 2>&1; grep -v amdgpu.ids $OUT/phase_px.log
fi
