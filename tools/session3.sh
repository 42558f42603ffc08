#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p $R/gpurun_out/s3
cd $R
echo "== axis test on the round-2 kernels (expected to FAIL: the bug)"
PYRITE_GPU_LIB=$R/pyrite_amd/csrc/variants/lib_r2head.so timeout -k 10 300 python3 -m pytest tests/test_gpu_parity.py -m gpu -q -k axis_parallel > gpurun_out/s3/axis_old.log 2>&1; echo "rc=$?"; tail -n 6 gpurun_out/s3/axis_old.log
echo "== full suite on the new kernels"
timeout -k 10 600 python3 -m pytest tests -m gpu -x -q > gpurun_out/s3/pytest.log 2>&1; echo "pytest rc=$?"; tail -n 4 gpurun_out/s3/pytest.log
echo "== A/B"
bash tools/ab.sh C3 128 main r2head
bash tools/ab.sh C5 64 main r2head
for v in main r2head; do
  if [ "$v" = "main" ]; then unset PYRITE_GPU_LIB; else export PYRITE_GPU_LIB=$R/pyrite_amd/csrc/variants/lib_$v.so; fi
  timeout -k 10 300 python3 bench.py --workload C2 --steps 3 --warmup 1 --no-cpu-baseline --no-traversal > gpurun_out/s3/c2_$v.json 2>/dev/null
  python3 -c "
import json,sys; d=json.load(open('gpurun_out/s3/c2_$v.json')); print('C2', '$v', d['value'], d['roofline']['frac'])"
done
