#!/bin/bash
# Run on the GPU box: a fuzz campaign of tests/test_gpu_fuzz.py on other seeds, every failure traced (tools/fuzz_trace.py says tie /
# order-dependent sphere hit / DEFECT).   gpurun -- 'bash tools/fuzz_campaign.sh BASE SCENES SOUPS [OUT [MESHES]]'
BASE=${1:-20000}; SCENES=${2:-4000}; SOUPS=${3:-2000}; MESHES=${5:-0}
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=${4:-$R/gpurun_out/fuzz_$BASE}
mkdir -p $OUT
cd $R
PYRITE_FUZZ_BASE=$BASE PYRITE_FUZZ_SEEDS=$SCENES PYRITE_FUZZ_SOUPS=$SOUPS PYRITE_FUZZ_MESHES=$MESHES timeout -k 10 1000 python -m pytest tests/test_gpu_fuzz.py -m gpu -q -p no:cacheprovider > $OUT/pytest.log 2>&1
tail -n 1 $OUT/pytest.log
grep '^FAILED' $OUT/pytest.log | sed -E 's/.*test_random_(scene|soup|mesh)[a-z_]*\[([0-9]+)\].*/\1 \2/' | sort -u > $OUT/failed.txt
while read kind seed; do
  echo "=== $kind $seed"
  timeout -k 10 150 python tools/fuzz_trace.py $kind $seed 2>&1 | grep -v amdgpu.ids | grep -E "differing pixels|<--|verdict|routine says" | cut -c1-300
done < $OUT/failed.txt > $OUT/traces.txt 2>&1
cat $OUT/traces.txt
python -c "
import json; d=json.load(open('$R/gpurun_out/parity_observed.json')); print('kernel forms', d['fuzz_scenes_by_kernel_form'], 'max relL2 seen (failing cases included)', d['max_rel_l2'])"
