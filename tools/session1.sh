#!/bin/bash
# Developer tool (GPU box), round 3 opening session: GPU tests, phase profiles of the stage scheduler, a PC-sampling attempt.
R=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p $R/gpurun_out/s1
cd $R
timeout -k 10 600 python3 -m pytest tests -m gpu -x -q > gpurun_out/s1/pytest.log 2>&1; echo "pytest rc=$?"; tail -n 5 gpurun_out/s1/pytest.log
export PYRITE_GPU_LIB=$R/pyrite_amd/csrc/variants/lib_prof.so
timeout -k 10 200 python3 tools/phase_profile.py C3 1920 1080 16 > gpurun_out/s1/phase_c3.txt 2>&1; echo "phase c3 rc=$?"; cat gpurun_out/s1/phase_c3.txt
timeout -k 10 200 python3 tools/phase_profile.py C5 1920 1080 16 > gpurun_out/s1/phase_c5.txt 2>&1; echo "phase c5 rc=$?"; cat gpurun_out/s1/phase_c5.txt
unset PYRITE_GPU_LIB
cd /tmp && export TMPDIR=/tmp
export ROCPROFILER_PC_SAMPLING_BETA_ENABLED=1
timeout -k 10 300 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-unit time --pc-sampling-method host_trap --pc-sampling-interval 50 --output-format csv -d $R/gpurun_out/s1/pcs -o pcs -- python3 $R/tools/sweep_c3.py C3 32 > $R/gpurun_out/s1/pcs.log 2>&1; echo "pcs rc=$?"; tail -n 8 $R/gpurun_out/s1/pcs.log
ls -la $R/gpurun_out/s1/pcs 2>/dev/null | head; find $R/gpurun_out/s1/pcs -type f | head
