#!/bin/bash
# Run on the GPU box (gpurun -- 'bash tools/refresh_profiles.sh [tag]'): the bench lines and rocprofv3 summaries that
# tools/collect_profiles.py copies into profiles/. Every step is bounded; PMC passes are separate runs (no trace domains);
# the profiled program sits directly behind `--`. A step that times out stops the script.
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/final
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
step() { # name, seconds, command...
  local name=$1 secs=$2; shift 2
  echo "[$name]"
  timeout -k 10 $secs "$@" > $OUT/$name.log 2>&1
  local rc=$?
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "$name timed out: stopping"; exit 1; fi
  [ $rc -ne 0 ] && { echo "$name failed rc=$rc"; tail -n 5 $OUT/$name.log; }
  return 0
}
C3="--workload C3 --no-cpu-baseline --no-traversal --no-c2 --no-c5 --no-c1 --no-live-traffic"
PMC="--pmc"
if [ "${ONLY:-all}" = "all" ] || [ "$ONLY" = "bench" ]; then
  T0=$(date +%s.%N)
  timeout -k 10 900 python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/c3_bench.json 2> $OUT/c3_bench.err; rc=$?
  python3 -c "import time,sys; print(\"%.1f s wall\" % (time.time() - float(sys.argv[1])))" $T0 > $OUT/c3_bench.wall; echo "[bench C3 driver args] rc=$rc $(cat $OUT/c3_bench.wall)"
fi
if [ "${ONLY:-all}" = "all" ] || [ "$ONLY" = "c3" ]; then
  step c3_trace 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c3_trace -o c3 -- python3 $R/bench.py $C3 --steps 2 --warmup 0
  step c3_fetch 300 rocprofv3 $PMC FETCH_SIZE --output-format csv -d $OUT/c3_fetch -o fetch -- python3 $R/bench.py $C3 --steps 1 --warmup 0
  step c3_write 300 rocprofv3 $PMC WRITE_SIZE --output-format csv -d $OUT/c3_write -o write -- python3 $R/bench.py $C3 --steps 1 --warmup 0
  step c3_sq1 300 rocprofv3 $PMC SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $OUT/c3_sq1 -o sq -- python3 $R/bench.py $C3 --spp 32 --steps 1 --warmup 0
  step c3_sq2 300 rocprofv3 $PMC SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_WAVES --output-format csv -d $OUT/c3_sq2 -o sq -- python3 $R/bench.py $C3 --spp 32 --steps 1 --warmup 0
  step c3_tcc 300 rocprofv3 $PMC TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/c3_tcc -o tcc -- python3 $R/bench.py $C3 --spp 32 --steps 1 --warmup 0
fi
if [ "${ONLY:-all}" = "all" ] || [ "$ONLY" = "isect" ]; then
  step isect_plain 200 python3 $R/tools/prof_intersect_c3.py
  step isect_trace 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/isect_trace -o isect -- python3 $R/tools/prof_intersect_c3.py
  step isect_fetch 300 rocprofv3 $PMC FETCH_SIZE --output-format csv -d $OUT/isect_fetch -o fetch -- python3 $R/tools/prof_intersect_c3.py
  step isect_tcc 300 rocprofv3 $PMC TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $OUT/isect_tcc -o tcc -- python3 $R/tools/prof_intersect_c3.py
  step isect_tcp 300 rocprofv3 $PMC TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum --output-format csv -d $OUT/isect_tcp -o tcp -- python3 $R/tools/prof_intersect_c3.py
fi
if [ "${ONLY:-all}" = "all" ] || [ "$ONLY" = "c5" ]; then
  step c5_bench 600 python3 $R/bench.py --workload C5 --steps 1 --warmup 0 --no-cpu-baseline --no-traversal --no-live-traffic
  step c5_trace 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c5_trace -o c5 -- python3 $R/bench.py --workload C5 --spp 256 --steps 2 --warmup 0 --no-cpu-baseline --no-traversal --no-live-traffic
  step c5_fetch 300 rocprofv3 $PMC FETCH_SIZE --output-format csv -d $OUT/c5_fetch -o fetch -- python3 $R/bench.py --workload C5 --steps 1 --warmup 0 --no-cpu-baseline --no-traversal --no-live-traffic
  step c5_write 300 rocprofv3 $PMC WRITE_SIZE --output-format csv -d $OUT/c5_write -o write -- python3 $R/bench.py --workload C5 --steps 1 --warmup 0 --no-cpu-baseline --no-traversal --no-live-traffic
  step c5_trace_full 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c5_trace_full -o c5 -- python3 $R/bench.py --workload C5 --steps 1 --warmup 0 --no-cpu-baseline --no-traversal --no-live-traffic
  step c5_sq1 300 rocprofv3 $PMC SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $OUT/c5_sq1 -o sq -- python3 $R/bench.py --workload C5 --spp 32 --steps 1 --warmup 0 --no-cpu-baseline --no-traversal --no-live-traffic
fi
if [ "${ONLY:-all}" = "all" ] || [ "$ONLY" = "c2" ]; then
  C2="--workload C2 --no-cpu-baseline --no-traversal --no-live-traffic"
  step c2_trace 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c2_trace -o c2 -- python3 $R/bench.py $C2 --steps 5 --warmup 1
  step c2_fetch 300 rocprofv3 $PMC FETCH_SIZE --output-format csv -d $OUT/c2_fetch -o fetch -- python3 $R/bench.py $C2 --steps 1 --warmup 0
  step c2_write 300 rocprofv3 $PMC WRITE_SIZE --output-format csv -d $OUT/c2_write -o write -- python3 $R/bench.py $C2 --steps 1 --warmup 0
fi
echo done
