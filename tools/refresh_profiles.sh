#!/bin/bash
# Run on the GPU box (gpurun -- 'bash tools/refresh_profiles.sh'): the bench lines and rocprofv3 summaries that get copied
# into profiles/ by tools/collect_profiles.py. Every step is bounded; PMC passes are separate runs (no trace domains).
set -e -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/final
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
echo "[1] bench C2"; timeout -k 10 600 python3 $R/bench.py > $OUT/c2_bench.json 2> $OUT/c2_bench.err
echo "[2] kernel trace C2"; timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c2_trace -o c2 -- python3 $R/bench.py --no-cpu-baseline --no-traversal > $OUT/c2_trace.log 2>&1
echo "[3] FETCH_SIZE"; timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/c2_fetch -o fetch -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-traversal > $OUT/c2_fetch.log 2>&1
echo "[4] WRITE_SIZE"; timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/c2_write -o write -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-traversal > $OUT/c2_write.log 2>&1
echo "[5] bench C3 full"; timeout -k 10 600 python3 $R/bench.py --workload C3 --steps 1 --warmup 0 --no-cpu-baseline --no-traversal > $OUT/c3_bench.json 2> $OUT/c3_bench.err
echo "[5b] kernel trace C3"; timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c3_trace -o c3 -- python3 $R/bench.py --workload C3 --steps 1 --warmup 0 --no-cpu-baseline --no-traversal > $OUT/c3_trace.log 2>&1
echo "[5c] C3 FETCH_SIZE"; timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/c3_fetch -o fetch -- python3 $R/bench.py --workload C3 --steps 1 --warmup 0 --no-cpu-baseline --no-traversal > $OUT/c3_fetch.log 2>&1
echo "[5d] C3 WRITE_SIZE"; timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/c3_write -o write -- python3 $R/bench.py --workload C3 --steps 1 --warmup 0 --no-cpu-baseline --no-traversal > $OUT/c3_write.log 2>&1
echo "[6] kernel trace intersect"; timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/isect_trace -o isect -- python3 $R/tools/prof_intersect_c3.py > $OUT/isect_trace.log 2>&1
echo done
