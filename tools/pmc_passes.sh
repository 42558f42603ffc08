#!/bin/bash
# Developer tool (GPU box): rocprofv3 --pmc passes over one short bench.py run each, ONE small counter group per pass, failing
# fast. Round 3 asked for a TA / TCP set in one pass; rocprofiler refused it at start-up ("error code 38: Request exceeds the
# capabilities of the hardware to collect": more counters of one block than the block has registers), aborted inside the
# tool's signal handler and then sat until `timeout` fired -- twice, ten GPU-minutes. Here: at most two counters of a block per
# pass, a watcher that ends the pass the moment the log shows a fatal line, and a summary per kernel.
#   bash tools/pmc_passes.sh OUTDIR "<group 1 counters>" "<group 2 counters>" ...        (bench args via PMC_BENCH_ARGS)
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$1; shift
mkdir -p "$OUT"
ARGS=${PMC_BENCH_ARGS:---workload C3 --spp 32 --steps 1 --warmup 0 --no-cpu-baseline --no-c2 --no-c5}
LIMIT=${PMC_PASS_SECONDS:-240}
cd /tmp && export TMPDIR=/tmp
n=0
for GROUP in "$@"; do
    n=$((n + 1))
    D=$OUT/pass$n
    rm -rf "$D"; mkdir -p "$D"
    echo "== pass $n: $GROUP" | tee -a "$OUT/passes.log"
    # the program itself behind `--` (no env / bash -c hop: the profiler's preload has initialised the GPU by then)
    rocprofv3 --pmc $GROUP --output-format csv -d "$D" -o pmc -- python3 "$R/bench.py" $ARGS > "$D/run.log" 2>&1 &
    PID=$!
    T=0
    while kill -0 $PID 2>/dev/null; do
        sleep 2; T=$((T + 2))
        if grep -q -E "^F[0-9]{8} |error code [0-9]+|rocprofv3_error_signal_handler" "$D/run.log" 2>/dev/null; then
            echo "   fatal line in run.log after ${T}s: ending the pass" | tee -a "$OUT/passes.log"
            grep -m 2 -E "error code|Could not" "$D/run.log" | cut -c1-240 | tee -a "$OUT/passes.log"
            kill $PID 2>/dev/null; sleep 1; kill -9 $PID 2>/dev/null
            break
        fi
        if [ $T -ge $LIMIT ]; then
            echo "   no end after ${LIMIT}s: ending the pass" | tee -a "$OUT/passes.log"
            kill $PID 2>/dev/null; sleep 2; kill -9 $PID 2>/dev/null
            break
        fi
    done
    wait $PID 2>/dev/null
    echo "   rc=$? after ${T}s" | tee -a "$OUT/passes.log"
done
python3 "$R/tools/pmc_passes_summary.py" "$OUT" | tee "$OUT/summary.txt"
