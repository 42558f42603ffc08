#!/bin/bash
# Developer tool (GPU box): one rocprofv3 --pmc pass over `bench.py --workload C3 --spp 32 --steps 1` and the sums of the
# counters over the render kernel's dispatches.   bash tools/pmc_once.sh "<COUNTER> <COUNTER> ..." [bench args]
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/pmc_once
rm -rf $OUT; mkdir -p $OUT
COUNTERS=$1; shift
ARGS=${@:---workload C3 --spp 32 --steps 1 --warmup 0 --no-cpu-baseline --no-traversal --no-c2}
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --pmc $COUNTERS --output-format csv -d $OUT -o pmc -- python3 $R/bench.py $ARGS > $OUT/run.log 2>&1
echo "rc=$?"; tail -n 2 $OUT/run.log | cut -c1-300
python3 - "$OUT" <<'PY'
import collections, csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    c = collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        if "render_kernel" in r["Kernel_Name"]:
            c[r["Counter_Name"]] += float(r["Counter_Value"])
    for k in sorted(c):
        print("%-28s %18.0f" % (k, c[k]))
PY
