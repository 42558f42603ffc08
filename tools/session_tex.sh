#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=${1:-$R/gpurun_out/tex}
mkdir -p $OUT
cd $R
timeout -k 10 300 python3 -m pytest tests/test_reference_images.py -m gpu -x -q 2>&1 | tail -n 3
timeout -k 10 200 python3 tools/prof_textures.py > $OUT/plain.txt 2>/dev/null; cat $OUT/plain.txt
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o tex -- python3 $R/tools/prof_textures.py > $OUT/trace.log 2>&1; echo "trace rc=$?"
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $OUT/sq1 -o sq -- python3 $R/tools/prof_textures.py > $OUT/sq1.log 2>&1; echo "sq1 rc=$?"
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_WAVES SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/sq2 -o sq -- python3 $R/tools/prof_textures.py > $OUT/sq2.log 2>&1; echo "sq2 rc=$?"
find $OUT -name "*kernel_stats.csv" | head -2; find $OUT -name "*counter_collection.csv" | head
python3 - "$OUT" <<'PY'
import collections, csv, glob, sys
for f in sorted(glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)):
    for i, row in enumerate(csv.reader(open(f))):
        if i < 4: print(",".join(row)[:260])
for f in sorted(glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)):
    c = collections.defaultdict(float); n = collections.Counter()
    for r in csv.DictReader(open(f)):
        if "render_kernel" in r["Kernel_Name"]:
            c[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
    for k in sorted(c): print("%-28s %18.0f  (%d dispatches)" % (k, c[k], n[k]))
PY
