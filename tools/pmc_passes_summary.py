#!/usr/bin/env python3
"""Per-kernel sums of the counters tools/pmc_passes.sh collected (one directory per pass): pmc_passes_summary.py OUTDIR.
Each pass also carries GRBM_GUI_ACTIVE when asked for, so busy / stall counters are printed next to it as a share."""
import collections
import csv
import glob
import os
import sys

out = sys.argv[1]
tot = collections.defaultdict(float)
dispatches = collections.defaultdict(set)
for f in sorted(glob.glob(os.path.join(out, "pass*", "**", "*counter_collection.csv"), recursive=True)):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0]
        k = k.replace("void pyr::(anonymous namespace)::", "")[:60]
        if "render_kernel" not in k and "intersect_kernel" not in k:
            continue
        tot[(k, row["Counter_Name"])] += float(row["Counter_Value"])
        dispatches[(k, row["Counter_Name"])].add(row["Dispatch_Id"])
kernels = sorted({k for k, _ in tot})
for k in kernels:
    print("== %s" % k)
    for (kk, c), v in sorted(tot.items()):
        if kk == k:
            print("   %-40s %20.0f   (%d dispatches)" % (c, v, len(dispatches[(kk, c)])))
