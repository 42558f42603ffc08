#!/usr/bin/env python3
"""Sum rocprofv3 --pmc counter_collection CSVs per (kernel, counter): pmc_summary.py file.csv [kernel-substring]"""
import collections
import csv
import sys

want = sys.argv[2] if len(sys.argv) > 2 else ""
tot = collections.defaultdict(float)
calls = collections.defaultdict(set)
with open(sys.argv[1]) as f:
    for row in csv.DictReader(f):
        k = row["Kernel_Name"].split("(")[0][:50]
        if want and want not in k:
            continue
        tot[(k, row["Counter_Name"])] += float(row["Counter_Value"])
        calls[k].add(row["Dispatch_Id"])
for (k, c), v in sorted(tot.items()):
    print("%-52s %-28s %18.0f  (%d dispatches)" % (k, c, v, len(calls[k])))
