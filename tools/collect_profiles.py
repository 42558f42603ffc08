#!/usr/bin/env python3
"""Copy the summaries tools/refresh_profiles.sh left under gpurun_out/final/ into profiles/ (tracked) and rebuild
profiles/traffic.json from the FETCH_SIZE / WRITE_SIZE passes.   python tools/collect_profiles.py [round-tag]"""
import csv
import datetime
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "final")
DST = os.path.join(ROOT, "profiles")
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"


def copy(src, dst):
    shutil.copyfile(os.path.join(SRC, src), os.path.join(DST, "%s_%s" % (tag, dst)))


def last_json_line(path):
    with open(path) as f:
        lines = [l for l in f.read().splitlines() if l.startswith("{")]
    return json.loads(lines[-1])


def counter_sum(path, counter, kernel_part):
    total, per_kernel = 0.0, {}
    with open(path) as f:
        for row in csv.DictReader(f):
            if row["Counter_Name"] == counter and kernel_part in row["Kernel_Name"]:
                per_kernel.setdefault(row["Kernel_Name"].split("(")[0], []).append(float(row["Counter_Value"]))
    return per_kernel


copy("c2_trace/c2_kernel_stats.csv", "c2_kernel_stats.csv")
if os.path.exists(os.path.join(SRC, "c3_trace", "c3_kernel_stats.csv")):
    copy("c3_trace/c3_kernel_stats.csv", "c3_kernel_stats.csv")
copy("isect_trace/isect_kernel_stats.csv", "intersect_kernel_stats.csv")
# keep only our kernels' rows of the (large) kernel trace
with open(os.path.join(SRC, "c2_trace", "c2_kernel_trace.csv")) as f, open(os.path.join(DST, tag + "_c2_kernel_trace_render.csv"), "w") as g:
    for i, line in enumerate(f):
        if i == 0 or "pyr::" in line:
            g.write(line)
pmc_files = ["c2_fetch/fetch_counter_collection.csv", "c2_write/write_counter_collection.csv"]
if os.path.exists(os.path.join(SRC, "c3_fetch", "fetch_counter_collection.csv")):
    pmc_files += ["c3_fetch/fetch_counter_collection.csv", "c3_write/write_counter_collection.csv"]
for name in pmc_files:
    out = os.path.join(DST, "%s_%s_pmc_%s.csv" % (tag, name[:2], "fetch_size" if "fetch" in name else "write_size"))
    with open(os.path.join(SRC, name)) as f, open(out, "w") as g:
        for i, line in enumerate(f):
            if i == 0 or "pyr::" in line:
                g.write(line)
for name, dst in (("c2_bench.json", "c2_bench.json"), ("c3_bench.json", "c3_bench_full.json")):
    with open(os.path.join(DST, "%s_%s" % (tag, dst)), "w") as g:
        json.dump(last_json_line(os.path.join(SRC, name)), g, indent=1)
        g.write("\n")

# timed launches are the <false, ...> (no counters) variant of the render kernel
fetch = counter_sum(os.path.join(SRC, "c2_fetch", "fetch_counter_collection.csv"), "FETCH_SIZE", "render_kernel<false")
write = counter_sum(os.path.join(SRC, "c2_write", "write_counter_collection.csv"), "WRITE_SIZE", "render_kernel<false")
(kernel, fv), = fetch.items()
(_, wv), = write.items()
fetch_kb, write_kb = sum(fv) / len(fv), sum(wv) / len(wv)
traffic = {
    "C2": {
        "hbm_bytes_per_launch": int((2.0 * fetch_kb + write_kb) * 1024.0),
        "source": "rocprofv3 --pmc FETCH_SIZE (%.1f KB) and --pmc WRITE_SIZE (%.1f KB) in separate passes on `bench.py --steps 1 --warmup 0`, %s, %s"
                  % (fetch_kb, write_kb, kernel.replace("void ", ""), datetime.date.today().isoformat()),
        "formula": "(2 * FETCH_SIZE + WRITE_SIZE) * 1024 -- FETCH_SIZE doubled per MI355X_MICROARCH.md 'HBM' (gfx950 tallies 128-B requests at 64 B); WRITE_SIZE is exact for float atomics",
        "note": "all of it is the film: two no-return f32 atomics per exposure, 32 B each at the memory side; the 2.8 KB scene is read from LDS",
    }
}
if os.path.exists(os.path.join(SRC, "c3_fetch", "fetch_counter_collection.csv")):
    fetch = counter_sum(os.path.join(SRC, "c3_fetch", "fetch_counter_collection.csv"), "FETCH_SIZE", "render_kernel_sm<false")
    write = counter_sum(os.path.join(SRC, "c3_write", "write_counter_collection.csv"), "WRITE_SIZE", "render_kernel_sm<false")
    (kernel3, fv), = fetch.items()
    (_, wv), = write.items()
    f3, w3 = sum(fv) / len(fv), sum(wv) / len(wv)
    traffic["C3"] = {
        "hbm_bytes_per_launch": int((2.0 * f3 + w3) * 1024.0),
        "source": "rocprofv3 --pmc FETCH_SIZE (%.1f KB) and --pmc WRITE_SIZE (%.1f KB) in separate passes on `bench.py --workload C3 --steps 1 --warmup 0`, %s, %s"
                  % (f3, w3, kernel3.replace("void ", ""), datetime.date.today().isoformat()),
        "formula": traffic["C2"]["formula"],
        "note": "reads: BVH nodes / primitives / shading records that missed L2 (the 56 MB tree sits in the 256 MB Infinity Cache, whose hits the counter still includes); writes: the film's f32 atomics",
    }
with open(os.path.join(DST, "traffic.json"), "w") as g:
    json.dump(traffic, g, indent=1)
    g.write("\n")
print(json.dumps(traffic, indent=1))
