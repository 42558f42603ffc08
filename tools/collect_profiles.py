#!/usr/bin/env python3
"""Copy the summaries tools/refresh_profiles.sh left under gpurun_out/final/ into profiles/ (tracked) and rebuild
profiles/traffic.json from the FETCH_SIZE / WRITE_SIZE passes.   python tools/collect_profiles.py [round-tag]

What lands in profiles/ (per round tag, e.g. r02):
  <tag>_<cfg>_kernel_stats.csv        rocprofv3 --kernel-trace --stats of the bench command (average launch duration per kernel)
  <tag>_<cfg>_pmc_fetch_size.csv / _write_size.csv   the raw counter rows of our kernels (separate --pmc passes)
  <tag>_<cfg>_sq_counters.txt         SQ counter sums of the dominant kernel + the derived lane occupancy
  <tag>_intersect16m_*.csv / .txt     the same for World::intersect on the 16 M-ray batch bench.py quotes
  <tag>_c3_bench_full.json            the bench line of the driver's command (`bench.py --gpus 1 --steps 20 --warmup 5`)
  traffic.json                        HBM bytes per launch (2 * FETCH_SIZE + WRITE_SIZE, MI355X_MICROARCH.md "HBM") and the
                                      rocprof average duration of the same kernel, read by bench.py"""
import collections
import csv
import datetime
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "final")
DST = os.path.join(ROOT, "profiles")
tag = sys.argv[1] if len(sys.argv) > 1 else "r03"


def have(*parts):
    return os.path.exists(os.path.join(SRC, *parts))


def copy(src, dst):
    shutil.copyfile(os.path.join(SRC, src), os.path.join(DST, "%s_%s" % (tag, dst)))


def last_json_line(path):
    with open(path) as f:
        lines = [l for l in f.read().splitlines() if l.startswith("{")]
    return json.loads(lines[-1])


def counters(path, kernel_part):
    """{counter: [value per dispatch]} for kernels whose name contains kernel_part."""
    out, name = collections.defaultdict(list), None
    with open(path) as f:
        for row in csv.DictReader(f):
            if kernel_part in row["Kernel_Name"]:
                out[row["Counter_Name"]].append(float(row["Counter_Value"]))
                name = row["Kernel_Name"].split("(")[0].replace("void ", "")
    return out, name


def keep_our_rows(src, dst):
    with open(os.path.join(SRC, src)) as f, open(os.path.join(DST, dst), "w") as g:
        for i, line in enumerate(f):
            if i == 0 or "pyr::" in line:
                g.write(line)


def kernel_average_ms(stats_csv, kernel_part):
    with open(stats_csv) as f:
        for row in csv.DictReader(f):
            if kernel_part in row["Name"]:
                return float(row["AverageNs"]) / 1e6, int(row["Calls"])
    return None, 0


FORMULA = ("(2 * FETCH_SIZE + WRITE_SIZE) * 1024 -- FETCH_SIZE doubled per MI355X_MICROARCH.md 'HBM' (gfx950 tallies 128-B requests at 64 B); "
           "WRITE_SIZE is exact for float atomics; Infinity-Cache hits are counted (the counters sit at the L2's fabric side)")
traffic = {}
if os.path.exists(os.path.join(DST, "traffic.json")):
    with open(os.path.join(DST, "traffic.json")) as f:
        traffic = json.load(f)

for cfg, timed_kernel, label in (("c3", "render_kernel_sm<false", "C3"), ("c2", "render_kernel<false", "C2"), ("c5", "render_kernel_sm<false", "C5")):
    trace = cfg + "_trace_full" if cfg == "c5" else cfg + "_trace"  # C5's plain trace is a reduced-spp run; the traffic belongs to the full launch
    if not have(trace, cfg + "_kernel_stats.csv"):
        continue
    if cfg != "c5":
        copy("%s/%s_kernel_stats.csv" % (trace, cfg), "%s_kernel_stats.csv" % cfg)
    else:
        copy("%s/%s_kernel_stats.csv" % (trace, cfg), "c5_full_kernel_stats.csv")
    ms, calls = kernel_average_ms(os.path.join(SRC, trace, cfg + "_kernel_stats.csv"), timed_kernel)
    if have(cfg + "_fetch", "fetch_counter_collection.csv") and have(cfg + "_write", "write_counter_collection.csv"):
        keep_our_rows(cfg + "_fetch/fetch_counter_collection.csv", "%s_%s_pmc_fetch_size.csv" % (tag, cfg))
        keep_our_rows(cfg + "_write/write_counter_collection.csv", "%s_%s_pmc_write_size.csv" % (tag, cfg))
        fetch, kernel = counters(os.path.join(SRC, cfg + "_fetch", "fetch_counter_collection.csv"), timed_kernel)
        write, _ = counters(os.path.join(SRC, cfg + "_write", "write_counter_collection.csv"), timed_kernel)
        fkb, wkb = sum(fetch["FETCH_SIZE"]) / len(fetch["FETCH_SIZE"]), sum(write["WRITE_SIZE"]) / len(write["WRITE_SIZE"])
        traffic[label] = {
            "hbm_bytes_per_launch": int((2.0 * fkb + wkb) * 1024.0),
            "kernel_ms": round(ms, 3) if ms else None,
            "source": "rocprofv3 --pmc FETCH_SIZE (%.1f KB) and --pmc WRITE_SIZE (%.1f KB) in separate passes on `bench.py --workload %s --steps 1 --warmup 0`, "
                      "%s; kernel_ms = rocprofv3 --kernel-trace --stats average over %d launches; %s" % (fkb, wkb, label, kernel, calls, datetime.date.today().isoformat()),
            "formula": FORMULA,
        }

for cfg, kernel_part in (("c3", "render_kernel_sm<false"), ("c5", "render_kernel_sm<false")):
    rows = {}
    for part in ("sq1", "sq2", "tcc"):
        path = os.path.join(SRC, "%s_%s" % (cfg, part))
        if os.path.isdir(path):
            for name in os.listdir(path):
                if name.endswith("counter_collection.csv"):
                    c, kernel = counters(os.path.join(path, name), kernel_part)
                    rows.update({k: sum(v) / len(v) for k, v in c.items()})
    if rows:
        with open(os.path.join(DST, "%s_%s_sq_counters.txt" % (tag, cfg)), "w") as g:
            g.write("# rocprofv3 --pmc (separate passes of <= 8 SQ counters), bench.py --workload %s --spp 32 --steps 1 --warmup 0, %s (the timed build), %s\n"
                    % (cfg.upper(), kernel_part, datetime.date.today().isoformat()))
            for k in sorted(rows):
                g.write("%-28s %18.0f\n" % (k, rows[k]))
            if "SQ_THREAD_CYCLES_VALU" in rows and "SQ_ACTIVE_INST_VALU" in rows:
                g.write("VALU lane occupancy = SQ_THREAD_CYCLES_VALU / (64 * SQ_ACTIVE_INST_VALU) = %.4f\n" % (rows["SQ_THREAD_CYCLES_VALU"] / (64.0 * rows["SQ_ACTIVE_INST_VALU"])))
            if "SQ_WAVE_CYCLES" in rows and "SQ_WAIT_ANY" in rows:
                g.write("wave cycles parked on a wait = SQ_WAIT_ANY / SQ_WAVE_CYCLES = %.4f\n" % (rows["SQ_WAIT_ANY"] / rows["SQ_WAVE_CYCLES"]))
            if "TCC_HIT_sum" in rows:
                g.write("L2 hit rate = %.4f\n" % (rows["TCC_HIT_sum"] / (rows["TCC_HIT_sum"] + rows["TCC_MISS_sum"])))

if have("isect_trace", "isect_kernel_stats.csv"):
    copy("isect_trace/isect_kernel_stats.csv", "intersect16m_kernel_stats.csv")
    ms, calls = kernel_average_ms(os.path.join(SRC, "isect_trace", "isect_kernel_stats.csv"), "intersect_kernel<false")
    lines = []
    if have("isect_plain.log"):
        lines.append(open(os.path.join(SRC, "isect_plain.log")).read().strip().splitlines()[-1])
    lines.append("rocprofv3 --kernel-trace --stats: intersect_kernel<false> average %.3f ms over %d launches" % (ms, calls))
    vals = {}
    for part in ("isect_fetch", "isect_tcc", "isect_tcp"):
        path = os.path.join(SRC, part)
        if os.path.isdir(path):
            for name in os.listdir(path):
                if name.endswith("counter_collection.csv"):
                    c, _ = counters(os.path.join(path, name), "intersect_kernel<false")
                    vals.update({k: sum(v) / len(v) for k, v in c.items()})
    for k in sorted(vals):
        lines.append("%-30s %16.0f per launch" % (k, vals[k]))
    if "FETCH_SIZE" in vals:
        lines.append("fabric-side read bytes per launch = 2 * FETCH_SIZE KB = %.2f GB (Infinity-Cache hits included; the 56 MB tree + 39 MB of primitives fit in the 256 MB cache)"
                     % (2 * vals["FETCH_SIZE"] * 1024 / 1e9))
    if "TCC_HIT_sum" in vals:
        lines.append("L2 hit rate = %.3f" % (vals["TCC_HIT_sum"] / (vals["TCC_HIT_sum"] + vals["TCC_MISS_sum"])))
    if "TCP_TCC_READ_REQ_sum" in vals:
        lines.append("L1 (TCP) hit rate = 1 - TCP_TCC_READ_REQ / TCP_TOTAL_CACHE_ACCESSES = %.3f" % (1 - vals["TCP_TCC_READ_REQ_sum"] / vals["TCP_TOTAL_CACHE_ACCESSES_sum"]))
    with open(os.path.join(DST, "%s_intersect16m_summary.txt" % tag), "w") as g:
        g.write("# World::intersect on the C3 mesh, 16 M incoherent rays (tools/prof_intersect_c3.py), %s\n" % datetime.date.today().isoformat())
        g.write("\n".join(lines) + "\n")

if have("c5_trace", "c5_kernel_stats.csv"):
    copy("c5_trace/c5_kernel_stats.csv", "c5_kernel_stats.csv")
if have("c5_bench.log"):
    with open(os.path.join(DST, "%s_c5_bench_full.json" % tag), "w") as g:
        json.dump(last_json_line(os.path.join(SRC, "c5_bench.log")), g, indent=1)
        g.write("\n")
if have("c3_bench.json"):
    with open(os.path.join(DST, "%s_c3_bench_full.json" % tag), "w") as g:
        json.dump(last_json_line(os.path.join(SRC, "c3_bench.json")), g, indent=1)
        g.write("\n")
with open(os.path.join(DST, "traffic.json"), "w") as g:
    json.dump(traffic, g, indent=1)
    g.write("\n")
print(json.dumps(traffic, indent=1))
