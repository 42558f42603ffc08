#!/usr/bin/env python3
"""Developer tool: run bench.py once per kernel variant / scheduler setting and print one line each.

    python tools/sweep_variants.py [--workload C3] [--spp 32] name[:ENV=VAL,...] ...

`name` is a build under pyrite_amd/csrc/variants/lib_<name>.so ("main" = the in-tree library); the optional ENV=VAL pairs are set
for that run (PYRITE_SM_LANES, PYRITE_SM_STEPS, ...). Every run is a child process under its own timeout."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = sys.argv[1:]
workload, spp, steps = "C3", "32", "2"
while args and args[0].startswith("--"):
    key, val = args[0], args[1]
    args = args[2:]
    if key == "--workload":
        workload = val
    elif key == "--spp":
        spp = val
    elif key == "--steps":
        steps = val
for spec in args:
    name, _, envs = spec.partition(":")
    env = dict(os.environ)
    if name != "main":
        env["PYRITE_GPU_LIB"] = os.path.join(ROOT, "pyrite_amd", "csrc", "variants", "lib_%s.so" % name)
    for pair in filter(None, envs.split(",")):
        k, v = pair.split("=")
        env[k] = v
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--workload", workload, "--spp", spp, "--steps", steps, "--warmup", "1", "--no-cpu-baseline",
           "--no-traversal", "--no-c2"]
    try:
        out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=400)
    except subprocess.TimeoutExpired:
        print("%-40s TIMEOUT" % spec, flush=True)
        sys.exit(1)  # a hung kernel: start nothing else
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    if not lines:
        print("%-40s FAILED rc=%d %s" % (spec, out.returncode, (out.stderr or out.stdout)[-400:].replace("\n", " | ")), flush=True)
        continue
    j = json.loads(lines[-1])
    r = j["roofline"]
    print("%-40s %8.1f Msamples/s  kernel %8.2f ms  frac %.4f  bytes/sample %.0f  check %s" % (spec, j["value"], r["kernel_ms"], r["frac"], r["bytes_per_sample"],
                                                                                              j["config"]["film_weight_check"]), flush=True)
