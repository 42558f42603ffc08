#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric (Msamples/s of the camera-to-light renderer) on the configuration the metric is quoted
on: C3 = configs[2], the Cornell box with the 819,212-triangle mesh at 1920 x 1080 x 1024 spp (the largest single-GPU
configuration; configs[3] is the same scene across 2/4/8 GPUs).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A *step* is one complete pass of the hot path over the workload: the whole image rendered into a film that is already
resident in HBM (scene uploaded and film allocated before the timed region). With N > 1 the image's tiles are dealt
round-robin to the ranks (strong scaling: the image is fixed), every rank renders its tiles in ONE launch into its own
buffer of ringed tile blocks and ONE RCCL gather brings the buffers to rank 0, which adds them into the film; gather and
assembly are inside the step (pyr_render_simple_sharded; see pyrite_amd/distributed.py). Timing: barrier + synchronize on
both sides of exactly K steps, max over ranks, rank 0 prints one JSON line. `--workload C2|C1|C5` runs another BASELINE
config instead.

Extra objects on the line:
  roofline      the dominant kernel of the workload. `achieved` = ALGORITHMIC bytes per launch (32 B
                per box test, 36 B per triangle test, 16 B per sphere / plane test, 52 B per shaded hit, 16 B per film
                exposure; counts from instrumented runs of the same launches -- same seeds -- outside the timed region) over
                the kernel's average launch duration measured with HIP events on the launch stream, against the 8 TB/s
                HBM3E peak. THIS `frac` is the number north_star's ">= 40 % of the HBM-read roofline" is compared with for
                the render; `traversal_roofline.frac` is the same for World::intersect alone. `traffic` = PMC HBM bytes per
                launch, measured in THIS run (N = 1, full workload): two child passes of one launch under `rocprofv3 --pmc`
                (FETCH_SIZE, WRITE_SIZE; ~25 s each), falling back to the passes committed under profiles/ when rocprofv3
                is missing or --no-live-traffic is given (`traffic_source` says which); `measured_hbm_GBps` = traffic / kernel time, what the fabric really moved. `bytes_per_sample` is
                printed because the algorithmic figure rewards wasted tests: a better tree lowers both it and `achieved`.
  issue         (N = 1, full workload, with the PMC passes) what binds the kernel when it is not HBM: `valu_busy`, the share of the time the
                vector ALUs execute an instruction, and `lane_occupancy`, the active lanes per vector instruction / 64, from one more PMC pass
                (SQ counters) of a 32-spp launch of the same kernel in this run.
  c2            (N = 1, default workload only) configs[1], the 36-triangle Cornell box at 1024^2 x 256 spp, three steps: its
                scene lives in LDS, so its algorithmic bytes are LDS reads and the object says "bound": "lds".
  c5            (N = 1, default workload only) configs[4]'s workload on one GPU: the same mesh made of dispersive glass, 20
                bounces, ONE timed step at the full 4096 spp (~20 s) after a warm-up at reduced spp, with its own roofline.
  c1            (N = 1, default workload only) configs[0], the reference's own CPU-runnable case (spheres only, 256^2 x 64 spp): five
                GPU steps, and `cpu` = the oracle's render of the whole configuration timed on the host's cores.
  cpu_baseline  the CPU oracle (a port of the reference's algorithm -- the Rust reference cannot be built here) rebuilt on
                this host with -O3 -march=native (oracle/Makefile `native`; the reference builds with target-cpu=native,
                .cargo/config:2) and timed on as many threads as the process may really use (the smaller of its CPU affinity,
                the host's physical cores and its cgroup CPU quota), rank 0, N = 1 only, on a bounded sample of the same
                workload. The line carries the CPU model, the core counts and a thread-scaling table.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (scene builder, builder keywords, width, height, spp)
    "C3": ("c3_mesh_in_box", {}, 1920, 1080, 1024),
    "C2": ("c2_cornell", {}, 1024, 1024, 256),
    "C1": ("c1_spheres", {}, 256, 256, 64),
    "C5": ("c3_mesh_in_box", {"glass": True, "bounces": 20}, 1920, 1080, 4096),
}
BYTES = dict(box_tests=32, triangle_tests=36, sphere_tests=16, plane_tests=16, shaded_hits=52, exposures=16)
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s (spec)


def algorithmic_bytes(counters):
    return sum(BYTES[k] * counters[k] for k in BYTES)


def host_cpu():
    """What this process may really use: CPU model, logical / physical core counts, affinity, cgroup CPU quota."""
    model, cores = "unknown", set()
    try:
        physical, core = None, None
        with open("/proc/cpuinfo") as f:
            for line in f:
                key, _, val = line.partition(":")
                key, val = key.strip(), val.strip()
                if key == "model name":
                    model = val
                elif key == "physical id":
                    physical = val
                elif key == "core id":
                    core = val
                elif not key and physical is not None and core is not None:
                    cores.add((physical, core))
                    physical = core = None
    except OSError:
        pass
    logical = os.cpu_count() or 1
    affinity = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else logical
    quota = None
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:  # cgroup v2: "<quota us> <period us>" or "max <period>"
            q, period = f.read().split()
            if q != "max":
                quota = float(q) / float(period)
    except (OSError, ValueError):
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f, open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as g:
                q, period = float(f.read()), float(g.read())
                if q > 0:
                    quota = q / period
        except (OSError, ValueError):
            pass
    physical_cores = len(cores) or logical
    usable = max(1, min(affinity, physical_cores, int(quota) if quota and quota >= 1 else affinity))
    return {"model": model, "logical_cpus": logical, "physical_cores": physical_cores, "affinity": affinity,
            "cgroup_cpu_quota": round(quota, 2) if quota else None, "threads_used": usable}


def native_oracle():
    """oracle/liboracle_native.so: the oracle rebuilt on THIS host with -O3 -march=native (the portable build the tests use is
    -O2 -march=x86-64-v3). Falls back to the portable library, and says so, when the host has no compiler."""
    import subprocess

    native = os.path.join(ROOT, "oracle", "liboracle_native.so")
    try:
        # -B: always rebuilt -- a copy built with another host's -march=native must never be loaded here
        subprocess.check_call(["make", "-s", "-B", "-C", os.path.join(ROOT, "oracle"), "native"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        return native, "-O3 -march=native -ffp-contract=off, built on this host"
    except (OSError, subprocess.CalledProcessError):
        return None, "-O2 -march=x86-64-v3 -ffp-contract=off (portable build: this host could not rebuild the oracle)"


def load_oracle():
    """tests/oracle.py bound to the build of the oracle made on this host (once per process). Returns (module, build flags)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle

    if not hasattr(load_oracle, "flags"):
        path, load_oracle.flags = native_oracle()
        if path:
            oracle.use_library(path)
    return oracle, load_oracle.flags


def c1_cpu(world, cam, renderer, width, height):
    """configs[0] is the reference's own CPU-runnable case: the oracle renders ALL of it (256 x 256 x 64 spp), timed, on the
    threads this process may really use."""
    oracle, flags = load_oracle()
    cpu = host_cpu()
    sc = oracle.OracleScene(world)
    film = renderer.new_film(width, height)
    t0 = time.perf_counter()
    c = sc.render(renderer, cam, film, threads=cpu["threads_used"])
    dt = time.perf_counter() - t0
    sc.close()
    return {"value": round(c["samples"] / dt / 1e6, 4), "unit": "Msamples/s", "cores": cpu["threads_used"], "kind": "port", "seconds": round(dt, 3),
            "sample": "the whole configuration: %dx%d x %d spp = %d samples" % (width, height, renderer.pixel_samples, c["samples"]), "build": flags}


def cpu_baseline(world, cam, renderer, width, height, target_seconds=15.0):
    """Time the oracle on the same scene and image with fewer samples per pixel (the loop is linear in spp, simple.rs:73), on
    as many threads as this process may really use, with a thread-scaling table (>= 2 s per point) so the figure can be judged."""
    import copy

    oracle, flags = load_oracle()
    cpu = host_cpu()
    threads = cpu["threads_used"]
    t0 = time.perf_counter()
    sc = oracle.OracleScene(world)
    build_s = time.perf_counter() - t0
    r = copy.copy(renderer)
    r.pixel_samples = 1
    tiles = renderer.num_tiles(width, height)

    def rate(n_threads, seconds):
        """Msamples/s of `n_threads` on tiles from the middle of the image, for about `seconds`."""
        n_tiles, done, spent = max(4, 2 * n_threads), 0, 0.0
        while spent < seconds:
            a = max(0, tiles // 2 - n_tiles // 2)
            film = renderer.new_film(width, height)
            t = time.perf_counter()
            c = sc.render(r, cam, film, threads=n_threads, tile_range=(a, min(tiles, a + n_tiles)))
            dt = time.perf_counter() - t
            done, spent = done + c["samples"], spent + dt
            n_tiles = min(tiles, int(n_tiles * max(1.5, min(8.0, 1.2 * (seconds - spent) / max(dt, 1e-3)))))
            if n_tiles >= tiles:
                break
        return done / max(spent, 1e-6) / 1e6

    points = sorted({1, 2, 4, 8, 16, 32, threads} & set(range(1, threads + 1)) | {threads})
    scaling = {str(n): round(rate(n, 2.0), 4) for n in points}
    spp = int(max(1, min(renderer.pixel_samples, round(scaling[str(threads)] * 1e6 * target_seconds / (width * height)))))
    r.pixel_samples = spp
    film = renderer.new_film(width, height)
    t0 = time.perf_counter()
    c = sc.render(r, cam, film, threads=threads)
    dt = time.perf_counter() - t0
    sc.close()
    return {
        "value": round(c["samples"] / dt / 1e6, 4),
        "unit": "Msamples/s",
        "cores": threads,
        "kind": "port",
        "sample": "same scene and %dx%d image at %d spp of %d (all tiles), %.1f s of CPU work (+ %.1f s BVH build, not counted)"
                  % (width, height, spp, renderer.pixel_samples, dt, build_s),
        "build": flags,
        "host": cpu,
        "threads_note": "threads = min(CPU affinity, physical cores, cgroup CPU quota): more threads than the quota only get throttled",
        "thread_scaling_Msamples_per_s": scaling,
    }


def traversal_roofline(device_index, n_rays=16_000_000):
    """World::intersect alone (pyr_scene_intersect: persistent waves, dynamic ray fetch) on the C3 scene -- the
    819,212-triangle mesh in the x10 Cornell box -- with incoherent rays (uniform origins in the box, uniform directions).
    This is north_star's "HBM-read roofline on BVH traversal": algorithmic bytes = 32 B per box test + 36 B per triangle
    test (counted by the instrumented build on the same batch), over the kernel time from HIP events."""
    import numpy as np

    from pyrite_amd import scenes

    world, _, _, _ = scenes.build(scenes.c3_mesh_in_box(64, 36, 1), seed=1)
    rng = np.random.RandomState(1)
    o = rng.uniform([-55, 1, 1], [-1, 55, 54], size=(n_rays, 3))
    d = rng.normal(size=(n_rays, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    rays = np.concatenate([o, d], axis=1).astype(np.float32)
    _, _, counters = world.intersect(rays, device=device_index, want_counters=True)
    best = None
    for _ in range(3):
        _, ms, _ = world.intersect(rays, device=device_index)
        best = ms if best is None else min(best, ms)
    nbytes = 32 * counters["box_tests"] + 36 * counters["triangle_tests"]
    achieved = nbytes / (best * 1e-3) / 1e9
    info = world.bvh_info(device_index)
    world.close()
    return {
        "kernel": "intersect_kernel<false>",
        "scene": "C3 mesh (819,212 triangles): %d four-child nodes of 128 B + %d two-triangle records of 80 B = %.1f MB walked"
                 % (info["num_wide_nodes"], info["num_pair_records"], (info["wide_node_bytes"] + info["pair_record_bytes"]) / 1e6),
        "rays": n_rays, "ray_kind": "incoherent (uniform origins and directions)", "kernel_ms": round(best, 3),
        "Mrays_per_s": round(n_rays / best / 1e3, 1), "box_tests_per_ray": round(counters["box_tests"] / n_rays, 2),
        "triangle_tests_per_ray": round(counters["triangle_tests"] / n_rays, 2), "bound": "hbm", "achieved": round(achieved, 1),
        "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
    }


def load_traffic(workload):
    """PMC traffic of the workload's full single-GPU launch from the committed rocprofv3 --pmc passes: (bytes per launch, the
    kernel time of those passes, where they came from). Not a measurement of this run -- the line says so (`traffic_source`)."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(path):
        with open(path) as f:
            table = json.load(f)
        entry = table.get(workload)
        if entry:
            return entry.get("hbm_bytes_per_launch"), entry.get("kernel_ms"), "profiles/traffic.json: " + entry.get("source", "rocprofv3 --pmc")
    return None, None, None


TRAFFIC_FORMULA = ("(2 * FETCH_SIZE + WRITE_SIZE) * 1024 -- FETCH_SIZE doubled per MI355X_MICROARCH.md 'HBM' (gfx950 tallies 128-B requests at "
                   "64 B); WRITE_SIZE is exact for float atomics; Infinity-Cache hits are counted (the counters sit at the L2's fabric side)")


def pmc_pass(workload, seed, kernel_part, counters, extra_args=(), pass_seconds=90):
    """One child run of this file's one launch under `rocprofv3 --pmc <counters>` (no trace domain beside them, the program itself
    behind `--`) -> ({counter: mean over the launches of the kernel whose name contains kernel_part}, the kernel's name), or
    (None, why not). The child is an ordinary child process in a process group of its own: a pass that outlives its limit is ended
    together with the program it profiles. The parent's scene stays where it is (a few GB of 288)."""
    import csv
    import glob
    import shutil
    import signal
    import subprocess
    import tempfile

    rocprof = shutil.which("rocprofv3")
    if rocprof is None:
        return None, "rocprofv3 is not on PATH"
    if any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", ""):
        # a profiler's library already sits in this process: a second one under it would start from a process that has
        # initialised the GPU before its own program is in place
        return None, "this run is itself being profiled"
    out = tempfile.mkdtemp(prefix="pyr_pmc_", dir="/tmp")
    cmd = [rocprof, "--pmc", *counters, "--output-format", "csv", "-d", out, "-o", "pmc", "--", sys.executable, os.path.abspath(__file__),
           "--workload", workload, "--steps", "1", "--warmup", "0", "--seed", str(seed), "--no-cpu-baseline", "--no-traversal", "--no-c2", "--no-c5",
           "--no-c1", "--no-live-traffic", *extra_args]
    try:
        child = subprocess.Popen(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, start_new_session=True)
        try:
            output, _ = child.communicate(timeout=pass_seconds)
        except subprocess.TimeoutExpired:
            os.killpg(child.pid, signal.SIGKILL)
            child.communicate()
            return None, "the %s pass took more than %d s" % ("/".join(counters), pass_seconds)
        if child.returncode != 0:
            return None, "the %s pass ended with code %d: %s" % ("/".join(counters), child.returncode, output.decode(errors="replace")[-200:])
        values, kernel = {}, None
        for path in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
            with open(path) as f:
                for row in csv.DictReader(f):
                    if kernel_part in row["Kernel_Name"] and row["Counter_Name"] in counters:
                        values.setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
                        kernel = row["Kernel_Name"].split("(")[0].replace("void ", "")
        missing = [c for c in counters if c not in values]
        if missing:
            return None, "the pass has no %s row of %s" % ("/".join(missing), kernel_part)
        return {c: sum(v) / len(v) for c, v in values.items()}, kernel
    finally:
        shutil.rmtree(out, ignore_errors=True)


def live_traffic(workload, seed, kernel_part):
    """PMC traffic of THIS build on THIS box: FETCH_SIZE and WRITE_SIZE in separate passes (MI355X_MICROARCH.md "HBM")
    -> (bytes per launch, a description) or (None, why not)."""
    kb, kernel, t0 = {}, None, time.time()
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        got, kernel = pmc_pass(workload, seed, kernel_part, [counter])
        if got is None:
            return None, kernel
        kb.update(got)
    return int((2.0 * kb["FETCH_SIZE"] + kb["WRITE_SIZE"]) * 1024.0), (
        "live: rocprofv3 --pmc FETCH_SIZE (%.1f KB) and --pmc WRITE_SIZE (%.1f KB), separate passes of one launch of this build on this box "
        "(`bench.py --workload %s --steps 1 --warmup 0 --seed %d`, %s; %.0f s for both); %s" % (kb["FETCH_SIZE"], kb["WRITE_SIZE"], workload, seed, kernel,
                                                                                      time.time() - t0, TRAFFIC_FORMULA))


def live_issue(workload, seed, kernel_part, samples, compute_units, spp=32):
    """What the HBM roofline does not show: how busy the vector ALUs are and how full their instructions. One more PMC pass, of a `spp`-spp
    launch of the same kernel (SQ counters count the same per sample at any length): SQ_ACTIVE_INST_VALU, SQ_WAVE_CYCLES and
    SQ_THREAD_CYCLES_VALU are quad-cycles (MI355X_MICROARCH.md, the PMC units table). The stage-scheduled kernel is persistent -- every
    wave lives as long as the launch -- so a wave's cycles ARE the kernel's:  VALU busy = (SQ_ACTIVE_INST_VALU / SIMDs) / (SQ_WAVE_CYCLES
    / SQ_WAVES);  lane occupancy = SQ_THREAD_CYCLES_VALU / (64 * SQ_ACTIVE_INST_VALU)."""
    counters = ["SQ_ACTIVE_INST_VALU", "SQ_THREAD_CYCLES_VALU", "SQ_WAVE_CYCLES", "SQ_WAVES", "SQ_INSTS_VALU"]
    got, kernel = pmc_pass(workload, seed, kernel_part, counters, ["--spp", str(spp)])
    if got is None:
        return {"source": "no pass in this run: %s" % kernel}
    simds = 4 * compute_units
    persistent = got["SQ_WAVES"] <= 8 * simds  # at most the resident set: nothing was launched behind it
    return {
        "valu_busy": round((got["SQ_ACTIVE_INST_VALU"] / simds) / (got["SQ_WAVE_CYCLES"] / got["SQ_WAVES"]), 4) if persistent else None,
        "lane_occupancy": round(got["SQ_THREAD_CYCLES_VALU"] / (64.0 * got["SQ_ACTIVE_INST_VALU"]), 4),
        "valu_instructions_per_sample": round(got["SQ_INSTS_VALU"] / (samples * spp), 1),
        "waves": int(got["SQ_WAVES"]), "simds": simds,
        "source": "live: rocprofv3 --pmc %s on one %d-spp launch of %s in this run; valu_busy = share of the quad-cycles a SIMD's waves are resident in "
                  "which its vector ALU executes an instruction; lane_occupancy = active lanes per vector instruction / 64" % (" ".join(counters), spp, kernel),
        "reading": "a kernel whose vector ALUs are busy most of the time at a lane occupancy well under 1 is bound by vector issue, not by HBM: "
                   "`roofline.frac` (algorithmic bytes over the HBM peak) is the contract's figure, this is the binding one",
    }


class Workload:
    """One BASELINE configuration on this rank's GPU: scene uploaded, film described, ready to be stepped."""

    def __init__(self, name, args, torch, local_rank, world_size, rehearsal):
        from pyrite_amd import abi, scenes

        self.name, self.torch, self.abi = name, torch, abi
        builder, kw, self.width, self.height, spp = WORKLOADS[name]
        self.builder = builder
        self.reduced = (args.spp is not None and args.spp != spp) or rehearsal
        self.spp = args.spp or spp
        project = getattr(scenes, builder)(width=self.width, height=self.height, pixel_samples=self.spp, **kw)
        self.world, self.cam, self.renderer, _ = scenes.build(project, seed=args.seed)
        for item in filter(None, args.dev.split(",")):
            key, val = item.split("=")
            setattr(self.renderer, key, int(val))
            self.reduced = True
        self.local_rank, self.world_size = local_rank, world_size
        self.device = torch.device("cuda", local_rank)
        self.world.scene(local_rank)  # BVH build + upload, outside the timed region
        r = self.renderer
        self.bins = r.spectrum_bins
        self.film_desc = abi.PyrFilmDesc(self.width, self.height, self.bins, r.spectrum_span[0], r.spectrum_span[1] - r.spectrum_span[0])
        self.stream = torch.cuda.current_stream(self.device)
        self.launch_events = []  # (start, stop) HIP events around every render launch of this rank, on the launch stream
        self.native, self.collective = None, "torch.distributed.gather (RCCL)"
        info = self.world.bvh_info(local_rank)
        self.lds_resident = info["node_bytes"] + info["primitive_bytes"] <= 8 * 1024
        # Sharding (pyrite_amd/distributed.py plan): a scene small enough to live in LDS costs about the same everywhere in
        # the image (C2: slowest of 8 contiguous shares 1.03x the mean), so each rank gets one contiguous band of rows; a big
        # scene does not (C3: 1.37x), so its tiles are dealt round-robin. One launch per rank either way. PYRITE_SHARDING overrides.
        self.sharding = os.environ.get("PYRITE_SHARDING") or ("contiguous" if world_size == 1 or self.lds_resident else "tiles")

    def render_share(self, share, buffer, flags=0):
        """One launch: this rank's share into its buffer (pixel rows, or ringed tile blocks)."""
        torch = self.torch
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(self.stream)
        self.renderer.render_device(buffer.data_ptr(), self.film_desc, self.cam, self.world, stream=self.stream.cuda_stream, device=self.local_rank,
                                    flags=flags, share=share)
        b.record(self.stream)
        self.launch_events.append((a, b))

    def use_native(self, comm):
        """Route the steps through pyr_render_simple_sharded (render + RCCL gather + assembly inside libpyrite_gpu.so)."""
        self.native = comm
        if comm.rank == 0:
            self.native_film = self.torch.zeros((self.height, self.width, self.bins, 2), dtype=self.torch.float32, device=self.device)

    def step(self):
        from pyrite_amd import distributed as pdist

        if getattr(self, "native", None) is not None:
            torch = self.torch
            film = getattr(self, "native_film", None)
            if film is not None:
                film.zero_()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(self.stream)
            self.native.render(self.renderer, self.cam, self.world, self.film_desc, film, stream=self.stream.cuda_stream)
            b.record(self.stream)
            self.launch_events.append((a, b))  # render + this rank's side of the gather (+ assembly on rank 0)
            return film
        return pdist.render_sharded(self.render_share, self.width, self.height, self.bins, self.renderer.tile_size, self.device,
                                    sharding=self.sharding)

    def counters_for_seeds(self, seeds_used):
        """Instrumented launches of rank 0's share, one per distinct seed of the timed steps, averaged by use."""
        from pyrite_amd import distributed as pdist

        torch = self.torch
        shares = pdist.plan(self.width, self.height, self.renderer.tile_size, self.world_size, self.sharding)
        total, keep = None, self.renderer.seed
        for seed in sorted(set(seeds_used)):
            weight = seeds_used.count(seed)
            self.renderer.seed = seed
            window = torch.zeros((max(1, shares[0].pixels(self.width)), self.bins, 2), dtype=torch.float32, device=self.device)
            self.render_share(shares[0], window, flags=self.abi.PYR_FLAG_COUNTERS)
            torch.cuda.synchronize(self.device)
            c = self.renderer.counters(self.world, self.local_rank)
            total = {k: c[k] * weight for k in c} if total is None else {k: total[k] + c[k] * weight for k in c}
            del window
        self.renderer.seed = keep
        n = len(seeds_used)
        return {k: int(round(v / n)) for k, v in total.items()}

    def kernel_name(self):
        forced = os.environ.get("PYRITE_SCHEDULER")
        staged = forced == "sm" or (forced != "sync" and not self.lds_resident)
        return "render_kernel_sm (stage-scheduled)" if staged else "render_kernel (bounce-synchronous)"

    def roofline(self, counters, kernel_ms, launches_per_step, live=None):
        """`live` = (bytes per launch, description) from live_traffic() of this run, or (None, why there is none)."""
        samples = self.width * self.height * self.spp
        traversal = 32 * counters["box_tests"] + 36 * counters["triangle_tests"] + 16 * (counters["sphere_tests"] + counters["plane_tests"])
        total = algorithmic_bytes(counters)
        achieved = total / (kernel_ms * 1e-3) / 1e9
        traffic, traffic_ms, traffic_source = load_traffic(self.name)
        committed = traffic
        if live is not None and live[0] is not None:
            traffic, traffic_ms, traffic_source = live[0], None, live[1]
            if committed:
                traffic_source += "; the committed passes (profiles/traffic.json) said %d" % committed
        elif self.reduced or self.world_size != 1:
            traffic = traffic_ms = None  # the committed counters are for the full single-GPU launch
            traffic_source = None
        elif traffic is not None and traffic_ms and abs(traffic_ms - kernel_ms) > 0.02 * kernel_ms:
            # the kernel the counters were collected on took another time than this run's: not the same build any more
            traffic_source = "dropped: %s took %.1f ms per launch, this run %.1f ms" % (traffic_source, traffic_ms, kernel_ms)
            traffic = traffic_ms = None
        if live is not None and live[0] is None:
            traffic_source = "%s; no live passes in this run: %s" % (traffic_source, live[1])
        out = {
            "bound": "lds" if self.lds_resident else "hbm",
            "achieved": round(achieved, 1),
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 4),
            "traffic": traffic,
            "traffic_source": traffic_source,
            "measured_hbm_GBps": round(traffic / ((traffic_ms or kernel_ms) * 1e-3) / 1e9, 1) if traffic else None,
            "measured_hbm_frac": round(traffic / ((traffic_ms or kernel_ms) * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if traffic else None,
            "frac_is": "algorithmic-bytes rate / HBM peak (the contract's roofline definition), NOT achieved memory bandwidth: that is measured_hbm_frac",
            "compared_with_target": "frac = algorithmic bytes / kernel time / 8 TB/s is what north_star's >= 0.40 is compared with",
            "kernel": self.kernel_name(),
            "kernel_ms": round(kernel_ms, 3),
            "launches_per_step": launches_per_step,
            "algorithmic_bytes_per_launch": int(total),
            "bytes_per_sample": round(total / max(counters["samples"], 1), 1),
            "traversal_bytes_per_launch": int(traversal),
            "traversal_frac": round(traversal / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
            "counters": counters,
        }
        if self.lds_resident:
            out["note"] = ("the scene (nodes + primitives) is staged in LDS: the algorithmic bytes are LDS reads and measure traversal work; "
                           "HBM only sees the film's atomics (`traffic`)")
        return out

    def describe(self, world_size, total_weight):
        r = self.renderer
        samples = self.width * self.height * self.spp
        expected_weight = float(samples) * r.spectrum_samples
        # every sample exposes its hero wavelength; its companions too unless a bounce dispersed (simple.rs:133-139)
        dispersive = bool(WORKLOADS[self.name][1].get("glass"))
        if dispersive:
            weight_ok = float(samples) * (1.0 - 1e-5) <= total_weight <= expected_weight
        else:
            weight_ok = abs(total_weight - expected_weight) <= 1e-5 * expected_weight
        return {
            "workload": "%s: %s, %dx%d, %d spp%s" % (self.name, self.builder, self.width, self.height, self.spp,
                                                    " (REDUCED: development run)" if self.reduced else ""),
            "bounces": r.bounces, "light_samples": r.light_samples, "spectrum_samples": r.spectrum_samples,
            "spectrum_bins": self.bins, "tile_size": r.tile_size, "triangles": len(self.world.flat.tri_material), "spheres": len(self.world.flat.spheres),
            "parallelism": "one launch on one GPU" if world_size == 1 and getattr(self, "native", None) is None
                           else "%s shares on %d GPUs (one launch per GPU), one film gather by %s" % (self.sharding, world_size, self.collective),
            # samples that map outside the image are dropped as in the reference (film.rs:51-54): a few per 1e7
            "film_weight": total_weight, "film_weight_expected": ("between %.0f and %.0f (dispersed paths expose the hero wavelength only)" % (samples, expected_weight))
            if dispersive else expected_weight,
            "film_weight_check": "ok" if weight_ok else "MISMATCH",
        }


def timed_steps(wl, steps, warmup, seed, fence, dist, world_size):
    """W untimed steps, then exactly K steps between two fences; returns (ms_per_step max over ranks, film on rank 0, seeds)."""
    torch = wl.torch
    for _ in range(warmup):
        wl.step()
    fence()
    wl.launch_events.clear()
    seeds_used = []
    t0 = time.perf_counter()
    film = None
    for k in range(steps):
        wl.renderer.seed = seed + k % 3  # seeds 1, 2, 3 in turn; a step is one full render either way
        seeds_used.append(wl.renderer.seed)
        film = wl.step()
    wl.renderer.seed = seed
    fence()
    elapsed = time.perf_counter() - t0
    if getattr(wl, "native", None) is not None:
        wl.native.status()  # raises if some rank's kernels flagged their film invalid, or the communicator died
    if world_size > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=wl.device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    return elapsed * 1e3 / max(steps, 1), film, seeds_used


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="C3", choices=sorted(WORKLOADS))
    ap.add_argument("--spp", type=int, default=None, help="override samples per pixel (development only: the line is then marked reduced)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-traversal", action="store_true", help="skip the BVH-traversal roofline measurement on the C3 scene")
    ap.add_argument("--no-c2", action="store_true", help="skip the extra C2 (configs[1]) measurement")
    ap.add_argument("--no-c5", action="store_true", help="skip the extra C5 (configs[4]'s workload on one GPU) measurement")
    ap.add_argument("--no-live-traffic", action="store_true", help="skip the two rocprofv3 --pmc child passes (N = 1, full workload) that measure `roofline.traffic` "
                                                                  "on this box; the line then carries the committed passes' figure")
    ap.add_argument("--no-c1", action="store_true", help="skip the extra C1 (configs[0]: GPU next to the oracle's full CPU render) measurement")
    ap.add_argument("--dev", default="", help="development overrides, e.g. spectrum_samples=1,light_samples=0,bounces=2 (marks the line reduced)")
    ap.add_argument("--seed", type=int, default=1)
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world_size = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world_size != args.gpus:
        if world_size == 1 and args.gpus > 1:
            sys.exit("--gpus %d needs the torch.distributed.run launcher (WORLD_SIZE is 1)" % args.gpus)
        sys.exit("--gpus %d does not match WORLD_SIZE %d" % (args.gpus, world_size))
    if not torch.cuda.is_available():
        sys.exit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    # PYRITE_BENCH_REHEARSAL=1 (development): all ranks share the visible GPUs and talk over gloo, to exercise the N > 1
    # code path on a one-GPU box; the line is marked and is not a measurement of anything.
    rehearsal = os.environ.get("PYRITE_BENCH_REHEARSAL") == "1"
    # PYRITE_BENCH_FORCE_NATIVE=1 (development, one rank): walk the N > 1 branch below -- communicator, probe step, the steps
    # through pyr_render_simple_sharded -- with a one-rank RCCL communicator, the only form of it a one-GPU box can run
    force_native = os.environ.get("PYRITE_BENCH_FORCE_NATIVE") == "1" and world_size == 1
    if force_native:
        os.environ["PYRITE_FORCE_RCCL"] = "1"
        os.environ.setdefault("PYRITE_SHARDING", "tiles")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
    if rehearsal:
        local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world_size > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=device)
        # bring up the communicator and the point-to-point channels the gather uses before anything is timed (RCCL creates
        # them lazily at the first collective of each kind): one tiny gather, the same call the step makes
        probe = torch.zeros(16, dtype=torch.float32, device="cpu" if rehearsal else device)
        dist.gather(probe, [torch.empty_like(probe) for _ in range(world_size)] if rank == 0 else None, dst=0)
        if not rehearsal:
            torch.cuda.synchronize(device)

    if force_native:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=device)

    def fence():
        if world_size > 1:
            dist.barrier()
        torch.cuda.synchronize(device)

    wl = Workload(args.workload, args, torch, local_rank, world_size, rehearsal)
    if rehearsal:
        wl.collective = "torch.distributed.gather over gloo (PYRITE_BENCH_REHEARSAL: ranks share the GPUs; not a measurement)"
    if (world_size > 1 or force_native) and not rehearsal and wl.sharding == "tiles" and os.environ.get("PYRITE_BENCH_COLLECTIVE", "native") == "native":
        # The gather inside the library (pyr_render_simple_sharded: grouped ncclSend / ncclRecv). If the communicator cannot be
        # made on this node, every rank falls back TOGETHER to torch.distributed.gather over the same plan, and the line says so.
        from pyrite_amd import distributed as pdist

        ok, comm, why = 1, None, ""
        try:
            comm = pdist.NativeSharded(local_rank)
        except Exception as e:  # noqa: BLE001 -- any failure means "use the other collective", reported below
            ok, why = 0, str(e)
        flag = torch.tensor([ok], dtype=torch.int32, device=device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 1:
            wl.use_native(comm)
            wl.collective = "pyr_render_simple_sharded (ncclSend/ncclRecv group inside libpyrite_gpu.so)"
            # one untimed step through it before anything is timed: a gather that fails on this node (the library reports it on
            # every rank, multi.cpp) sends all ranks to the other collective together instead of ending the run
            try:
                wl.step()
                torch.cuda.synchronize(device)
                comm.status()
            except Exception as e:  # noqa: BLE001
                ok, why = 0, str(e)
            flag.fill_(ok)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            wl.launch_events.clear()
            if int(flag.item()) != 1:
                comm.close()
                wl.native = None
                if hasattr(wl, "native_film"):
                    del wl.native_film
                wl.collective = "torch.distributed.gather (RCCL); the native gather failed on some rank%s" % (": " + why if why else "")
        else:
            if comm is not None:
                comm.close()
            wl.collective = "torch.distributed.gather (RCCL); the native communicator failed on some rank%s" % (": " + why if why else "")
        if force_native:
            wl.collective += " -- PYRITE_BENCH_FORCE_NATIVE: one rank, a rehearsal of the code path"
    ms_per_step, film, seeds_used = timed_steps(wl, args.steps, args.warmup, args.seed, fence, dist, world_size)
    samples = wl.width * wl.height * wl.spp
    value = samples / (ms_per_step * 1e-3) / 1e6

    line, mismatch = None, False
    if rank == 0:
        # sanity: every sample exposed spectrum_samples wavelengths into the gathered film
        total_weight = float(film[..., 1].sum(dtype=torch.float64).item()) if film is not None else 0.0
        config = wl.describe(world_size, total_weight)
        mismatch = config["film_weight_check"] != "ok"
        line = {
            "metric": "Msamples/sec",
            "value": round(value, 3),
            "unit": "Msamples/s",
            "n_gpus": world_size,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3),
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "seeds": sorted(set(seeds_used)),
            "config": config,
        }
    del film

    if rank == 0:
        # kernel duration from HIP events on the launch stream: one launch per step and rank (the figure is rank 0's kernel
        # time and rank 0's algorithmic bytes, i.e. per GPU)
        launches = len(wl.launch_events)
        kernel_ms = sum(a.elapsed_time(b) for a, b in wl.launch_events) / max(args.steps, 1)
        counters = wl.counters_for_seeds(seeds_used)
        live = None
        if world_size == 1 and not wl.reduced and not args.no_live_traffic and not force_native:
            staged = wl.kernel_name().startswith("render_kernel_sm")
            live = live_traffic(wl.name, args.seed, "render_kernel_sm<false" if staged else "render_kernel<false")
        line["roofline"] = wl.roofline(counters, kernel_ms, launches // max(args.steps, 1), live)
        if live is not None and live[0] is not None:
            line["issue"] = live_issue(wl.name, args.seed, "render_kernel_sm<false" if staged else "render_kernel<false", wl.width * wl.height,
                                       torch.cuda.get_device_properties(local_rank).multi_processor_count)
        if world_size == 1 and not args.no_traversal:
            line["traversal_roofline"] = traversal_roofline(local_rank)
        if world_size == 1 and args.workload == "C3" and not args.no_c2 and not wl.reduced:
            # configs[1] beside the headline: three steps of the LDS-resident Cornell box
            c2_args = argparse.Namespace(**{**vars(args), "spp": None, "dev": ""})
            c2 = Workload("C2", c2_args, torch, local_rank, 1, False)
            c2_ms, c2_film, c2_seeds = timed_steps(c2, 3, 1, args.seed, fence, dist, 1)
            c2_kernel_ms = sum(a.elapsed_time(b) for a, b in c2.launch_events) / 3
            c2_config = c2.describe(1, float(c2_film[..., 1].sum(dtype=torch.float64).item()))
            mismatch = mismatch or c2_config["film_weight_check"] != "ok"
            del c2_film
            line["c2"] = {"value": round(c2.width * c2.height * c2.spp / (c2_ms * 1e-3) / 1e6, 3), "unit": "Msamples/s", "steps": 3, "warmup": 1,
                          "ms_per_step": round(c2_ms, 3), "config": c2_config,
                          "roofline": c2.roofline(c2.counters_for_seeds(c2_seeds), c2_kernel_ms, 1)}
            c2.world.close()
        if world_size == 1 and args.workload == "C3" and not args.no_c5 and not wl.reduced:
            # configs[4]'s workload beside the headline: the glass mesh, 20 bounces -- a warm-up at 64 spp, then ONE timed step at
            # the full 4096 spp and its counters pass
            warm_args = argparse.Namespace(**{**vars(args), "spp": 64, "dev": ""})
            c5_warm = Workload("C5", warm_args, torch, local_rank, 1, False)
            c5_warm.step()
            fence()
            c5_warm.world.close()
            del c5_warm
            c5_args = argparse.Namespace(**{**vars(args), "spp": None, "dev": ""})
            c5 = Workload("C5", c5_args, torch, local_rank, 1, False)
            c5_ms, c5_film, c5_seeds = timed_steps(c5, 1, 0, args.seed, fence, dist, 1)
            c5_kernel_ms = sum(a.elapsed_time(b) for a, b in c5.launch_events)
            c5_config = c5.describe(1, float(c5_film[..., 1].sum(dtype=torch.float64).item()))
            mismatch = mismatch or c5_config["film_weight_check"] != "ok"
            del c5_film
            line["c5"] = {"value": round(c5.width * c5.height * c5.spp / (c5_ms * 1e-3) / 1e6, 3), "unit": "Msamples/s", "steps": 1, "warmup": "1 at 64 spp",
                          "ms_per_step": round(c5_ms, 3), "config": c5_config, "roofline": c5.roofline(c5.counters_for_seeds(c5_seeds), c5_kernel_ms, 1)}
            c5.world.close()
            del c5
        if world_size == 1 and args.workload == "C3" and not args.no_c1 and not args.no_cpu_baseline and not wl.reduced:
            # configs[0], the reference's own CPU-runnable case, whole: five GPU steps (a 5 ms launch each) next to the oracle's
            # full render of the same 256 x 256 x 64 spp on the host's cores
            c1_args = argparse.Namespace(**{**vars(args), "spp": None, "dev": ""})
            c1 = Workload("C1", c1_args, torch, local_rank, 1, False)
            c1_ms, c1_film, c1_seeds = timed_steps(c1, 5, 1, args.seed, fence, dist, 1)
            c1_kernel_ms = sum(a.elapsed_time(b) for a, b in c1.launch_events) / 5
            c1_config = c1.describe(1, float(c1_film[..., 1].sum(dtype=torch.float64).item()))
            mismatch = mismatch or c1_config["film_weight_check"] != "ok"
            del c1_film
            line["c1"] = {"value": round(c1.width * c1.height * c1.spp / (c1_ms * 1e-3) / 1e6, 3), "unit": "Msamples/s", "steps": 5, "warmup": 1,
                          "ms_per_step": round(c1_ms, 3), "config": c1_config, "roofline": c1.roofline(c1.counters_for_seeds(c1_seeds), c1_kernel_ms, 1),
                          "cpu": c1_cpu(c1.world, c1.cam, c1.renderer, c1.width, c1.height),
                          "note": "configs[0]: spheres only, the scene lives in LDS; the CPU figure is the oracle's render of the WHOLE configuration"}
            c1.world.close()
        if world_size == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(wl.world, wl.cam, wl.renderer, wl.width, wl.height)
        print(json.dumps(line), flush=True)
    if world_size > 1:
        dist.barrier()  # rank 0 ran the instrumented pass above: leave together
        dist.destroy_process_group()
    elif force_native:
        dist.destroy_process_group()
    if mismatch:
        sys.exit("film weight check failed: the film does not hold samples x spectrum_samples exposures")


if __name__ == "__main__":
    main()
