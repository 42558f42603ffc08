#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric (Msamples/s of the camera-to-light renderer) on BASELINE.json's configs[1].

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A *step* is one complete pass of the hot path over the workload: the whole C2 image (1024 x 1024, 256 spp = 268.4 M
samples) rendered into a film that is already resident in HBM (scene uploaded and film allocated before the timed
region). With N > 1 the image's tiles are split into N contiguous raster ranges (strong scaling: the image is fixed), every
rank renders its range into its own film window and ONE RCCL gather brings the windows to rank 0; the gather is inside the
step. Timing: barrier + synchronize on both sides of exactly K steps, max over ranks, rank 0 prints one JSON line.

Extra objects on the line:
  roofline      algorithmic bytes (32 B per box test, 36 B per triangle test, 16 B per sphere / plane
                test, 52 B per shaded hit, 16 B per film exposure; counts from an instrumented run of the same launch,
                outside the timed region) over the kernel's average launch duration measured with HIP events on the launch
                stream, against the 8 TB/s HBM3E peak. `traffic` (PMC HBM bytes) comes from the rocprofv3 --pmc passes
                whose summary is committed under profiles/ (null when no summary for this workload is present).
  cpu_baseline  the CPU oracle (oracle/liboracle.so, a port of the reference's algorithm -- the Rust reference cannot be
                built here) timed on this host's cores, rank 0, N = 1 only, on a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (scene builder name, width, height, spp)
    "C2": ("c2_cornell", 1024, 1024, 256),
    "C1": ("c1_spheres", 256, 256, 64),
    "C3": ("c3_mesh_in_box", 1920, 1080, 1024),
}
BYTES = dict(box_tests=32, triangle_tests=36, sphere_tests=16, plane_tests=16, shaded_hits=52, exposures=16)
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s (spec)


def algorithmic_bytes(counters):
    return sum(BYTES[k] * counters[k] for k in BYTES)


def cpu_baseline(world, cam, renderer, width, height, target_seconds=15.0):
    """Time the oracle on the same scene and image with fewer samples per pixel (the loop is linear in spp, simple.rs:73)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import copy

    import oracle

    threads = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    sc = oracle.OracleScene(world)
    r = copy.copy(renderer)
    # calibrate on a few central tiles, then size the sample for ~target_seconds
    r.pixel_samples = 1
    tiles = renderer.num_tiles(width, height)
    probe = (tiles // 2, min(tiles, tiles // 2 + 4 * threads))
    film = renderer.new_film(width, height)
    t0 = time.perf_counter()
    c = sc.render(r, cam, film, threads=threads, tile_range=probe)
    rate = c["samples"] / max(time.perf_counter() - t0, 1e-6)
    spp = int(max(1, min(renderer.pixel_samples, round(rate * target_seconds / (width * height)))))
    r.pixel_samples = spp
    film = renderer.new_film(width, height)
    t0 = time.perf_counter()
    c = sc.render(r, cam, film, threads=threads)
    dt = time.perf_counter() - t0
    return {
        "value": round(c["samples"] / dt / 1e6, 4),
        "unit": "Msamples/s",
        "cores": threads,
        "kind": "port",
        "sample": "same scene and %dx%d image at %d spp of %d (all tiles), %.1f s of CPU work" % (width, height, spp, renderer.pixel_samples, dt),
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
        "kernel": "intersect_kernel<false>", "scene": "C3 mesh (819,212 triangles, %d nodes of 64 B)" % info["num_nodes"],
        "rays": n_rays, "ray_kind": "incoherent (uniform origins and directions)", "kernel_ms": round(best, 3),
        "Mrays_per_s": round(n_rays / best / 1e3, 1), "box_tests_per_ray": round(counters["box_tests"] / n_rays, 2),
        "triangle_tests_per_ray": round(counters["triangle_tests"] / n_rays, 2), "bound": "hbm", "achieved": round(achieved, 1),
        "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
    }


def load_traffic(workload):
    path = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(path):
        with open(path) as f:
            table = json.load(f)
        entry = table.get(workload)
        if entry:
            return entry.get("hbm_bytes_per_launch")
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="C2", choices=sorted(WORKLOADS))
    ap.add_argument("--spp", type=int, default=None, help="override samples per pixel (development only: the line is then marked reduced)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-traversal", action="store_true", help="skip the BVH-traversal roofline measurement on the C3 scene")
    ap.add_argument("--dev", default="", help="development overrides, e.g. spectrum_samples=1,light_samples=0,bounces=2 (marks the line reduced)")
    ap.add_argument("--seed", type=int, default=1)
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    from pyrite_amd import abi, scenes
    from pyrite_amd import distributed as pdist

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

    if world_size > 1:
        # bring up the communicator and the point-to-point channels the gather uses before anything is timed (RCCL creates
        # them lazily at the first collective of each kind): one tiny gather, the same call the step makes
        probe = torch.zeros(16, dtype=torch.float32, device="cpu" if rehearsal else device)
        dist.gather(probe, [torch.empty_like(probe) for _ in range(world_size)] if rank == 0 else None, dst=0)
        if not rehearsal:
            torch.cuda.synchronize(device)

    builder, width, height, spp = WORKLOADS[args.workload]
    reduced = (args.spp is not None and args.spp != spp) or rehearsal
    spp = args.spp or spp
    project = getattr(scenes, builder)(width=width, height=height, pixel_samples=spp)
    world, cam, renderer, _ = scenes.build(project, seed=args.seed)
    for item in filter(None, args.dev.split(",")):
        key, val = item.split("=")
        setattr(renderer, key, int(val))
        reduced = True
    world.scene(local_rank)  # BVH build + upload, outside the timed region
    bins = renderer.spectrum_bins
    film_desc = abi.PyrFilmDesc(width, height, bins, renderer.spectrum_span[0], renderer.spectrum_span[1] - renderer.spectrum_span[0])
    stream = torch.cuda.current_stream(device)

    launch_events = []  # (start, stop) HIP events around every render launch of this rank, on the launch stream

    def render_window(tile_range, rows, window, flags=0):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(stream)
        renderer.render_device(window.data_ptr(), film_desc, cam, world, stream=stream.cuda_stream, device=local_rank, flags=flags,
                               tile_range=tile_range, film_rows=rows)
        b.record(stream)
        launch_events.append((a, b))

    # Sharding (pyrite_amd/distributed.py plan): a scene small enough to live in LDS costs about the same everywhere in the
    # image (C2: slowest of 8 contiguous shares 1.03x the mean), so each rank gets one contiguous band = one launch; a big
    # scene does not (C3: 1.37x), so its tile rows are dealt round-robin. PYRITE_SHARDING overrides.
    info0 = world.bvh_info(local_rank)
    sharding = os.environ.get("PYRITE_SHARDING") or ("contiguous" if world_size == 1 or info0["node_bytes"] + info0["primitive_bytes"] <= 8 * 1024 else "cyclic")

    def step():
        return pdist.render_sharded(render_window, width, height, bins, renderer.tile_size, device, sharding=sharding)

    def fence():
        if world_size > 1:
            dist.barrier()
        torch.cuda.synchronize(device)

    for _ in range(args.warmup):
        step()
    fence()
    launch_events.clear()
    start_evt, stop_evt = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    start_evt.record(stream)
    film = None
    for k in range(args.steps):
        renderer.seed = args.seed + k % 3  # seeds 1, 2, 3 in turn; a step is one full render either way
        film = step()
    renderer.seed = args.seed
    stop_evt.record(stream)
    fence()
    elapsed = time.perf_counter() - t0
    if world_size > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ms_per_step = elapsed * 1e3 / max(args.steps, 1)
    samples = width * height * spp
    value = samples / (ms_per_step * 1e-3) / 1e6

    line = None
    if rank == 0:
        # sanity: every sample exposed spectrum_samples wavelengths into the gathered film
        total_weight = float(film[..., 1].sum(dtype=torch.float64).item()) if film is not None else 0.0
        expected_weight = float(samples) * renderer.spectrum_samples
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
            "seeds": sorted({args.seed + k % 3 for k in range(args.steps)}),
            "config": {
                "workload": "%s: %s, %dx%d, %d spp%s" % (args.workload, builder, width, height, spp, " (REDUCED spp: development run)" if reduced else ""),
                "bounces": renderer.bounces, "light_samples": renderer.light_samples, "spectrum_samples": renderer.spectrum_samples,
                "spectrum_bins": bins, "tile_size": renderer.tile_size, "triangles": len(world.flat.tri_material), "spheres": len(world.flat.spheres),
                "parallelism": "one launch on one GPU" if world_size == 1 else "%s tile bands on %d GPUs, one film gather" % (sharding, world_size),
                # samples that map outside the image are dropped as in the reference (film.rs:51-54): a few per 1e7
                "film_weight": total_weight, "film_weight_expected": expected_weight,
                "film_weight_check": "ok" if abs(total_weight - expected_weight) <= 1e-5 * expected_weight else "MISMATCH",
            },
        }
    del film

    if rank == 0:
        # kernel duration from HIP events on the launch stream: one launch per step at N = 1, one per band of this rank's
        # share otherwise (the figure is then rank 0's kernel time and rank 0's algorithmic bytes, i.e. per GPU)
        kernel_ms = sum(a.elapsed_time(b) for a, b in launch_events) / max(args.steps, 1)
        # algorithmic bytes per step: instrumented run of the same launches, outside the timed region
        shares = pdist.plan(width, height, renderer.tile_size, world_size, sharding)
        counters = None
        for tile_range, (first_row, rows) in shares[0]:
            window = torch.zeros((rows, width, bins, 2), dtype=torch.float32, device=device)
            render_window(tile_range, (first_row, rows), window, flags=abi.PYR_FLAG_COUNTERS)
            torch.cuda.synchronize(device)
            c = renderer.counters(world, local_rank)
            counters = c if counters is None else {k: counters[k] + c[k] for k in c}
            del window
        info = world.bvh_info(local_rank)
        lds_resident = info["node_bytes"] + info["primitive_bytes"] <= 8 * 1024
        forced = os.environ.get("PYRITE_SCHEDULER")
        staged = forced == "sm" or (forced != "sync" and not lds_resident)
        kernel_name = "render_kernel_sm (stage-scheduled)" if staged else "render_kernel (bounce-synchronous)"
        traversal = 32 * counters["box_tests"] + 36 * counters["triangle_tests"] + 16 * (counters["sphere_tests"] + counters["plane_tests"])
        total = algorithmic_bytes(counters)
        achieved = total / (kernel_ms * 1e-3) / 1e9
        line["roofline"] = {
            "bound": "hbm",
            "achieved": round(achieved, 1),
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 4),
            "traffic": load_traffic(args.workload),
            "kernel": kernel_name,
            "kernel_ms": round(kernel_ms, 3),
            "launches_per_step": len(shares[0]),
            "algorithmic_bytes_per_launch": int(total),
            "traversal_bytes_per_launch": int(traversal),
            "traversal_frac": round(traversal / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
            "counters": counters,
        }
        if world_size == 1 and not args.no_traversal:
            line["traversal_roofline"] = traversal_roofline(local_rank)
        if world_size == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(world, cam, renderer, width, height)
    if rank == 0:
        print(json.dumps(line), flush=True)
    if world_size > 1:
        dist.barrier()  # rank 0 ran the instrumented pass above: leave together
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
