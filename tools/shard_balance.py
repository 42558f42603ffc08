#!/usr/bin/env python3
"""Developer tool: time every rank's share of a sharded render on ONE GPU (load balance of pyrite_amd.distributed.plan).
    python tools/shard_balance.py [C2|C3] world_size spp"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from pyrite_amd import abi, scenes  # noqa: E402
from pyrite_amd import distributed as pdist  # noqa: E402

which, world, spp = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
if which == "C2":
    W, H = 1024, 1024
    project = scenes.c2_cornell(W, H, spp)
else:
    W, H = 1920, 1080
    project = scenes.c3_mesh_in_box(W, H, spp)
world_, cam, r, _ = scenes.build(project, seed=1)
world_.scene(0)
dev = torch.device("cuda", 0)
desc = abi.PyrFilmDesc(W, H, r.spectrum_bins, r.spectrum_span[0], r.spectrum_span[1] - r.spectrum_span[0])
stream = torch.cuda.current_stream(dev)
for sharding in ("contiguous", "tiles"):
    shares = pdist.plan(W, H, r.tile_size, world, sharding)
    times = []
    for share in shares:
        buffer = torch.zeros((max(1, share.pixels(W)), r.spectrum_bins, 2), dtype=torch.float32, device=dev)
        ms = 0.0
        for _ in range(2):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(stream)
            if share.tile_count:
                r.render_device(buffer.data_ptr(), desc, cam, world_, stream=stream.cuda_stream, device=0, share=share)
            b.record(stream)
            torch.cuda.synchronize(dev)
            ms = a.elapsed_time(b)
        times.append(ms)
    mean = sum(times) / len(times)
    print("%-10s per-rank ms: %s | max / mean = %.3f (strong-scaling efficiency bound %.3f)"
          % (sharding, " ".join("%.1f" % t for t in times), max(times) / mean, mean / max(times)), flush=True)
