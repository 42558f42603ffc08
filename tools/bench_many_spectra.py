#!/usr/bin/env python3
"""Developer tool (GPU box): what the eager replay is worth -- the C3 scene as it is (four spectrum-reading programs: their values sit in LDS
value slots, looked up once per replay item) and with EXTRA unused materials of distinct spectra, which push the scene past the seven slots:
the replay then evaluates a program's spectrum record by record.      python tools/bench_many_spectra.py [spp] [extra materials]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from pyrite_amd import abi, scenes  # noqa: E402
from pyrite_amd.project import material, spectrum  # noqa: E402
from pyrite_amd.renderer import Camera, Renderer, World  # noqa: E402

spp = int(sys.argv[1]) if len(sys.argv) > 1 else 64
extra = int(sys.argv[2]) if len(sys.argv) > 2 else 6
W, H = 1920, 1080
dev = torch.device("cuda", 0)
project = scenes.c3_mesh_in_box(W, H, spp)
for n_extra in (0, extra):
    flat = scenes.c3_flat()
    rng = np.random.default_rng(1)
    for k in range(n_extra):
        flat.add_material({"surface": material.diffuse(color=spectrum(format="array", min=400.0, max=700.0, points=[float(x) for x in rng.uniform(0.1, 0.9, 7 + k)]))})
    world = World(flat)
    r = Renderer.from_project(project["renderer"], seed=1)
    cam = Camera.from_project(project["camera"])
    world.scene(0)
    film = torch.zeros((H, W, r.spectrum_bins, 2), dtype=torch.float32, device=dev)
    desc = abi.PyrFilmDesc(W, H, r.spectrum_bins, r.spectrum_span[0], r.spectrum_span[1] - r.spectrum_span[0])
    stream = torch.cuda.current_stream(dev)
    times = []
    for _ in range(5):
        film.zero_()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(stream)
        r.render_device(film.data_ptr(), desc, cam, world, stream=stream.cuda_stream, device=0)
        b.record(stream)
        torch.cuda.synchronize(dev)
        times.append(a.elapsed_time(b))
    ms = sorted(times[2:])[1]
    print("C3 at %d spp, %d extra spectra: %.1f ms  %.1f Msamples/s" % (spp, n_extra, ms, W * H * spp / ms / 1e3), flush=True)
    world.close()
