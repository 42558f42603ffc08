#!/usr/bin/env python3
"""Developer tool (GPU box): the tests' generated textures scene (scenes.textures_example: a lamp coloured by a mono texture times a
spectrum times a number -- a PRODUCT tape form, device_scene.h) at 1024 x 512 x 200 spp, with the hit tape and with PYRITE_HIT_TAPE=0
(every wavelength online).     python tools/bench_generated_textures.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pyrite_amd import abi, scenes
dev = torch.device("cuda", 0)
for hit_tape in ("1", "0"):
    os.environ["PYRITE_HIT_TAPE"] = hit_tape
    project = scenes.textures_example(1024, 512, 200)
    world, cam, r, _ = scenes.build(project, seed=1)
    world.scene(0)
    print(r.path_info(world))
    W, H = 1024, 512
    film = torch.zeros((H, W, r.spectrum_bins, 2), dtype=torch.float32, device=dev)
    desc = abi.PyrFilmDesc(W, H, r.spectrum_bins, r.spectrum_span[0], r.spectrum_span[1] - r.spectrum_span[0])
    stream = torch.cuda.current_stream(dev)
    best = None
    for _ in range(4):
        film.zero_()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(stream); r.render_device(film.data_ptr(), desc, cam, world, stream=stream.cuda_stream, device=0); b.record(stream)
        torch.cuda.synchronize(dev)
        ms = a.elapsed_time(b); best = ms if best is None else min(best, ms)
    print("generated textures scene, hit tape", hit_tape, "%.1f ms %.1f Msamples/s" % (best, W * H * r.pixel_samples / best / 1e3), flush=True)
    world.close()
