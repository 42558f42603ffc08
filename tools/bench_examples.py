#!/usr/bin/env python3
"""Developer tool: the reference's example scenes (interpreter programs) at their projects' own sizes: the default (the stage
scheduler; with the hit tape of round 4 where the scene's colour programs allow it) and PYRITE_HIT_TAPE=0 (the online form of
round 3).     python tools/bench_examples.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from pyrite_amd import abi, scenes  # noqa: E402

dev = torch.device("cuda", 0)
cases = {
    "spheres 512x256x600": scenes.spheres_example(512, 256, 600),
    "diamonds 512x300x200 (256 bounces)": scenes.diamonds_example(512, 300, 200, bounces=256),
    "lamps 384x256x256": scenes.lamps_example(384, 256, 256),
    "textures 1024x512x400": scenes.textures_reference_example(os.path.join(ROOT, "tests", "golden", "textures"), 1024, 512, 400),
}
only = os.environ.get("EXAMPLES")  # e.g. EXAMPLES=textures,spheres
for name, project in cases.items():
  if only and name.split()[0] not in only.split(","):
      continue
  for hit_tape in (os.environ.get("HIT_TAPE_MODES", "1,0").split(",")):
    os.environ["PYRITE_HIT_TAPE"] = hit_tape  # read when the scene is created
    world, cam, r, _ = scenes.build(project, seed=1)
    world.scene(0)
    W, H = project["image"]["width"], project["image"]["height"]
    film = torch.zeros((H, W, r.spectrum_bins, 2), dtype=torch.float32, device=dev)
    desc = abi.PyrFilmDesc(W, H, r.spectrum_bins, r.spectrum_span[0], r.spectrum_span[1] - r.spectrum_span[0])
    stream = torch.cuda.current_stream(dev)
    for sched in ("hit tape" if hit_tape == "1" else "online",):
        best = None
        try:
            for _ in range(3):
                film.zero_()
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(stream)
                r.render_device(film.data_ptr(), desc, cam, world, stream=stream.cuda_stream, device=0)
                b.record(stream)
                torch.cuda.synchronize(dev)
                ms = a.elapsed_time(b)
                best = ms if best is None else min(best, ms)
            print("%-36s %-8s %9.2f ms %8.1f Msamples/s  weight %.6g" % (name, sched, best, W * H * r.pixel_samples / best / 1e3, float(film[..., 1].sum(dtype=torch.float64))), flush=True)
        except Exception as e:  # noqa: BLE001
            print("%-36s %-8s failed: %s" % (name, sched, e), flush=True)
    world.close()
