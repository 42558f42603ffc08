#!/usr/bin/env python3
"""Developer study: where does the 0.90x floor luminance of the spheres example against the reference's
hq_example.png come from? Renders pyrite/test/spheres/spheres.lua's scene on the GPU at the project's own size and sample
count under a few hypotheses and prints, per region of the image and per channel (linear light, 8 x 8 block means), the
ratio render / reference.   python tools/spheres_image_study.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pyrite_amd import develop, images, scenes  # noqa: E402

ref = np.load(os.path.join(ROOT, "tests", "golden", "reference_example_images.npz"))["spheres"].astype(np.float64)
LUMA = np.array([0.2126, 0.7152, 0.0722])
REGIONS = {
    "floor front centre (lamp-lit)": (slice(27, 32), slice(24, 40)),
    "floor front sides": (slice(27, 32), slice(4, 20)),
    "floor far (near horizon)": (slice(17, 20), slice(2, 10)),
    "left ball (red curve)": (slice(8, 18), slice(4, 12)),
    "right ball (mirror / green mix)": (slice(8, 18), slice(52, 60)),
}


def render(**overrides):
    project = scenes.spheres_example(512, 256, 600)
    project["renderer"] = project["renderer"].with_(**overrides)
    world, cam, r, film = scenes.build(project, seed=1)
    r.render(film, cam, world)
    rgb = develop.develop(film)
    lin = images.srgb_to_linear(rgb.astype(np.float64) / 255.0).astype(np.float64)
    world.close()
    return lin.reshape(32, 8, 64, 8, 3).mean((1, 3))


print("noise floor of the comparison: two renders of the project with different seeds differ by (median |ratio - 1| per cell, floor region):")
a = render()
project = scenes.spheres_example(512, 256, 600)
world, cam, r, film = scenes.build(project, seed=2)
r.render(film, cam, world)
b = images.srgb_to_linear(develop.develop(film).astype(np.float64) / 255.0).astype(np.float64).reshape(32, 8, 64, 8, 3).mean((1, 3))
fl = (slice(27, 32), slice(4, 60))
print("   %.4f" % np.median(np.abs((a[fl] @ LUMA) / (b[fl] @ LUMA) - 1)))
for label, kw in (("project as today's reference reads it (bounces 8, light_samples 4, 64 bins)", {}), ("bounces 16", {"bounces": 16}), ("bounces 32", {"bounces": 32}),
                  ("light_samples 1", {"light_samples": 1}), ("spectrum_resolution 50 (the lua's ignored `spectrum_bins = 50` honoured)", {"spectrum_resolution": 50}), ("spectrum_samples 1", {"spectrum_samples": 1})):
    mine = a if not kw else render(**kw)
    print(label)
    for name, sl in REGIONS.items():
        m, rr = mine[sl].reshape(-1, 3).mean(0), ref[sl].reshape(-1, 3).mean(0)
        print("   %-34s Y %.3f | R %.3f G %.3f B %.3f   (reference linear RGB %.3f %.3f %.3f)" % (name, (m @ LUMA) / (rr @ LUMA), *(m / np.maximum(rr, 1e-6)), *rr))
