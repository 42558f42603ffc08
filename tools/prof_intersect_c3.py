import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
from pyrite_amd import scenes
from bench_intersect import rays_random
world, cam, r, film = scenes.build(scenes.c3_mesh_in_box(64, 36, 1), seed=1)
rays = rays_random(8_000_000, [-55, 1, 1], [-1, 55, 54])
for _ in range(2):
    hits, ms, _ = world.intersect(rays)
print("C3 random 8M rays: %.3f ms, %.0f Mrays/s" % (ms, len(rays) / ms / 1e3))
