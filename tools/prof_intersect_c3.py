"""Profiling target (tools/refresh_profiles.sh): World::intersect on the C3 mesh with the same 16 M incoherent rays
bench.py's traversal_roofline quotes (uniform origins in the box, uniform directions, seed 1), three timed launches."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pyrite_amd import scenes  # noqa: E402

n_rays = int(sys.argv[1]) if len(sys.argv) > 1 else 16_000_000
world, cam, r, film = scenes.build(scenes.c3_mesh_in_box(64, 36, 1), seed=1)
rng = np.random.RandomState(1)
o = rng.uniform([-55, 1, 1], [-1, 55, 54], size=(n_rays, 3))
d = rng.normal(size=(n_rays, 3))
d /= np.linalg.norm(d, axis=1, keepdims=True)
rays = np.concatenate([o, d], axis=1).astype(np.float32)
_, _, counters = world.intersect(rays, want_counters=True)
best = None
for _ in range(3):
    hits, ms, _ = world.intersect(rays)
    best = ms if best is None else min(best, ms)
nbytes = 32 * counters["box_tests"] + 36 * counters["triangle_tests"]
print("C3 mesh, %d incoherent rays: best of 3 %.3f ms (HIP events), %.0f Mrays/s, algorithmic %.1f GB/s = %.4f of 8 TB/s; %.2f box + %.2f triangle tests per ray"
      % (n_rays, best, n_rays / best / 1e3, nbytes / best / 1e6, nbytes / best / 1e6 / 8000.0, counters["box_tests"] / n_rays, counters["triangle_tests"] / n_rays))
