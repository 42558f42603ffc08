#!/usr/bin/env python3
"""Developer tool (GPU box): the random soups of tests/test_gpu_fuzz.py at other scene scales -- everything (triangles, spheres, camera, ray
batch) multiplied by SCALE -- against the oracle: closest hits (bit-exact or ties) and the film on both schedulers. DIST_EPSILON is an
absolute 1e-4 in the reference (math.rs:4), so scale is not neutral: at 1e4 an ulp of a coordinate is larger than it.
    python tools/fuzz_scale.py SCALE [SEEDS] [FIRST]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle  # noqa: E402
from pyrite_amd.project import camera, transform, vector  # noqa: E402
from pyrite_amd.renderer import Camera, Renderer, World  # noqa: E402
from test_gpu_fuzz import random_soup  # noqa: E402
from test_gpu_parity import random_rays, rel_l2  # noqa: E402

scale = float(sys.argv[1])
seeds = int(sys.argv[2]) if len(sys.argv) > 2 else 50
first = int(sys.argv[3]) if len(sys.argv) > 3 else 0
f32 = np.float32
totals = dict(rays=0, differ=0, ties=0, not_ties=0, film_cases=0, film_bad_pixels=0, film_bad_cases=0)
for seed in range(first, first + seeds):
    flat = random_soup(2000 + seed)
    flat.tri_positions = [(np.asarray(batch, dtype=f32) * f32(scale)).astype(f32) for batch in flat.tri_positions]  # one array per add_triangles call
    flat.spheres = [[f32(c * scale) for c in s] for s in flat.spheres]
    world = World(flat)
    rays = random_rays(20000, seed, [-5 * scale] * 3, [5 * scale] * 3)
    oh, _ = oracle.OracleScene(world).intersect(rays)
    gh, _, _ = world.intersect(rays)
    differ = np.nonzero((oh["shape"] != gh["shape"]) | (oh["distance"] != gh["distance"]))[0]
    ties = sum(1 for i in differ if oh["distance"][i] == gh["distance"][i])
    totals["rays"] += len(rays); totals["differ"] += len(differ); totals["ties"] += ties; totals["not_ties"] += len(differ) - ties
    for i in [i for i in differ if oh["distance"][i] != gh["distance"][i]][:2]:
        print("  seed %d ray %d: oracle %r gpu %r ray %r" % (seed, i, oh[i], gh[i], rays[i]))
    r = Renderer(pixel_samples=3, bounces=6, light_samples=2, spectrum_samples=5, tile_size=16, seed=seed)
    cam = Camera.from_project(camera.perspective(fov=60, transform=transform.look_at(**{"from": vector(0, -9 * scale, 1 * scale), "to": vector(0, 0, 0), "up": vector(z=1)})))
    cfilm = r.new_film(40, 30)
    oracle.OracleScene(world).render(r, cam, cfilm, threads=8)
    for sched in ("sm", "sync"):
        os.environ["PYRITE_SCHEDULER"] = sched
        g = r.new_film(40, 30)
        r.render(g, cam, world)
        e = rel_l2(g, cfilm)
        bad = int(((e > 1e-5) | (g.grains[..., 1] != cfilm.grains[..., 1]).any(axis=-1).reshape(-1)).sum())
        totals["film_cases"] += 1; totals["film_bad_pixels"] += bad; totals["film_bad_cases"] += 1 if bad else 0
        if bad:
            print("  seed %d %s: %d differing pixels" % (seed, sched, bad))
    world.close()
print("scale %g, %d soups:" % (scale, seeds), totals)
