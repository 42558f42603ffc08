#!/usr/bin/env python3
"""Generate the golden fixtures in this directory with the CPU oracle (oracle/liboracle.so).

    python tests/golden/make_golden.py [case ...]

The reference cannot run here (Rust, no toolchain) and has no vectors of its own, so these fixtures pin the ORACLE's
output (inputs: the scene builders in pyrite_amd/scenes.py + seeds; outputs: film grains and ray hits) so that a
regression in either the oracle or the HIP path shows up against committed data. Each .npz holds
  grains   float32 [h, w, bins, 2]   the film after one render
  rays     float32 [n, 6]            seeded test rays
  hit_distance / hit_shape / hit_u / hit_v   the oracle's World::intersect answers for those rays
  meta     json: scene name, image size, renderer parameters, seed
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import oracle  # noqa: E402
from pyrite_amd import scenes  # noqa: E402
from pyrite_amd.project import renderer  # noqa: E402

CASES = {
    "c1_spheres": lambda: scenes.c1_spheres(width=16, height=16, pixel_samples=8),
    "c2_cornell": lambda: scenes.c2_cornell(width=16, height=16, pixel_samples=8),
    "spheres_example": lambda: scenes.spheres_example(width=24, height=12, pixel_samples=8),
    "diamonds_example": lambda: scenes.diamonds_example(width=16, height=10, pixel_samples=8, bounces=16),
    "lamps_example": lambda: scenes.lamps_example(width=18, height=12, pixel_samples=8),
    "textures_example": lambda: scenes.textures_example(width=18, height=12, pixel_samples=8),
}
SEED = 3


def golden_rays(n, seed):
    rng = np.random.RandomState(seed)
    o = rng.uniform([-6, -2, 0.05], [1, 6, 5.4], size=(n, 3))
    d = rng.normal(size=(n, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    return np.concatenate([o, d], axis=1).astype(np.float32)


def build_case(name):
    project = CASES[name]()
    r = project["renderer"]
    project["renderer"] = renderer.simple(pixel_samples=r.pixel_samples, bounces=r.bounces, light_samples=r.light_samples,
                                          spectrum_samples=r.spectrum_samples, tile_size=8)
    return project


def main():
    for name in (sys.argv[1:] or CASES):  # name the cases to regenerate, or none for all
        project = build_case(name)
        world, cam, r, film = scenes.build(project, seed=SEED)
        sc = oracle.OracleScene(world)
        counters = sc.render(r, cam, film, threads=1)
        rays = golden_rays(1500, 11)
        hits, _ = sc.intersect(rays)
        meta = dict(scene=name, width=film.width, height=film.height, bins=film.bins, seed=SEED, pixel_samples=r.pixel_samples, bounces=r.bounces,
                    light_samples=r.light_samples, spectrum_samples=r.spectrum_samples, tile_size=r.tile_size, counters=counters)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), grains=film.grains, rays=rays, hit_distance=hits["distance"], hit_shape=hits["shape"],
                            hit_u=hits["u"], hit_v=hits["v"], meta=json.dumps(meta))
        print(name, film.grains.shape, "weight", film.total_weight(), counters)


if __name__ == "__main__":
    main()
