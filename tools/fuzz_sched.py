#!/usr/bin/env python3
"""Developer tool (GPU box): python tools/fuzz_sched.py scene|soup|mesh SEED [REPEATS] -- one fuzz case on every scheduler, several
times: which scheduler differs from the oracle, where, and is it the same every time?"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
kind, seed = sys.argv[1], int(sys.argv[2])
repeats = int(sys.argv[3]) if len(sys.argv) > 3 else 3
import oracle
from pyrite_amd import scenes
from pyrite_amd.project import camera, transform, vector
from pyrite_amd.renderer import Camera, Renderer, World
from test_gpu_fuzz import random_project, random_soup
from test_gpu_parity import rel_l2

if kind in ("scene", "mesh"):
    project = random_project(500000 + seed, knot=True) if kind == "mesh" else random_project(1000 + seed)
    world, cam, r, _ = scenes.build(project, seed=seed)
    W, H = project["image"]["width"], project["image"]["height"]
else:
    world = World(random_soup(2000 + seed))
    r = Renderer(pixel_samples=3, bounces=6, light_samples=2, spectrum_samples=5, tile_size=16, seed=seed)
    cam = Camera.from_project(camera.perspective(fov=60, transform=transform.look_at(**{"from": vector(0, -9, 1), "to": vector(0, 0, 0), "up": vector(z=1)})))
    W, H = 40, 30
cfilm = r.new_film(W, H)
cc = oracle.OracleScene(world).render(r, cam, cfilm, threads=8)
for rep in range(repeats):
    for sched in ("sm", "sync"):
        os.environ["PYRITE_SCHEDULER"] = sched
        g = r.new_film(W, H)
        gc = r.render(g, cam, world, counters=True)
        e = rel_l2(g, cfilm).reshape(H, W)
        wdiff = (g.grains[..., 1] != cfilm.grains[..., 1]).any(axis=-1)
        bad = np.argwhere((e > 1e-5) | wdiff)
        print(rep, sched, "differing pixels", [(int(x), int(y), float(e[y, x]), bool(wdiff[y, x])) for y, x in bad][:6],
              {k: gc[k] - cc[k] for k in ("samples", "extension_rays", "shadow_rays", "shaded_hits", "exposures")}, flush=True)
