#!/usr/bin/env python3
"""Developer check: C3 (819k-triangle mesh in the x10 Cornell box) -- BVH build time, ray parity, small-render parity."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle
from pyrite_amd import scenes
from gpu_check import rel_l2, random_rays

glass = "--glass" in sys.argv
p = scenes.c3_mesh_in_box(width=96, height=54, pixel_samples=4, glass=glass, bounces=20 if glass else None)
world, cam, r, film = scenes.build(p, seed=1)
t = time.time(); world.scene(0); print("pyr_scene_create (BVH build + upload): %.2f s" % (time.time() - t), world.bvh_info())
osc = oracle.OracleScene(world)
rays = random_rays(200000, 3, [-55, 0, 0], [0, 55, 54])
oh, oc = osc.intersect(rays)
gh, ms, gc = world.intersect(rays, want_counters=True)
exact = oh["distance"] == gh["distance"]
print("rays: %.3f ms for %d rays (%.1f Mrays/s); distance exact %.5f; shape same %.5f; hits %.3f" % (ms, len(rays), len(rays) / ms / 1e3, exact.mean(), (oh["shape"] == gh["shape"]).mean(), (oh["shape"] != 0xFFFFFFFF).mean()))
print("per ray: oracle box %.1f tri %.1f | gpu box %.1f tri %.1f" % (oc["box_tests"] / len(rays), oc["triangle_tests"] / len(rays), gc["box_tests"] / len(rays), gc["triangle_tests"] / len(rays)))
ofilm = r.new_film(film.width, film.height)
t = time.time(); ocount = osc.render(r, cam, ofilm, threads=16); t1 = time.time() - t
gcount = r.render(film, cam, world, counters=True)
print("oracle render %.2f s" % t1, {k: ocount[k] for k in ("samples", "extension_rays", "shadow_rays", "exposures")})
print("gpu counters          ", {k: gcount[k] for k in ("samples", "extension_rays", "shadow_rays", "exposures")})
e = rel_l2(film.develop(), ofilm.develop()).reshape(-1)
print("weights identical:", np.array_equal(ofilm.grains[..., 1], film.grains[..., 1]), "relL2 median %.3g p99 %.3g max %.3g frac<=1e-5 %.5f" % (np.median(e), np.percentile(e, 99), e.max(), (e <= 1e-5).mean()))
