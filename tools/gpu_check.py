#!/usr/bin/env python3
"""Developer check on a GPU box: HIP path vs CPU oracle on small renders and ray batches; prints parity statistics."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import oracle  # noqa: E402
from pyrite_amd import scenes  # noqa: E402


def rel_l2(a, b):
    num = np.sqrt(((a - b) ** 2).sum(axis=-1))
    den = np.sqrt((b ** 2).sum(axis=-1)) + 1e-6
    return num / den


def random_rays(n, seed, lo, hi):
    rng = np.random.RandomState(seed)
    o = rng.uniform(lo, hi, size=(n, 3))
    d = rng.normal(size=(n, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    return np.concatenate([o, d], axis=1).astype(np.float32)


def check(name, project, seed=1, threads=8):
    world, cam, r, film = scenes.build(project, seed=seed)
    print("==", name, "tris", len(world.flat.tri_material), "spheres", len(world.flat.spheres), "lamps", len(world.flat.lamps))
    print("   bvh", world.bvh_info())
    osc = oracle.OracleScene(world)

    rays = random_rays(200000, 3, [-6, -1, -0.5], [0.5, 6, 6])
    t0 = time.time()
    oh, oc = osc.intersect(rays)
    t1 = time.time()
    gh, ms, gc = world.intersect(rays, want_counters=True)
    same_shape = oh["shape"] == gh["shape"]
    hit = oh["shape"] != 0xFFFFFFFF
    dt = np.abs(oh["distance"][hit & same_shape] - gh["distance"][hit & same_shape])
    print("   intersect: oracle %.2fs, gpu %.3f ms; shape mismatches %d / %d; max |dt| %.3g; hits %d" % (
        t1 - t0, ms, int((~same_shape).sum()), len(rays), float(dt.max()) if len(dt) else 0.0, int(hit.sum())))
    print("   counters oracle box %d tri %d sph %d | gpu box %d tri %d sph %d" % (
        oc["box_tests"], oc["triangle_tests"], oc["sphere_tests"], gc["box_tests"], gc["triangle_tests"], gc["sphere_tests"]))

    ofilm = r.new_film(film.width, film.height)
    t0 = time.time()
    ocount = osc.render(r, cam, ofilm, threads=threads)
    t1 = time.time()
    gcount = r.render(film, cam, world, counters=True)
    t2 = time.time()
    print("   render: oracle %.2fs gpu(with counters, incl. copies) %.2fs" % (t1 - t0, t2 - t1))
    print("   oracle counters", ocount)
    print("   gpu    counters", gcount)
    od, gd = ofilm.develop(), film.develop()
    w_equal = np.array_equal(ofilm.grains[..., 1], film.grains[..., 1])
    e = rel_l2(gd, od).reshape(-1)
    print("   weights identical:", w_equal, " total weight", ofilm.total_weight(), film.total_weight())
    print("   relL2 per pixel: median %.3g  p90 %.3g  p99 %.3g  max %.3g  frac<=1e-3: %.4f  mean %.3g" % (
        np.median(e), np.percentile(e, 90), np.percentile(e, 99), e.max(), float((e <= 1e-3).mean()), e.mean()))
    print("   mean developed oracle %.5f gpu %.5f" % (od.mean(), gd.mean()))


if __name__ == "__main__":
    size = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    spp = int(sys.argv[2]) if len(sys.argv) > 2 else 16
    check("C1 spheres", scenes.c1_spheres(width=size, height=size, pixel_samples=spp))
    check("C2 cornell", scenes.c2_cornell(width=size, height=size, pixel_samples=spp))
