#!/usr/bin/env python3
"""Traversal micro-benchmark: World::intersect (pyr_scene_intersect) on large ray batches, coherent and incoherent."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from pyrite_amd import scenes

def rays_random(n, lo, hi, seed=1):
    rng = np.random.RandomState(seed)
    o = rng.uniform(lo, hi, size=(n, 3)); d = rng.normal(size=(n, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
    return np.concatenate([o, d], axis=1).astype(np.float32)

def rays_camera(n, scale, seed=2):
    side = int(np.sqrt(n)); ys, xs = np.mgrid[0:side, 0:side]
    u = (xs.reshape(-1) + 0.5) / side * 2 - 1; v = (ys.reshape(-1) + 0.5) / side * 2 - 1
    vp = 1 / np.tan(np.radians(37.7 / 2))
    d = np.stack([u / vp, np.ones_like(u), -v / vp], axis=1); d /= np.linalg.norm(d, axis=1, keepdims=True)
    o = np.tile(np.array([[-2.78 * scale, -8 * scale, 2.73 * scale]]), (len(d), 1))
    return np.concatenate([o, d], axis=1).astype(np.float32)

def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
    run(n)

def run(n):
    for name, proj, scale in (("C2", scenes.c2_cornell(64, 64, 1), 1.0), ("C3", scenes.c3_mesh_in_box(64, 36, 1), 10.0)):
        world, cam, r, film = scenes.build(proj, seed=1)
        info = world.bvh_info()
        for kind, rays in (("camera", rays_camera(n, scale)), ("random", rays_random(n, [-5.5 * scale, 0.1 * scale, 0.1 * scale], [-0.1 * scale, 5.5 * scale, 5.4 * scale]))):
            hits, ms, c = world.intersect(rays, want_counters=True)
            hits, ms, _ = world.intersect(rays)
            nodes = c["box_tests"] / 2
            print("%s %-6s %d rays: %.3f ms  %.0f Mrays/s  node visits/ray %.1f tri/ray %.1f  -> %.1f G node visits/s, algorithmic %.0f GB/s" % (
                name, kind, len(rays), ms, len(rays) / ms / 1e3, nodes / len(rays), c["triangle_tests"] / len(rays), nodes / ms / 1e6,
                (32 * c["box_tests"] + 36 * c["triangle_tests"]) / ms / 1e6))


if __name__ == "__main__":
    main()
