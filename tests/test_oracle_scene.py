"""Scene-level checks of the CPU oracle: the restated reference BVH, closest-hit against brute force, analytic
furnace values, determinism, tile sharding and film windows, and the committed golden films."""
import ctypes as C
import json
import os

import numpy as np
import pytest

import oracle
from oracle import F3, F6
from pyrite_amd import scenes
from pyrite_amd.project import camera, material, renderer, shape, transform, vector

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
L = oracle.lib()


def rays_in_box(n, seed):
    rng = np.random.RandomState(seed)
    o = rng.uniform([-5.5, 0.1, 0.1], [-0.1, 5.5, 5.4], size=(n, 3))
    d = rng.normal(size=(n, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    return np.concatenate([o, d], axis=1).astype(np.float32)


@pytest.fixture(scope="module")
def cornell():
    world, cam, r, film = scenes.build(scenes.c2_cornell(32, 32, 4), seed=1)
    return world, cam, r, oracle.OracleScene(world)


def test_reference_bvh_is_preorder_with_one_item_per_leaf(cornell):
    world, _, _, sc = cornell
    nodes = sc.bvh_nodes()
    n_items = 36
    assert len(nodes) == 2 * n_items - 1  # binary tree, one item per leaf (bvh.rs:58-66)
    items = [item for _, size, item in nodes if size == 0]
    assert sorted(i & 0x3FFFFFFF for i in items) == list(range(n_items))

    def check(i):  # returns the number of nodes in the subtree rooted at i
        aabb, size, _ = nodes[i]
        if size == 0:
            return 1
        first = i + 1
        n_first = check(first)
        second = first + n_first
        n_second = check(second)
        assert size == n_first + n_second  # subtree_size counts descendants (bvh.rs:48)
        for child in (first, second):  # children are inside the parent box
            assert np.all(nodes[child][0][:3] >= aabb[:3] - 1e-6) and np.all(nodes[child][0][3:] <= aabb[3:] + 1e-6)
        return 1 + size

    assert check(0) == len(nodes)


def test_closest_hit_equals_brute_force(cornell):
    world, _, _, sc = cornell
    rays = rays_in_box(300, 5)
    hits, counters = sc.intersect(rays)
    pos = np.concatenate([np.asarray(p).reshape(-1, 9) for p in world.flat.tri_positions])
    dist, u, v = C.c_float(), C.c_float(), C.c_float()
    for r, h in zip(rays, hits):
        best = (np.inf, 0xFFFFFFFF)
        for i, p in enumerate(pos):
            if L.oracle_triangle_intersect(F3(*p[0:3]), F3(*p[3:6]), F3(*p[6:9]), F6(*r), C.byref(dist), C.byref(u), C.byref(v)):
                if 1e-4 < dist.value < best[0]:  # world.rs:290
                    best = (dist.value, (1 << 30) | i)
        assert h["distance"] == np.float32(best[0])
        if best[1] != 0xFFFFFFFF:
            p = pos[h["shape"] & 0x3FFFFFFF]  # ties (shared edges) may name another triangle at the same distance
            assert L.oracle_triangle_intersect(F3(*p[0:3]), F3(*p[3:6]), F3(*p[6:9]), F6(*r), C.byref(dist), C.byref(u), C.byref(v))
            assert dist.value == h["distance"]
    assert counters["box_tests"] > 0 and counters["triangle_tests"] < 300 * 36  # the tree prunes


def test_convex_diffuse_sphere_under_a_uniform_sky_reflects_albedo():
    # One convex Lambertian sphere of albedo 0.5 under sky radiance 1: every path is sphere -> sky, so the expected value
    # is exactly albedo * E[2 cos] * sky = 0.5 on the sphere and 1 off it (diffuse.rs:27-29 with uniform-hemisphere sampling).
    project = {
        "image": {"width": 32, "height": 32},
        "renderer": renderer.simple(pixel_samples=64, light_samples=0, spectrum_samples=4, tile_size=16),
        "camera": camera.perspective(fov=20, transform=transform.look_at(**{"from": vector(0, 0, 10), "to": vector(0, 0, 0)})),
        "world": {"sky": 1.0, "objects": [shape.sphere(position=vector(0, 0, 0), radius=1.0, material={"surface": material.diffuse(color=0.5)})]},
    }
    world, cam, r, film = scenes.build(project, seed=2)
    oracle.OracleScene(world).render(r, cam, film, threads=4)
    dev = film.grains[..., 0].sum(-1) / np.maximum(film.grains[..., 1].sum(-1), 1)
    assert dev[14:18, 14:18].mean() == pytest.approx(0.5, abs=0.02)  # centre of the sphere
    assert dev[0:3, 0:3].mean() == pytest.approx(1.0, abs=1e-6)  # sky only
    assert film.total_weight() == 32 * 32 * 64 * 4  # every sample exposes all S wavelengths, weight 1 each


def test_same_seed_same_film_threads_do_not_matter(cornell):
    world, cam, r, sc = cornell
    a, b, c = r.new_film(32, 32), r.new_film(32, 32), r.new_film(32, 32)
    sc.render(r, cam, a, threads=1)
    sc.render(r, cam, b, threads=8)
    assert np.array_equal(a.grains[..., 1], b.grains[..., 1])
    assert np.allclose(a.grains, b.grains, rtol=1e-6, atol=0)
    r2 = type(r)(**{**r.__dict__, "seed": 99})
    sc.render(r2, cam, c, threads=8)
    assert not np.array_equal(a.grains, c.grains)


def test_tile_ranges_add_up_to_the_whole_image():
    world, cam, r, film = scenes.build(scenes.c1_spheres(48, 40, 2), seed=4)
    r.tile_size = 16  # 3 x 3 tiles, the last row 8 pixels high
    sc = oracle.OracleScene(world)
    whole = r.new_film(48, 40)
    sc.render(r, cam, whole, threads=2)
    parts = r.new_film(48, 40)
    for lo, hi in ((0, 4), (4, 5), (5, 9)):
        sc.render(r, cam, parts, threads=2, tile_range=(lo, hi))
    assert np.array_equal(whole.grains[..., 1], parts.grains[..., 1])
    assert np.allclose(whole.grains, parts.grains, rtol=1e-6)


def test_film_window_keeps_only_its_rows():
    world, cam, r, film = scenes.build(scenes.c1_spheres(32, 32, 2), seed=4)
    r.tile_size = 8
    sc = oracle.OracleScene(world)
    whole = r.new_film(32, 32)
    window = np.zeros((10, 32, whole.bins, 2), dtype=np.float32)  # rows 7..16: tile row 1 (8..15) plus a one-row halo on both sides
    sc.render(r, cam, whole, threads=1, tile_range=(4, 8), film_rows=(7, 10), window=window)
    only_row1 = r.new_film(32, 32)
    sc.render(r, cam, only_row1, threads=1, tile_range=(4, 8))
    assert np.array_equal(window, only_row1.grains[7:17])
    assert only_row1.grains[:7].sum() == 0 and only_row1.grains[17:].sum() == 0  # leaks never travel further than one row


@pytest.mark.parametrize("name", ["c1_spheres", "c2_cornell", "spheres_example", "diamonds_example", "lamps_example", "textures_example"])
def test_oracle_reproduces_the_committed_golden_films(name):
    import importlib.util

    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(GOLDEN, "make_golden.py"))
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)
    data = np.load(os.path.join(GOLDEN, name + ".npz"))
    meta = json.loads(str(data["meta"]))
    world, cam, r, film = scenes.build(mg.build_case(name), seed=meta["seed"])
    sc = oracle.OracleScene(world)
    counters = sc.render(r, cam, film, threads=1)
    assert counters == meta["counters"]
    assert np.array_equal(film.grains[..., 1], data["grains"][..., 1])
    assert np.allclose(film.grains, data["grains"], rtol=1e-6, atol=1e-12)
    hits, _ = sc.intersect(data["rays"])
    assert np.array_equal(hits["shape"], data["hit_shape"])
    assert np.array_equal(hits["distance"], data["hit_distance"])
