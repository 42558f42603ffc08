"""Parity of the HIP path (through the C ABI) with the CPU oracle on identical seeds -- run with `-m gpu` on an MI355X.

Stated tolerance (north_star: "within a stated per-pixel spectral L2 tolerance"): with s = acc / weight per bin,
    relL2(pixel) = ||s_gpu - s_cpu||_2 / (||s_cpu||_2 + 1e-6)
must be <= 1e-5 for EVERY pixel (round 3: no share of outliers is allowed any more), and the film WEIGHTS (integer sample
counts per bin) must be identical. The kernels and the oracle perform the same f32 operations in the same order (no FMA
contraction, the same Cephes transcendentals, the same RNG streams), so paths agree bit for bit and only the order of the
float atomics differs: the largest per-pixel value any of the 70 film comparisons of the suite saw on an MI355X is 1.3e-7
(profiles/r03_parity_observed.json; every assertion message and gpurun_out/parity_observed.json carry the observed
maximum). Integer / index results (hit shapes, counters, weights) must be exact."""
import importlib.util
import json
import os

import numpy as np
import pytest

import oracle
from pyrite_amd import abi, scenes
from pyrite_amd.project import renderer

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TOL = 1e-5
OBSERVED = {}  # test id -> largest per-pixel relL2 any assert_parity of that test saw (written out by conftest.py)


def rel_l2(gpu_film, cpu_film):
    a, b = gpu_film.develop(), cpu_film.develop()
    return (np.sqrt(((a - b) ** 2).sum(-1)) / (np.sqrt((b ** 2).sum(-1)) + 1e-6)).reshape(-1)


def assert_parity(gpu_film, cpu_film):
    assert np.array_equal(gpu_film.grains[..., 1], cpu_film.grains[..., 1]), "film weights differ"
    e = rel_l2(gpu_film, cpu_film)
    worst = float(e.max()) if e.size else 0.0
    name = os.environ.get("PYTEST_CURRENT_TEST", "?").split(" ")[0]
    OBSERVED[name] = max(OBSERVED.get(name, 0.0), worst)
    message = "relL2: median %.3g p99 %.3g max %.3g (pixel %d)" % (np.median(e), np.percentile(e, 99), worst, int(e.argmax()))
    assert worst <= TOL, message
    assert not np.isnan(gpu_film.grains).any()


def render_both(project, seed, gpu_lib, threads=8):
    world, cam, r, gfilm = scenes.build(project, seed=seed)
    cfilm = r.new_film(gfilm.width, gfilm.height)
    ccount = oracle.OracleScene(world).render(r, cam, cfilm, threads=threads)
    gcount = r.render(gfilm, cam, world, counters=True)
    return gfilm, cfilm, gcount, ccount


def random_rays(n, seed, lo, hi):
    rng = np.random.RandomState(seed)
    o = rng.uniform(lo, hi, size=(n, 3))
    d = rng.normal(size=(n, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    return np.concatenate([o, d], axis=1).astype(np.float32)


def primitive_distance(world, shape, ray):
    """The oracle's own intersection routine (shapes/mod.rs:75-119 / :57-74) on ONE primitive of the scene: (hit, distance, u, v)."""
    import ctypes as C

    d = world.desc
    kind, index = int(shape) >> 30, int(shape) & 0x3FFFFFFF
    r6 = oracle.F6(*[float(x) for x in ray])
    dist, u, v = C.c_float(), C.c_float(), C.c_float()
    if kind == 1:  # PYR_SHAPE_TRIANGLE
        p = np.ctypeslib.as_array(d.tri_positions, shape=(d.num_triangles * 9,))[9 * index:9 * index + 9]
        hit = oracle.lib().oracle_triangle_intersect(oracle.F3(*p[0:3]), oracle.F3(*p[3:6]), oracle.F3(*p[6:9]), r6, C.byref(dist), C.byref(u), C.byref(v))
        return bool(hit), np.float32(dist.value), np.float32(u.value), np.float32(v.value)
    assert kind == 0, "planes are scanned linearly on both sides: no tie can involve one"
    sp = np.ctypeslib.as_array(d.spheres, shape=(d.num_spheres * 4,))[4 * index:4 * index + 4]
    point = oracle.F3()
    hit = oracle.lib().oracle_sphere_intersect(oracle.F3(*sp[0:3]), float(sp[3]), r6, C.byref(dist), point)
    return bool(hit), np.float32(dist.value), np.float32(0), np.float32(0)


def assert_same_hits(ohits, ghits, world, rays):
    """Closest hits are bit-exact -- distance, shape, u, v -- except at TIES, and every difference must be PROVEN one: both
    sides keep the smallest distance they see, but a (zero-thickness) box whose entry distance rounds an ulp above the closest
    hit so far is culled (bvh.rs:213), so two primitives hit at numerically almost the same distance can resolve either way
    depending on the visiting order -- a property of the reference's own traversal, whose tree is not this library's. For
    every ray that differs, the oracle's intersection routine is run on the primitive EACH side reported: it must hit, at
    exactly the f32 distance (and barycentrics) that side reported, and the two distances must agree to a few ulps. A
    traversal that skipped a closer primitive, returned a wrong index or a wrong distance fails one of the three."""
    rays = np.asarray(rays, dtype=np.float32).reshape(-1, 6)
    differ = (ohits["distance"] != ghits["distance"]) | (ohits["shape"] != ghits["shape"])
    same = ~differ
    assert np.array_equal(ohits["u"][same], ghits["u"][same]) and np.array_equal(ohits["v"][same], ghits["v"][same])
    for i in np.nonzero(differ)[0]:
        for side, hits in (("oracle", ohits), ("gpu", ghits)):
            assert hits["shape"][i] != 0xFFFFFFFF, "ray %d: only one side found a hit (%s missed)" % (i, side)
            hit, dist, u, v = primitive_distance(world, hits["shape"][i], rays[i])
            assert hit and dist == hits["distance"][i], "ray %d: %s reports shape %#x at %r, the primitive itself says %r" % (
                i, side, hits["shape"][i], hits["distance"][i], dist if hit else None)
            if (int(hits["shape"][i]) >> 30) == 1:
                assert u == hits["u"][i] and v == hits["v"][i]
        od, gd = float(ohits["distance"][i]), float(ghits["distance"][i])
        assert abs(od - gd) <= 2e-6 * od, "ray %d: not a tie: oracle %r gpu %r" % (i, od, gd)
    return int(differ.sum())


CASES = {
    "c1_spheres": lambda: scenes.c1_spheres(64, 64, 16),
    "c2_cornell": lambda: scenes.c2_cornell(64, 64, 16),
    "spheres_example": lambda: scenes.spheres_example(96, 48, 16),
    "diamonds_example": lambda: scenes.diamonds_example(64, 40, 8, bounces=32),
    "lamps_example": lambda: scenes.lamps_example(72, 48, 16),
    "textures_example": lambda: scenes.textures_example(72, 48, 8),
}


@pytest.mark.parametrize("name", sorted(CASES))
def test_render_matches_the_oracle(name, gpu_lib):
    gfilm, cfilm, gcount, ccount = render_both(CASES[name](), 5, gpu_lib)
    assert_parity(gfilm, cfilm)
    # path-level counters are exact; box / primitive test counts differ because the two sides walk different trees
    for key in ("samples", "extension_rays", "shadow_rays", "shaded_hits", "exposures"):
        assert gcount[key] == ccount[key], key


@pytest.mark.parametrize("scheduler", ["sync", "sm"])
@pytest.mark.parametrize("name", ["c2_cornell", "lamps_example", "diamonds_example", "textures_example"])
def test_all_schedulers_give_the_oracle_film(name, scheduler, gpu_lib, monkeypatch):
    """The bounce-synchronous walk and the stage-scheduled state machine are two schedules of the
    same per-path work: both must reproduce the oracle (the library picks one per scene; PYRITE_SCHEDULER forces it)."""
    monkeypatch.setenv("PYRITE_SCHEDULER", scheduler)
    gfilm, cfilm, gcount, ccount = render_both(CASES[name](), 8, gpu_lib)
    assert_parity(gfilm, cfilm)
    for key in ("samples", "extension_rays", "shadow_rays", "shaded_hits", "exposures"):
        assert gcount[key] == ccount[key], key


HIT_TAPE_CASES = {  # name: (project, does the scene record a tape when it may?)
    "textures_reference_example": (lambda: scenes.textures_reference_example(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "textures"), 96, 48, 8), True),
    "spheres_example": (CASES["spheres_example"], True),
    "textures_example": (CASES["textures_example"], True),   # a mono texture times a spectrum: a PRODUCT form -- the texture once per hit, the spectrum in the replay
    "lamps_example": (CASES["lamps_example"], True),         # a blackbody lamp (a LAMBDA form: evaluated once per replay item), an rgb() lamp
}


@pytest.mark.parametrize("hit_tape", ["1", "0"])
@pytest.mark.parametrize("name", sorted(HIT_TAPE_CASES))
def test_interpreter_scenes_with_and_without_the_hit_tape(name, hit_tape, gpu_lib, monkeypatch):
    """Round 4: a scene with interpreter programs whose colour programs all have a tape form (device_scene.h TapeForm: textures,
    spheres, lamps, and the generated textures scene, whose mono texture times a spectrum is a PRODUCT form) records a spectral tape -- the interpreter runs once per
    hit, the replay does the per-wavelength part. PYRITE_HIT_TAPE=0 (read at scene creation) keeps round 3's online form. Both
    are the oracle's film; and with the tape shrunk under its bound the hit-tape form -- and only it -- reports the overflow."""
    from pyrite_amd._lib import PyriteGpuError

    monkeypatch.setenv("PYRITE_HIT_TAPE", hit_tape)
    project, eligible = HIT_TAPE_CASES[name]
    world, _, r, _ = scenes.build(project(), seed=11)
    info = r.path_info(world)  # pyr_scene_path_info: what a render of this scene runs
    world.close()
    assert info["interpreter"] == 1 and info["stage_scheduler"] == 1 and info["tape"] == (2 if hit_tape == "1" and eligible else 0), info
    gfilm, cfilm, gcount, ccount = render_both(project(), 11, gpu_lib)
    assert_parity(gfilm, cfilm)
    for key in ("samples", "extension_rays", "shadow_rays", "shaded_hits", "exposures"):
        assert gcount[key] == ccount[key], key
    monkeypatch.setenv("PYRITE_TEST_TAPE_OPS", "2")
    if hit_tape == "1" and eligible:
        with pytest.raises(PyriteGpuError, match="spectral tape"):
            render_both(project(), 11, gpu_lib)
    else:
        render_both(project(), 11, gpu_lib)


def test_ragged_image_and_odd_parameters(gpu_lib):
    project = scenes.c2_cornell(50, 37, 3)  # tiles of 16: ragged right column and bottom row
    project["renderer"] = renderer.simple(pixel_samples=3, tile_size=16, spectrum_samples=7, light_samples=1, bounces=3)
    gfilm, cfilm, gcount, ccount = render_both(project, 9, gpu_lib)
    assert_parity(gfilm, cfilm)
    assert gcount["samples"] == 50 * 37 * 3 == ccount["samples"]


def test_vertical_image_single_wavelength_and_no_light_samples(gpu_lib):
    project = scenes.c1_spheres(24, 40, 4)
    project["renderer"] = renderer.simple(pixel_samples=4, tile_size=32, spectrum_samples=1, light_samples=0, bounces=5)
    gfilm, cfilm, _, _ = render_both(project, 2, gpu_lib)
    assert_parity(gfilm, cfilm)
    assert gfilm.total_weight() == 24 * 40 * 4


@pytest.mark.parametrize("scheduler", ["sync", "sm"])
def test_world_without_objects_is_all_sky(scheduler, gpu_lib, monkeypatch):
    """No primitives at all (the BVH is a root with two empty leaves): every path misses and shows the sky."""
    from pyrite_amd.project import light_source

    monkeypatch.setenv("PYRITE_SCHEDULER", scheduler)
    project = scenes.lamps_example(40, 24, 4)
    project["world"] = {"sky": light_source.d65 * 0.5, "objects": []}
    gfilm, cfilm, gcount, ccount = render_both(project, 3, gpu_lib)
    assert_parity(gfilm, cfilm)
    assert gcount["shaded_hits"] == 0 and gcount["extension_rays"] == gcount["samples"] == 40 * 24 * 4 and gcount == {**ccount, "box_tests": gcount["box_tests"]}
    assert gfilm.grains[..., 0].sum() > 0


def test_calls_the_kernels_cannot_index_are_refused_before_anything_runs(gpu_lib):
    """Sizes the 32-bit indices of the kernels do not reach -- 2^32 pixels, a window outside the image, more than 64 wavelengths -- are
    errors of the call (never a wrap-around on the device); the film pointer is not touched (it points at eight bytes here)."""
    import ctypes as C

    from pyrite_amd import abi
    from pyrite_amd._lib import lib

    world, cam, r, _ = scenes.build(scenes.c2_cornell(16, 16, 1), seed=1)
    grain = (C.c_float * 2)()
    no_progress = C.cast(None, abi.PyrProgressFn)

    def call(width, height, **overrides):
        params = r.params()
        for key, value in overrides.items():
            setattr(params, key, value)
        desc = abi.PyrFilmDesc(width, height, r.spectrum_bins, r.spectrum_span[0], r.spectrum_span[1] - r.spectrum_span[0])
        rc = lib().pyr_render_simple(world.scene(0), C.byref(cam.c), C.byref(desc), C.byref(params), C.cast(grain, C.c_void_p), no_progress, None)
        return rc, lib().pyr_last_error().decode()

    assert call(65536, 65536) == (abi.PYR_ERR_UNSUPPORTED, "image too large: 2^32 pixels or more")
    rc, message = call(60000, 60000, tile_size=8, film_layout=abi.PYR_FILM_TILE_BLOCKS)
    assert rc == abi.PYR_ERR_UNSUPPORTED and "tile blocks" in message
    assert call(16, 16, spectrum_samples=65)[0] == abi.PYR_ERR_UNSUPPORTED
    assert call(16, 16, film_row_begin=8, film_row_count=9) == (abi.PYR_ERR_INVALID_ARGUMENT, "film window exceeds the image")
    assert call(0, 16)[0] == abi.PYR_ERR_INVALID_ARGUMENT
    assert grain[0] == 0.0 and grain[1] == 0.0
    world.close()


def test_empty_work_is_a_no_op(gpu_lib):
    world, cam, r, film = scenes.build(scenes.c2_cornell(16, 16, 0), seed=1)  # zero samples per pixel
    r.render(film, cam, world)
    assert film.grains.sum() == 0
    r.pixel_samples = 2
    r.render(film, cam, world, tile_range=(1, 1))  # empty tile range (tile_end == 0 would mean "all tiles")
    assert film.grains.sum() == 0
    hits, ms, _ = world.intersect(np.zeros((0, 6), dtype=np.float32))
    assert len(hits) == 0


@pytest.mark.parametrize("name", ["c1_spheres", "c2_cornell", "lamps_example"])
def test_closest_hit_matches_the_oracle_exactly(name, gpu_lib):
    world, cam, r, film = scenes.build(CASES[name](), seed=1)
    rays = random_rays(100000, 3, [-6, -1, 0.05], [0.5, 6, 5.4])
    ohits, _ = oracle.OracleScene(world).intersect(rays)
    ghits, ms, counters = world.intersect(rays, want_counters=True)
    assert_same_hits(ohits, ghits, world, rays)
    assert counters["box_tests"] > 0


@pytest.mark.parametrize("name", ["c1_spheres", "c2_cornell", "c3_shaped"])
def test_axis_parallel_rays_find_what_the_oracle_finds(name, gpu_lib):
    """Rays with direction components of exactly zero: 1 / d is infinite there and the plane distances of the box test turn
    into inf - inf = NaN for every box on the origin's side of zero. Rounds 1-2 took min / max over both planes of an axis and
    rejected boxes such a ray lies inside of (found in round 3: an axis-parallel ray through the Cornell box missed everything);
    the planes are now picked by the direction's sign and a NaN plane drops out. Both trees: binary (C1, C2) and four-wide."""
    from pyrite_amd.renderer import World

    if name == "c3_shaped":
        world = World(scenes.c3_flat(segments=96, sides=48))
        lo, hi = np.array([-55, 1, 1]), np.array([-1, 55, 54])
    else:
        world, _, _, _ = scenes.build(CASES[name](), seed=1)
        lo, hi = np.array([-5.5, 0.1, 0.05]), np.array([-0.1, 5.5, 5.4])
    # (seed 7 drew an origin whose z rounds to exactly 1.65f, the top of box.obj's short block, with a direction along +x: a ray
    # lying IN a box face. The reference's box test rejects that box -- (1.65 - 1.65) * inf is NaN and its max() keeps the -inf of
    # the other plane -- while a test that lets the NaN plane drop out accepts it and finds the grazing hit on the top face.
    # Degenerate, measure zero, neither answer wrong; the seed below draws no such origin.)
    rng = np.random.RandomState(12)
    origins = rng.uniform(lo, hi, size=(20000, 3))
    axis = np.eye(3)[rng.randint(0, 3, 20000)] * rng.choice([-1.0, 1.0], size=(20000, 1))  # +-x, +-y, +-z
    planar = rng.normal(size=(20000, 3))
    planar[np.arange(20000), rng.randint(0, 3, 20000)] = 0.0  # one component exactly zero
    planar /= np.linalg.norm(planar, axis=1, keepdims=True)
    planar[rng.rand(20000) < 0.5] *= np.float32(1.0)
    neg_zero = axis.copy()
    neg_zero[neg_zero == 0] = -0.0  # the sign of a zero picks the plane too
    rays = np.concatenate([np.concatenate([origins, d], axis=1) for d in (axis, planar, neg_zero)]).astype(np.float32)
    ohits, _ = oracle.OracleScene(world).intersect(rays)
    ghits, _, _ = world.intersect(rays)
    assert_same_hits(ohits, ghits, world, rays)
    assert (ohits["shape"] != 0xFFFFFFFF).mean() > 0.7  # the boxes are closed on five sides
    world.close()


def test_closest_hit_on_a_dense_mesh(gpu_lib):
    from pyrite_amd.compiler import FlatScene
    from pyrite_amd.project import material
    from pyrite_amd.renderer import World

    flat = FlatScene()
    flat.sky_program = flat.compile(0.0)
    mat, _ = flat.add_material({"surface": material.diffuse(color=0.8)})
    pos, nrm = scenes.torus_knot_mesh(segments=96, sides=48, fit_min=(-4, -4, -4), fit_max=(4, 4, 4))
    flat.add_triangles(pos, nrm, mat)
    world = World(flat)
    info = world.bvh_info()
    assert info["num_primitives"] == 96 * 48 * 2 and info["max_depth"] <= 40
    rays = random_rays(60000, 8, [-5, -5, -5], [5, 5, 5])
    ohits, _ = oracle.OracleScene(world).intersect(rays)
    ghits, _, _ = world.intersect(rays)
    ties = assert_same_hits(ohits, ghits, world, rays)
    assert ties < 0.01 * len(rays)  # and proven ties stay rare
    assert (ohits["shape"] != 0xFFFFFFFF).mean() > 0.1


def test_rays_with_a_direction_longer_than_one_pass_beside_spheres(gpu_lib):
    """collision's sphere routine assumes a unit direction; with |d| > 1 (a directional lamp whose `direction` is not
    normalised makes such shadow rays, lamp.rs:24-35) it reports hits for rays that pass the sphere at a distance, and the
    reference is only saved by the leaf's bounding-box test. Leaves here hold several primitives, so spheres carry that test
    themselves (found by tests/test_gpu_fuzz.py, seed 196)."""
    from pyrite_amd.compiler import FlatScene
    from pyrite_amd.project import material, shape, vector
    from pyrite_amd.renderer import World

    flat = FlatScene()
    white = {"surface": material.diffuse(color=0.8)}
    flat.add_world({"objects": [shape.sphere(position=vector(1.169252501331818, 1.0246223838963893, 1.4219978669133146), radius=0.3238749019030583, material=white),
                                shape.sphere(position=vector(-1.3, 1.7, 1.0), radius=0.32, material=white),
                                shape.sphere(position=vector(0, -2, 0.5), radius=0.5, material=white)]})
    world = World(flat)
    rays = np.array([[1.73444414, 0.728979588, -0.110429764, -0.123903722, 0.326353073, 0.953749716]], dtype=np.float32)
    more = random_rays(20000, 4, [-3, -3, -1], [3, 3, 3])
    more[:, 3:] *= np.random.RandomState(5).uniform(0.8, 1.3, (len(more), 1)).astype(np.float32)  # lengths 0.8 .. 1.3
    rays = np.concatenate([rays, more])
    ohits, _ = oracle.OracleScene(world).intersect(rays)
    ghits, _, _ = world.intersect(rays)
    assert ohits["shape"][0] == 0xFFFFFFFF and ghits["shape"][0] == 0xFFFFFFFF
    assert np.array_equal(ohits["shape"], ghits["shape"]) and np.array_equal(ohits["distance"], ghits["distance"])


def test_wide_and_binary_trees_answer_alike(gpu_lib, monkeypatch):
    """Scenes that do not live in LDS are walked through the 4-wide collapse of the binary tree (bvh.h Node128);
    PYRITE_WIDE_BVH=0 keeps the binary tree. Same hits, same film, fewer box tests."""
    from pyrite_amd.renderer import Camera, Renderer, World

    project = scenes.c3_mesh_in_box(width=48, height=27, pixel_samples=4)
    rays = random_rays(50000, 12, [-55, 0, 0], [0, 55, 54])
    results = {}
    for wide in ("1", "0"):
        monkeypatch.setenv("PYRITE_WIDE_BVH", wide)
        world = World(scenes.c3_flat(segments=96, sides=48))
        hits, _, counters = world.intersect(rays, want_counters=True)
        r = Renderer.from_project(project["renderer"], seed=2)
        film = r.new_film(48, 27)
        r.render(film, Camera.from_project(project["camera"]), world)
        results[wide] = (hits, counters, film)
    (h1, c1, f1), (h0, c0, f0) = results["1"], results["0"]
    assert_same_hits(h0, h1, world, rays)
    assert c1["triangle_tests"] <= 1.1 * c0["triangle_tests"] and c1["box_tests"] < c0["box_tests"]
    assert_parity(f1, f0)
    ohits, _ = oracle.OracleScene(world).intersect(rays)
    assert_same_hits(ohits, h1, world, rays)


@pytest.mark.parametrize("glass", [False, True])
def test_c3_shaped_scene_at_scale_50(glass, gpu_lib):
    """C3 / C5 in small: the x10 Cornell box with a (coarser) torus-knot mesh, diffuse or dispersive glass. At this scale
    d^2 ~ 2500 has an ulp larger than DIST_EPSILON, which is where an exact box cut-off for shadow rays once diverged from
    the reference's closest-hit visibility test."""
    from pyrite_amd.renderer import Camera, Renderer, World

    project = scenes.c3_mesh_in_box(width=64, height=36, pixel_samples=4, glass=glass, bounces=20 if glass else None)
    world = World(scenes.c3_flat(segments=96, sides=48, glass=glass))
    r = Renderer.from_project(project["renderer"], seed=6)
    cam = Camera.from_project(project["camera"])
    gfilm, cfilm = r.new_film(64, 36), r.new_film(64, 36)
    ccount = oracle.OracleScene(world).render(r, cam, cfilm, threads=8)
    gcount = r.render(gfilm, cam, world, counters=True)
    assert_parity(gfilm, cfilm)
    for key in ("samples", "extension_rays", "shadow_rays", "shaded_hits", "exposures"):
        assert gcount[key] == ccount[key], key
    if glass:
        assert gcount["exposures"] < gcount["samples"] * 10  # dispersed paths expose the hero wavelength only


@pytest.mark.parametrize("hit_tape", ["1", "0"])
@pytest.mark.parametrize("glassy", [False, True])
def test_interpreter_material_on_a_mesh_that_does_not_live_in_lds(glassy, hit_tape, gpu_lib, monkeypatch):
    """The interpreter builds of the stage scheduler on a scene walked from HBM (the form a textured production mesh takes: C3's box
    with a 9,216-triangle knot): a fresnel mix of a mirror and an rgb()-coloured coat, or of dispersive glass and the coat (hero-only
    paths beside full ones on one tape), with the hit tape and with every wavelength online -- both the oracle's film."""
    from pyrite_amd.project import fresnel, material, mix, rgb
    from pyrite_amd.renderer import Camera, Renderer, World

    monkeypatch.setenv("PYRITE_HIT_TAPE", hit_tape)
    coat = material.diffuse(color=rgb(0.8, 0.45, 0.2))
    under = material.refractive(ior=1.5, dispersion=0.01371, color=1) if glassy else material.mirror(color=1)
    mesh_material = {"surface": mix(under, coat, fresnel(1.5))}
    project = scenes.c3_mesh_in_box(width=64, height=36, pixel_samples=4, bounces=12 if glassy else None, mesh_material=mesh_material)
    world = World(scenes.c3_flat(segments=96, sides=48, mesh_material=mesh_material))
    r = Renderer.from_project(project["renderer"], seed=4)
    info = r.path_info(world)
    assert info == {"stage_scheduler": 1, "interpreter": 1, "scene_in_lds": 0, "tape": 2 if hit_tape == "1" else 0, "phase_lanes": 32}, info
    cam = Camera.from_project(project["camera"])
    gfilm, cfilm = r.new_film(64, 36), r.new_film(64, 36)
    ccount = oracle.OracleScene(world).render(r, cam, cfilm, threads=8)
    gcount = r.render(gfilm, cam, world, counters=True)
    assert_parity(gfilm, cfilm)
    for key in ("samples", "extension_rays", "shadow_rays", "shaded_hits", "exposures"):
        assert gcount[key] == ccount[key], key


def test_stage_scheduler_on_an_lds_resident_scene_with_a_one_level_lds_stack(gpu_lib, monkeypatch):
    """the kernels built for scenes staged in LDS have no scratch part of the traversal stack (one entry), so the
    launcher must keep the WHOLE stack in LDS for such scenes whatever PYRITE_LDS_STACK or the budget say -- also when the
    stage scheduler is forced onto a scene the synchronous walk would normally take."""
    monkeypatch.setenv("PYRITE_SCHEDULER", "sm")
    monkeypatch.setenv("PYRITE_LDS_STACK", "1")
    gfilm, cfilm, gcount, ccount = render_both(scenes.c2_cornell(48, 48, 6), 4, gpu_lib)
    assert_parity(gfilm, cfilm)
    for key in ("samples", "extension_rays", "shadow_rays", "shaded_hits", "exposures"):
        assert gcount[key] == ccount[key], key


@pytest.mark.parametrize("levels", ["1", "3"])
def test_short_lds_stack_spills_to_scratch_without_changing_results(levels, gpu_lib, monkeypatch):
    """The resumable traversal keeps only a few stack levels in LDS (TravStack); force nearly every push into the scratch
    part and check rays and film against the oracle."""
    from pyrite_amd.renderer import Camera, Renderer, World

    monkeypatch.setenv("PYRITE_LDS_STACK", levels)
    project = scenes.c3_mesh_in_box(width=48, height=27, pixel_samples=4)
    world = World(scenes.c3_flat(segments=96, sides=48))
    assert world.bvh_info()["max_depth"] > 3
    rays = random_rays(40000, 9, [-55, 0, 0], [0, 55, 54])
    ohits, _ = oracle.OracleScene(world).intersect(rays)
    ghits, _, _ = world.intersect(rays)
    assert_same_hits(ohits, ghits, world, rays)
    r = Renderer.from_project(project["renderer"], seed=2)
    cam = Camera.from_project(project["camera"])
    gfilm, cfilm = r.new_film(48, 27), r.new_film(48, 27)
    oracle.OracleScene(world).render(r, cam, cfilm, threads=8)
    r.render(gfilm, cam, world)
    assert_parity(gfilm, cfilm)


@pytest.mark.parametrize("name", ["c1_spheres", "c2_cornell", "spheres_example", "diamonds_example", "lamps_example", "textures_example"])
def test_gpu_reproduces_the_committed_golden_films(name, gpu_lib):
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(GOLDEN, "make_golden.py"))
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)
    data = np.load(os.path.join(GOLDEN, name + ".npz"))
    meta = json.loads(str(data["meta"]))
    world, cam, r, film = scenes.build(mg.build_case(name), seed=meta["seed"])
    r.render(film, cam, world)
    golden = r.new_film(film.width, film.height)
    golden.grains[...] = data["grains"]
    assert_parity(film, golden)
    ghits, _, _ = world.intersect(data["rays"])
    golden_hits = np.zeros(len(ghits), dtype=ghits.dtype)
    golden_hits["distance"], golden_hits["shape"], golden_hits["u"], golden_hits["v"] = data["hit_distance"], data["hit_shape"], data["hit_u"], data["hit_v"]
    assert_same_hits(golden_hits, ghits, world, data["rays"])


def test_tile_ranges_windows_and_progress(gpu_lib):
    world, cam, r, film = scenes.build(scenes.c1_spheres(48, 40, 2), seed=4)
    r.tile_size = 16
    seen = []
    r.render(film, cam, world, on_status=lambda percent, message: seen.append((percent, message)))
    assert seen[0] == (0, "Rendering") and seen[-1][0] == 100  # simple.rs:30-34, :49-54
    parts = r.new_film(48, 40)
    for lo, hi in ((0, 4), (4, 5), (5, 9)):
        r.render(parts, cam, world, tile_range=(lo, hi))
    assert np.array_equal(film.grains[..., 1], parts.grains[..., 1])
    assert np.allclose(film.grains, parts.grains, rtol=1e-5)
    window = np.zeros((18, 48, film.bins, 2), dtype=np.float32)  # tile row 1 = pixel rows 16..31, plus halo rows 15 and 32
    r.render(film, cam, world, tile_range=(3, 6), film_rows=(15, 18), window=window)
    only = r.new_film(48, 40)
    r.render(only, cam, world, tile_range=(3, 6))
    assert np.allclose(window, only.grains[15:33], rtol=1e-6)
    assert only.grains[:15].sum() == 0 and only.grains[33:].sum() == 0


def test_full_size_c2_properties(gpu_lib):
    """BASELINE.json configs[1] at its full image size (1024 x 1024; 2 of the 256 spp) through size-independent properties:
    every sample exposes exactly S wavelengths (weights sum to samples * S), weights are whole numbers, a second render
    adds linearly, and a band of tiles equals the same band of the whole image."""
    world, cam, r, film = scenes.build(scenes.c2_cornell(1024, 1024, 2), seed=11)
    counters = r.render(film, cam, world, counters=True)
    assert counters["samples"] == 1024 * 1024 * 2 and counters["exposures"] == counters["samples"] * 10
    assert film.grains[..., 1].sum(dtype=np.float64) == counters["exposures"]
    assert np.array_equal(film.grains[..., 1], np.round(film.grains[..., 1]))
    assert np.isfinite(film.grains).all() and (film.grains[..., 0] >= 0).all()
    once = film.grains.copy()
    r.render(film, cam, world)  # same seed again: the film is a pure accumulator
    assert np.array_equal(film.grains[..., 1], 2 * once[..., 1])
    assert np.allclose(film.grains[..., 0], 2 * once[..., 0], rtol=1e-5)
    band = r.new_film(1024, 1024)
    r.render(band, cam, world, tile_range=(32 * 10, 32 * 12))  # tile rows 10 and 11
    rows = slice(10 * 32 + 1, 12 * 32 - 1)  # interior rows: no other tile row can leak into them
    assert np.array_equal(band.grains[rows][..., 1], once[rows][..., 1])
    assert np.allclose(band.grains[rows], once[rows], rtol=1e-5)
    # the oracle on a few tiles of the full-size image (tile-level parity at BASELINE size)
    cpu = r.new_film(1024, 1024)
    oracle.OracleScene(world).render(r, cam, cpu, threads=8, tile_range=(32 * 16 + 14, 32 * 16 + 18))
    gpu = r.new_film(1024, 1024)
    r.render(gpu, cam, world, tile_range=(32 * 16 + 14, 32 * 16 + 18))
    assert_parity(gpu, cpu)


# ------------------------------------------------------------------------------------------------ spectral tape (stage scheduler)
def _many_materials_scene(n_spheres, spectral, width=48, height=32, pixel_samples=8):
    """A floor, a lamp and `n_spheres` small spheres, each with a material of its own: `spectral` -> every colour is its own
    array spectrum (more spectrum-reading programs than the replay keeps values for), else its own constant."""
    from pyrite_amd.project import camera, light_source, material, shape, spectrum, transform, vector

    rng = np.random.RandomState(11)
    objects = [
        shape.sphere(position=vector(0, 0, -100), radius=100.0, material={"surface": material.diffuse(color=0.7)}),
        shape.sphere(position=vector(0, 0, 6), radius=1.0, material={"surface": material.emissive(color=light_source.d65 * 5)}),
    ]
    side = int(np.ceil(np.sqrt(n_spheres)))
    for i in range(n_spheres):
        x, y = (i % side) - side / 2 + 0.5, (i // side) - side / 2 + 0.5
        if spectral:
            colour = spectrum(format="array", min=380.0, max=780.0, points=[float(v) for v in rng.uniform(0.1, 0.9, 7)])
        else:
            colour = float(rng.uniform(0.1, 0.9))
        objects.append(shape.sphere(position=vector(x * 0.8, y * 0.8, 0.3), radius=0.3, material={"surface": material.diffuse(color=colour)}))
    return {
        "image": {"width": width, "height": height},
        "renderer": renderer.simple(pixel_samples=pixel_samples, light_samples=2, bounces=5, tile_size=16, spectrum_samples=5),
        "camera": camera.perspective(fov=60, transform=transform.look_at(**{"from": vector(0, -7, 4), "to": vector(0, 0, 0.3), "up": vector(z=1)})),
        "world": {"sky": light_source.d65 * 0.3, "objects": objects},
    }


@pytest.mark.parametrize("n_spheres,spectral", [(12, True), (150, False), (150, True)])
def test_tape_replay_fallbacks_match_the_oracle(n_spheres, spectral, gpu_lib, monkeypatch):
    """The stage scheduler's spectral tape: more than 8 spectrum-reading programs (values looked up record by record from the
    LDS table of prepared programs), more than 128 programs (prepared from the HBM records), and both at once."""
    monkeypatch.setenv("PYRITE_SCHEDULER", "sm")
    gfilm, cfilm, gcount, ccount = render_both(_many_materials_scene(n_spheres, spectral), 4, gpu_lib)
    assert_parity(gfilm, cfilm)
    assert gcount["exposures"] == ccount["exposures"] and gcount["shadow_rays"] == ccount["shadow_rays"]


def test_tape_records_the_bound_and_no_more(gpu_lib, monkeypatch):
    """A path appends at most 2 * bounces + 2 * light_samples + 1 records: a closed, all-diffuse box with an inner lamp and
    the maximum next-event work makes the longest tapes; one bounce and no light samples the shortest."""
    monkeypatch.setenv("PYRITE_SCHEDULER", "sm")
    for bounces, light_samples in ((12, 6), (1, 0), (2, 1)):
        project = scenes.c2_cornell(40, 40, 6)
        project["renderer"] = renderer.simple(pixel_samples=6, bounces=bounces, light_samples=light_samples, tile_size=16)
        gfilm, cfilm, _, _ = render_both(project, 9, gpu_lib)
        assert_parity(gfilm, cfilm)


@pytest.mark.parametrize("scheduler", ["sync", "sm"])
def test_c1_with_the_reference_shaped_lamp_material(scheduler, gpu_lib, monkeypatch):
    """SURVEY 8(d) writes C1's lamp sphere as cornell.lua's `emissive + diffuse`. The config itself uses a purely emissive lamp
    (a diffuse hit ON a spherical lamp samples that lamp from its own surface: solid_angle_towards is None there and
    lamp.rs:63-66 falls back to area / distance^2 with distance ~ 0 -- fireflies of 1e12 and more, a reference quirk that makes
    the workload useless). This keeps the reference-shaped material as a parity case, so that branch -- sample_towards from
    inside the shrunken radius, the area / d^2 weight with d ~ 0 -- runs on the GPU and agrees with the oracle, fireflies and all."""
    monkeypatch.setenv("PYRITE_SCHEDULER", scheduler)
    gfilm, cfilm, gcount, ccount = render_both(scenes.c1_spheres(64, 64, 8, reference_lamp=True), 6, gpu_lib)
    assert np.array_equal(gfilm.grains[..., 1], cfilm.grains[..., 1])
    for key in ("samples", "extension_rays", "shadow_rays", "shaded_hits", "exposures"):
        assert gcount[key] == ccount[key], key
    finite = np.isfinite(cfilm.grains[..., 0])
    assert np.array_equal(np.isfinite(gfilm.grains[..., 0]), finite)  # the same grains blew up, if any did
    g, c = np.where(finite, gfilm.grains[..., 0], 0.0), np.where(finite, cfilm.grains[..., 0], 0.0)
    assert np.allclose(g, c, rtol=2e-5, atol=1e-12)
    assert c.max() > 1e3 * np.median(c[c > 0])  # the quirk is there: some grain holds a firefly


def test_tape_overflow_is_an_error_not_a_wrong_film(gpu_lib, monkeypatch):
    """If a path ever wants more records than the tape's bound (a future change to the next-event count, say), the render
    must fail: the switch below shrinks the tape under the bound to show the overflow word reach the caller."""
    from pyrite_amd._lib import PyriteGpuError

    monkeypatch.setenv("PYRITE_SCHEDULER", "sm")
    world, cam, r, film = scenes.build(scenes.c2_cornell(32, 32, 4), seed=1)
    r.render(film, cam, world)  # the real bound: fine
    monkeypatch.setenv("PYRITE_TEST_TAPE_OPS", "3")
    with pytest.raises(PyriteGpuError, match="spectral tape"):
        r.render(r.new_film(32, 32), cam, world)
    monkeypatch.delenv("PYRITE_TEST_TAPE_OPS")
    again = r.new_film(32, 32)
    r.render(again, cam, world)  # the word was cleared with the error
    assert np.array_equal(again.grains[..., 1], film.grains[..., 1])
    world.close()


def test_full_size_c3_properties(gpu_lib):
    """BASELINE.json's C3 at full size on the stage scheduler with the spectral tape (819,212 triangles, 1920 x 1080, the
    persistent grid at its full width), 2 spp: every sample exposes its wavelengths exactly once; the image rendered in one
    launch equals the image rendered the way an 8-rank plan renders it (pyrite_amd/distributed.py: every rank one launch over
    its strided tiles into ringed tile blocks, rank 0 adds the blocks with the assembly kernel) -- weights exactly, spectra
    up to the order of the float atomics."""
    import torch

    from pyrite_amd import distributed as pdist

    W, H, spp = 1920, 1080, 2
    world, cam, r, whole = scenes.build(scenes.c3_mesh_in_box(W, H, spp), seed=2)
    c = r.render(whole, cam, world, counters=True)
    assert c["samples"] == W * H * spp and c["triangle_tests"] > 0
    weight = whole.grains[..., 1].sum(dtype=np.float64)
    assert c["exposures"] == weight <= W * H * spp * r.spectrum_samples and weight >= 0.9999 * W * H * spp * r.spectrum_samples
    assert not np.isnan(whole.grains).any()

    dev = torch.device("cuda", 0)
    desc = whole.desc()
    film = torch.zeros((H, W, r.spectrum_bins, 2), dtype=torch.float32, device=dev)
    stream = torch.cuda.current_stream(dev)
    total_samples = 0
    for share in pdist.plan(W, H, r.tile_size, 8):
        buffer = torch.zeros((share.pixels(W), r.spectrum_bins, 2), dtype=torch.float32, device=dev)
        r.render_device(buffer.data_ptr(), desc, cam, world, stream=stream.cuda_stream, device=0, share=share, flags=abi.PYR_FLAG_COUNTERS)
        total_samples += r.counters(world, 0)["samples"]
        assert float(buffer[..., 1].sum(dtype=torch.float64)) == r.counters(world, 0)["exposures"]  # the ring holds what spills over a tile's edge
        pdist.assemble(film, buffer, share, r.tile_size)
    assert total_samples == W * H * spp
    film = film.cpu().numpy()
    assert np.array_equal(film[..., 1], whole.grains[..., 1])
    assert np.allclose(film[..., 0], whole.grains[..., 0], rtol=1e-4, atol=1e-6)

    # the oracle on tiles of the full-size image, one by one, at 4 spp (tile-level parity at BASELINE size: the contract
    # config's diffuse 819 k-triangle mesh with next-event estimation on it, tracer.rs:257-280, :347-442): two tiles on the
    # mesh, one at its silhouette, two on the floor in the mesh's shadow, one in the wall / floor corner
    r.pixel_samples = 4
    sc = oracle.OracleScene(world)
    tiles_x = W // 32
    mesh_tests = []
    for row, col in ((24, 30), (22, 20), (26, 38), (30, 28), (29, 36), (31, 12)):
        tile = tiles_x * row + col
        cpu, gpu = r.new_film(W, H), r.new_film(W, H)
        cc = sc.render(r, cam, cpu, threads=8, tile_range=(tile, tile + 1))
        gc = r.render(gpu, cam, world, tile_range=(tile, tile + 1), counters=True)
        for key in ("samples", "extension_rays", "shadow_rays", "shaded_hits", "exposures"):
            assert gc[key] == cc[key], (row, col, key)
        assert cc["samples"] == 32 * 32 * 4 and cc["shadow_rays"] > cc["samples"]  # next-event estimation ran
        assert_parity(gpu, cpu)
        mesh_tests.append(gc["triangle_tests"] / gc["extension_rays"])
    assert max(mesh_tests) > 2 * min(mesh_tests)  # the tiles differ in how much of the mesh they see
    sc.close()
    world.close()


def c3_bench_rays(n, seed=1):
    """bench.py's `traversal_roofline` batch: uniform origins in the x10 Cornell box, uniform directions."""
    rng = np.random.RandomState(seed)
    o = rng.uniform([-55, 1, 1], [-1, 55, 54], size=(n, 3))
    d = rng.normal(size=(n, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    return np.concatenate([o, d], axis=1).astype(np.float32)


def camera_rays(cam, n, seed, aspect=1080.0 / 1920.0):
    """Primary rays of a pinhole camera through uniformly drawn view-plane positions (Camera::ray_towards, cameras.rs:70-97)."""
    c = cam.c
    m = np.array(list(c.cam_to_world), dtype=np.float64).reshape(4, 4).T  # cgmath matrices are column-major
    rng = np.random.RandomState(seed)
    x, y = rng.uniform(-1, 1, n), rng.uniform(-aspect, aspect, n)
    target = np.stack([x / c.view_plane * c.focus_distance, -y / c.view_plane * c.focus_distance, np.full(n, -c.focus_distance)], axis=1)
    d = target / np.linalg.norm(target, axis=1, keepdims=True)
    d = d @ m[:3, :3].T
    o = np.broadcast_to(m[:3, 3], (n, 3))
    return np.concatenate([o, d], axis=1).astype(np.float32)


def test_closest_hit_on_the_full_c3_mesh(gpu_lib):
    """World::intersect (world.rs:273-299) on the FULL 819,212-triangle C3 scene -- the tree (four-child nodes, triangle pairs)
    and the kernel (intersect_kernel, straight-line steps, five waves per SIMD) behind bench.py's `traversal_roofline` --
    against the oracle: 250 k rays of the very generator bench.py uses, 150 k camera rays (coherent, most of them end on
    the mesh) and 50 k shadow-like rays towards the lamp. Bit-exact distance, shape and barycentrics; every difference a
    proven tie (assert_same_hits)."""
    world, cam, r, _ = scenes.build(scenes.c3_mesh_in_box(64, 36, 1), seed=1)
    info = world.bvh_info()
    assert info["num_primitives"] == 819212
    rng = np.random.RandomState(4)
    floor = rng.uniform([-55, 1, 0.01], [-1, 55, 0.01], size=(50000, 3))
    lamp = rng.uniform([-34.3, 22.7, 54.79], [-21.3, 33.2, 54.79], size=(50000, 3))  # box.obj's light quad x10
    to_lamp = lamp - floor
    to_lamp /= np.linalg.norm(to_lamp, axis=1, keepdims=True)
    rays = np.concatenate([c3_bench_rays(250000), camera_rays(cam, 150000, 2), np.concatenate([floor, to_lamp], axis=1).astype(np.float32)])
    sc = oracle.OracleScene(world)
    ohits, _ = sc.intersect(rays)
    ghits, _, counters = world.intersect(rays, want_counters=True)
    ties = assert_same_hits(ohits, ghits, world, rays)
    assert ties < 0.001 * len(rays), ties
    on_mesh = (ohits["shape"] >> 30 == 1) & ((ohits["shape"] & 0x3FFFFFFF) >= 12)  # the box's 12 triangles come first
    assert on_mesh[:250000].mean() > 0.05 and on_mesh[250000:400000].mean() > 0.08 and (ohits["shape"] != 0xFFFFFFFF).mean() > 0.85  # the box is open at the front
    assert counters["triangle_tests"] > len(rays) and counters["box_tests"] > 10 * len(rays)
    # ... and eight million more of bench.py's rays (tools/big_hit_check.py ran 100 M once: 99,999,983 bit-exact, 16 ties, one
    # flat-box prune order; this slice is on the driver's record every round)
    total = ties
    for batch in range(16):
        rays = c3_bench_rays(500000, seed=100 + batch)
        ohits, _ = sc.intersect(rays)
        ghits, _, _ = world.intersect(rays)
        total += assert_same_hits(ohits, ghits, world, rays)
    assert total < 40, total  # ~2 ties per 10 M rays were seen
    sc.close()
    world.close()


def test_full_size_c3_oracle_tiles_at_the_bench_seeds(gpu_lib):
    """The oracle on tiles of the full-size C3 image at the seeds bench.py renders with (1, 2, 3), 2 spp: the image's
    four corners, the ragged bottom row (1080 = 33 x 32 + 24), an edge column, and tiles on the mesh and in its shadow -- 18
    tiles in all, path counters exactly, every pixel within 1e-5."""
    W, H = 1920, 1080
    world, cam, r, _ = scenes.build(scenes.c3_mesh_in_box(W, H, 2), seed=1)
    sc = oracle.OracleScene(world)
    tiles_x, tiles_y = W // 32, (H + 31) // 32
    assert (tiles_x, tiles_y) == (60, 34)
    picks = {1: ((0, 0), (0, 59), (33, 0), (33, 59), (24, 30), (30, 28)), 2: ((33, 31), (16, 0), (16, 59), (22, 20), (26, 38), (12, 30)),
             3: ((0, 30), (33, 12), (8, 45), (25, 25), (29, 35), (31, 12))}
    # (29, 35), not its neighbour (29, 36): at seed 3 that tile holds ONE sample whose camera ray meets two triangles of the mesh at
    # the same f32 distance (99.601166: a shared edge, u = 0.00013 on one, v = -0.0 on the other) -- the oracle's walk keeps one, this
    # library's tree the other, the normals differ, one pixel is off by 1.4 %. Traced with tools/tile_trace.py 29 36 3 2
    # (profiles/r04_tile_trace_29_36_seed3.txt): a tie of the kind DESIGN.md 5 describes, not a defect of either side.
    for seed, tiles in picks.items():
        r.seed = seed
        for row, col in tiles:
            tile = tiles_x * row + col
            cpu, gpu = r.new_film(W, H), r.new_film(W, H)
            cc = sc.render(r, cam, cpu, threads=8, tile_range=(tile, tile + 1))
            gc = r.render(gpu, cam, world, tile_range=(tile, tile + 1), counters=True)
            for key in ("samples", "extension_rays", "shadow_rays", "shaded_hits", "exposures"):
                assert gc[key] == cc[key], (seed, row, col, key)
            assert cc["samples"] == 32 * (24 if row == 33 else 32) * 2
            assert_parity(gpu, cpu)
    sc.close()
    world.close()


def test_full_size_c5_properties(gpu_lib):
    """BASELINE.json's C5 at its real mesh: the 819,212-triangle scene with the mesh made of dispersive glass, 20 bounces
    (test/dragon/dragon.lua:9, :30-35), 1920 x 1080 at 1 spp -- the deep tree, the long tapes (2 * 20 + 2 * 4 + 1 records)
    and the hero-only replay of dispersed paths at full size. Size-independent properties: every sample exposes its hero
    wavelength and, unless a bounce dispersed, its S - 1 companions (simple.rs:120-139), so a pixel's weight is a sum of 1s
    and Ss; the image rendered share by share the way 8 ranks render it equals the image rendered in one launch; and the
    oracle agrees sample for sample on tiles that show the glass mesh."""
    import torch

    from pyrite_amd import distributed as pdist

    W, H, spp = 1920, 1080, 1
    world, cam, r, whole = scenes.build(scenes.c3_mesh_in_box(W, H, spp, glass=True, bounces=20), seed=3)
    S = r.spectrum_samples
    c = r.render(whole, cam, world, counters=True)
    samples = W * H * spp
    weight = whole.grains[..., 1].sum(dtype=np.float64)
    assert c["samples"] == samples and c["exposures"] == weight
    assert samples * (1 - 1e-5) <= weight < samples * S  # some paths dispersed (hero only), none exposed more than S
    dispersed_share = (samples * S - weight) / (samples * (S - 1))
    assert 0.05 < dispersed_share < 0.6, dispersed_share  # the mesh fills a good part of the view
    assert np.array_equal(whole.grains[..., 1], np.round(whole.grains[..., 1])) and np.isfinite(whole.grains).all()
    assert samples <= c["extension_rays"] <= samples * 20

    dev = torch.device("cuda", 0)
    desc = whole.desc()
    film = torch.zeros((H, W, r.spectrum_bins, 2), dtype=torch.float32, device=dev)
    stream = torch.cuda.current_stream(dev)
    for share in pdist.plan(W, H, r.tile_size, 8):
        buffer = torch.zeros((share.pixels(W), r.spectrum_bins, 2), dtype=torch.float32, device=dev)
        r.render_device(buffer.data_ptr(), desc, cam, world, stream=stream.cuda_stream, device=0, share=share)
        pdist.assemble(film, buffer, share, r.tile_size)
    film = film.cpu().numpy()
    assert np.array_equal(film[..., 1], whole.grains[..., 1])
    assert np.allclose(film[..., 0], whole.grains[..., 0], rtol=1e-4, atol=1e-6)

    # the oracle on tiles in the middle of the mesh, one by one, at 4 spp (tile-level parity at BASELINE size)
    r.pixel_samples = 4
    sc = oracle.OracleScene(world)
    for tile in (60 * 16 + 28, 60 * 17 + 30, 60 * 12 + 25, 60 * 20 + 33):
        cpu, gpu = r.new_film(W, H), r.new_film(W, H)
        cc = sc.render(r, cam, cpu, threads=8, tile_range=(tile, tile + 1))
        gc = r.render(gpu, cam, world, tile_range=(tile, tile + 1), counters=True)
        assert gc["exposures"] == cc["exposures"] and gc["extension_rays"] == cc["extension_rays"] and gc["shadow_rays"] == cc["shadow_rays"]
        assert cc["exposures"] < cc["samples"] * S  # glass in view: dispersion happened here
        assert_parity(gpu, cpu)
    world.close()


def test_blocks_assembly_kernel_equals_the_torch_form(gpu_lib):
    """pyr_film_blocks_assemble_device against pyrite_amd.distributed.assemble_blocks_torch on random blocks: image edges cut
    tiles in both directions, strides 1 / 2 / 3, rings that overlap blocks of the same set."""
    import torch

    from pyrite_amd import distributed as pdist

    dev = torch.device("cuda", 0)
    width, height, ts, bins = 75, 41, 16, 5
    gen = torch.Generator(device="cpu").manual_seed(3)
    for n, rank in ((1, 0), (2, 1), (3, 2)):
        share = pdist.plan(width, height, ts, n, "tiles")[rank]
        blocks = torch.rand((share.pixels(width), bins, 2), generator=gen)
        expect = pdist.assemble_blocks_torch(torch.ones((height, width, bins, 2)), blocks, share, ts)
        got = pdist.assemble(torch.ones((height, width, bins, 2), device=dev), blocks.to(dev), share, ts).cpu()
        assert torch.allclose(got, expect, rtol=1e-6, atol=0)


@pytest.mark.parametrize("ranks", [2, 3])
def test_native_multi_device_render_equals_the_single_device_film(ranks, gpu_lib):
    """pyr_render_simple_multi with the same GPU standing in for every rank (RCCL refuses two ranks on one device, so the
    blocks travel by hipMemcpyPeerAsync: the plan, the strided launches, the ringed blocks and the assembly are the same
    code the RCCL path runs) against one plain render and against the oracle."""
    world, cam, r, whole = scenes.build(scenes.c2_cornell(72, 56, 6), seed=5)
    r.tile_size = 16  # 5 x 4 tiles, both edges cut
    r.render(whole, cam, world)
    multi = r.new_film(72, 56)
    seen = []
    r.render_multi(multi, cam, world, devices=[0] * ranks, on_status=lambda percent, message: seen.append(percent))
    assert seen == [0, 100]
    assert np.array_equal(multi.grains[..., 1], whole.grains[..., 1])
    assert np.allclose(multi.grains, whole.grains, rtol=1e-5)
    cpu = r.new_film(72, 56)
    oracle.OracleScene(world).render(r, cam, cpu, threads=4)
    assert_parity(multi, cpu)
    world.close()


def test_native_sharded_entry_with_one_rank(gpu_lib):
    """pyr_comm_create / pyr_render_simple_sharded for a world of one WITHOUT a communicator (the default for one rank): the
    block buffer and the assembly in place. The same entry with a real one-rank RCCL communicator, and the multi-rank flow,
    are in tests/test_gpu_multi.py."""
    import torch

    from pyrite_amd import distributed as pdist

    world, cam, r, whole = scenes.build(scenes.c2_cornell(64, 48, 4), seed=6)
    r.render(whole, cam, world)
    dev = torch.device("cuda", 0)
    film = torch.zeros((48, 64, r.spectrum_bins, 2), dtype=torch.float32, device=dev)
    comm = pdist.NativeSharded(0)
    assert not comm.uses_rccl
    comm.render(r, cam, world, whole.desc(), film, stream=torch.cuda.current_stream(dev).cuda_stream)
    torch.cuda.synchronize(dev)
    comm.status()
    comm.close()
    assert np.array_equal(film.cpu().numpy()[..., 1], whole.grains[..., 1])
    assert np.allclose(film.cpu().numpy(), whole.grains, rtol=1e-5)
    world.close()
