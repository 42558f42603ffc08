"""The HIP path against ANALYTIC radiance values -- no oracle anywhere in this file ("energy / furnace
tests"). The parity tests prove GPU == oracle; these stand outside that pair and would catch a misreading of the reference
that the kernels and the oracle share: component selection weights (materials/mod.rs:48-54, :213-221), uniform-hemisphere
sampling with the 2 |n.o| Lambert weight (math.rs:155-164, diffuse.rs:27-29), emission accounting and the next-event gate
(tracer.rs:257-280, :303-318), light sampling weights (lamp.rs:54-77, shapes/mod.rs:180-204, tracer.rs:365), the bounce
limit (tracer.rs:221) and Film::expose's weights (film.rs:89-95, simple.rs:133-139)."""
import numpy as np
import pytest

from pyrite_amd import scenes
from pyrite_amd.compiler import FlatScene
from pyrite_amd.project import camera, material, renderer, shape, transform, vector
from pyrite_amd.renderer import Camera, Renderer, World

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu_lib():
    from pyrite_amd import _lib

    lib = _lib.lib()
    if lib.pyr_device_count() < 1:
        pytest.fail("no HIP device: the analytic tests need a GPU and pyrite_amd has no CPU path")
    return lib


def mean_radiance(film):
    """Mean developed value over all pixels and bins: sum(acc) / sum(weight) (every exposure has weight 1, film.rs:128-143)."""
    return float(film.grains[..., 0].sum(dtype=np.float64) / film.grains[..., 1].sum(dtype=np.float64))


def closed_box(albedo, emission, half=1.0):
    """A watertight cube of 12 triangles seen from inside, every face `emissive(emission) + diffuse(albedo)`; every triangle is
    a lamp (World::from_project registers emissive shapes as lights, world.rs:184-236)."""
    flat = FlatScene()
    flat.sky_program = flat.compile(0.0)
    mat, _ = flat.add_material({"surface": material.emissive(color=emission) + material.diffuse(color=albedo)})
    h = half
    corners = np.array([[x, y, z] for x in (-h, h) for y in (-h, h) for z in (-h, h)], dtype=np.float32)
    quads = [(0, 1, 3, 2, (1, 0, 0)), (4, 6, 7, 5, (-1, 0, 0)), (0, 4, 5, 1, (0, 1, 0)), (2, 3, 7, 6, (0, -1, 0)), (0, 2, 6, 4, (0, 0, 1)),
             (1, 5, 7, 3, (0, 0, -1))]  # inward normals
    pos, nrm = [], []
    for a, b, c, d, n in quads:
        for tri in ((a, b, c), (a, c, d)):
            pos.append(corners[list(tri)])
            nrm.append(np.tile(np.array(n, dtype=np.float32), (3, 1)))
    flat.add_triangles(np.array(pos), np.array(nrm), mat, emissive=True)
    return World(flat)


@pytest.mark.parametrize("scheduler", ["sync", "sm"])
@pytest.mark.parametrize("light_samples,bounces", [(0, 8), (4, 8), (2, 3), (4, 1)])
def test_furnace_closed_box_has_the_geometric_series_radiance(scheduler, light_samples, bounces, gpu_lib, monkeypatch):
    """Inside a closed box whose walls all emit E and reflect a Lambertian fraction rho, the radiance after B path segments is
    E (1 + rho + ... + rho^(B-1)) in every direction. Pyrite's estimator reaches it like this: a hit picks the emissive or the
    diffuse component with probability 1/2 and weight 2; emission ends the path, a diffuse bounce multiplies the throughput by
    rho * 2 |n.o| whose mean over the uniform hemisphere is rho. With next-event estimation (light_samples > 0) the first two
    diffuse events sample the twelve wall triangles instead of waiting to hit one (tracer.rs:257-280) -- a different
    estimator of the same series, so both must give the same number. (A last segment that picks the diffuse component with
    next-event estimation still on adds the light one bounce further: the series then has one term more, see below.)"""
    monkeypatch.setenv("PYRITE_SCHEDULER", scheduler)
    rho, E = 0.5, 1.0
    world = closed_box(rho, E)
    cam = Camera.from_project(camera.perspective(fov=70, transform=transform.look_at(**{"from": vector(0.1, -0.2, 0.15), "to": vector(0.4, 1, 0.2), "up": vector(z=1)})))
    r = Renderer(pixel_samples=1024, bounces=bounces, light_samples=light_samples, spectrum_samples=4, spectrum_bins=8, tile_size=16, seed=5)
    film = r.new_film(64, 64)
    r.render(film, cam, world)
    assert film.total_weight() == 64 * 64 * 1024 * 4
    # emission is seen at hit k (k = 0 .. B-1) with throughput rho^k; next-event estimation at diffuse event j <= 1 stands for
    # the emission of hit j + 1, also when j + 1 == B (the path is cut there, the light sample was already taken)
    terms = bounces + (1 if light_samples > 0 and bounces <= 2 else 0)
    expected = E * sum(rho ** k for k in range(terms))
    got = mean_radiance(film)
    assert got == pytest.approx(expected, rel=0.01), (got, expected)
    world.close()


@pytest.mark.parametrize("scheduler", ["sync", "sm"])
def test_convex_diffuse_sphere_under_a_uniform_sky_reflects_albedo(scheduler, gpu_lib, monkeypatch):
    """One convex Lambertian sphere of albedo 0.5 under sky radiance 1: every path is sphere -> sky, so the value is exactly
    albedo * E[2 cos] * sky = 0.5 on the sphere and exactly 1 off it."""
    monkeypatch.setenv("PYRITE_SCHEDULER", scheduler)
    project = {
        "image": {"width": 32, "height": 32},
        "renderer": renderer.simple(pixel_samples=256, light_samples=0, spectrum_samples=4, tile_size=16),
        "camera": camera.perspective(fov=20, transform=transform.look_at(**{"from": vector(0, 0, 10), "to": vector(0, 0, 0)})),
        "world": {"sky": 1.0, "objects": [shape.sphere(position=vector(0, 0, 0), radius=1.0, material={"surface": material.diffuse(color=0.5)})]},
    }
    world, cam, r, film = scenes.build(project, seed=2)
    r.render(film, cam, world)
    dev = film.grains[..., 0].sum(-1) / np.maximum(film.grains[..., 1].sum(-1), 1)
    assert dev[14:18, 14:18].mean() == pytest.approx(0.5, abs=0.01)  # centre of the sphere
    assert dev[0:3, 0:3].mean() == pytest.approx(1.0, abs=1e-6)  # sky only
    assert film.total_weight() == 32 * 32 * 256 * 4
    world.close()


def test_mirror_box_preserves_radiance(gpu_lib):
    """A perfect mirror (colour 1) in front of a uniform sky shows the sky unchanged whatever the number of reflections:
    mirror.rs:5-21 has probability 1 and no cosine factor."""
    project = {
        "image": {"width": 32, "height": 32},
        "renderer": renderer.simple(pixel_samples=64, light_samples=2, spectrum_samples=3, tile_size=16, bounces=6),
        "camera": camera.perspective(fov=40, transform=transform.look_at(**{"from": vector(0, -6, 2), "to": vector(0, 0, 0.5), "up": vector(z=1)})),
        "world": {"sky": 0.75, "objects": [shape.sphere(position=vector(-1.1, 0, 1), radius=1.0, material={"surface": material.mirror(color=1)}),
                                           shape.sphere(position=vector(1.1, 0, 1), radius=1.0, material={"surface": material.mirror(color=1)})]},
    }
    world, cam, r, film = scenes.build(project, seed=3)
    r.render(film, cam, world)
    dev = film.grains[..., 0] / np.maximum(film.grains[..., 1], 1)
    seen = film.grains[..., 1] > 0
    # a path that is still between the two mirrors after 6 segments contributes 0; everything else is exactly the sky
    values = dev[seen]
    assert np.all((np.abs(values - 0.75) < 1e-6) | (values < 0.75))
    assert (np.abs(values - 0.75) < 1e-6).mean() > 0.95
    world.close()
