"""The BASELINE.json configurations as project trees.

  C1  Cornell-shaped box made of spheres only, 256x256, 64 spp           (reference CPU-runnable case)
  C2  test/cornell/box.obj (36 triangles, `light` = emissive quad), 1024x1024, 256 spp
  C3  C2 scaled x10 without the two blocks + a 819,200-triangle torus-knot "dragon stand-in", 1920x1080, 1024 spp
  C4  = C3 across 2/4/8 GPUs
  C5  C3 with the mesh made of dispersive glass (test/dragon/dragon.lua:30-35), bounces 20, 4096 spp

Materials, spectra and camera follow pyrite/test/cornell/cornell.lua:4-35 (spectra re-encoded in data/cornell_spectra.json).
`dragon.obj` is absent from the reference mount (.MISSING_LARGE_BLOBS); if a file named dragon.obj sits next to this
module it is used in place of the torus knot."""
from __future__ import annotations

import json
import os

import numpy as np

from .compiler import DATA_DIR, load_obj
from .project import camera, material, renderer, shape, spectrum, transform, vector

f32 = np.float32


def cornell_spectra():
    with open(os.path.join(DATA_DIR, "cornell_spectra.json")) as f:
        raw = json.load(f)
    return {k: spectrum(format="array", min=v["min"], max=v["max"], points=v["points"]) for k, v in raw.items()}


def cornell_materials():
    s = cornell_spectra()
    return {
        "light": {"surface": material.emissive(color=s["lamp"] * 3) + material.diffuse(color=0.78)},  # cornell.lua:4-7
        "white": {"surface": material.diffuse(color=s["white"])},
        "green": {"surface": material.diffuse(color=s["green"])},
        "red": {"surface": material.diffuse(color=s["red"])},
    }


def cornell_camera(scale=1.0):
    return camera.perspective(  # cornell.lua:28-35
        fov=37.7,
        transform=transform.look_at(**{"from": vector(-2.78 * scale, -8 * scale, 2.73 * scale), "to": vector(-2.78 * scale, 0, 2.73 * scale),
                                       "up": vector(z=1)}),
    )


def _simple(pixel_samples, **kw):
    return renderer.simple(pixel_samples=pixel_samples, **kw)


def c1_spheres(width=256, height=256, pixel_samples=64, reference_lamp=False):
    """`reference_lamp=True` gives the lamp sphere SURVEY 8(d)'s material -- cornell.lua:4-7's `emissive + diffuse` -- instead of
    the purely emissive one the config uses (see the comment at the lamp below): a parity case, not a workload."""
    m = cornell_materials()
    R = 100.0
    x0, x1, y1, z0, z1 = -5.56, 0.0, 5.592, 0.0, 5.488
    cx, cy, cz = (x0 + x1) / 2, y1 / 2, (z0 + z1) / 2
    objects = [
        shape.sphere(position=vector(x0 - R, cy, cz), radius=R, material=m["red"]),     # left wall
        shape.sphere(position=vector(x1 + R, cy, cz), radius=R, material=m["green"]),   # right wall
        shape.sphere(position=vector(cx, y1 + R, cz), radius=R, material=m["white"]),   # back wall
        shape.sphere(position=vector(cx, cy, z0 - R), radius=R, material=m["white"]),   # floor
        shape.sphere(position=vector(cx, cy, z1 + R), radius=R, material=m["white"]),   # ceiling
        shape.sphere(position=vector(-3.7, 3.3, 0.9), radius=0.9, material=m["white"]),
        shape.sphere(position=vector(-1.6, 1.7, 0.8), radius=0.8, material=m["white"]),
        # The lamp sphere is purely emissive (like test/spheres/spheres.lua:29-36). With cornell.lua's `emissive + diffuse`
        # light material a diffuse hit ON a spherical lamp samples that lamp from its own surface, where
        # solid_angle_towards returns None (shapes/mod.rs:253-271) and lamp.rs:63-66 falls back to area / distance^2 with
        # distance ~ 0: unbounded weights (fireflies of 1e12+) that are a reference quirk, not a useful parity workload.
        shape.sphere(position=vector(-2.78, 2.795, 4.9), radius=0.5,
                     material=m["light"] if reference_lamp else {"surface": material.emissive(color=cornell_spectra()["lamp"] * 3)}),
    ]
    return {
        "image": {"width": width, "height": height},
        "renderer": _simple(pixel_samples),
        "camera": cornell_camera(),
        "world": {"objects": objects},
    }


def _box_mesh(drop=()):
    mesh = load_obj(os.path.join(DATA_DIR, "cornell_box.obj"))
    mesh["objects"] = [o for o in mesh["objects"] if o["name"] not in drop]
    return mesh


def _box_materials(m, names):
    table = {"light": m["light"], "left": m["red"], "right": m["green"], "tall": m["white"], "short": m["white"], "back": m["white"],
             "ceiling": m["white"], "floor": m["white"]}  # cornell.lua:41-51
    return {k: table[k] for k in names}


def c2_cornell(width=1024, height=1024, pixel_samples=256):
    m = cornell_materials()
    mesh = _box_mesh()
    return {
        "image": {"width": width, "height": height},
        "renderer": _simple(pixel_samples),
        "camera": cornell_camera(),
        "world": {"objects": [shape.mesh(file=mesh, materials=_box_materials(m, [o["name"] for o in mesh["objects"]]))]},
    }


def torus_knot_mesh(segments=640, sides=640, p=2, q=3, tube=0.42, noise_seed=7, noise_amp=0.15,
                    fit_min=(-44.0, 12.0, 0.5), fit_max=(-12.0, 44.0, 32.0)):
    """(p,q) torus-knot tube, `segments` x `sides` x 2 triangles, radius modulated by seeded value noise, smooth
    vertex normals, scaled uniformly and translated to fit the given box. Returns (positions [n,3,3], normals [n,3,3])."""
    t = np.arange(segments, dtype=np.float64) * (2 * np.pi / segments)

    def curve(t):
        r = 2.0 + np.cos(q * t)
        return np.stack([r * np.cos(p * t), r * np.sin(p * t), np.sin(q * t)], axis=-1)

    c = curve(t)
    h = 1e-4
    tangent = curve(t + h) - curve(t - h)
    tangent /= np.linalg.norm(tangent, axis=1, keepdims=True)
    # frame from the direction towards the torus axis: smooth and closed for a torus knot
    radial = np.stack([np.cos(p * t), np.sin(p * t), np.zeros_like(t)], axis=-1)
    n1 = radial - (radial * tangent).sum(1, keepdims=True) * tangent
    n1 /= np.linalg.norm(n1, axis=1, keepdims=True)
    n2 = np.cross(tangent, n1)

    rng = np.random.RandomState(noise_seed)
    grid = 16
    lattice = rng.rand(grid, grid)
    a = np.arange(segments) * grid / segments
    b = np.arange(sides) * grid / sides
    a0, b0 = a.astype(int) % grid, b.astype(int) % grid
    fa, fb = (a - np.floor(a))[:, None], (b - np.floor(b))[None, :]
    fa, fb = fa * fa * (3 - 2 * fa), fb * fb * (3 - 2 * fb)
    v00 = lattice[a0][:, b0]
    v10 = lattice[(a0 + 1) % grid][:, b0]
    v01 = lattice[a0][:, (b0 + 1) % grid]
    v11 = lattice[(a0 + 1) % grid][:, (b0 + 1) % grid]
    noise = (v00 * (1 - fa) + v10 * fa) * (1 - fb) + (v01 * (1 - fa) + v11 * fa) * fb
    radius = tube * (1.0 + noise_amp * (2.0 * noise - 1.0))

    phi = np.arange(sides, dtype=np.float64) * (2 * np.pi / sides)
    ring = np.cos(phi)[None, :, None] * n1[:, None, :] + np.sin(phi)[None, :, None] * n2[:, None, :]
    verts = c[:, None, :] + radius[:, :, None] * ring  # [segments, sides, 3]

    lo, hi = verts.reshape(-1, 3).min(0), verts.reshape(-1, 3).max(0)
    fit_min, fit_max = np.asarray(fit_min, dtype=np.float64), np.asarray(fit_max, dtype=np.float64)
    scale = ((fit_max - fit_min) / (hi - lo)).min()
    verts = (verts - (lo + hi) / 2) * scale + (fit_min + fit_max) / 2
    verts = verts.astype(f32)

    i = np.arange(segments)[:, None]
    j = np.arange(sides)[None, :]
    i1, j1 = (i + 1) % segments, (j + 1) % sides
    idx = lambda ii, jj: (ii * sides + jj)  # noqa: E731
    quad = np.stack([idx(i, j) + 0 * j, idx(i1, j) + 0 * j, idx(i1, j1), idx(i, j1) + 0 * i], axis=-1).reshape(-1, 4)
    faces = np.concatenate([quad[:, [0, 1, 2]], quad[:, [0, 2, 3]]], axis=0)
    flat = verts.reshape(-1, 3)
    tri = flat[faces]  # [n,3,3]
    fn = np.cross(tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0]).astype(np.float64)
    vn = np.zeros((len(flat), 3), dtype=np.float64)
    for k in range(3):
        np.add.at(vn, faces[:, k], fn)
    vn /= np.linalg.norm(vn, axis=1, keepdims=True)
    normals = vn.astype(f32)[faces]
    return np.ascontiguousarray(tri, dtype=f32), np.ascontiguousarray(normals, dtype=f32)


def _mesh_dict_from_arrays(name, positions, normals):
    n = len(positions)
    ar = np.arange(3 * n).reshape(n, 3)
    polys = [[(int(a), None, int(a)), (int(b), None, int(b)), (int(c), None, int(c))] for a, b, c in ar]
    return {"position": positions.reshape(-1, 3), "texture": np.zeros((0, 2), dtype=f32), "normal": normals.reshape(-1, 3),
            "objects": [{"name": name, "polys": polys}]}


def c3_mesh_in_box(width=1920, height=1080, pixel_samples=1024, segments=640, sides=640, glass=False, bounces=None, mesh_material=None):
    """C3 (and C5 with glass=True). Built through FlatScene's bulk triangle path: see `c3_flat`. `mesh_material` (development:
    tools/bench_interp_mesh.py) gives the mesh another material, e.g. one that runs interpreter programs."""
    return {"image": {"width": width, "height": height}, "renderer": _simple(pixel_samples, bounces=bounces),
            "camera": cornell_camera(scale=10.0), "world": None,
            "flat": lambda: c3_flat(segments=segments, sides=sides, glass=glass, mesh_material=mesh_material)}


def c3_flat(segments=640, sides=640, glass=False, mesh_material=None):
    from .compiler import FlatScene  # local import: scenes are plain data otherwise

    m = cornell_materials()
    box = _box_mesh(drop=("tall", "short"))
    flat = FlatScene()
    flat.add_world({"objects": [shape.mesh(file=box, scale=10.0, materials=_box_materials(m, [o["name"] for o in box["objects"]]))]})
    dragon = os.path.join(os.path.dirname(os.path.abspath(__file__)), "dragon.obj")
    if mesh_material is not None:
        pass
    elif glass:
        mesh_material = {"surface": material.refractive(ior=1.5, dispersion=0.01371, color=1)}  # dragon.lua:30-35
    else:
        mesh_material = {"surface": material.diffuse(color=0.8)}
    mat, _ = flat.add_material(mesh_material)
    if os.path.exists(dragon):
        mesh = load_obj(dragon)
        pos, nrm = [], []
        for o in mesh["objects"]:
            for poly in o["polys"]:
                if len(poly) == 3 and all(ix[2] is not None for ix in poly):
                    pos.append([mesh["position"][ix[0]] for ix in poly])
                    nrm.append([mesh["normal"][ix[2]] for ix in poly])
        positions, normals = np.asarray(pos, dtype=f32), np.asarray(nrm, dtype=f32)
    else:
        positions, normals = torus_knot_mesh(segments=segments, sides=sides)
    flat.add_triangles(positions, normals, mat)
    return flat


def build(project, seed=1, base_dir="."):
    """Project tree -> (World, Camera, Renderer, Film) through the same from_project constructors the reference uses
    (main.rs:111-134 parse_project, :190-195 Film::new). Mesh and texture paths are relative to `base_dir`, the directory
    of the project file (project/mod.rs:73-76)."""
    from .renderer import Camera, Renderer, World

    r = Renderer.from_project(project["renderer"], seed=seed)
    cam = Camera.from_project(project["camera"])
    if project.get("world") is None and "flat" in project:
        world = World(project["flat"]())
    else:
        world = World.from_project(project["world"], base_dir)
    film = r.new_film(int(project["image"]["width"]), int(project["image"]["height"]))
    return world, cam, r, film


CONFIGS = {
    "C1": c1_spheres,
    "C2": c2_cornell,
    "C3": c3_mesh_in_box,
    "C5": lambda **kw: c3_mesh_in_box(glass=True, bounces=20, **{"pixel_samples": 4096, **kw}),
}


# ------------------------------------------------------------------------------------------------
# Small scenes that follow the reference's other example projects; used by the parity tests to reach the
# code paths C1-C5 do not (interpreted programs, curve spectra, mirror / dispersive glass, thin lens, point and
# directional lamps, planes, sky).
# ------------------------------------------------------------------------------------------------
def spheres_example(width=128, height=64, pixel_samples=16):
    """pyrite/test/spheres/spheres.lua:1-69 (simple renderer, D65 ball lamp, fresnel mirror/diffuse mix, curve spectra)."""
    from .project import fresnel, light_source, mix

    ball = shape.sphere(radius=1.5, position=vector(0, 1.4, 10), material=None)
    green = spectrum(format="curve", points=[(400, 0), (450, 0.3), (500, 0), (550, 1), (600, 0)])
    red = spectrum(format="curve", points=[(580, 0), (600, 1), (610, 1), (650, 0)])
    objects = [
        shape.sphere(radius=50.0, position=vector(0, -50, 10), material={"surface": material.diffuse(color=1)}),
        ball.with_(position=vector(0, 1.5, 10), material={"surface": material.emissive(color=light_source.d65 * 3)}),
        ball.with_(position=vector(-3, 1.4, 10),
                   material={"surface": mix(material.mirror(color=1), material.diffuse(color=green), fresnel(1.5))}),
        ball.with_(position=vector(3, 1.4, 10), material={"surface": material.diffuse(color=red)}),
    ]
    return {
        "image": {"width": width, "height": height},
        "camera": camera.perspective(fov=53, transform=transform.look_at(**{"from": vector(0, 1, 0), "to": vector(0, 1, 1)})),
        "renderer": renderer.simple(pixel_samples=pixel_samples, spectrum_samples=10, spectrum_bins=50, tile_size=32, light_samples=4),
        "world": {"objects": objects},
    }


def diamonds_example(width=128, height=75, pixel_samples=8, bounces=32):
    """pyrite/test/diamonds/diamonds.lua:1-60 (dispersive glass mesh, plexi mirror with a fresnel mix colour, two quad
    lamps, thin lens, spectrum_samples = 1). `bounces` is 256 in the project file; tests use fewer."""
    from .project import fresnel, light_source, mix

    diamond = {"surface": material.refractive(ior=2.37782, dispersion=0.01371, color=1)}
    plexi = {"surface": material.mirror(color=mix(0, 0.2, fresnel(1.1)))}
    mesh = shape.mesh(file=os.path.join(DATA_DIR, "diamonds.obj"), materials={
        "diamonds": diamond,
        "light_left": {"surface": material.emissive(color=light_source.d65)},
        "light_right": {"surface": material.emissive(color=light_source.d65 * 2)},
        "bottom": plexi,
    })
    return {
        "image": {"width": width, "height": height},
        "renderer": renderer.simple(pixel_samples=pixel_samples, spectrum_samples=1, spectrum_bins=50, tile_size=32, bounces=bounces),
        "camera": camera.perspective(fov=12.5, focus_distance=11.08, aperture=0.02,
                                     transform=transform.look_at(**{"from": vector(-6.55068, -8.55076, 4.0), "to": vector(0.1, 0, 0.1),
                                                                    "up": vector(z=1)})),
        "world": {"objects": [mesh]},
    }


def lamps_example(width=96, height=64, pixel_samples=16):
    """Point lamp + directional lamp + sky over a plane floor with a glass sphere (constant ior: companions survive),
    a blackbody-coloured diffuse sphere and an rgb() coloured sphere: the lamp kinds and opcodes no other scene reaches
    (lamp.rs:24-52, tracer.rs:444-459, shapes/mod.rs:441-452, Blackbody / RgbSpectrumValue)."""
    from .project import blackbody, light, light_source, rgb

    objects = [
        shape.plane(origin=vector(0, 0, 0), normal=vector(z=1), material={"surface": material.diffuse(color=0.5)}),
        shape.sphere(position=vector(-1.2, 0, 1), radius=1.0, material={"surface": material.refractive(ior=1.5, color=1)}),
        shape.sphere(position=vector(1.2, 0.5, 0.7), radius=0.7, material={"surface": material.diffuse(color=blackbody(3000) * 2e-13)}),
        shape.sphere(position=vector(0.2, -1.6, 0.5), radius=0.5, material={"surface": material.diffuse(color=rgb(0.8, 0.3, 0.1))}),
        light.point(position=vector(3, -3, 5), color=light_source.d65 * 40),
        light.directional(direction=vector(-0.3, 0.2, 0.933), width=0.98, color=light_source.a * 2),
    ]
    return {
        "image": {"width": width, "height": height},
        "renderer": renderer.simple(pixel_samples=pixel_samples, light_samples=2, bounces=6, tile_size=16),
        "camera": camera.perspective(fov=45, transform=transform.look_at(**{"from": vector(0, -7, 2.5), "to": vector(0, 0, 0.8), "up": vector(z=1)})),
        "world": {"sky": light_source.d65 * 0.2, "objects": objects},
    }


def _generated_textures(seed=3, size=16):
    """Small seeded images standing in for texture files: an sRGB checker, a smooth 16-bit height-like mono image and a
    tangent-space normal map (linear RGB, blue-ish), so that tests need no image files."""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:size, 0:size]
    checker = np.where(((xx // 2 + yy // 2) % 2)[..., None] == 0, np.array([230, 60, 40]), np.array([40, 90, 220])).astype(np.uint8)
    checker = np.clip(checker.astype(int) + rng.integers(-20, 20, checker.shape), 0, 255).astype(np.uint8)
    mono = (32768 + 20000 * np.sin(xx * 0.9) * np.cos(yy * 0.7) + rng.integers(-3000, 3000, (size, size))).astype(np.uint16)
    bump = rng.normal(0.0, 0.25, (size, size, 2))
    nz = np.sqrt(np.clip(1.0 - (bump ** 2).sum(-1), 0.05, 1.0))
    normal_map = np.clip((np.concatenate([bump, nz[..., None]], -1) * 0.5 + 0.5) * 255.0, 0, 255).astype(np.uint8)
    rgba = np.concatenate([checker, rng.integers(100, 255, (size, size, 1)).astype(np.uint8)], -1)
    return {"checker": checker, "mono": mono, "normal_map": normal_map, "rgba": rgba}


def textures_example(width=96, height=64, pixel_samples=8):
    """What pyrite/test/textures exercises (the directory is not in the reference checkout): colour and mono textures on a
    plane, a sphere and a uv-mapped mesh, normal maps on all three shape kinds, a textured emissive sphere (the light
    sample's texture coordinates, lamp.rs:64-77) -- texture.rs:87-150, execution_context.rs:114-139,
    materials/mod.rs:68-80, shapes/mod.rs:346-385 / :454-469 / :550-558, world.rs:308-374."""
    from .project import light, light_source, mix, rgb, texture

    tex = _generated_textures()
    checker, mono, nmap, rgba = tex["checker"], tex["mono"], tex["normal_map"], tex["rgba"]
    # a uv-mapped, slightly tilted quad (two triangles) with its own vertex normals
    quad = {
        "position": np.array([[-1.5, 1.0, 0.2], [0.5, 1.0, 0.2], [0.5, 2.6, 1.4], [-1.5, 2.6, 1.4]], dtype=f32),
        "texture": np.array([[0, 0], [2, 0], [2, 1.5], [0, 1.5]], dtype=f32),
        "normal": np.array([[0, -0.6, 0.8], [0.1, -0.6, 0.8], [0, -0.55, 0.83], [-0.1, -0.6, 0.8]], dtype=f32),
        "objects": [{"name": "quad", "polys": [[(0, 0, 0), (1, 1, 1), (2, 2, 2)], [(0, 0, 0), (2, 2, 2), (3, 3, 3)]]}],
    }
    objects = [
        shape.plane(origin=vector(0, 0, 0), normal=vector(z=1), texture_scale=vector(1.5, 2.5),
                    material={"surface": material.diffuse(color=texture(checker) * 0.9), "normal_map": texture(nmap, "linear")}),
        shape.sphere(position=vector(-1.6, -0.4, 0.9), radius=0.9, texture_scale=vector(0.25, 0.5),
                     material={"surface": material.diffuse(color=texture(rgba)), "normal_map": texture(nmap, "linear")}),
        shape.sphere(position=vector(1.5, 0.2, 0.7), radius=0.7,
                     material={"surface": mix(material.mirror(color=1), material.diffuse(color=rgb(0.9, 0.8, 0.3)), texture(mono, "mono", "linear"))}),
        shape.mesh(file=quad, transform=transform.look_at(**{"from": vector(0.3, 0.2, 0), "to": vector(0.3, 0.2, -1), "up": vector(0.1, 1, 0)}),
                   scale=1.1, materials={"quad": {"surface": material.diffuse(color=texture(checker)), "normal_map": texture(nmap, "linear")}}),
        shape.sphere(position=vector(0.2, -1.4, 2.6), radius=0.35, texture_scale=vector(0.5, 0.5),
                     material={"surface": material.emissive(color=light_source.d65 * texture(mono, "mono") * 25)}),
        light.point(position=vector(3, -3, 4), color=light_source.a * 6),
    ]
    return {
        "image": {"width": width, "height": height},
        "renderer": renderer.simple(pixel_samples=pixel_samples, light_samples=2, bounces=5, tile_size=16, spectrum_samples=6, spectrum_bins=24),
        "camera": camera.perspective(fov=50, transform=transform.look_at(**{"from": vector(0, -6, 2.6), "to": vector(0, 0, 0.8), "up": vector(z=1)})),
        "world": {"sky": light_source.d65 * 0.15, "objects": objects},
    }


def textures_reference_example(texture_dir, width=1024, height=512, pixel_samples=400):
    """pyrite/test/textures/textures.lua:1-91 (simple renderer): a mirror / textured-diffuse fresnel mix floor with a normal
    map on a plane, two D65 ball lamps, the uv-mapped colour-checker quad, a textured + normal-mapped sphere and a cube. The
    texture images come from `texture_dir` (tests/golden/textures holds shrunk copies); the cube's `fabric` textures are
    not in the reference checkout, so the cube is plain diffuse here."""
    from .project import fresnel, light_source, mix, texture

    def tex(name, *modifiers):
        return texture(os.path.join(texture_dir, name), *modifiers)

    light_ball = shape.sphere(material={"surface": material.emissive(color=light_source.d65 * 20)}, position=vector(0, 0, 0), radius=1)
    floor_material = {
        "surface": mix(material.mirror(color=1), material.diffuse(color=tex("tiles_color.png")), fresnel(1.5)),
        "normal_map": tex("tiles_normal.png", "linear") * vector(1, -1, 1),
    }
    objects = [
        shape.plane(origin=vector(), normal=vector(y=1), material=floor_material, texture_scale=5),
        light_ball.with_(position=vector(-1, 12, 2), radius=3),
        light_ball.with_(position=vector(15, 3, 4)),
        shape.mesh(file=os.path.join(texture_dir, "color_checker.obj"),
                   materials={"color_checker": {"surface": material.diffuse(color=tex("color_checker.png"))}}),
        shape.sphere(position=vector(-3, 1, 0), radius=1, texture_scale=vector(0.5, 1),
                     material={"surface": material.diffuse(color=tex("tactile_paving_color.png")),
                               "normal_map": tex("tactile_paving_normal.png", "linear") * vector(1, -1, 1)}),
        shape.mesh(file=os.path.join(texture_dir, "cube.obj"),
                   transform=transform.look_at(**{"from": vector(2, 0.5, 1), "to": vector(-1, 0.5, 2)}),
                   materials={"cube": {"surface": material.diffuse(color=0.6)}}),
    ]
    return {
        "image": {"width": width, "height": height},
        "renderer": renderer.simple(pixel_samples=pixel_samples, spectrum_samples=10, spectrum_bins=50, tile_size=32, bounces=8, light_samples=2),
        "camera": camera.perspective(fov=53, transform=transform.look_at(**{"from": vector(0, 2, 12), "to": vector(0, 2, 0)})),
        "world": {"objects": objects},
    }
