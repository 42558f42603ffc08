"""The project-file loader (next row f2): the declarative Lua subset project files are written in, evaluated against the
prelude's names, must build the same typed tree -- and so the same flattened scene -- as the Python surface does."""
import ctypes as C
import os

import numpy as np
import pytest

import oracle
from pyrite_amd import abi, lua_project, scenes
from pyrite_amd.lua_project import LuaError, evaluate
from pyrite_amd.project import (blackbody, camera, fresnel, light, light_source, material, mix, renderer, rgb, shape, spectrum, texture, transform,
                                vector)

HERE = os.path.dirname(os.path.abspath(__file__))
PROJECTS = os.path.join(HERE, "golden", "projects")
TEXTURES = os.path.join(HERE, "golden", "textures")


def test_values_tables_and_operators():
    assert evaluate("return 1 + 2 * 3 - 4 / 2") == 5.0
    assert evaluate("return 2 ^ 3 ^ 2, 7") == 512.0  # right associative; extra return values are dropped
    assert evaluate("return -2 ^ 2") == -4.0  # unary minus binds looser than ^
    assert evaluate('return "a" .. "b" .. 1') == "ab1"
    assert evaluate("return {1, 2, {3, 4}}") == [1.0, 2.0, [3.0, 4.0]]
    assert evaluate("return {a = 1, ['b c'] = 2, nested = {x = true, y = nil}}") == {"a": 1.0, "b c": 2.0, "nested": {"x": True, "y": None}}
    assert evaluate("local t = {10, 20, n = 'x'} return {t[2], t.n, #t}") == [20.0, "x", 2.0]
    assert evaluate("local a, b = 1 return {a, b == nil, not a, 1 < 2 and 'yes' or 'no'}") == [1.0, True, False, "yes"]
    assert evaluate("x = 3 -- a global\n--[[ long\ncomment ]] local y = x * 2 return y") == 6.0
    assert evaluate("return [[long\nstring]]") == "long\nstring"
    assert evaluate("local t = {} t.a = 1 t.b = {2} return t") == {"a": 1.0, "b": [2.0]}


def test_prelude_calls_build_the_typed_tree():
    v = evaluate("return vector(1, 2, 3)")
    assert v.type == "vector" and (v.x, v.y, v.z, v.w) == (1.0, 2.0, 3.0, 0.0)
    assert evaluate("return vector {z = 1}").z == 1.0
    e = evaluate("return light_source.d65 * 3 + 1")
    assert e.type == "binary" and e.operator == "add" and e.lhs.operator == "mul" and e.lhs.lhs.get("name") == "d65" and e.rhs == 1.0
    t = evaluate('return texture("a.png", "linear")')
    assert t.type == "color_texture" and t.path == "a.png" and t.linear and evaluate('return texture "b.png"').path == "b.png"
    m = evaluate("return material.emissive {color = 2} + material.diffuse {color = 0.78}")
    assert m.type == "binary" and m.lhs.type == "emissive" and m.rhs.color == 0.78
    s = evaluate("local ball = shape.sphere {radius = 1.5, position = vector(0, 1.4, 10)} return ball:with{position = ball.position:with{x = -3}}")
    assert s.type == "sphere" and s.material is None and s.radius == 1.5 and (s.position.x, s.position.y) == (-3.0, 1.4)
    look = evaluate("return transform.look_at {from = vector(0, 0, 15), to = vector()}")
    assert look.from_.z == 15.0 and look.up is None
    r = evaluate("return renderer.simple {pixel_samples = 500, spectrum_bins = 50, made_up = 1}")
    assert r.pixel_samples == 500.0 and r.spectrum_resolution is None
    g = evaluate("return material.refractive {ior = 1.5, _ior = 2.4, color = 1}")  # unknown fields are ignored (dragon.lua)
    assert g.ior == 1.5 and g.dispersion is None


def test_errors_name_the_file_and_line():
    with pytest.raises(LuaError, match=r"p.lua:2: attempt to call a nil value \(global 'sphere'\)"):
        evaluate("local a = 1\nreturn sphere {radius = 1}", "p.lua")
    with pytest.raises(LuaError, match="function definitions are not supported"):
        evaluate("local f = function(x) return x end return f(1)")
    with pytest.raises(LuaError, match="'for' is not supported"):
        evaluate("for i = 1, 3 do end")
    with pytest.raises(LuaError, match="attempt to index a nil value"):
        evaluate("return nothing.here")
    with pytest.raises(LuaError, match="arithmetic on a table and a number"):
        evaluate("return {} + 1")
    with pytest.raises(LuaError, match="module 'missing' not found"):
        evaluate('return require "missing"', base_dir=PROJECTS)
    with pytest.raises(LuaError, match="expected }"):
        evaluate("return {1, 2")


def python_gallery():
    """tests/golden/projects/gallery.lua + materials.lua, written with the Python surface."""
    curve = spectrum(format="curve", points=[[400, 0], [450, 0.3], [500, 0], [550, 1], [600, 0]])
    glass = material.refractive(ior=1.5, color=1)
    m = {
        "lamp": {"surface": material.emissive(color=light_source.d65 * 4)},
        "floor": {"surface": mix(material.mirror(color=1), material.diffuse(color=texture(os.path.join(TEXTURES, "tiles_color.png"))), fresnel(1.5)),
                  "normal_map": texture(os.path.join(TEXTURES, "tiles_normal.png"), "linear") * vector(1, -1, 1)},
        "green": {"surface": material.diffuse(color=curve)},
        "warm": {"surface": material.diffuse(color=curve.with_(points=[[580, 0], [600, 1], [610, 1], [650, 0]]))},
        "dense_glass": {"surface": glass.with_(ior=1.7, dispersion=0.01)},
        "rgb_paint": {"surface": material.diffuse(color=rgb(0.8, 0.3, 0.1) * 0.9 + 0.05)},
        "glow": {"surface": material.emissive(color=blackbody(3200) * 2e-13) + material.diffuse(color=0.5)},
    }
    ball = shape.sphere(radius=0.8, position=vector(0, 0.8, 0), material=None)
    lamp_ball = ball.with_(material=m["lamp"], radius=0.5, position=ball.position.with_(y=4, z=1))
    objects = [
        shape.plane(origin=vector(), normal=vector(y=1), material=m["floor"], texture_scale=4),
        lamp_ball,
        ball.with_(material=m["green"], position=ball.position.with_(x=-2)),
        ball.with_(material=m["warm"], position=ball.position.with_(x=2)),
        ball.with_(material=m["dense_glass"], radius=0.6, position=vector(-0.7, 0.6, 1.5)),
        ball.with_(material=m["rgb_paint"], radius=0.4, position=vector(0.9, 0.4, 2)),
        ball.with_(material=m["glow"], radius=0.3, position=vector(0, 0.3, 3)),
        shape.mesh(file=os.path.join(TEXTURES, "color_checker.obj"), scale=0.5,
                   materials={"color_checker": {"surface": material.diffuse(color=texture(os.path.join(TEXTURES, "color_checker.png")))}}),
        light.point(position=vector(-4, 5, 4), color=light_source.a * 8),
        light.directional(direction=vector(0.3, 0.9, 0.3), width=0.98, color=light_source.d65 * 0.5),
    ]
    return {
        "image": {"width": 96, "height": 64},
        "renderer": renderer.simple(pixel_samples=8, spectrum_samples=6, tile_size=16, bounces=6, light_samples=2),
        "camera": camera.perspective(fov=50, focus_distance=8.0, aperture=0.001,
                                     transform=transform.look_at(**{"from": vector(0, 2, 8), "to": vector(0, 1, 0)})),
        "world": {"sky": light_source.d65 * 0.1, "objects": objects},
    }


def desc_bytes(world):
    """Every array of the flattened scene, for equality checks."""
    d = world.desc
    out = {}
    for name, ctype in abi.PyrSceneDesc._fields_:
        value = getattr(d, name)
        if isinstance(value, (int, float)):
            out[name] = value
    flat = world.flat
    out["programs"], out["instrs"], out["materials"], out["components"] = flat.programs, flat.instrs, flat.materials, flat.components
    out["spectra"], out["spectrum_data"] = flat.spectra, list(flat.spectrum_data)
    out["spheres"], out["planes"], out["lamps"] = np.asarray(flat.spheres).tolist(), np.asarray(flat.planes).tolist(), repr(flat.lamps)
    out["tris"] = [np.asarray(t).reshape(-1).tolist() for t in flat.tri_positions]
    out["textures"] = [(f, t.tobytes()) for f, t in flat.textures]
    return out


def test_project_file_flattens_to_the_same_scene_as_the_python_surface():
    project, base_dir = lua_project.load_project(os.path.join(PROJECTS, "gallery.lua"))
    assert project["image"] == {"width": 96, "height": 64} and base_dir == PROJECTS
    lua_world, lua_cam, lua_r, lua_film = scenes.build(project, seed=5, base_dir=base_dir)
    py_world, py_cam, py_r, py_film = scenes.build(python_gallery(), seed=5)
    assert desc_bytes(lua_world) == desc_bytes(py_world)
    assert bytes(lua_cam.c) == bytes(py_cam.c)
    assert (lua_r.pixel_samples, lua_r.bounces, lua_r.light_samples, lua_r.spectrum_samples, lua_r.spectrum_bins, lua_r.tile_size) == (8, 6, 2, 6, 64, 16)
    oracle.OracleScene(lua_world).render(lua_r, lua_cam, lua_film, threads=4)
    oracle.OracleScene(py_world).render(py_r, py_cam, py_film, threads=4)
    assert np.array_equal(lua_film.grains[..., 1], py_film.grains[..., 1]) and lua_film.grains[..., 0].sum() > 0
    assert np.allclose(lua_film.grains, py_film.grains, rtol=1e-4, atol=1e-7)  # same samples, float-add order only


@pytest.mark.gpu
def test_gpu_renders_a_project_file(gpu_lib):
    from test_gpu_parity import assert_parity

    project, base_dir = lua_project.load_project(os.path.join(PROJECTS, "gallery.lua"))
    world, cam, r, film = scenes.build(project, seed=5, base_dir=base_dir)
    cpu = r.new_film(film.width, film.height)
    oracle.OracleScene(world).render(r, cam, cpu, threads=8)
    r.render(film, cam, world)
    assert_parity(film, cpu)


REFERENCE_TESTS = "/root/reference/pyrite/test"


@pytest.mark.skipif(not os.path.isdir(REFERENCE_TESTS), reason="the reference checkout only exists in the build container")
@pytest.mark.parametrize("name,restated", [("spheres", lambda: scenes.spheres_example(512, 256, 600)),
                                           ("diamonds", lambda: scenes.diamonds_example(512, 300, 200, bounces=256))])
def test_the_reference_projects_load_and_equal_the_restated_scenes(name, restated):
    """pyrite/test/<name>/<name>.lua read by the loader == the scene pyrite_amd/scenes.py restates by hand (the one the
    reference-image tests render): same camera, renderer parameters and flattened world."""
    project, base_dir = lua_project.load_project(os.path.join(REFERENCE_TESTS, name, name + ".lua"))
    lua_world, lua_cam, lua_r, lua_film = scenes.build(project, seed=1, base_dir=base_dir)
    py_world, py_cam, py_r, py_film = scenes.build(restated(), seed=1)
    assert desc_bytes(lua_world) == desc_bytes(py_world)
    assert bytes(lua_cam.c) == bytes(py_cam.c)
    for key in ("pixel_samples", "bounces", "light_samples", "spectrum_samples", "spectrum_bins", "tile_size"):
        assert getattr(lua_r, key) == getattr(py_r, key), key
    assert (lua_film.width, lua_film.height, lua_film.bins) == (py_film.width, py_film.height, py_film.bins)


@pytest.mark.skipif(not os.path.isdir(REFERENCE_TESTS), reason="the reference checkout only exists in the build container")
def test_every_reference_project_file_parses():
    import glob

    for path in sorted(glob.glob(os.path.join(REFERENCE_TESTS, "*", "*.lua"))):
        if os.path.basename(path) in ("materials.lua", "lamp.lua") or path.endswith("cornell/colors.lua"):
            continue  # modules that other projects `require`
        project, _ = lua_project.load_project(path)
        assert project["renderer"].type in ("simple", "bidirectional", "photon_mapping") and len(project["world"]["objects"]) >= 1, path
