"""Host logic: the project surface, the program compiler, material / world flattening and OBJ ingest
(pyrite/src/program/compiler.rs, materials/mod.rs:90-227, world.rs:39-271)."""
import math
import os

import numpy as np
import pytest

from pyrite_amd import abi, scenes
from pyrite_amd.compiler import FlatScene, ProjectError, camera_from_project, eval_number, eval_vector, load_obj, renderer_from_project
from pyrite_amd.project import blackbody, camera, fresnel, light_source, material, mix, renderer, rgb, shape, spectrum, texture, transform, vector

import oracle


def ops(flat, program):
    p = flat.programs[program]
    return [flat.instrs[p["first"] + k]["op"] for k in range(p["n"])]


def test_number_compiles_to_a_constant_program():
    flat = FlatScene()
    p = flat.compile(0.78)
    assert flat.programs[p]["kind"] == abi.PROGRAM_CONSTANT and flat.programs[p]["constant"] == float(np.float32(0.78))
    assert flat.instrs == []


def test_spectrum_times_number_lowering_order():
    # cornell.lua:5 `lamp.color * 3`: SpectrumValue, then the constant is materialised, then Binary Mul (compiler.rs:764-785)
    flat = FlatScene()
    lamp = spectrum(format="array", min=400, max=700, points=[1, 2, 3])
    p = flat.compile(lamp * 3)
    assert ops(flat, p) == [abi.OP_SPECTRUM, abi.OP_NUMBER, abi.OP_BINARY]
    mul = flat.instrs[flat.programs[p]["first"] + 2]
    assert (mul["a"], mul["b"], mul["operator"], mul["deps"]) == (0, 1, abi.BIN_MUL, abi.DEP_WAVELENGTH)
    p2 = flat.compile(3 * lamp)
    assert ops(flat, p2) == [abi.OP_SPECTRUM, abi.OP_NUMBER, abi.OP_BINARY]
    mul2 = flat.instrs[flat.programs[p2]["first"] + 2]
    assert (mul2["a"], mul2["b"]) == (1, 0)  # the number stays the left operand


def test_the_same_spectrum_table_gets_one_id():
    flat = FlatScene()
    s = spectrum(format="array", min=400, max=700, points=[1, 2])
    flat.compile(s * 2)
    flat.compile(s)
    flat.compile(light_source.d65)
    flat.compile(light_source.d65 * 3)
    assert len(flat.spectra) == 2
    d65 = flat.spectra[1]
    assert (d65[0], d65[1], d65[2], d65[4]) == (abi.SPECTRUM_ARRAY, 300.0, 830.0, 107)  # build.rs:131-187


def test_rgb_expression_is_converted_through_the_basis():
    flat = FlatScene()
    p = flat.compile(rgb(0.8, 0.3, 0.1) * 0.5)
    assert ops(flat, p) == [abi.OP_RGB, abi.OP_RGB, abi.OP_BINARY, abi.OP_RGB_SPECTRUM]  # constant widened to rgb, then RgbSpectrumValue
    assert flat.uses_rgb_basis
    assert flat.programs[p]["rgbs"] == 3 and flat.programs[p]["numbers"] == 1


def test_vector_cannot_be_a_colour_and_a_missing_texture_file_is_an_error():
    flat = FlatScene()
    with pytest.raises(ProjectError, match="vector as a number"):
        flat.compile(vector(1, 2, 3) * 2)
    with pytest.raises(ProjectError, match="could not load .* as color texture"):  # textures.rs:78-83
        flat.compile(texture("no_such_file.png"))


def test_mix_material_probabilities():
    # materials/mod.rs:176-195: lhs gets clamp(amount), rhs gets 1 - lhs probability; pushed rhs first
    flat = FlatScene()
    m, emissive = flat.add_material({"surface": mix(material.mirror(color=1), material.diffuse(color=0.8), 0.25)})
    first, n, first_e, n_e, _ = flat.materials[m]
    assert (n, n_e, emissive) == (2, 0, False)
    diffuse, mirror = flat.components[first], flat.components[first + 1]
    assert (diffuse["bsdf"], mirror["bsdf"]) == (abi.BSDF_DIFFUSE, abi.BSDF_MIRROR)
    assert flat.programs[mirror["probability"]]["constant"] == 0.25
    assert flat.programs[diffuse["probability"]]["constant"] == 0.75
    assert diffuse["compensation"] == 2.0 and mirror["compensation"] == 2.0


def test_fresnel_mix_builds_clamp_and_subtraction():
    flat = FlatScene()
    m, _ = flat.add_material({"surface": mix(material.mirror(color=1), material.diffuse(color=0.8), fresnel(1.5))})
    first = flat.materials[m][0]
    diffuse, mirror = flat.components[first], flat.components[first + 1]
    assert ops(flat, mirror["probability"]) == [abi.OP_FRESNEL, abi.OP_CLAMP]
    assert ops(flat, diffuse["probability"]) == [abi.OP_FRESNEL, abi.OP_CLAMP, abi.OP_NUMBER, abi.OP_BINARY]  # 1 - clamp(fresnel)


def test_added_materials_share_the_parent_probability_and_collect_emissive():
    flat = FlatScene()
    m, emissive = flat.add_material({"surface": material.emissive(color=2) + material.diffuse(color=0.78)})
    first, n, first_e, n_e, _ = flat.materials[m]
    assert (n, n_e, emissive) == (2, 1, True)
    assert [flat.components[first + k]["bsdf"] for k in range(2)] == [abi.BSDF_DIFFUSE, abi.BSDF_EMISSIVE]
    assert flat.components[first_e]["bsdf"] == abi.BSDF_EMISSIVE
    assert flat.components[first_e]["compensation"] == 1.0 and flat.components[first]["compensation"] == 2.0
    assert all(flat.components[first + k]["probability"] == -1 for k in range(2))


def test_constant_expression_evaluation():
    assert eval_number(mix(2, 4, 0.25)) == 2.5
    assert list(eval_vector(vector(1, 2, 3) * 2)) == [2, 4, 6, 0]
    with pytest.raises(ProjectError):
        eval_number(blackbody(4000))


def test_look_at_camera_matches_cgmath_conventions():
    cam = camera_from_project(scenes.cornell_camera())
    m = np.array(cam.cam_to_world[:]).reshape(4, 4).T  # column-major -> rows
    assert np.allclose(m[:3, 3], [-2.78, -8, 2.73])
    assert np.allclose(m[:3, 2], [0, -1, 0], atol=1e-7)  # camera looks down -Z: -f column
    assert np.allclose(m[:3, 1], [0, 0, 1], atol=1e-7)  # up
    assert cam.view_plane == pytest.approx(1 / math.tan(math.radians(37.7 / 2)), rel=1e-6)
    assert (cam.focus_distance, cam.aperture) == (1.0, 0.0)


def test_renderer_defaults_and_scope():
    r = renderer_from_project(renderer.simple(pixel_samples=7, spectrum_bins=50))  # unknown key ignored -> 64 bins
    assert r == dict(bounces=8, pixel_samples=7, light_samples=4, spectrum_samples=10, spectrum_bins=64, spectrum_span=(380.0, 780.0), tile_size=32)
    with pytest.raises(ProjectError, match="out of scope"):
        renderer_from_project(renderer.bidirectional(pixel_samples=1))


def test_cornell_box_ingest():
    world, cam, r, film = scenes.build(scenes.c2_cornell(32, 32, 1))
    flat = world.flat
    assert len(flat.tri_material) == 36 and len(flat.lamps) == 2  # the two `light` triangles (world.rs:225-229)
    assert [l["shape_kind"] for l in flat.lamps] == [abi.SHAPE_TRIANGLE] * 2
    light_tris = [l["shape_index"] for l in flat.lamps]
    pos = np.concatenate([np.asarray(p).reshape(-1, 9) for p in flat.tri_positions])
    assert np.allclose(pos[light_tris][:, 2::3], 5.48)  # the quad at z = 5.48
    nrm = np.concatenate([np.asarray(p).reshape(-1, 9) for p in flat.tri_normals])
    assert np.allclose(np.linalg.norm(nrm.reshape(-1, 3), axis=1), 1.0, atol=1e-6)
    with pytest.raises(ProjectError, match="missing material for 'tall'"):
        FlatScene().add_world({"objects": [shape.mesh(file=os.path.join(scenes.DATA_DIR, "cornell_box.obj"), materials={})]})


def test_obj_loader_indices_polys_and_flat_normals(tmp_path):
    path = tmp_path / "t.obj"
    path.write_text("o a\nv 0 0 0\nv 1 0 0\nv 0 1 0\nv 1 1 0\nvt 0 0\nvn 0 0 1\nf 1/1/1 2/1/1 3/1/1\nf -1 -2 -3\nf 1 2 4 3\no b\nf 1//1 2//1 4//1\n")
    mesh = load_obj(str(path))
    assert [o["name"] for o in mesh["objects"]] == ["a", "b"]
    assert mesh["objects"][0]["polys"][0] == [(0, 0, 0), (1, 0, 0), (2, 0, 0)]
    assert mesh["objects"][0]["polys"][1] == [(3, None, None), (2, None, None), (1, None, None)]  # negative = relative
    flat = FlatScene()
    white = {"surface": material.diffuse(color=1)}
    flat.add_world({"objects": [shape.mesh(file=str(path), materials={"a": white, "b": white}, scale=2.0)]})
    assert len(flat.tri_material) == 3  # the quad is skipped (world.rs:218-232)
    pos = np.concatenate([np.asarray(p).reshape(-1, 9) for p in flat.tri_positions])
    assert pos.max() == 2.0  # scale applied
    nrm = np.concatenate([np.asarray(p).reshape(-1, 9) for p in flat.tri_normals])
    assert np.allclose(nrm[1].reshape(3, 3), [[0, 0, 1]] * 3)  # flat normal: (v2-v1) x (v3-v1) = (-1,0,0) x (0,-1,0)


def test_compiled_programs_evaluate_like_the_expressions():
    flat = FlatScene()
    s = spectrum(format="array", min=400, max=700, points=[1.0, 3.0, 2.0])
    progs = {
        "scaled": flat.compile(s * 3),
        "mixed": flat.compile(mix(s, 10, 0.25)),
        "black": flat.compile(blackbody(3000) * 1e-13),
        "fres": flat.compile(fresnel(1.5)),
        "rgb": flat.compile(rgb(1, 1, 1)),
    }
    flat.sky_program = flat.compile(0.0)
    from pyrite_amd.renderer import World

    sc = oracle.OracleScene(World(flat))
    assert sc.run_program(progs["scaled"], 550.0)[0] == 9.0
    assert sc.run_program(progs["mixed"], 550.0) == (3.0 * 0.75 + 10 * 0.25, True)
    m = 600e-9
    assert sc.run_program(progs["black"], 600.0)[0] == pytest.approx(3.74183e-16 * m ** -5 / (math.exp(1.4388e-2 / (m * 3000)) - 1) * 1e-13, rel=1e-4)
    assert sc.run_program(progs["fres"], 500.0, normal=(0, 0, 1), incident=(0, 0, -1)) == (pytest.approx(0.04, rel=1e-5), False)
    white = sc.run_program(progs["rgb"], 550.0)[0]
    assert 0.9 < white < 1.1  # Burns' basis spectra sum to ~1 for white
