"""The C++ host layer (include/pyrite_host.hpp, libpyrite_host.so) against the Python front-end and the oracle.

CPU: every scene of pyrite_host_tool, written against the C++ surface, flattens to the same bytes as the same scene written
against the Python surface (program compiler, material flattening, world flattening, OBJ ingest, tangent frames, textures,
camera and renderer parameters). GPU: Renderer::render through the C++ seam gives the oracle's film."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from pyrite_amd import abi, build as gpu_build, images, scenes
from pyrite_amd.compiler import DATA_DIR, FlatScene, camera_from_project, renderer_from_project

HOST_LIB, HOST_TOOL = gpu_build.HOST_OUT, gpu_build.HOST_TOOL

SCENES = {
    "c1": scenes.c1_spheres,
    "c2": scenes.c2_cornell,
    "spheres": scenes.spheres_example,
    "diamonds": scenes.diamonds_example,
    "lamps": scenes.lamps_example,
    "textures": scenes.textures_example,
}


@pytest.fixture(scope="module")
def host():
    gpu_build.build_host()
    lib = C.CDLL(HOST_LIB)
    lib.pyrh_serialize_desc.restype = C.c_uint64
    lib.pyrh_serialize_desc.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64]
    return lib


def scene_bytes(lib, desc):
    n = lib.pyrh_serialize_desc(C.byref(desc), None, 0)
    buf = (C.c_uint8 * n)()
    assert lib.pyrh_serialize_desc(C.byref(desc), buf, n) == n
    return bytes(buf)


def data_dir_for(name, tmp_path):
    """The textures scene reads its texel arrays (linear f32, as Texture::from_path leaves them) from files."""
    if name != "textures":
        return DATA_DIR
    tex = scenes._generated_textures()
    for fname, source, linear, mono in (("checker", tex["checker"], False, False), ("nmap_linear", tex["normal_map"], True, False),
                                        ("rgba", tex["rgba"], False, False), ("mono_linear", tex["mono"], True, True),
                                        ("mono_srgb", tex["mono"], False, True)):
        images.linearise(source, linear, mono).astype("<f4").tofile(os.path.join(tmp_path, fname + ".f32"))
    return str(tmp_path)


@pytest.mark.parametrize("name", sorted(SCENES))
def test_cpp_front_end_flattens_like_the_python_front_end(host, name, tmp_path):
    project = SCENES[name]()
    flat = FlatScene().add_world(project["world"], DATA_DIR)
    expected = scene_bytes(host, flat.desc())
    out = os.path.join(tmp_path, "scene.bin")
    subprocess.check_call([HOST_TOOL, "dump", name, data_dir_for(name, tmp_path), out], stdout=subprocess.DEVNULL)
    with open(out, "rb") as f:
        got = f.read()
    assert got[:len(expected)] == expected, "flattened scene differs"
    rest = got[len(expected):]
    cam = np.frombuffer(rest[:C.sizeof(abi.PyrCamera)], dtype=np.float32)
    want = np.frombuffer(bytes(camera_from_project(project["camera"])), dtype=np.float32)
    # cos / sin of the half angle come from two libms: allow an ulp or two on view_plane, nothing elsewhere
    assert np.array_equal(cam[:16], want[:16]) and np.array_equal(cam[17:], want[17:])
    assert abs(cam[16] - want[16]) <= 4e-7 * abs(want[16])
    r = renderer_from_project(project["renderer"])
    params = np.frombuffer(rest[C.sizeof(abi.PyrCamera):], dtype=np.uint32)
    assert list(params[[0, 2, 3, 4, 5]]) == [r["bounces"], r["light_samples"], r["spectrum_samples"], r["spectrum_bins"], r["tile_size"]]


def test_missing_mesh_material_is_a_project_error(host, tmp_path):
    """world.rs:199-208: an OBJ object without a material entry is reported, not skipped."""
    obj = os.path.join(tmp_path, "cornell_box.obj")
    with open(os.path.join(DATA_DIR, "cornell_box.obj")) as f, open(obj, "w") as g:
        g.write(f.read() + "\no extra\nf 1 2 3\n")
    with open(os.path.join(DATA_DIR, "cornell_spectra.json")) as f, open(os.path.join(tmp_path, "cornell_spectra.json"), "w") as g:
        g.write(f.read())
    p = subprocess.run([HOST_TOOL, "dump", "c2", str(tmp_path), os.path.join(tmp_path, "x.bin")], capture_output=True, text=True)
    assert p.returncode == 1 and "missing material for 'extra'" in p.stderr


def test_png_writer_and_evaluate_at(host, tmp_path):
    """save_png output decodes to the pixels given (stored deflate blocks, CRCs, Adler-32)."""
    lib = host
    lib.pyrh_test_png.restype = C.c_int
    lib.pyrh_test_png.argtypes = [C.c_char_p, C.c_void_p, C.c_uint32, C.c_uint32]
    rng = np.random.default_rng(5)
    rgb = rng.integers(0, 256, (150, 301, 3), dtype=np.uint8)  # > 65535 bytes of scanlines: more than one stored block
    path = os.path.join(tmp_path, "x.png")
    assert lib.pyrh_test_png(path.encode(), rgb.ctypes.data, 301, 150) == 0
    assert np.array_equal(images.read_png(path), rgb)


@pytest.mark.gpu
@pytest.mark.parametrize("name,width,height,spp", [("c2", 64, 64, 8), ("lamps", 48, 32, 8), ("textures", 48, 32, 4)])
def test_cpp_renderer_seam_matches_the_oracle(host, name, width, height, spp, tmp_path):
    import oracle

    film_path, png_path = os.path.join(tmp_path, "film.bin"), os.path.join(tmp_path, "out.png")
    out = subprocess.check_output([HOST_TOOL, "render", name, data_dir_for(name, tmp_path), str(width), str(height), str(spp), "3", film_path, png_path], text=True)
    assert "The scene contains" in out and "100 %" in out
    project = SCENES[name](width, height, spp)
    world, cam, r, cpu_film = scenes.build(project, seed=3, base_dir=DATA_DIR)
    gpu = np.fromfile(film_path, dtype=np.float32).reshape(height, width, r.spectrum_bins, 2)
    oracle.OracleScene(world).render(r, cam, cpu_film, threads=4)
    assert np.array_equal(gpu[..., 1], cpu_film.grains[..., 1]), "film weights differ from the oracle"
    a = np.divide(gpu[..., 0], gpu[..., 1], out=np.zeros_like(gpu[..., 0]), where=gpu[..., 1] > 0)
    b = cpu_film.develop()
    e = np.sqrt(((a - b) ** 2).sum(-1)) / (np.sqrt((b ** 2).sum(-1)) + 1e-6)
    assert (e <= 1e-5).mean() >= 0.999, "per-pixel spectral relL2 vs oracle: max %.3g" % e.max()
    # the developed image written by the C++ side equals the Python front-end's development of the same film
    from pyrite_amd import develop
    from pyrite_amd.film import Film

    f = Film(width, height, r.spectrum_bins, r.spectrum_span)
    f.grains[...] = gpu
    assert np.array_equal(images.read_png(png_path), develop.develop(f))


@pytest.mark.skipif(not os.path.exists("/root/reference/pyrite/test/textures/color_checker.jpg"), reason="the reference checkout only exists in the build container")
def test_cpp_jpeg_textures_match_the_python_reader(host):
    host.pyrh_test_load_texture.restype = C.c_int64
    host.pyrh_test_load_texture.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    path = "/root/reference/pyrite/test/textures/color_checker.jpg"
    want = images.load_texture(path, False, False)
    buf = np.zeros(want.size, dtype=np.float32)
    w, h = C.c_uint32(), C.c_uint32()
    assert host.pyrh_test_load_texture(path.encode(), 0, 0, buf.ctypes.data, buf.size, C.byref(w), C.byref(h)) == want.size
    assert np.array_equal(buf.view(np.uint32), np.ascontiguousarray(want).reshape(-1).view(np.uint32))


# ------------------------------------------------------------------------------------------------ project files (lua_project.cpp)
PROJECTS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "projects")
REFERENCE_TESTS = "/root/reference/pyrite/test"


def flatten_with_python(path, texel_dir):
    """Python front-end: project file -> scene bytes; leaves the textures it decoded as texel files for the C++ side."""
    from pyrite_amd import lua_project

    project, base_dir = lua_project.load_project(path)
    flat = FlatScene().add_world(project["world"], base_dir)
    for (source, linear, mono), tid in flat._texture_ids.items():
        texels = np.ascontiguousarray(flat.textures[tid][1], dtype="<f4")
        name = "%s.%s.%s.f32" % (os.path.basename(source), "linear" if linear else "srgb", "mono" if mono else "color")
        with open(os.path.join(texel_dir, name), "wb") as f:
            f.write(np.array([texels.shape[1], texels.shape[0]], dtype="<u4").tobytes())
            f.write(texels.tobytes())
    return project, flat


def check_project_file(host, path, tmp_path, native_images=True):
    project, flat = flatten_with_python(path, str(tmp_path))
    expected = scene_bytes(host, flat.desc())
    out = os.path.join(tmp_path, "scene.bin")
    subprocess.check_call([HOST_TOOL, "dump-project", path, "-" if native_images else str(tmp_path), out], stdout=subprocess.DEVNULL)
    with open(out, "rb") as f:
        got = f.read()
    assert got[:len(expected)] == expected, "flattened scene differs"
    rest = got[len(expected):]
    cam = np.frombuffer(rest[:C.sizeof(abi.PyrCamera)], dtype=np.float32)
    want = np.frombuffer(bytes(camera_from_project(project["camera"])), dtype=np.float32)
    assert np.array_equal(cam[:16], want[:16]) and np.array_equal(cam[17:], want[17:]) and abs(cam[16] - want[16]) <= 4e-7 * abs(want[16])
    r = renderer_from_project(project["renderer"])
    params = np.frombuffer(rest[C.sizeof(abi.PyrCamera):], dtype=np.uint32)
    image = project.get("image") or {}
    assert list(params) == [r["bounces"], r["pixel_samples"], r["light_samples"], r["spectrum_samples"], r["spectrum_bins"], r["tile_size"],
                            image.get("width", 0), image.get("height", 0)]


@pytest.mark.parametrize("texels", ["decoded by the C++ side", "handed over as texel files"])
def test_cpp_project_file_reader_matches_the_python_one(host, tmp_path, texels):
    """gallery.lua: locals, require, :with{} chains, table / string / parenthesised calls, arithmetic on expressions, textures,
    a normal map, an OBJ mesh, every lamp kind. Its PNG textures are decoded natively (images.cpp) or come through the hook."""
    check_project_file(host, os.path.join(PROJECTS, "gallery.lua"), tmp_path, native_images=texels.startswith("decoded"))


@pytest.mark.parametrize("name", sorted(os.listdir(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "textures"))))
def test_cpp_image_decoder_matches_the_python_one(host, name, tmp_path):
    """PNG (inflate, filters, colour types) and baseline JPEG -> linear texels, bit for bit, for every texture fixture."""
    host.pyrh_test_load_texture.restype = C.c_int64
    host.pyrh_test_load_texture.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "textures", name)
    if os.path.splitext(name)[1].lower() not in (".png", ".jpg", ".jpeg"):
        pytest.skip("not an image")
    for linear, mono in ((False, False), (True, False), (False, True), (True, True)):
        want = images.load_texture(path, linear, mono)
        buf = np.zeros(want.size, dtype=np.float32)
        w, h = C.c_uint32(), C.c_uint32()
        n = host.pyrh_test_load_texture(path.encode(), int(linear), int(mono), buf.ctypes.data, buf.size, C.byref(w), C.byref(h))
        assert n == want.size and (h.value, w.value) == want.shape[:2]
        assert np.array_equal(buf.view(np.uint32), np.ascontiguousarray(want).reshape(-1).view(np.uint32))


def test_cpp_project_file_errors_name_file_and_line(host, tmp_path):
    def run(text):
        path = os.path.join(tmp_path, "p.lua")
        with open(path, "w") as f:
            f.write(text)
        p = subprocess.run([HOST_TOOL, "dump-project", path, "-", os.path.join(tmp_path, "x.bin")], capture_output=True, text=True)
        assert p.returncode == 1
        return p.stderr

    assert "p.lua:2: attempt to call a nil value (global 'sphere')" in run("local a = 1\nreturn sphere {radius = 1}")
    assert "function definitions are not supported" in run("local f = function(x) return x end return f(1)")
    assert "'for' is not supported" in run("for i = 1, 3 do end")
    assert "renderer.bidirectional is out of scope" in run(
        "return {camera = camera.perspective {fov = 40, transform = transform.look_at {from = vector(0, 0, 5), to = vector()}},"
        " renderer = renderer.bidirectional {pixel_samples = 1}, world = {objects = {}}}")
    assert "a.png: no such file" in run(
        "return {camera = camera.perspective {fov = 40, transform = transform.look_at {from = vector(0, 0, 5), to = vector()}},"
        " renderer = renderer.simple {pixel_samples = 1}, world = {objects = {shape.sphere {position = vector(), radius = 1,"
        " material = {surface = material.diffuse {color = texture 'a.png'}}}}}}")


@pytest.mark.skipif(not os.path.isdir(REFERENCE_TESTS), reason="the reference checkout only exists in the build container")
@pytest.mark.parametrize("name", ["spheres/spheres.lua", "diamonds/diamonds.lua", "textures/textures.lua", "colors/colors.lua", "rgb_emission/rgb_emission.lua",
                                  "rgb_reflection/rgb_reflection.lua"])
def test_cpp_reads_the_reference_project_files_like_the_python_reader(host, name, tmp_path):
    from pyrite_amd import lua_project
    from pyrite_amd.compiler import ProjectError

    path = os.path.join(REFERENCE_TESTS, name)
    try:
        project, _ = flatten_with_python(path, str(tmp_path))
        renderer_from_project(project["renderer"])
    except (ProjectError, lua_project.LuaError) as error:
        pytest.skip("the Python reader does not take this project either: %s" % error)
    check_project_file(host, path, tmp_path)


@pytest.mark.gpu
def test_cpp_renders_a_project_file(host, tmp_path):
    """`pyrite project.lua` through the C++ layer: project file -> film -> render.png; the film equals the oracle's."""
    import oracle
    from pyrite_amd import lua_project

    path = os.path.join(PROJECTS, "gallery.lua")
    project, _ = flatten_with_python(path, str(tmp_path))
    png, film_path = os.path.join(tmp_path, "render.png"), os.path.join(tmp_path, "film.bin")
    out = subprocess.check_output([HOST_TOOL, "render-project", path, str(tmp_path), "5", png, film_path], text=True)
    assert "The scene contains" in out and "Saving final result" in out
    project, base_dir = lua_project.load_project(path)
    world, cam, r, cpu_film = scenes.build(project, seed=5, base_dir=base_dir)
    gpu = np.fromfile(film_path, dtype=np.float32).reshape(cpu_film.grains.shape)
    oracle.OracleScene(world).render(r, cam, cpu_film, threads=4)
    assert np.array_equal(gpu[..., 1], cpu_film.grains[..., 1])
    a = np.divide(gpu[..., 0], gpu[..., 1], out=np.zeros_like(gpu[..., 0]), where=gpu[..., 1] > 0)
    b = cpu_film.develop()
    e = np.sqrt(((a - b) ** 2).sum(-1)) / (np.sqrt((b ** 2).sum(-1)) + 1e-6)
    assert (e <= 1e-5).mean() >= 0.999
    assert images.read_png(png).shape == (cpu_film.height, cpu_film.width, 3)


@pytest.mark.gpu
def test_cpp_renderer_seam_over_several_devices_matches_the_oracle(host, tmp_path):
    """Renderer::render(film, camera, world, devices) -> pyr_render_simple_multi, with one GPU standing in for three ranks
    (the blocks then travel by hipMemcpyPeerAsync; with distinct devices the same call gathers over RCCL)."""
    import oracle

    film_path = os.path.join(tmp_path, "film.bin")
    env = dict(os.environ, PYRITE_DEVICES="0,0,0")
    out = subprocess.check_output([HOST_TOOL, "render", "c2", data_dir_for("c2", tmp_path), "72", "56", "6", "3", film_path], text=True, env=env)
    assert "100 %" in out
    world, cam, r, cpu_film = scenes.build(SCENES["c2"](72, 56, 6), seed=3, base_dir=DATA_DIR)
    gpu = np.fromfile(film_path, dtype=np.float32).reshape(56, 72, r.spectrum_bins, 2)
    oracle.OracleScene(world).render(r, cam, cpu_film, threads=4)
    assert np.array_equal(gpu[..., 1], cpu_film.grains[..., 1]), "film weights differ from the oracle"
    assert np.allclose(gpu[..., 0], cpu_film.grains[..., 0], rtol=1e-5, atol=1e-9)


@pytest.mark.gpu
def test_cpp_world_intersect_matches_the_oracle(host, tmp_path):
    """World::intersect through the C++ layer: closest hits of a ray batch equal the oracle's."""
    import oracle
    from test_gpu_parity import assert_same_hits, random_rays

    rays = random_rays(20000, 3, [-5.5, 0.1, 0.1], [-0.1, 5.5, 5.4])
    rays_path, hits_path = os.path.join(tmp_path, "rays.f32"), os.path.join(tmp_path, "hits.bin")
    rays.astype("<f4").tofile(rays_path)
    subprocess.check_call([HOST_TOOL, "intersect", "c2", DATA_DIR, rays_path, hits_path], stdout=subprocess.DEVNULL)
    got = np.fromfile(hits_path, dtype=np.dtype([("distance", "<f4"), ("shape", "<u4"), ("u", "<f4"), ("v", "<f4")]))
    world, _, _, _ = scenes.build(scenes.c2_cornell(8, 8, 1), seed=1)
    want, _ = oracle.OracleScene(world).intersect(rays)
    assert_same_hits(want, got, world, rays)


@pytest.mark.gpu
def test_cpp_development_with_filter_and_white_balance(host, tmp_path):
    """image.filter / image.white (main.rs:190-238): the C++ layer evaluates the two programs at the sampling wavelengths and
    develops on the GPU; its PNG equals the Python front-end's development of the same film (cornell.lua's white = blackbody)."""
    from pyrite_amd import develop, lua_project
    from pyrite_amd.film import Film

    path = os.path.join(tmp_path, "balanced.lua")
    with open(path, "w") as f:
        f.write("""
local warm = spectrum {format = "curve", points = {{380, 0.2}, {500, 0.6}, {650, 1.0}, {780, 0.9}}}
return {
    image = {width = 40, height = 24, white = blackbody(4000), filter = warm * 0.9 + 0.05},
    renderer = renderer.simple {pixel_samples = 8, tile_size = 16},
    camera = camera.perspective {fov = 50, transform = transform.look_at {from = vector(0, 2, 8), to = vector(0, 1, 0)}},
    world = {sky = light_source.d65 * 0.5, objects = {
        shape.sphere {position = vector(0, -100, 0), radius = 100, material = {surface = material.diffuse {color = 0.6}}},
        shape.sphere {position = vector(0, 1, 0), radius = 1, material = {surface = material.diffuse {color = rgb(0.9, 0.4, 0.2)}}},
        shape.sphere {position = vector(2, 3, 2), radius = 0.5, material = {surface = material.emissive {color = light_source.a * 6}}},
    }},
}
""")
    png, film_path = os.path.join(tmp_path, "out.png"), os.path.join(tmp_path, "film.bin")
    subprocess.check_call([HOST_TOOL, "render-project", path, "-", "2", png, film_path], stdout=subprocess.DEVNULL)
    project, _ = lua_project.load_project(path)
    film = Film(40, 24, 64, (380.0, 780.0))
    film.grains[...] = np.fromfile(film_path, dtype=np.float32).reshape(film.grains.shape)
    want = develop.develop(film, filter=project["image"]["filter"], white=project["image"]["white"])
    got = images.read_png(png)
    assert got.shape == want.shape and np.abs(got.astype(int) - want.astype(int)).max() <= 1 and (got == want).mean() > 0.99
    assert got.std() > 5  # an image, not a constant


@pytest.mark.timeout(900)
def test_host_cpp_and_oracle_under_sanitizers(tmp_path):
    """the host-side C++ that parses untrusted input (project files, OBJ meshes, image files), the tree
    builder and the CPU oracle in ONE executable built with -fsanitize=address,undefined (tests/host_asan_driver.cpp), no GPU:
    the suite's own project files load, flatten, build their trees and render through the oracle; mutated and truncated copies of
    them, binary junk, a mesh with junk in it and a truncated PNG are refused with a message -- no out-of-bounds access, overflow
    or leak either way."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    csrc = os.path.join(root, "pyrite_amd", "csrc")
    exe = str(tmp_path / "host_asan")
    san = ["-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-ffp-contract=off"]
    subprocess.check_call(["gcc", "-c", *san, os.path.join(csrc, "jpeg.c"), "-o", str(tmp_path / "jpeg.o")])
    subprocess.check_call(["g++", "-std=c++17", *san, "-I" + os.path.join(root, "include"), "-o", exe, os.path.join(root, "tests", "host_asan_driver.cpp"),
                           os.path.join(csrc, "host", "pyrite_host.cpp"), os.path.join(csrc, "host", "lua_project.cpp"), os.path.join(csrc, "host", "images.cpp"),
                           str(tmp_path / "jpeg.o"), os.path.join(csrc, "bvh.cpp"), os.path.join(root, "oracle", "oracle.cpp"),
                           "-L" + csrc, "-lpyrite_gpu", "-Wl,-rpath," + csrc, "-pthread"])
    projects = os.path.join(root, "tests", "golden", "projects")
    good = [os.path.join(projects, n) for n in sorted(os.listdir(projects)) if n.endswith(".lua") and n != "materials.lua"]  # a module gallery.lua requires, not a project
    assert len(good) >= 1
    # leak detection stays off: the sanitized executable links libpyrite_gpu.so for the symbols World::from_project names (never
    # called here), and the HIP runtime it pulls in keeps allocations of its own alive at exit
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1", LD_PRELOAD="")
    run = subprocess.run([exe] + good, capture_output=True, text=True, env=env)
    assert run.returncode == 0, run.stderr[-4000:]
    assert run.stdout.split()[0] == str(len(good)) and "0 refused" in run.stdout, run.stdout + run.stderr[-2000:]

    rng = np.random.default_rng(7)
    hostile = []
    for k, path in enumerate(good):
        text = open(path, "rb").read()
        d = tmp_path / ("case%d" % k)
        d.mkdir()
        for name in os.listdir(projects):  # meshes / images the project names, next to every mutated copy
            if not name.endswith(".lua"):
                os.symlink(os.path.join(projects, name), d / name)
        for j, cut in enumerate((0, 1, len(text) // 3, len(text) // 2, len(text) - 2)):
            p = d / ("cut%d.lua" % j)
            p.write_bytes(text[:cut])
            hostile.append(str(p))
        for j in range(12):  # a few bytes overwritten: unbalanced brackets, broken numbers, stray operators
            b = bytearray(text)
            for pos in rng.integers(0, len(b), 4):
                b[pos] = int(rng.choice(list(b"{}()[],.=\"'-0179e+\x00\xff \n")))
            p = d / ("mut%d.lua" % j)
            p.write_bytes(bytes(b))
            hostile.append(str(p))
    junk = tmp_path / "junk.lua"
    junk.write_bytes(rng.integers(0, 256, 4096, dtype=np.uint8).tobytes())
    deep = tmp_path / "deep.lua"
    deep.write_text("return " + "{" * 5000 + "}" * 5000)
    mesh_dir = tmp_path / "mesh"
    mesh_dir.mkdir()
    (mesh_dir / "bad.obj").write_text("o thing\nv 0 0 0\nv 1 nan 0\nv 1e999 1 0\nvn 0 0\nvt 1\nf 1/9/9 2 3\nf 1 2 99999999999\nf -5 2 3\nf\n")
    (mesh_dir / "cut.png").write_bytes(b"\x89PNG\r\n\x1a\n" + b"\x00\x00\x00\rIHDR" + b"\x00" * 6)
    scene = ("return {camera = camera.perspective {fov = 40, transform = transform.look_at {from = vector(0, -5, 1), to = vector()}},"
             " renderer = renderer.simple {pixel_samples = 1}, world = {objects = {%s}}}")
    (mesh_dir / "mesh.lua").write_text(scene % "shape.mesh {file = 'bad.obj', materials = {thing = {surface = material.diffuse {color = 0.5}}}}")
    (mesh_dir / "tex.lua").write_text(scene % "shape.sphere {position = vector(), radius = 1, material = {surface = material.diffuse {color = texture 'cut.png'}}}")
    hostile += [str(junk), str(deep), str(mesh_dir / "mesh.lua"), str(mesh_dir / "tex.lua"), str(tmp_path / "does_not_exist.lua")]
    run = subprocess.run([exe] + hostile, capture_output=True, env=env)
    assert run.returncode == 0, run.stderr[-4000:].decode("utf-8", "replace")
    assert b"deep.lua: " in run.stderr and b"too many syntax levels" in run.stderr  # 5000 nested tables once ran the parser off the stack
    loaded, refused = (int(x) for x in run.stdout.decode().replace(",", "").split() if x.isdigit())
    assert loaded + refused == len(hostile) and refused >= len(hostile) // 2, run.stdout
