"""Next row f4: textures and normal maps -- image ingest, the bicubic lookup, tangent frames, the texture opcodes and
normal-map programs in the compiler and in the oracle. (GPU parity for the same scene is in test_gpu_parity.py.)"""
import numpy as np
import pytest

import oracle
from pyrite_amd import abi, compiler, develop, images, scenes
from pyrite_amd.project import material, rgb, shape, texture, vector

f32 = np.float32


# ---------------------------------------------------------------------------------------------- image ingest
def test_png_reader_round_trips_the_writer(tmp_path):
    rng = np.random.default_rng(5)
    img = rng.integers(0, 256, (7, 11, 3), dtype=np.uint8)
    path = str(tmp_path / "t.png")
    develop.save_png(path, img)
    assert np.array_equal(images.read_png(path), img)


def test_png_reader_handles_every_filter_type_and_gray_alpha(tmp_path):
    import struct
    import zlib

    rng = np.random.default_rng(6)
    h, w = 5, 6
    for color_type, channels in ((0, 1), (4, 2), (2, 3), (6, 4)):
        img = rng.integers(0, 256, (h, w, channels), dtype=np.uint8)
        bpp = channels
        raw = bytearray()
        prev = np.zeros(w * channels, dtype=np.int32)
        for y in range(h):
            line = img[y].reshape(-1).astype(np.int32)
            ftype = y % 5
            out = np.zeros_like(line)
            for i in range(len(line)):
                a = line[i - bpp] if i >= bpp else 0
                b = prev[i]
                c = prev[i - bpp] if i >= bpp else 0
                if ftype == 0:
                    pred = 0
                elif ftype == 1:
                    pred = a
                elif ftype == 2:
                    pred = b
                elif ftype == 3:
                    pred = (a + b) >> 1
                else:
                    p = a + b - c
                    pa, pb, pc = abs(p - a), abs(p - b), abs(p - c)
                    pred = a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)
                out[i] = (line[i] - pred) & 0xFF
            raw.append(ftype)
            raw.extend(out.astype(np.uint8).tobytes())
            prev = line

        def chunk(kind, body):
            return struct.pack(">I", len(body)) + kind + body + struct.pack(">I", zlib.crc32(kind + body) & 0xFFFFFFFF)

        data = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, color_type, 0, 0, 0)) + chunk(b"IDAT", zlib.compress(bytes(raw))) + chunk(b"IEND", b"")
        path = tmp_path / ("f%d.png" % color_type)
        path.write_bytes(data)
        assert np.array_equal(images.read_png(str(path)), img), color_type


def test_linearisation_follows_the_srgb_transfer_function():
    ramp = np.arange(256, dtype=np.uint8).reshape(1, 256, 1)
    lin = images.linearise(ramp, linear=False, mono=True)[0]
    assert lin[0] == 0.0 and lin[255] == 1.0 and np.all(np.diff(lin) > 0)
    assert lin[10] == f32(10 / 255 / 12.92)  # the linear toe
    assert abs(float(lin[128]) - ((128 / 255 + 0.055) / 1.055) ** 2.4) < 1e-7
    assert np.array_equal(images.linearise(ramp, linear=True, mono=True)[0], (ramp[0, :, 0].astype(f32) / f32(255)))
    # colour image -> LinSrgba with alpha 1; gray -> r = g = b; rgba keeps its alpha linear; 16-bit is /65535
    rgbimg = np.array([[[255, 0, 0], [0, 255, 0]]], dtype=np.uint8)
    col = images.linearise(rgbimg, False, False)
    assert col.shape == (1, 2, 4) and np.array_equal(col[0, 0], [1, 0, 0, 1]) and np.array_equal(col[0, 1], [0, 1, 0, 1])
    mono = images.linearise(rgbimg, False, True)
    assert mono.shape == (1, 2) and np.allclose(mono[0], images.LUMA_WEIGHTS[:2])
    rgba = images.linearise(np.array([[[255, 255, 255, 51]]], dtype=np.uint8), False, False)
    assert rgba[0, 0, 3] == f32(51) / f32(255)
    assert images.linearise(np.array([[65535, 0]], dtype=np.uint16), True, True)[0, 0] == 1.0


# ---------------------------------------------------------------------------------------------- bicubic lookup
def test_texture_lookup_at_texel_centres_wraps_and_interpolates():
    rng = np.random.default_rng(2)
    tex = rng.random((4, 8)).astype(f32)  # [h, w], row 0 is the top of the image
    h, w = tex.shape
    for j in range(h):
        for i in range(w):
            x, y = (i + 0.5) / w, 1.0 - (j + 0.5) / h  # texture.rs:95, :104: y runs upwards
            assert oracle.texture_get(tex, x, y)[0] == tex[j, i]
    # one period further out in both directions is the same texel (rem_euclid wrap-around)
    assert oracle.texture_get(tex, (3 + 0.5) / w + 1.0, 1.0 - 0.5 / h - 2.0)[0] == tex[0, 3]
    assert oracle.texture_get(tex, (0.5) / w - 1.0, 1.0 - 1.5 / h)[0] == tex[1, 0]
    # a linear ramp along x: texture.rs:322-334's cubic through (0, 1, 2, 3) at 0.5 is 1.5
    ramp = np.tile(np.arange(8, dtype=f32), (4, 1))
    assert oracle.texture_get(ramp, 3.0 / 8, 0.5)[0] == 2.5 - 0.0  # x = 3/8 * 8 - 0.5 = 2.5: between texels 2 and 3
    # hand-evaluated cubic: v = (1, 2, 4, 8), pos = 0.25 -> a = 5, b = -6, c = 3, d = 2 -> 2 + (3 + (-6 + 1.25) * .25) * .25
    row = np.tile(np.array([1, 2, 4, 8, 0, 0, 0, 0], dtype=f32), (4, 1))
    expect = f32(2) + (f32(3) + (f32(-6) + f32(5) * f32(0.25)) * f32(0.25)) * f32(0.25)
    assert oracle.texture_get(row, (1.25 + 0.5) / 8, 0.5)[0] == expect
    # colour textures interpolate every channel
    col = rng.random((4, 4, 4)).astype(f32)
    got = oracle.texture_get(col, 0.3, 0.6)
    for ch in range(4):
        assert got[ch] == oracle.texture_get(np.ascontiguousarray(col[..., ch]), 0.3, 0.6)[0]


# ---------------------------------------------------------------------------------------------- tangent frames
def test_quaternion_from_an_orthonormal_basis_rotates_the_axes_onto_it():
    assert np.array_equal(oracle.quat_from_cols([1, 0, 0], [0, 1, 0], [0, 0, 1]), [1, 0, 0, 0])
    rng = np.random.default_rng(9)
    for _ in range(20):  # all four branches of the conversion (trace >= 0 and the three largest-diagonal cases)
        a = rng.normal(size=3)
        a /= np.linalg.norm(a)
        b = np.cross(a, rng.normal(size=3))
        b /= np.linalg.norm(b)
        c = np.cross(a, b)
        q = oracle.quat_from_cols(a, b, c)
        assert abs(np.linalg.norm(q) - 1) < 1e-5
        for axis, col in zip(np.eye(3), (a, b, c)):
            assert np.allclose(oracle.quat_rotate(q, axis), col, atol=2e-6)
        assert np.allclose(compiler._quat_from_cols(a.astype(f32), b.astype(f32), c.astype(f32)), q, atol=1e-6)


def _flat(objects, sky=0.0):
    flat = compiler.FlatScene()
    flat.add_world({"sky": sky, "objects": objects})
    return flat


def test_triangle_frames_and_texture_coordinates():
    from pyrite_amd.renderer import World

    tex = scenes._generated_textures()
    quad = {"position": np.array([[0, 0, 0], [2, 0, 0], [2, 2, 0], [0, 2, 0]], dtype=f32),
            "texture": np.array([[0, 0], [1, 0], [1, 1], [0, 1]], dtype=f32), "normal": np.zeros((0, 3), dtype=f32),
            "objects": [{"name": "q", "polys": [[(0, 0, None), (1, 1, None), (2, 2, None)], [(0, 0, None), (2, 2, None), (3, 3, None)]]}]}
    flat = _flat([shape.mesh(file=quad, materials={"q": {"surface": material.diffuse(color=texture(tex["checker"])), "normal_map": vector(0, 0, 1)}})])
    # u along +x, v along +y, flat normal +z: tangent space == world space, the frame is the identity rotation
    assert np.allclose(np.concatenate(flat.tri_frames).reshape(-1, 4), [1, 0, 0, 0], atol=1e-6)
    osc = oracle.OracleScene(World(flat))
    n, t, frame, shading = osc.surface_data([1.5, 0.5, 3, 0, 0, -1])
    assert np.allclose(n, [0, 0, 1]) and np.allclose(t, [0.75, 0.25], atol=1e-6) and np.allclose(frame, [1, 0, 0, 0], atol=1e-6)
    assert np.allclose(shading, [0, 0, 1], atol=1e-6)  # the constant map (0, 0, 1) leaves the normal alone
    # a normal map that leans towards +tangent tilts the shading normal towards +x
    flat2 = _flat([shape.mesh(file=quad, materials={"q": {"surface": material.diffuse(color=0.5), "normal_map": vector(1, 0, 1)}})])
    _, _, _, shading2 = oracle.OracleScene(World(flat2)).surface_data([1.5, 0.5, 3, 0, 0, -1])
    assert np.allclose(shading2, np.array([1, 0, 1]) / np.sqrt(2), atol=1e-6)


def test_plane_and_sphere_texture_coordinates():
    from pyrite_amd.renderer import World

    flat = _flat([shape.plane(origin=vector(0, 0, 0), normal=vector(z=1), texture_scale=vector(2, 4), material={"surface": material.diffuse(color=0.5)}),
                  shape.sphere(position=vector(0, 0, 5), radius=1, texture_scale=vector(0.5, 0.25), material={"surface": material.diffuse(color=0.5)})])
    osc = oracle.OracleScene(World(flat))
    n, t, frame, _ = osc.surface_data([3, 5, 2, 0, 0, -1])  # the plane under (3, 5)
    # shapes/mod.rs:454-468: the hit position expressed in the plane's tangent frame, divided by the scale
    local = oracle.quat_rotate(frame * np.array([1, -1, -1, -1], dtype=f32), [3, 5, 0])
    assert np.allclose(n, [0, 0, 1]) and np.allclose(t, [local[0] / 2, local[1] / 4], atol=1e-6) and abs(local[2]) < 1e-6
    assert abs(np.hypot(*local[:2]) - np.hypot(3, 5)) < 1e-5
    n, t, _, _ = osc.surface_data([0, -9, 5, 0, 1, 0])  # the sphere's equator, point (0, -1, 5): normal (0, -1, 0)
    lat, lon = np.arccos(n[1]), np.arctan2(n[0], n[2])
    assert np.allclose(n, [0, -1, 0], atol=1e-6) and np.allclose(t, [lon / np.pi * 0.5 / 0.5, (1 - lat / np.pi) / 0.25], atol=1e-5)


# ---------------------------------------------------------------------------------------------- compiler + VM
def test_texture_expressions_compile_to_the_texture_opcodes():
    tex = scenes._generated_textures()
    flat = compiler.FlatScene()
    colour = flat.compile(texture(tex["checker"]))
    prog = flat.programs[colour]
    ops = [i["op"] for i in flat.instrs[prog["first"]:prog["first"] + prog["n"]]]
    assert ops == [abi.OP_COLOR_TEXTURE, abi.OP_RGB_SPECTRUM]  # an RGB value used as a number goes through the RGB basis
    first = flat.instrs[prog["first"]]
    assert first["b"] == abi.INPUT_TEXTURE and first["deps"] == abi.DEP_TEXTURE and first["a"] == 0
    mono = flat.compile(texture(tex["mono"], "mono", "linear") * 2)
    ops = [i["op"] for i in flat.instrs[flat.programs[mono]["first"]:flat.programs[mono]["first"] + flat.programs[mono]["n"]]]
    assert abi.OP_MONO_TEXTURE in ops and abi.OP_BINARY in ops
    assert flat.compile(texture(tex["checker"])) != colour and len(flat.textures) == 2  # same image, same kind -> same texture id
    nm = flat.compile(texture(tex["normal_map"], "linear"), allow_wavelength=False, output="vector")
    p = flat.programs[nm]
    assert p["output_kind"] == abi.OUTPUT_VECTOR
    assert [i["op"] for i in flat.instrs[p["first"]:p["first"] + p["n"]]] == [abi.OP_COLOR_TEXTURE, abi.OP_RGB_TO_VECTOR]
    with pytest.raises(compiler.ProjectError):  # NormalInput has no wavelength (tracer.rs:58-68)
        from pyrite_amd.project import light_source

        flat.compile(light_source.d65 * texture(tex["checker"]), allow_wavelength=False, output="vector")
    d = flat.desc()
    assert d.num_textures == 3 and d.textures[0].format == abi.TEXTURE_COLOR and d.textures[1].format == abi.TEXTURE_MONO
    assert d.textures[1].offset == 16 * 16 * 4 and d.num_texture_floats == 16 * 16 * (4 + 1 + 4)


def test_a_constant_texture_renders_exactly_like_the_constant_colour():
    """Bicubic interpolation of a constant field returns the constant exactly, and a texture feeds the same RgbSpectrumValue
    an rgb() literal does: the two films must be identical bit for bit."""
    from pyrite_amd.renderer import Camera, Renderer, World

    flat_colour = np.full((4, 4, 3), 0.5, dtype=f32)
    films = []
    for colour in (texture(flat_colour, "linear"), rgb(0.5, 0.5, 0.5)):
        project = scenes.lamps_example(32, 24, 4)
        project["world"]["objects"][0] = shape.plane(origin=vector(0, 0, 0), normal=vector(z=1), material={"surface": material.diffuse(color=colour)})
        world, cam, r, film = scenes.build(project, seed=4)
        oracle.OracleScene(world).render(r, cam, film, threads=4)
        films.append(film.grains.copy())
    assert np.array_equal(films[0], films[1]) and films[0][..., 0].sum() > 0


def test_textured_scene_renders_and_is_deterministic():
    world, cam, r, film = scenes.build(scenes.textures_example(48, 32, 4), seed=2)
    osc = oracle.OracleScene(world)
    c1 = osc.render(r, cam, film, threads=4)
    film2 = r.new_film(48, 32)
    c2 = osc.render(r, cam, film2, threads=1)
    assert c1 == c2 and np.array_equal(film.grains[..., 1], film2.grains[..., 1])
    assert np.isfinite(film.grains).all() and film.grains[..., 0].sum() > 0
    assert np.allclose(film.grains[..., 0], film2.grains[..., 0], rtol=1e-4, atol=1e-6)  # only the float-add order differs


def _tiny_jpeg(levels, blocks_x):
    """A baseline JPEG written by hand: one component, all-ones quantisation, flat 8 x 8 blocks (DC coefficients only), a
    4-bit code for each of the 12 DC categories and a 1-bit end-of-block code."""
    import struct

    def segment(marker, body):
        return b"\xff" + bytes([marker]) + struct.pack(">H", len(body) + 2) + body

    blocks_y = len(levels) // blocks_x
    out = b"\xff\xd8"
    out += segment(0xDB, bytes([0]) + bytes([1] * 64))
    out += segment(0xC0, struct.pack(">BHHB", 8, blocks_y * 8, blocks_x * 8, 1) + bytes([1, 0x11, 0]))
    out += segment(0xC4, bytes([0x00]) + bytes([0, 0, 0, 12] + [0] * 12) + bytes(range(12)))  # DC: categories 0..11, 4 bits each
    out += segment(0xC4, bytes([0x10]) + bytes([1] + [0] * 15) + bytes([0x00]))  # AC: only EOB, code "0"
    out += segment(0xDA, bytes([1, 1, 0x00, 0, 63, 0]))
    bits, pred = "", 0
    for level in levels:
        dc = 8 * (level - 128)  # F(0,0) of a flat block
        diff, pred = dc - pred, dc
        cat = abs(diff).bit_length()
        bits += format(cat, "04b")
        if cat:
            bits += format(diff if diff > 0 else diff + (1 << cat) - 1, "0%db" % cat)
        bits += "0"
    bits += "1" * (-len(bits) % 8)
    data = bytes(int(bits[i:i + 8], 2) for i in range(0, len(bits), 8)).replace(b"\xff", b"\xff\x00")
    return out + data + b"\xff\xd9"


def test_jpeg_reader_survives_truncated_and_corrupt_files(tmp_path):
    """Texture files are untrusted: every truncation of a valid file, a few thousand byte flips and a set of hostile headers go
    through the reader built with AddressSanitizer + UBSan (CPU build; GPU sanitizers are not available). Nothing may read
    or write out of bounds, shift by a bad count or leak; a refused file must come with a message."""
    import os
    import struct
    import subprocess

    def segment(marker, body):
        return b"\xff" + bytes([marker]) + struct.pack(">H", len(body) + 2) + body

    good = _tiny_jpeg([16, 128, 255, 0, 77, 200, 3, 90], 4)
    cases = [good]
    cases += [good[:n] for n in range(len(good))]  # every truncation
    rng = np.random.default_rng(9)
    for _ in range(4000):  # byte flips, mostly in the headers
        b = bytearray(good)
        for _ in range(int(rng.integers(1, 4))):
            b[int(rng.integers(2, len(b)))] = int(rng.integers(0, 256))
        cases.append(bytes(b))
    soi = b"\xff\xd8"
    sof = lambda w, h, nc, comps: segment(0xC0, struct.pack(">BHHB", 8, h, w, nc) + comps)  # noqa: E731
    cases += [
        soi + segment(0xDB, bytes([0x10]) + bytes(20)),  # 16-bit table that stops early
        soi + segment(0xC4, bytes([0x00]) + bytes([255] * 16)),  # Huffman counts far beyond the segment
        soi + segment(0xC4, bytes([0x00]) + bytes([0] * 15 + [200]) + bytes(10)),  # 200 values announced, 10 present
        soi + sof(65535, 65535, 3, bytes([1, 0x22, 0, 2, 0x11, 1, 3, 0x11, 1])),  # 12.8 GB image
        soi + sof(16, 16, 3, bytes([1, 0x22, 0])),  # three components announced, one described
        soi + sof(16, 16, 1, bytes([1, 0x11, 9])),  # quantisation table 9
        soi + sof(16, 16, 1, bytes([1, 0x11, 0])) + segment(0xDA, bytes([4, 1, 0, 2, 0, 3, 0, 4, 0])),  # scan with 4 components
        soi + sof(16, 16, 3, bytes([1, 0x11, 0, 2, 0x11, 0, 3, 0x11, 0])) + segment(0xDA, bytes([1, 1, 0x00, 0, 63, 0])),  # non-interleaved scan
        soi + sof(16, 16, 1, bytes([1, 0x11, 0])) + segment(0xDA, bytes([1, 1, 0x77, 0, 63, 0])),  # Huffman table 7
        soi + b"\xff\xdb\x00\x01",  # segment length below 2
        soi + b"\xff\xd0" * 40,  # restart markers where segments belong
        soi + segment(0xDD, b""),  # empty restart-interval segment
    ]
    # a DC category of 15 (shift counts beyond baseline's 11 bits) with a table that allows it
    hostile = good.replace(bytes([0, 0, 0, 12] + [0] * 12) + bytes(range(12)), bytes([0, 0, 0, 12] + [0] * 12) + bytes([15] * 12))
    cases.append(hostile)
    corpus = tmp_path / "corpus.bin"
    with open(corpus, "wb") as f:
        for c in cases:
            f.write(struct.pack("<I", len(c)) + c)
    exe = tmp_path / "jpeg_asan"
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.check_call(["gcc", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-o", str(exe),
                           os.path.join(root, "pyrite_amd", "csrc", "jpeg.c"), os.path.join(root, "tests", "jpeg_asan_driver.c"), "-lm"])
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", LD_PRELOAD="")
    run = subprocess.run([str(exe), str(corpus)], capture_output=True, text=True, env=env)
    assert run.returncode == 0, run.stderr[-3000:]
    decoded, refused = (int(x) for x in run.stdout.replace(",", "").split() if x.isdigit())
    assert decoded >= 1 and refused >= len(good) - 40 and decoded + refused == len(cases)
    for n in (0, 1, 3, 40, len(good) // 2):  # and the Python surface reports them as errors
        bad = tmp_path / ("cut%d.jpg" % n)
        bad.write_bytes(good[:n])
        with pytest.raises(ValueError):
            images.read_image(str(bad))


def test_jpeg_reader_decodes_a_hand_written_baseline_file(tmp_path):
    levels = [16, 128, 255, 0, 77, 200]
    path = tmp_path / "flat.jpg"
    path.write_bytes(_tiny_jpeg(levels, 3))
    img = images.read_image(str(path))
    assert img.shape == (16, 24, 3) and img.dtype == np.uint8
    for k, level in enumerate(levels):
        block = img[(k // 3) * 8:(k // 3) * 8 + 8, (k % 3) * 8:(k % 3) * 8 + 8]
        assert np.all(block == level), (k, level, block[0, 0])
    bad = tmp_path / "bad.jpg"
    bad.write_bytes(b"not a jpeg at all")
    with pytest.raises(ValueError, match="not a JPEG"):
        images.read_image(str(bad))
    with pytest.raises(ValueError, match="unsupported image format"):
        images.read_image(str(tmp_path / "x.tiff"))
