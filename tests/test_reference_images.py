"""The only outputs of the reference that exist for the simple renderer are the example images it keeps next to two test
projects (tests/golden/make_reference_fixtures.py). The oracle renders the same projects -- scenes.spheres_example and
scenes.diamonds_example restate pyrite/test/spheres/spheres.lua and pyrite/test/diamonds/diamonds.lua -- and must agree
with them in luminance level and structure. The pin is weak by nature (8-bit images of an earlier build, independent
noise, colour rendition of the development step drifted), so the tolerances are loose; what it does rule out is a wrong
radiometric constant, a wrong camera, a wrong BSDF weight or a wrong lamp term anywhere on the path."""
import os

import numpy as np
import pytest

import oracle
from pyrite_amd import images, scenes

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_example_images.npz")
LUMA = np.array([0.2126, 0.7152, 0.0722])


def oracle_block_means(project, block, threads=8):
    world, cam, r, film = scenes.build(project, seed=1)
    oracle.OracleScene(world).render(r, cam, film, threads=threads)
    rgb = oracle.film_develop(film)  # main.rs:315-418 restated
    lin = images.srgb_to_linear(rgb.astype(np.float64) / 255.0).astype(np.float64)
    h, w = lin.shape[0] // block * block, lin.shape[1] // block * block
    return lin[:h, :w].reshape(h // block, block, w // block, block, 3).mean((1, 3))


def test_spheres_example_matches_the_reference_image():
    data = np.load(GOLDEN)
    ref = data["spheres"].astype(np.float64)  # 8 x 8 block means of the 512 x 256 image -> 32 x 64 cells
    mine = oracle_block_means(scenes.spheres_example(256, 128, 48), 4)  # half size, 4 x 4 blocks: the same cells
    assert mine.shape == ref.shape == (32, 64, 3)
    yr, ym = ref @ LUMA, mine @ LUMA
    assert np.corrcoef(yr.ravel(), ym.ravel())[0, 1] > 0.99
    assert (yr[2:22, 24:40] > 0.95).mean() > 0.8 and (ym[2:22, 24:40] > 0.95).mean() > 0.8  # the lamp saturates in both
    assert yr[0:2].max() < 0.01 and ym[0:2].max() < 0.01  # black above the horizon
    floor = (slice(27, 32), slice(4, 60))
    ratio = ym[floor] / yr[floor]
    # 0.90 -- a difference of colour rendition, not of light transport: see test_gpu_renders_match_the_reference_images
    assert 0.87 < np.median(ratio) < 0.94, np.median(ratio)
    # the left ball is the red / orange one, the right ball the green one, in both
    for img in (ref, mine):
        left, right = img[8:18, 4:12].mean((0, 1)), img[8:18, 52:60].mean((0, 1))
        assert left[0] > 2 * left[2] and left[0] > left[1]
        assert right[1] > right[0] and right[1] > right[2]


@pytest.mark.timeout(600)
def test_diamonds_example_matches_the_reference_image():
    """Dispersive glass (ior 2.37782 + 0.01371 / lambda^2), 256 bounces, thin lens, a fresnel-mixed mirror floor, two quad
    lamps, one wavelength per sample: with the project's own 200 spp the oracle's image has 0.96x the reference image's
    mean luminance and correlates 0.999 with it (at few spp most of a pixel's 50 bins are empty and develop to zero, which
    is why the comparison needs the full sample count)."""
    data = np.load(GOLDEN)
    ref = data["diamonds"].astype(np.float64)  # 37 x 64 cells of 8 x 8 pixels
    mine = oracle_block_means(scenes.diamonds_example(256, 150, 200, bounces=256), 4)[:37]
    assert mine.shape == ref.shape == (37, 64, 3)
    yr, ym = ref @ LUMA, mine @ LUMA
    assert np.corrcoef(yr.ravel(), ym.ravel())[0, 1] > 0.985
    assert 0.85 < ym.mean() / yr.mean() < 1.1, ym.mean() / yr.mean()
    cells = (yr > 0.01) & (yr < 0.9)
    assert 0.8 < np.median(ym[cells] / yr[cells]) < 1.15


TEXTURES = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "textures")
# colour-checker patches (8 x 8 pixel cells of the 1024 x 512 image: row, column) and the cube, whose textures are missing
PATCHES = {"brown": (19, 65), "orange": (25, 65), "blue": (31, 65), "white": (38, 65), "green": (31, 71), "red": (31, 77), "yellow": (31, 83),
           "cyan": (31, 98)}
CUBE = (slice(40, 58), slice(76, 98))


def check_textures_image(mine, ref):
    yr, ym = ref @ LUMA, mine @ LUMA
    mask = np.ones_like(yr, dtype=bool)
    mask[CUBE] = False
    assert np.corrcoef(yr[mask], ym[mask])[0, 1] > 0.99  # measured 0.997
    cells = mask & (yr > 0.02) & (yr < 0.9)
    assert 0.9 < np.median(ym[cells] / yr[cells]) < 1.1  # measured 0.99
    for name, (y, x) in PATCHES.items():  # texture lookup, sRGB decoding, RGB -> spectrum, lamps, development: all in one number
        assert np.allclose(mine[y, x], ref[y, x], rtol=0.15, atol=0.02), (name, mine[y, x], ref[y, x])


def test_textures_example_matches_the_reference_image():
    """pyrite/test/textures: colour textures on a quad, a plane and a sphere, normal maps on plane and sphere, a fresnel mix
    of mirror and textured diffuse. This image was rendered by the current development code, so colours are comparable:
    the colour-checker patches come out within a few percent per channel (brown 0.124 0.044 0.022 vs 0.124 0.043 0.022)."""
    data = np.load(GOLDEN)
    ref = data["textures"].astype(np.float64)  # 64 x 128 cells
    mine = oracle_block_means(scenes.textures_reference_example(TEXTURES, 512, 256, 48), 4)
    assert mine.shape == ref.shape == (64, 128, 3)
    check_textures_image(mine, ref)


@pytest.mark.gpu
def test_gpu_renders_match_the_reference_images(gpu_lib):
    """The HIP path at the projects' own sizes and sample counts (512 x 256 x 600 spp, 512 x 300 x 200 spp x 256 bounces),
    developed on the GPU (pyr_film_develop), against the reference's example images."""
    from pyrite_amd import develop

    data = np.load(GOLDEN)
    for name, project, floor in (("spheres", scenes.spheres_example(512, 256, 600), (slice(27, 32), slice(4, 60))),
                                 ("diamonds", scenes.diamonds_example(512, 300, 200, bounces=256), None)):
        world, cam, r, film = scenes.build(project, seed=1)
        r.render(film, cam, world)
        rgb = develop.develop(film)
        lin = images.srgb_to_linear(rgb.astype(np.float64) / 255.0).astype(np.float64)
        ref = data[name].astype(np.float64)
        h, w = ref.shape[0] * 8, ref.shape[1] * 8
        mine = lin[:h, :w].reshape(h // 8, 8, w // 8, 8, 3).mean((1, 3))
        yr, ym = ref @ LUMA, mine @ LUMA
        assert np.corrcoef(yr.ravel(), ym.ravel())[0, 1] > 0.99, name
        if floor is not None:
            # The white floor (albedo 1 under the D65 lamp) comes out at 0.90x the image's luminance. tools/spheres_image_study.py
            # (profiles/r02_spheres_image_study.txt) took that number apart at the project's own 600 spp, where two seeds differ by
            # 0.3 % per cell: it does not move with bounces (8 / 16 / 32), light samples (4 / 1), spectrum samples (10 / 1) or
            # bins (64 / the lua's ignored `spectrum_bins = 50`), so it is not light transport; it differs per channel
            # (R 0.96, G 0.90, B 0.87: the image's floor is neutral, today's development renders a D65-lit white slightly warm,
            # as the reference's CURRENT code does too -- the textures image, rendered by it, matches per channel within a few
            # percent) and saturated colours differ most (the red ball's green channel is 0.07x: the image is less saturated).
            # That is the spectrum -> RGB step of an earlier build. The windows are what the noise floor supports.
            assert 0.88 < np.median(ym[floor] / yr[floor]) < 0.93
            per_channel = np.median(mine[floor] / ref[floor], axis=(0, 1))
            assert np.all(np.abs(per_channel - np.array([0.953, 0.889, 0.868])) < 0.025), per_channel
        else:
            assert 0.85 < ym.mean() / yr.mean() < 1.1
    world, cam, r, film = scenes.build(scenes.textures_reference_example(TEXTURES, 1024, 512, 400), seed=1)
    r.render(film, cam, world)
    lin = images.srgb_to_linear(develop.develop(film).astype(np.float64) / 255.0).astype(np.float64)
    check_textures_image(lin.reshape(64, 8, 128, 8, 3).mean((1, 3)), data["textures"].astype(np.float64))
