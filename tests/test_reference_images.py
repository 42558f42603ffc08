"""The only outputs of the reference that exist for the simple renderer are the example images it keeps next to two test
projects (tests/golden/make_reference_fixtures.py). The oracle renders the same projects -- scenes.spheres_example and
scenes.diamonds_example restate pyrite/test/spheres/spheres.lua and pyrite/test/diamonds/diamonds.lua -- and must agree
with them in luminance level and structure. The pin is weak by nature (8-bit images, independent noise; two of the three are
from a build whose spectrum -> RGB step differed from today's), but it is the only output of the reference there is, so the
windows are as tight as the noise allows: tools/reference_image_study.py (profiles/r03_reference_image_study.txt) renders
each project with two seeds and every window below is the observed value +- a few times the seed-to-seed difference. What
this rules out is a wrong radiometric constant, camera, BSDF weight, lamp term or texture convention anywhere on the path;
for the spheres image, whose colours are an earlier build's, quantities in which the colour step cancels are asserted too."""
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


CENTRE, SIDES, FLOOR = (slice(27, 32), slice(24, 40)), (slice(27, 32), slice(4, 20)), (slice(27, 32), slice(4, 60))


def centre_to_sides(img):
    """Floor in front of the lamp : floor to its sides, per channel. A ratio of two regions of ONE channel: whatever the
    development step does to a channel cancels, what remains is light transport (lamp falloff, the balls' shadows and bounce)."""
    return img[CENTRE].reshape(-1, 3).mean(0) / img[SIDES].reshape(-1, 3).mean(0)


def check_spheres_transport(mine, ref, cells_allowed):
    relative = centre_to_sides(mine) / centre_to_sides(ref)
    # study: 1.012 .. 1.018 per channel at 600 spp (two seeds 0.001 apart), 1.011 .. 1.021 at the CPU test's 48 spp
    assert np.all((relative > 0.995) & (relative < 1.035)), relative
    saturated_ref, saturated_mine = (ref @ LUMA) > 0.95, (mine @ LUMA) > 0.95  # the lamp's disc: position, size, camera
    assert saturated_ref.sum() > 250 and (saturated_ref ^ saturated_mine).sum() <= cells_allowed, (saturated_ref ^ saturated_mine).sum()


def test_spheres_example_matches_the_reference_image():
    data = np.load(GOLDEN)
    ref = data["spheres"].astype(np.float64)  # 8 x 8 block means of the 512 x 256 image -> 32 x 64 cells
    mine = oracle_block_means(scenes.spheres_example(256, 128, 48), 4)  # half size, 4 x 4 blocks: the same cells
    assert mine.shape == ref.shape == (32, 64, 3)
    yr, ym = ref @ LUMA, mine @ LUMA
    assert np.corrcoef(yr.ravel(), ym.ravel())[0, 1] > 0.998  # 0.9991 with either seed
    assert yr[0:2].max() < 0.01 and ym[0:2].max() < 0.01  # black above the horizon
    ratio = ym[FLOOR] / yr[FLOOR]
    # 0.897 -- a difference of colour rendition, not of light transport: see test_gpu_renders_match_the_reference_images
    assert 0.885 < np.median(ratio) < 0.91, np.median(ratio)
    per_channel = np.median(mine[FLOOR] / ref[FLOOR], axis=(0, 1))
    assert np.all(np.abs(per_channel - np.array([0.942, 0.885, 0.864])) < 0.02), per_channel
    check_spheres_transport(mine, ref, cells_allowed=14)  # 4 and 8 cells of 311 with seeds 1 and 2
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
    assert np.corrcoef(yr.ravel(), ym.ravel())[0, 1] > 0.994  # 0.9964 / 0.9966 with seeds 1 / 2 at this half-size render
    assert 0.95 < ym.mean() / yr.mean() < 0.985, ym.mean() / yr.mean()  # 0.967 / 0.969 (the GPU at full size: 0.960)
    cells = (yr > 0.01) & (yr < 0.9)
    assert 0.91 < np.median(ym[cells] / yr[cells]) < 0.975  # 0.952 / 0.935
    per_channel = mine.mean((0, 1)) / ref.mean((0, 1))  # 0.989 0.964 0.937: the same R > G > B drift as the spheres image's
    assert np.all(np.abs(per_channel - np.array([0.989, 0.964, 0.938])) < 0.02), per_channel


TEXTURES = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "textures")
# colour-checker patches (8 x 8 pixel cells of the 1024 x 512 image: row, column) and the cube, whose textures are missing
PATCHES = {"brown": (19, 65), "orange": (25, 65), "blue": (31, 65), "white": (38, 65), "green": (31, 71), "red": (31, 77), "yellow": (31, 83),
           "cyan": (31, 98)}
CUBE = (slice(40, 58), slice(76, 98))


def check_textures_image(mine, ref, correlation, median_window, rtol, atol):
    yr, ym = ref @ LUMA, mine @ LUMA
    mask = np.ones_like(yr, dtype=bool)
    mask[CUBE] = False
    assert np.corrcoef(yr[mask], ym[mask])[0, 1] > correlation
    cells = mask & (yr > 0.02) & (yr < 0.9)
    assert median_window[0] < np.median(ym[cells] / yr[cells]) < median_window[1], np.median(ym[cells] / yr[cells])
    for name, (y, x) in PATCHES.items():  # texture lookup, sRGB decoding, RGB -> spectrum, lamps, development: all in one number
        assert np.allclose(mine[y, x], ref[y, x], rtol=rtol, atol=atol), (name, mine[y, x], ref[y, x])


def test_textures_example_matches_the_reference_image():
    """pyrite/test/textures: colour textures on a quad, a plane and a sphere, normal maps on plane and sphere, a fresnel mix
    of mirror and textured diffuse. This image was rendered by the current development code, so colours are comparable:
    the colour-checker patches come out within a few percent per channel (brown 0.124 0.044 0.022 vs 0.124 0.043 0.022)."""
    data = np.load(GOLDEN)
    ref = data["textures"].astype(np.float64)  # 64 x 128 cells
    mine = oracle_block_means(scenes.textures_reference_example(TEXTURES, 512, 256, 48), 4)
    assert mine.shape == ref.shape == (64, 128, 3)
    # at this test's 48 spp and half size: correlation 0.9966, median cell ratio 0.988 / 0.989, a patch (one cell) within 0.8 of
    # the old window; the GPU test below renders the project's own 400 spp and holds the tight windows
    check_textures_image(mine, ref, correlation=0.995, median_window=(0.975, 1.002), rtol=0.15, atol=0.01)


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
        if floor is not None:
            assert np.corrcoef(yr.ravel(), ym.ravel())[0, 1] > 0.998, name
            # The white floor (albedo 1 under the D65 lamp) comes out at 0.90x the image's luminance. tools/spheres_image_study.py
            # (profiles/r02_spheres_image_study.txt) took that number apart at the project's own 600 spp, where two seeds differ by
            # 0.3 % per cell: it does not move with bounces (8 / 16 / 32), light samples (4 / 1), spectrum samples (10 / 1) or
            # bins (64 / the lua's ignored `spectrum_bins = 50`), so it is not light transport; it differs per channel
            # (R 0.96, G 0.90, B 0.87: the image's floor is neutral, today's development renders a D65-lit white slightly warm,
            # as the reference's CURRENT code does too -- the textures image, rendered by it, matches per channel within a few
            # percent) and saturated colours differ most (the red ball's green channel is 0.07x: the image is less saturated).
            # That is the spectrum -> RGB step of an earlier build. The windows are what the noise floor supports.
            assert 0.885 < np.median(ym[floor] / yr[floor]) < 0.91
            per_channel = np.median(mine[floor] / ref[floor], axis=(0, 1))
            assert np.all(np.abs(per_channel - np.array([0.953, 0.889, 0.868])) < 0.015), per_channel
            # ... and where the colour step cancels, the transport itself: 1.012 .. 1.018 per channel, 5 of 311 lamp cells differ
            check_spheres_transport(mine, ref, cells_allowed=10)
        else:
            # diamonds (r03 study, two seeds): correlation 0.9984 / 0.9983, mean luminance 0.9598 / 0.9593, median cell ratio
            # 0.944 / 0.950, per channel 0.987 0.955 0.931 +- 0.003 -- the same warm drift as the spheres image's, at 4 %
            assert np.corrcoef(yr.ravel(), ym.ravel())[0, 1] > 0.997, name
            assert 0.945 < ym.mean() / yr.mean() < 0.975, ym.mean() / yr.mean()
            cells = (yr > 0.01) & (yr < 0.9)
            assert 0.925 < np.median(ym[cells] / yr[cells]) < 0.97
            per_channel = mine.mean((0, 1)) / ref.mean((0, 1))
            assert np.all(np.abs(per_channel - np.array([0.987, 0.955, 0.931])) < 0.015), per_channel
    world, cam, r, film = scenes.build(scenes.textures_reference_example(TEXTURES, 1024, 512, 400), seed=1)
    r.render(film, cam, world)
    lin = images.srgb_to_linear(develop.develop(film).astype(np.float64) / 255.0).astype(np.float64)
    # textures (this image is from today's development code): correlation 0.9990, median cell ratio 0.996 / 0.995, every
    # colour-checker patch within 7 % per channel (two seeds differ by up to 4.7 % on a patch, which is ONE 8 x 8 cell)
    check_textures_image(lin.reshape(64, 8, 128, 8, 3).mean((1, 3)), data["textures"].astype(np.float64), correlation=0.998, median_window=(0.985, 1.006),
                         rtol=0.07, atol=0.004)
