#!/usr/bin/env python3
"""Developer study: how tightly can the three example images the reference rendered pin this library?
Renders each project on the GPU at the project's own size and sample count with TWO seeds, prints the seed-to-seed noise of every
statistic tests/test_reference_images.py asserts, the statistic against the reference image, and -- for the spheres image, whose
colour rendition belongs to an earlier build -- quantities in which the colour step cancels (ratios between regions of one
channel, the lamp's saturated disc).      python tools/reference_image_study.py > profiles/r03_reference_image_study.txt"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from pyrite_amd import develop, images, scenes  # noqa: E402
from test_reference_images import CUBE, PATCHES, TEXTURES  # noqa: E402

data = np.load(os.path.join(ROOT, "tests", "golden", "reference_example_images.npz"))
LUMA = np.array([0.2126, 0.7152, 0.0722])


def cells(project, seed, shape):
    world, cam, r, film = scenes.build(project, seed=seed)
    r.render(film, cam, world)
    lin = images.srgb_to_linear(develop.develop(film).astype(np.float64) / 255.0).astype(np.float64)
    world.close()
    h, w = shape[0] * 8, shape[1] * 8
    return lin[:h, :w].reshape(shape[0], 8, shape[1], 8, 3).mean((1, 3))


def report(name, value_a, value_b, fmt="%.4f"):
    a, b = np.atleast_1d(value_a), np.atleast_1d(value_b)
    print("   %-58s seed 1 %s | seed 2 %s | |difference| %s" % (name, " ".join(fmt % v for v in a), " ".join(fmt % v for v in b), " ".join(fmt % abs(x - y) for x, y in zip(a, b))))


print("== diamonds (512 x 300, 200 spp, 256 bounces)")
ref = data["diamonds"].astype(np.float64)
m = [cells(scenes.diamonds_example(512, 300, 200, bounces=256), s, ref.shape[:2]) for s in (1, 2)]
yr = ref @ LUMA
ym = [x @ LUMA for x in m]
report("mean luminance / reference", ym[0].mean() / yr.mean(), ym[1].mean() / yr.mean())
mid = (yr > 0.01) & (yr < 0.9)
report("median cell ratio (0.01 < Y_ref < 0.9)", np.median(ym[0][mid] / yr[mid]), np.median(ym[1][mid] / yr[mid]))
report("correlation of cell luminance", np.corrcoef(yr.ravel(), ym[0].ravel())[0, 1], np.corrcoef(yr.ravel(), ym[1].ravel())[0, 1])
report("per-channel mean / reference (R G B)", m[0].mean((0, 1)) / ref.mean((0, 1)), m[1].mean((0, 1)) / ref.mean((0, 1)))

print("== textures (1024 x 512, 400 spp)")
ref = data["textures"].astype(np.float64)
m = [cells(scenes.textures_reference_example(TEXTURES, 1024, 512, 400), s, ref.shape[:2]) for s in (1, 2)]
yr = ref @ LUMA
ym = [x @ LUMA for x in m]
mask = np.ones_like(yr, dtype=bool)
mask[CUBE] = False
mid = mask & (yr > 0.02) & (yr < 0.9)
report("median cell ratio (0.02 < Y_ref < 0.9, cube excluded)", np.median(ym[0][mid] / yr[mid]), np.median(ym[1][mid] / yr[mid]))
report("correlation of cell luminance", np.corrcoef(yr[mask], ym[0][mask])[0, 1], np.corrcoef(yr[mask], ym[1][mask])[0, 1])
worst = 0.0
for name, (y, x) in PATCHES.items():
    report("patch %-7s render / reference (R G B)" % name, m[0][y, x] / np.maximum(ref[y, x], 1e-6), m[1][y, x] / np.maximum(ref[y, x], 1e-6), "%.3f")
    print("      reference linear RGB %.3f %.3f %.3f | render %.3f %.3f %.3f | absolute difference %.4f %.4f %.4f" % (*ref[y, x], *m[0][y, x], *np.abs(m[0][y, x] - ref[y, x])))

print("== spheres (512 x 256, 600 spp): quantities in which the colour step cancels")
ref = data["spheres"].astype(np.float64)
m = [cells(scenes.spheres_example(512, 256, 600), s, ref.shape[:2]) for s in (1, 2)]
centre, sides = (slice(27, 32), slice(24, 40)), (slice(27, 32), slice(4, 20))


def region_ratio(img):
    return img[centre].reshape(-1, 3).mean(0) / img[sides].reshape(-1, 3).mean(0)


print("   floor front centre : front sides, per channel -- reference %s" % " ".join("%.4f" % v for v in region_ratio(ref)))
report("floor front centre : front sides (R G B)", region_ratio(m[0]), region_ratio(m[1]))
report("... divided by the reference's (R G B)", region_ratio(m[0]) / region_ratio(ref), region_ratio(m[1]) / region_ratio(ref))
yr = ref @ LUMA
ym = [x @ LUMA for x in m]
sat_r = yr > 0.95
for k in (0, 1):
    sat_m = ym[k] > 0.95
    print("   lamp disc: %d cells saturated in the reference, %d in the render (seed %d), %d in both, %d in one only" % (sat_r.sum(), sat_m.sum(), k + 1, (sat_r & sat_m).sum(), (sat_r ^ sat_m).sum()))
far, near = (slice(22, 24), slice(4, 60)), (slice(29, 32), slice(4, 60))
falloff = lambda img: (img[far] @ LUMA).mean() / (img[near] @ LUMA).mean()  # noqa: E731
print("   floor luminance far rows : near rows -- reference %.4f" % falloff(ref))
report("floor luminance far rows : near rows", falloff(m[0]), falloff(m[1]))
report("floor luminance / reference (the colour offset itself)", np.median(ym[0][27:32, 4:60] / yr[27:32, 4:60]), np.median(ym[1][27:32, 4:60] / yr[27:32, 4:60]))
