#!/usr/bin/env python3
"""turn the reference-image pin from an asserted offset into evidence. CPU only (the oracle renders).

    python tools/reference_pin_study.py > profiles/r04_reference_pin_study.txt      (needs /root/reference; ~2 minutes)

Renders pyrite/test/spheres and pyrite/test/diamonds at the projects' own sizes and sample counts with two seeds, keeps the
UNCLAMPED linear RGB of every pixel (tests/reference_pin.py: the development step is linear up to its clamp) and asks of the
reference's 8-bit images: (1) which transfer function wrote them; (2) with that function undone, do the quantities a
per-channel gain cannot touch agree; (3) does the spheres image's colour transform predict the diamonds image."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle  # noqa: E402
import reference_pin as rp  # noqa: E402
from pyrite_amd import images, scenes  # noqa: E402

np.set_printoptions(precision=4, suppress=True, linewidth=200)
REFERENCE = "/root/reference/pyrite/test"
THREADS = int(os.environ.get("STUDY_THREADS", "8"))


def render(project, seed):
    world, cam, r, film = scenes.build(project, seed=seed)
    t = time.time()
    oracle.OracleScene(world).render(r, cam, film, threads=THREADS)
    sys.stderr.write("rendered seed %d in %.1f s\n" % (seed, time.time() - t))
    return film


def fmt(v):
    return " ".join("%.4f" % x for x in np.atleast_1d(v))


print("== spheres (512 x 256, 600 spp; oracle, seeds 1 and 2) against pyrite/test/spheres/hq_example.png")
ref8 = images.read_png(os.path.join(REFERENCE, "spheres", "hq_example.png"))[..., :3]
films = [render(scenes.spheres_example(512, 256, 600), s) for s in (1, 2)]
lin = [rp.linear_rgb(f.grains) for f in films]
dev = oracle.film_develop(films[0])
mine8 = np.floor(255.0 * rp.srgb_encode(lin[0]) + 0.5)
d = np.abs(mine8[:-1] - dev[:-1].astype(np.float64))
print("   numpy development vs oracle_film_develop (seed 1): %.4f %% of channel values equal, max difference %d eight-bit unit(s)" % (100.0 * (d == 0).mean(), d.max()))

print("-- (1) which transfer function wrote the image? row means of the floor under the lamp (columns 200..311, rows 190..255: v = 250 down to 74)")
for model in ("srgb", "gamma"):
    for k, l in enumerate(lin):
        gains, gammas, rms = rp.fit_transfer(l, ref8, rp.FLOOR_ROWS, rp.FLOOR_COLUMNS, model)
        print("   %-44s seed %d: gain %s%s | rms residual %s eight-bit units" % ("piecewise sRGB (today's main.rs:315-327), gain only" if model == "srgb" else "power law v = (g L)^(1/gamma), gain + gamma",
                                                                                  k + 1, fmt(gains), "" if model == "srgb" else " | gamma " + fmt(gammas), fmt(rms)))
gains, gammas, _ = rp.fit_transfer(lin[0], ref8, rp.FLOOR_ROWS, rp.FLOOR_COLUMNS, "gamma")
print("   rounding alone leaves 0.29 units rms per pixel and ~0.02 per row mean; the 0.2 .. 0.27 left by the power law is the two images' independent noise.")
print("   -> the image was encoded with a power law of exponent 1 / 2.2 (fitted gamma %s), not with the piecewise sRGB function." % fmt(gammas))

print("-- (2) transport-only quantities, render / reference (1.0 = agreement), reference decoded with piecewise sRGB (round 3) and with gamma 2.2")
ref_srgb, ref_gamma = rp.blocks(rp.srgb_decode(ref8 / 255.0)), rp.blocks((ref8 / 255.0) ** rp.REFERENCE_GAMMA)
for k, l in enumerate(lin):
    mine = rp.blocks(np.clip(l, 0, 1))
    for name, ref in (("sRGB decode", ref_srgb), ("gamma 2.2 decode", ref_gamma)):
        cs = rp.region_ratio(mine, rp.CENTRE, rp.SIDES) / rp.region_ratio(ref, rp.CENTRE, rp.SIDES)
        fn = rp.region_ratio(mine, rp.BAND_FAR, rp.BAND_NEAR) / rp.region_ratio(ref, rp.BAND_FAR, rp.BAND_NEAR)
        pool = (mine[rp.POOL] @ rp.LUMA).mean() / (mine[rp.BAND_NEAR] @ rp.LUMA).mean() / ((ref[rp.POOL] @ rp.LUMA).mean() / (ref[rp.BAND_NEAR] @ rp.LUMA).mean())
        print("   seed %d %-17s floor centre : sides (R G B) %s | far band : near band, 4.1x apart (R G B) %s | clipped pool rows : near band (luminance, no gain applied) %.4f" % (k + 1, name, fmt(cs), fmt(fn), pool))
    t = rp.spheres_transport(l, ref8, gains)
    print("   seed %d gamma 2.2 + the fitted gains applied BEFORE the clip: pool rows : near band (R G B) %s   (the pool is clipped in both; the gain decides how much)" % (k + 1, fmt(t["pool_near"])))
print("   round 3's +1.2 .. 1.8 % (centre : sides) and +8 % (far : near incl. the pool) were the wrong decode: a ratio of two levels read through the wrong curve.")

print("-- (3) what is left is one gain per channel: reference = g_c x render")
floor = (slice(27, 32), slice(4, 60))
for k, l in enumerate(lin):
    mine = rp.blocks(np.clip(l, 0, 1))
    print("   seed %d floor cells, render / reference per channel: median %s | 10th .. 90th percentile %s .. %s" % (
        k + 1, fmt(np.median(mine[floor] / ref_gamma[floor], axis=(0, 1))), fmt(np.percentile(mine[floor] / ref_gamma[floor], 10, axis=(0, 1))), fmt(np.percentile(mine[floor] / ref_gamma[floor], 90, axis=(0, 1)))))
print("   fitted gains %s: the floor is albedo 1 under `light_source.d65 * 3`; today's code renders it (R G B) %s, the image has %s (gamma-decoded, green = 1)" % (
    fmt(gains), fmt(rp.blocks(np.clip(lin[0], 0, 1))[floor].mean((0, 1)) / rp.blocks(np.clip(lin[0], 0, 1))[floor].mean((0, 1))[1]), fmt(ref_gamma[floor].mean((0, 1)) / ref_gamma[floor].mean((0, 1))[1])))
print("   Colours: a 3 x 3 matrix fitted on floor + both balls (cells with reference luminance 0.01 .. 0.9):")
yr = ref_gamma @ rp.LUMA
sel = (yr > 0.01) & (yr < 0.9)
sel[:8] = False
A, B = rp.blocks(np.clip(lin[0], 0, 1))[sel], ref_gamma[sel]
M3, res, *_ = np.linalg.lstsq(A, B, rcond=None)
print("   reference = render @ M, M^T =\n%s" % M3.T)
pred_diag, pred_m3 = A * gains, A @ M3
print("   rms error over those %d cells: gains only %.4f, 3 x 3 matrix %.4f (linear units; cell noise ~0.002): the old build's response curves were not a matrix away from" % (sel.sum(), np.sqrt(np.mean((pred_diag - B) ** 2)), np.sqrt(np.mean((pred_m3 - B) ** 2))))
print("   CIE XYZ -> sRGB -- off-diagonal terms desaturate (the image's balls are paler than today's) -- and their data is not in the checkout (main.rs:172: `rgb_curves = None; /* ... */`).")

print("== diamonds (512 x 300, 200 spp, 256 bounces, one wavelength per sample; oracle, seeds 1 and 2) against pyrite/test/diamonds/hq_example.png")
ref8d = images.read_png(os.path.join(REFERENCE, "diamonds", "hq_example.png"))[..., :3]
films_d = [render(scenes.diamonds_example(512, 300, 200, bounces=256), s) for s in (1, 2)]
refd = rp.blocks((ref8d / 255.0) ** rp.REFERENCE_GAMMA)
refd_srgb = rp.blocks(rp.srgb_decode(ref8d / 255.0))
yr = refd @ rp.LUMA
mid = (yr > 0.01) & (yr < 0.9)
for k, f in enumerate(films_d):
    l = rp.linear_rgb(f.grains)
    mine = rp.blocks(np.clip(l, 0, 1))
    gained = rp.blocks(np.clip(l * gains, 0, 1))
    gained_m3 = rp.blocks(np.clip(l @ M3, 0, 1))
    print("   seed %d mean over the image, render / reference (R G B): sRGB decode %s | gamma 2.2 decode %s | gamma 2.2 + the spheres image's gains %s | + its 3 x 3 matrix %s" % (
        k + 1, fmt(mine.mean((0, 1)) / refd_srgb.mean((0, 1))), fmt(mine.mean((0, 1)) / refd.mean((0, 1))), fmt(gained.mean((0, 1)) / refd.mean((0, 1))), fmt(gained_m3.mean((0, 1)) / refd.mean((0, 1)))))
    r = gained.mean((0, 1)) / refd.mean((0, 1))
    print("          chromatic part of that prediction (R / G, B / G of the ratios; 1.0 = the spheres gains predict the diamonds image's colour balance): %s" % fmt([r[0] / r[1], r[2] / r[1]]))
    r0 = mine.mean((0, 1)) / refd.mean((0, 1))
    print("          ... and without the gains: %s" % fmt([r0[0] / r0[1], r0[2] / r0[1]]))
    print("          correlation of cell luminance %.4f; mid cells (0.01 < Y_ref < 0.9) sum ratio with gains %s" % (np.corrcoef(yr.ravel(), (mine @ rp.LUMA).ravel())[0, 1], fmt(gained[mid].sum(0) / refd[mid].sum(0))))
