"""How the reference's example images are read (tests/test_reference_images.py, tools/reference_pin_study.py).

The images under pyrite/test/*/hq_example.png are 8-bit and were written by whatever build of the reference rendered them. Round 4
measured what that build's last step was instead of assuming today's (main.rs:315-327): on the white floor of the spheres image,
which falls off by a factor of 16 under the lamp, a pure power law v = (g_c * L)^(1 / gamma) with gamma = 2.18 .. 2.19 in every
channel reproduces the image's row means to 0.2 .. 0.27 eight-bit units (the rounding noise), where today's piecewise sRGB
transfer leaves a systematic 0.85 .. 0.88 (profiles/r04_reference_pin_study.txt). So these two images are decoded with gamma 2.2,
and what is compared are quantities a per-channel gain g_c cannot touch -- ratios between regions of ONE channel -- next to the
fitted gamma itself. The per-channel gains (the colour rendition of that build's response curves, main.rs:172's commented-out
`rgb_curves`, whose data is not in the checkout) are reported, never asserted around their own value.

Everything here is numpy restating the development step for analysis; it is cross-checked against the oracle's
film_develop in the tests. Test infrastructure only."""
import numpy as np

from pyrite_amd.compiler import tables

LUMA = np.array([0.2126, 0.7152, 0.0722])
XYZ_TO_SRGB = np.array([[3.2404542, -1.5371385, -0.4985314], [-0.9692660, 1.8760108, 0.0415560], [0.0556434, -0.2040259, 1.0572252]])
REFERENCE_GAMMA = 2.2  # what the fit finds (2.17 .. 2.20 per channel and quantisation model)


def _table(T, mn, mx, w):
    n = len(T)
    fi = np.clip((np.asarray(w, dtype=np.float64) - mn) / (mx - mn), 0.0, 1.0) * (n - 1)
    i0 = np.minimum(np.floor(fi).astype(int), n - 2)
    m = (fi - i0)[:, None]
    return T[i0] * (1 - m) + T[i0 + 1] * m


def tristimulus_weights(bins, lo=380.0, width=400.0, step=2.0):
    """W[bins, 3] with XYZ / xyz_scale = spectrum @ W: spectrum_to_tristimulus (main.rs:371-418) is linear in the pixel's
    piecewise-constant spectrum -- trapezoids at `step` nm against the CIE tables, divided by the total width."""
    tb = tables()
    xyz = np.asarray(tb["xyz"], dtype=np.float64).reshape(-1, 3)
    wl = np.arange(lo, lo + width + step / 2, step)
    R = _table(xyz, float(tb["xyz_min"]), float(tb["xyz_max"]), wl)
    idx = np.minimum(np.floor((wl - lo) / width * bins), bins - 1).astype(int)
    tw = np.full(len(wl), step)
    tw[0] = tw[-1] = step / 2
    W = np.zeros((bins, 3))
    np.add.at(W, idx, R * tw[:, None])
    return W / width


def linear_rgb(grains, wl_start=380.0, wl_width=400.0):
    """Unclamped linear sRGB [h, w, 3] (f64) of a film's grains [h, w, bins, 2]: main.rs:315-327 before `into_encoding`."""
    acc, weight = grains[..., 0].astype(np.float64), grains[..., 1].astype(np.float64)
    spectrum = np.where(weight > 0, acc / np.maximum(weight, 1e-30), 0.0)
    xyz = spectrum @ tristimulus_weights(spectrum.shape[-1], wl_start, wl_width) * 3.444
    return xyz @ XYZ_TO_SRGB.T


def srgb_encode(v):
    v = np.clip(v, 0.0, 1.0)
    return np.where(v <= 0.0031308, 12.92 * v, 1.055 * v ** (1 / 2.4) - 0.055)


def srgb_decode(v):
    return np.where(v <= 0.04045, v / 12.92, ((v + 0.055) / 1.055) ** 2.4)


def blocks(a, b=8):
    h, w = a.shape[0] // b * b, a.shape[1] // b * b
    return a[:h, :w].reshape(h // b, b, w // b, b, -1).mean((1, 3))


def fit_transfer(mine_lin, ref8, rows, cols, model="gamma"):
    """Per channel: least squares of the reference's 8-bit row means over `rows` (pairs of rows, columns `cols`) against
    round(255 * f(g * L)) of the render's linear values, pixel by pixel (so clipping and rounding are part of the model).
    model "gamma": f = x^(1/gamma), parameters (g, gamma); "srgb": f = the piecewise sRGB transfer, parameter g.
    Returns gains[3], gammas[3] (nan for "srgb"), rms[3] in eight-bit units."""
    from scipy.optimize import least_squares

    gains, gammas, rms = np.zeros(3), np.full(3, np.nan), np.zeros(3)
    for c in range(3):
        target = np.array([ref8[y:y + 2, cols, c].astype(np.float64).mean() for y in rows])

        def residual(p):
            out = []
            for y in rows:
                x = np.clip(p[0] * mine_lin[y:y + 2, cols, c], 0.0, 1.0)
                e = x ** (1.0 / p[1]) if model == "gamma" else srgb_encode(x)
                out.append(np.floor(255.0 * e + 0.5).mean())
            return np.array(out) - target

        s = least_squares(residual, [1.1, 2.2] if model == "gamma" else [1.1], diff_step=1e-3)
        gains[c], rms[c] = s.x[0], np.sqrt(np.mean(s.fun ** 2))
        if model == "gamma":
            gammas[c] = s.x[1]
    return gains, gammas, rms


# regions of the 512 x 256 spheres image, in pixels
FLOOR_COLUMNS = slice(200, 312)  # under the lamp, between the balls: the floor's own white, least tinted by the balls
FLOOR_ROWS = list(range(190, 256, 2))  # from just under the saturated pool of light (v ~ 250) to the image's edge (v ~ 74)


def region_ratio(img, a, b):
    return img[a].reshape(-1, 3).mean(0) / img[b].reshape(-1, 3).mean(0)


# 8 x 8 cells: floor in front of the lamp / to its sides; two unsaturated bands of floor 4x apart in brightness
CENTRE, SIDES = (slice(27, 32), slice(24, 40)), (slice(27, 32), slice(4, 20))
BAND_FAR, BAND_NEAR = (slice(24, 26), slice(4, 60)), (slice(29, 32), slice(4, 60))
POOL = (slice(22, 24), slice(4, 60))  # the rows that hold the saturated pool of light under the lamp


def spheres_transport(mine_lin, ref8, gains=None):
    """The transport-only quantities of the spheres image, render / reference, the reference decoded with gamma 2.2:
    centre : sides and far band : near band per channel (ratios of two regions of one channel: any per-channel gain cancels);
    and -- given the gains -- the rows that hold the clipped pool of light, where the gain decides how much is clipped and so
    has to be applied to the render BEFORE the clip."""
    ref = blocks((ref8[..., :3].astype(np.float64) / 255.0) ** REFERENCE_GAMMA)
    mine = blocks(np.clip(mine_lin, 0.0, 1.0))
    out = {"centre_sides": region_ratio(mine, CENTRE, SIDES) / region_ratio(ref, CENTRE, SIDES),
           "far_near": region_ratio(mine, BAND_FAR, BAND_NEAR) / region_ratio(ref, BAND_FAR, BAND_NEAR)}
    if gains is not None:
        gained = blocks(np.clip(mine_lin * gains, 0.0, 1.0))
        out["pool_near"] = region_ratio(gained, POOL, BAND_NEAR) / region_ratio(ref, POOL, BAND_NEAR)
    return out
