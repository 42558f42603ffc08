"""Film development, the step after the hot path: developed pixel spectra -> CIE XYZ ->
sRGB, as pyrite/src/main.rs:190-238 (filter / white balance) and :315-418 (spectrum_to_xyz) do it, on the GPU through
`pyr_film_develop`. The image-settings programs (`filter`, `white`: project/mod.rs:111-118) only depend on the wavelength
(SpectrumSamplingInput, main.rs:470-476) and are evaluated here, once per sampling wavelength, in f32."""
from __future__ import annotations

import ctypes as C
import struct
import zlib

import numpy as np

from . import abi
from ._lib import check, lib
from .compiler import ProjectError, is_number, tables
from .film import Film

f32 = np.float32


def _array_get(data, mn, mx, w):
    """Spectrum::Array::get (project/spectra.rs:32-55) for one f32 wavelength."""
    n = len(data)
    if w <= mn:
        return f32(data[0])
    if w >= mx:
        return f32(data[-1])
    normalized = f32(f32(w - mn) / f32(mx - mn))
    fi = f32(normalized * f32(f32(n) - f32(1)))
    i0 = int(np.trunc(fi))
    mix = f32(fi - f32(np.trunc(fi)))
    return f32(f32(f32(data[i0]) * f32(f32(1) - mix)) + f32(f32(data[i0 + 1]) * mix))


def _curve_get(points, w):
    """Interpolated::get (math.rs:22-72): zero at and outside the end points."""
    pts = np.asarray(points, dtype=f32).reshape(-1, 2)
    if len(pts) == 0 or pts[0, 0] >= w or pts[-1, 0] <= w:
        return f32(0)
    lo, hi = 0, len(pts) - 1
    while hi > lo + 1:
        mid = (lo + hi) // 2
        if pts[mid, 0] == w:
            return f32(pts[mid, 1])
        if pts[mid, 0] > w:
            hi = mid
        else:
            lo = mid
    (x0, y0), (x1, y1) = pts[lo], pts[hi]
    return f32(y0 + f32(f32(y1 - y0) * f32(f32(w - x0) / f32(x1 - x0))))


def evaluate_at(expression, wavelength):
    """Value of a number expression at one wavelength, with the VM's f32 arithmetic (program/execution_context.rs:69-283).
    Only wavelength-dependent expressions are allowed here, as in the reference (main.rs:478-518)."""
    w = f32(wavelength)
    e = expression
    if is_number(e):
        return f32(e)
    t = e.type
    with np.errstate(all="ignore"):
        if t == "spectrum":
            name = e.get("name")
            if name is not None:
                tb = tables()
                return _array_get(tb[name], f32(tb["light_min"]), f32(tb["light_max"]), w)
            if e.get("format") == "array":
                return _array_get(np.asarray(e.points, dtype=f32), f32(e.min), f32(e.max), w)
            return _curve_get(e.points, w)
        if t == "blackbody":  # math.rs:177-182
            temperature = evaluate_at(e.temperature, w)
            wl = f32(w * f32(1.0e-9))
            a2 = f32(wl * wl)
            a4 = f32(a2 * a2)
            power = f32(f32(3.74183e-16) * f32(f32(1) / f32(wl * a4)))
            return f32(power / f32(f32(np.exp(np.float64(f32(f32(1.4388e-2) / f32(wl * temperature))))) - f32(1)))
        if t == "binary":
            l, r = evaluate_at(e.lhs, w), evaluate_at(e.rhs, w)
            return f32({"add": l + r, "sub": l - r, "mul": l * r, "div": l / r}[e.operator])
        if t == "mix":
            amount = min(max(evaluate_at(e.amount, w), f32(0)), f32(1))
            return f32(f32(evaluate_at(e.lhs, w) * f32(f32(1) - amount)) + f32(evaluate_at(e.rhs, w) * amount))
        if t == "clamp":
            return f32(max(min(evaluate_at(e.value, w), evaluate_at(e.max, w)), evaluate_at(e.min, w)))
    if t == "fresnel":
        raise ProjectError("the surface normal cannot be used while sampling a constant spectrum")
    raise ProjectError("cannot sample a %s expression as a spectrum" % t)


def sampling_wavelengths(film: Film, step_size):
    """wl_i of spectrum_to_tristimulus (main.rs:393-411): start at the span's minimum, add `step_size` in f32 while below the
    maximum; one more sample than steps."""
    lo, hi = f32(film.wavelength_start), f32(film.wavelength_start + film.wavelength_width)
    out = [lo]
    while out[-1] < hi:
        out.append(f32(out[-1] + f32(step_size)))
    return np.asarray(out, dtype=f32)


def develop_params(film: Film, step_size=2.0, filter=None, white=None):
    """PyrDevelopParams for `film` (+ the numpy arrays it borrows)."""
    tb = tables()
    wl = sampling_wavelengths(film, step_size)
    keep = {"xyz": np.ascontiguousarray(tb["xyz"], dtype=f32)}
    p = abi.PyrDevelopParams()
    p.step_size, p.xyz_scale, p.sample_count = float(step_size), 3.444, len(wl)
    p.xyz_table = keep["xyz"].ctypes.data_as(C.POINTER(C.c_float))
    p.xyz_count, p.xyz_min, p.xyz_max = len(keep["xyz"]), float(tb["xyz_min"]), float(tb["xyz_max"])
    if filter is not None:  # main.rs:197-202
        keep["filter"] = np.asarray([evaluate_at(filter, w) for w in wl], dtype=f32)
        p.filter = keep["filter"].ctypes.data_as(C.POINTER(C.c_float))
    if white is not None:  # main.rs:204-222
        w, hi = f32(film.wavelength_start), f32(film.wavelength_start + film.wavelength_width)
        mx, d65_mx = f32(0), f32(0)
        while w < hi:
            mx = max(mx, evaluate_at(white, w))
            d65_mx = max(d65_mx, _array_get(tb["d65"], f32(tb["light_min"]), f32(tb["light_max"]), w))
            w = f32(w + f32(1.0))
        keep["white_div"] = np.asarray([max(f32(evaluate_at(white, x) / mx), f32(0.000001)) for x in wl], dtype=f32)
        keep["white_mul"] = np.asarray([f32(_array_get(tb["d65"], f32(tb["light_min"]), f32(tb["light_max"]), x) / d65_mx) for x in wl], dtype=f32)
        p.white_div = keep["white_div"].ctypes.data_as(C.POINTER(C.c_float))
        p.white_mul = keep["white_mul"].ctypes.data_as(C.POINTER(C.c_float))
    return p, keep


def develop(film: Film, step_size=2.0, filter=None, white=None, device=0):
    """uint8 [height, width, 3] sRGB image of `film`, developed on the GPU."""
    p, keep = develop_params(film, step_size, filter, white)
    desc = film.desc()
    grains = np.ascontiguousarray(film.grains)
    out = np.zeros((film.height, film.width, 3), dtype=np.uint8)
    check(lib().pyr_film_develop(C.byref(desc), grains.ctypes.data, C.byref(p), out.ctypes.data, int(device)))
    del keep
    return out


def save_png(path, rgb):
    """Minimal PNG writer (8-bit RGB, no interlace) -- the image::save of main.rs:327."""
    rgb = np.ascontiguousarray(rgb, dtype=np.uint8)
    h, w, _ = rgb.shape
    raw = b"".join(b"\x00" + rgb[y].tobytes() for y in range(h))

    def chunk(tag, data):
        body = tag + data
        return struct.pack(">I", len(data)) + body + struct.pack(">I", zlib.crc32(body) & 0xFFFFFFFF)

    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 0)) + chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b""))
