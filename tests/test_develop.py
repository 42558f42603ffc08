"""Film development ("next" row f1): the oracle's restatement of main.rs:315-418 against hand-derived values, and (gpu) the
HIP kernel against the oracle."""
import math

import numpy as np
import pytest

import oracle
from pyrite_amd import scenes
from pyrite_amd.compiler import ProjectError, tables
from pyrite_amd.develop import develop, evaluate_at, sampling_wavelengths, save_png
from pyrite_amd.film import Film
from pyrite_amd.project import blackbody, fresnel, light_source, mix, spectrum


def srgb8(v):
    v = min(max(v, 0.0), 1.0)
    e = 12.92 * v if v <= 0.0031308 else 1.055 * v ** (1 / 2.4) - 0.055
    return int(min(max(e, 0.0), 1.0) * 255 + 0.5)


def flat_film(value, width=4, height=3):
    film = Film(width, height, 64)
    film.grains[..., 0] = value * 2.0  # acc = 2 * value with weight 2 -> developed value
    film.grains[..., 1] = 2.0
    return film


def test_sampling_wavelengths():
    film = Film(2, 2, 64)
    wl = sampling_wavelengths(film, 2.0)
    assert wl[0] == 380.0 and wl[-1] == 780.0 and len(wl) == 201
    wl = sampling_wavelengths(film, 30.0)  # the preview step overshoots the span once (main.rs:395)
    assert list(wl[-2:]) == [770.0, 800.0] and len(wl) == 15


def test_flat_spectrum_develops_to_the_observer_integral():
    # spectrum == c everywhere: XYZ = c * 3.444 * (trapezoid integral of the observer over 380..780) / 400
    c = 0.5
    t = tables()
    wl = np.arange(380, 781, 2)
    idx = (wl - 360).astype(int)  # the tables are 1 nm from 360
    xyz = np.array([np.trapezoid(t["xyz"][idx, k].astype(np.float64), wl) for k in range(3)]) / 400.0 * 3.444 * c
    m = np.array([[3.2404542, -1.5371385, -0.4985314], [-0.9692660, 1.8760108, 0.0415560], [0.0556434, -0.2040259, 1.0572252]])
    expect = [srgb8(v) for v in m @ xyz]
    img = oracle.film_develop(flat_film(c))
    assert np.abs(img[0, 0].astype(int) - np.array(expect)).max() <= 1
    assert (img.reshape(-1, 3)[:-1] == img[0, 0]).all()
    assert (img[-1, -1] == 0).all()  # DevelopedPixels never yields the last pixel (film.rs:299)
    assert abs(int(img[0, 0, 0]) - int(img[0, 0, 1])) < 25  # an equal-energy spectrum is close to neutral


def test_undeveloped_grains_are_black_and_brightness_is_monotonic():
    film = Film(3, 2, 64)
    assert (oracle.film_develop(film) == 0).all()  # weight 0 -> 0 (film.rs:138-142)
    dark, bright = oracle.film_develop(flat_film(0.05)), oracle.film_develop(flat_film(0.4))
    assert (bright[0, 0] > dark[0, 0]).all()
    assert (oracle.film_develop(flat_film(50.0))[0, 0] == 255).all()  # clamped


def test_filter_and_white_balance():
    base = oracle.film_develop(flat_film(0.3))
    half = oracle.film_develop(flat_film(0.3), filter=0.5)
    assert (half[0, 0] < base[0, 0]).all()
    red_pass = spectrum(format="curve", points=[(590, 0), (600, 1), (779, 1), (780, 0)])
    red = oracle.film_develop(flat_film(0.3), filter=red_pass)
    assert red[0, 0, 0] > red[0, 0, 2] and red[0, 0, 2] == 0
    # a film exposed with a 4000 K blackbody looks neutral-ish once balanced against the same white (cornell.lua:16)
    film = Film(2, 2, 64)
    for b in range(64):
        w = 380 + (b + 0.5) * 400 / 64
        film.grains[..., b, 0] = evaluate_at(blackbody(4000), w) * 1e-14
        film.grains[..., b, 1] = 1.0
    raw = oracle.film_develop(film)[0, 0].astype(int)
    balanced = oracle.film_develop(film, white=blackbody(4000))[0, 0].astype(int)
    assert raw[0] - raw[2] > 40  # 4000 K is reddish
    assert abs(balanced[0] - balanced[2]) < abs(raw[0] - raw[2]) / 2


def test_expression_sampling_follows_the_vm():
    s = spectrum(format="array", min=400, max=700, points=[1.0, 3.0, 2.0])
    assert evaluate_at(s * 3, 550.0) == 9.0
    assert evaluate_at(mix(s, 10, 0.25), 550.0) == np.float32(3.0 * 0.75 + 2.5)
    assert evaluate_at(light_source.d65, 560.0) == np.float32(1.0)  # data/d65.csv is normalised to 1 at 560 nm
    with pytest.raises(ProjectError, match="surface normal"):
        evaluate_at(fresnel(1.5), 500.0)


def test_png_writer_round_trip(tmp_path):
    import struct
    import zlib

    rgb = (np.arange(5 * 7 * 3) % 256).astype(np.uint8).reshape(5, 7, 3)
    path = tmp_path / "x.png"
    save_png(str(path), rgb)
    data = path.read_bytes()
    assert data[:8] == b"\x89PNG\r\n\x1a\n"
    w, h = struct.unpack(">II", data[16:24])
    assert (w, h) == (7, 5)
    i = data.index(b"IDAT")
    n = struct.unpack(">I", data[i - 4:i])[0]
    raw = zlib.decompress(data[i + 4:i + 4 + n])
    rows = np.frombuffer(raw, dtype=np.uint8).reshape(5, 1 + 7 * 3)
    assert (rows[:, 0] == 0).all() and (rows[:, 1:].reshape(5, 7, 3) == rgb).all()


@pytest.mark.gpu
def test_gpu_development_matches_the_oracle(gpu_lib):
    world, cam, r, film = scenes.build(scenes.c2_cornell(96, 64, 16), seed=3)
    r.render(film, cam, world)
    for kwargs in ({}, {"step_size": 30.0}, {"white": blackbody(4000)}, {"filter": spectrum(format="curve", points=[(450, 0), (500, 1), (600, 1), (650, 0)])}):
        gpu, cpu = develop(film, **kwargs), oracle.film_develop(film, **kwargs)
        assert np.array_equal(gpu, cpu), kwargs
    assert develop(film).reshape(-1, 3)[:-1].max() > 100 and (develop(film)[-1, -1] == 0).all()
    rng = np.random.RandomState(4)
    noise = Film(33, 17, 50, (400.0, 700.0))
    noise.grains[..., 0] = rng.gamma(0.5, 1.0, size=noise.grains.shape[:-1])
    noise.grains[..., 1] = rng.randint(0, 3, size=noise.grains.shape[:-1])
    assert np.array_equal(develop(noise), oracle.film_develop(noise))
