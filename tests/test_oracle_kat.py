"""Known-answer tests that pin the CPU oracle, function by function, to the reference lines it restates.

The reference has no tests of its own: every expected value here is derived by hand from the
cited source lines (or from a published constant of a third-party algorithm) and written out in the test."""
import ctypes as C
import math

import numpy as np
import pytest

import oracle
from oracle import F3, F6, U4

L = oracle.lib()
f32 = np.float32


def ray(o, d):
    return F6(*o, *d)


# ---------------------------------------------------------------- RNG (rand_xorshift 0.3.0 / rand 0.8.5)
def test_xorshift128_known_answer():
    # Marsaglia's xorshift128 with his published initial state has the well-known first output 3701687786.
    st = U4(123456789, 362436069, 521288629, 88675123)
    assert L.oracle_rng_next_u32(st) == 3701687786
    assert L.oracle_rng_next_u32(st) == 458299110
    assert L.oracle_rng_next_u32(st) == 2500872618


def test_seeding_is_a_function_of_seed_tile_iteration():
    a, b, c, d = U4(), U4(), U4(), U4()
    L.oracle_rng_seed(1, 0, 0, a)
    L.oracle_rng_seed(1, 0, 0, b)
    L.oracle_rng_seed(1, 0, 1, c)
    L.oracle_rng_seed(1, 1, 0, d)
    assert list(a) == list(b)
    assert list(a) != list(c) and list(a) != list(d) and list(c) != list(d)
    assert any(a)


def test_gen_f32_uses_the_top_24_bits():
    st = U4(123456789, 362436069, 521288629, 88675123)
    v = L.oracle_rng_gen_f32(st)
    assert v == f32(3701687786 >> 8) * f32(2.0 ** -24)  # rand 0.8.5 Standard for f32
    assert 0.0 <= v < 1.0


def test_gen_range_f32_formula_and_bounds():
    st = U4(123456789, 362436069, 521288629, 88675123)
    v = L.oracle_rng_gen_range_f32(st, 380.0, 420.0)
    bits = (3701687786 >> 9) | 0x3F800000
    value0_1 = np.frombuffer(np.uint32(bits).tobytes(), dtype=f32)[0] - f32(1)
    assert v == f32(value0_1 * f32(40.0)) + f32(380.0)  # UniformFloat::sample_single: value0_1 * scale + low
    st = U4(1, 2, 3, 4)
    for _ in range(20000):
        v = L.oracle_rng_gen_range_f32(st, 740.0, 780.0)
        assert 740.0 <= v < 780.0


def test_integer_ranges_are_uniform_and_in_range():
    st = U4(9, 8, 7, 6)
    counts = np.zeros(10, dtype=int)
    for _ in range(20000):
        counts[L.oracle_rng_gen_range_usize(st, 10)] += 1
    assert counts.min() > 1700 and counts.max() < 2300
    counts = np.zeros(3, dtype=int)
    for _ in range(9000):
        counts[L.oracle_rng_choose_index(st, 3)] += 1
    assert counts.min() > 2700
    assert L.oracle_rng_choose_index(st, 1) == 0


def test_choose_of_one_rejects_half_of_the_draws():
    # gen_range(0..1u32): zone = (1 << 31) - 1, so a draw with the top bit set is rejected (rand 0.8.5 sample_single).
    st = U4(123456789, 362436069, 521288629, 88675123)  # first draw 3701687786 has the top bit set -> rejected
    ref = U4(123456789, 362436069, 521288629, 88675123)
    L.oracle_rng_choose_index(st, 1)
    L.oracle_rng_next_u32(ref)  # rejected draw
    L.oracle_rng_next_u32(ref)  # 458299110 < 2^31 -> accepted
    assert list(st) == list(ref)


# ---------------------------------------------------------------- math.rs
def test_slab_hit_miss_and_inside():
    d = C.c_float()
    box = F6(0, 0, 0, 1, 1, 1)
    assert L.oracle_aabb_intersection_distance(box, ray((-1, 0.5, 0.5), (1, 0, 0)), C.byref(d)) == 1 and d.value == 1.0
    assert L.oracle_aabb_intersection_distance(box, ray((0.5, 0.5, 0.5), (0, 0, 1)), C.byref(d)) == 1 and d.value == 0.0  # inside: max(tmin, 0)
    assert L.oracle_aabb_intersection_distance(box, ray((2, 0.5, 0.5), (1, 0, 0)), C.byref(d)) == 0  # behind the ray: tmax < 0
    assert L.oracle_aabb_intersection_distance(box, ray((-1, 2, 0.5), (1, 0, 0)), C.byref(d)) == 0  # parallel and outside: inf slabs


def test_slab_nan_from_zero_times_inf_follows_min_max_nan_rules():
    # Origin exactly on a slab plane with a zero direction component: t1 = (0 - 0) * inf = NaN, t2 = +inf.
    # Rust's f32::min/max ignore the NaN operand, so min(t1, t2) = +inf, tmin = +inf and the box is MISSED
    # (math.rs:196-200 evaluated literally); an epsilon inside the slab it is hit.
    d = C.c_float()
    box = F6(0, 0, 0, 1, 1, 1)
    assert L.oracle_aabb_intersection_distance(box, ray((-1, 0.0, 0.5), (1, 0, 0)), C.byref(d)) == 0
    assert L.oracle_aabb_intersection_distance(box, ray((-1, 1.0, 0.5), (1, 0, 0)), C.byref(d)) == 0
    assert L.oracle_aabb_intersection_distance(box, ray((-1, 1e-6, 0.5), (1, 0, 0)), C.byref(d)) == 1 and d.value == 1.0


def test_schlick_and_fresnel():
    n, i = F3(0, 0, 1), F3(0, 0, -1)
    assert L.oracle_schlick(1.0, 1.5, n, i) == pytest.approx(0.04, rel=1e-6)  # ((1-1.5)/(1+1.5))^2 at normal incidence
    graze = F3(math.sqrt(1 - 0.01 ** 2), 0, -0.01)
    assert L.oracle_schlick(1.0, 1.5, n, graze) == pytest.approx(0.04 + 0.96 * 0.99 ** 5, rel=1e-5)
    # from the dense side beyond the critical angle: total internal reflection
    inside = F3(math.sin(math.radians(60)), 0, -math.cos(math.radians(60)))
    assert L.oracle_schlick(1.5, 1.0, n, inside) == 1.0
    # fresnel() picks the side from the sign of incident . normal (math.rs:167-175)
    assert L.oracle_fresnel(1.5, 1.0, n, i) == pytest.approx(0.04, rel=1e-6)
    assert L.oracle_fresnel(1.5, 1.0, n, F3(0, 0, 1)) == pytest.approx(0.04, rel=1e-6)


def test_ortho_branches():
    out = F3()
    L.oracle_ortho(F3(0, 1, 0), out)  # |x| < eps -> cross with X
    assert list(out) == [0.0, 0.0, -1.0]
    L.oracle_ortho(F3(1, 0, 0), out)  # |y| < eps -> cross with Y
    assert list(out) == [0.0, 0.0, 1.0]
    L.oracle_ortho(F3(1, 1, 0), out)  # |z| < eps -> cross with Z
    assert list(out) == [1.0, -1.0, 0.0]
    L.oracle_ortho(F3(1, 2, 3), out)  # generic: v x (-y, x, 0)
    assert np.allclose(list(out), np.cross([1, 2, 3], [-2, 1, 0]))
    assert abs(np.dot(list(out), [1, 2, 3])) < 1e-6


def test_sample_sphere_and_hemisphere_distribution():
    st = U4(5, 6, 7, 8)
    out = F3()
    pts = []
    for _ in range(4000):
        L.oracle_sample_sphere(st, out)
        pts.append(list(out))
    pts = np.array(pts)
    assert np.allclose(np.linalg.norm(pts, axis=1), 1.0, atol=1e-5)
    assert np.abs(pts.mean(axis=0)).max() < 0.05  # uniform on the sphere
    dirs = []
    normal = np.array([0.3, -0.5, 0.8]) / np.linalg.norm([0.3, -0.5, 0.8])
    for _ in range(4000):
        L.oracle_sample_hemisphere(st, F3(*normal), out)
        dirs.append(list(out))
    dirs = np.array(dirs)
    assert np.allclose(np.linalg.norm(dirs, axis=1), 1.0, atol=1e-4)
    cosines = dirs @ normal
    assert cosines.min() >= -1e-6
    assert cosines.mean() == pytest.approx(0.5, abs=0.02)  # uniform hemisphere (pdf 1/2pi), hence the 2|n.o| weight


def test_sample_cone_stays_inside_the_cone():
    st = U4(11, 12, 13, 14)
    out = F3()
    axis = np.array([0.0, 0.6, 0.8])
    cos_half = 0.9
    for _ in range(2000):
        L.oracle_sample_cone(st, F3(*axis), cos_half, out)
        v = np.array(list(out))
        assert abs(np.linalg.norm(v) - 1) < 1e-5
        assert v @ axis >= cos_half - 1e-6
    assert L.oracle_solid_angle(1.0) == 0.0
    assert L.oracle_solid_angle(0.0) == pytest.approx(2 * math.pi, rel=1e-6)


def test_blackbody_matches_plancks_law():
    lam, t = 550.0, 5000.0
    m = lam * 1e-9
    expect = 3.74183e-16 * m ** -5 / (math.exp(1.4388e-2 / (m * t)) - 1)
    assert L.oracle_blackbody(lam, t) == pytest.approx(expect, rel=2e-5)


# ---------------------------------------------------------------- shapes/mod.rs
def test_moller_trumbore_canonical():
    dist, u, v = C.c_float(), C.c_float(), C.c_float()
    tri = (F3(0, 0, 0), F3(1, 0, 0), F3(0, 1, 0))
    assert L.oracle_triangle_intersect(*tri, ray((0.25, 0.25, 1), (0, 0, -1)), C.byref(dist), C.byref(u), C.byref(v)) == 1
    assert (dist.value, u.value, v.value) == (1.0, 0.25, 0.25)
    # two-sided: the same hit from below
    assert L.oracle_triangle_intersect(*tri, ray((0.25, 0.25, -2), (0, 0, 1)), C.byref(dist), C.byref(u), C.byref(v)) == 1
    assert dist.value == 2.0
    # outside the triangle (u + v > 1), behind the origin, and closer than DIST_EPSILON
    assert L.oracle_triangle_intersect(*tri, ray((0.75, 0.75, 1), (0, 0, -1)), C.byref(dist), C.byref(u), C.byref(v)) == 0
    assert L.oracle_triangle_intersect(*tri, ray((0.25, 0.25, 1), (0, 0, 1)), C.byref(dist), C.byref(u), C.byref(v)) == 0
    assert L.oracle_triangle_intersect(*tri, ray((0.25, 0.25, 5e-5), (0, 0, -1)), C.byref(dist), C.byref(u), C.byref(v)) == 0


def test_triangle_determinant_cull_is_absolute():
    # |det| < 1e-4 is a miss whatever the scale (shapes/mod.rs:85): a 0.005-sided triangle has |det| <= 2.5e-5.
    dist, u, v = C.c_float(), C.c_float(), C.c_float()
    tiny = (F3(0, 0, 0), F3(0.005, 0, 0), F3(0, 0.005, 0))
    assert L.oracle_triangle_intersect(*tiny, ray((0.001, 0.001, 1), (0, 0, -1)), C.byref(dist), C.byref(u), C.byref(v)) == 0
    big = (F3(0, 0, 0), F3(0.05, 0, 0), F3(0, 0.05, 0))
    assert L.oracle_triangle_intersect(*big, ray((0.01, 0.01, 1), (0, 0, -1)), C.byref(dist), C.byref(u), C.byref(v)) == 1


def test_sphere_front_hit_miss_and_inside_quirk():
    dist, p = C.c_float(), F3()
    assert L.oracle_sphere_intersect(F3(0, 0, 5), 1.0, ray((0, 0, 0), (0, 0, 1)), C.byref(dist), p) == 1
    assert dist.value == 4.0 and list(p) == [0.0, 0.0, 4.0]
    assert L.oracle_sphere_intersect(F3(0, 0, 5), 1.0, ray((0, 2, 0), (0, 0, 1)), C.byref(dist), p) == 0  # d2 > r^2
    assert L.oracle_sphere_intersect(F3(0, 0, 5), 1.0, ray((0, 0, 0), (0, 0, -1)), C.byref(dist), p) == 0  # tca < 0
    # origin inside, heading towards the centre: collision returns the point BEHIND the origin, pyrite reports |P - o|
    assert L.oracle_sphere_intersect(F3(0, 0, 5), 1.0, ray((0, 0, 4.5), (0, 0, 1)), C.byref(dist), p) == 1
    assert dist.value == 0.5 and list(p) == [0.0, 0.0, 4.0]


# ---------------------------------------------------------------- spectra
def test_array_spectrum_interpolates_and_clamps():
    data = np.array([1.0, 3.0, 2.0], dtype=f32)
    get = lambda w: L.oracle_spectrum_get(0, 400.0, 600.0, data.ctypes.data, 3, w)  # noqa: E731
    assert get(300.0) == 1.0 and get(400.0) == 1.0  # w <= min -> first
    assert get(600.0) == 2.0 and get(900.0) == 2.0  # w >= max -> last
    assert get(450.0) == 2.0  # halfway between points 0 and 1
    assert get(500.0) == 3.0
    assert get(550.0) == 2.5


def test_curve_spectrum_is_zero_at_and_outside_the_end_points():
    pts = np.array([[400, 0.5], [450, 0.3], [500, 0.0], [550, 1.0], [600, 0.25]], dtype=f32)
    get = lambda w: L.oracle_spectrum_get(1, 0.0, 0.0, pts.ctypes.data, 5, w)  # noqa: E731
    assert get(399.0) == 0.0 and get(400.0) == 0.0 and get(600.0) == 0.0 and get(700.0) == 0.0  # math.rs:37-45
    assert get(450.0) == f32(0.3)  # exact x of an interior point
    assert get(525.0) == 0.5
    assert get(575.0) == pytest.approx(0.625)


# ---------------------------------------------------------------- refraction
def test_refract_normal_incidence_probabilities():
    st = U4(1, 2, 3, 4)
    out = F3()
    seen = set()
    for _ in range(200):
        p = L.oracle_refract(st, 1.5, 1.0, F3(0, 0, -1), F3(0, 0, 1), out)
        z = round(out[2], 6)
        seen.add(z)
        re = 0.04
        big_p = 0.25 + 0.5 * re
        if z > 0:  # reflected: weight re / P
            assert p == pytest.approx(re / big_p, rel=1e-5)
        else:  # transmitted straight through: weight (1 - re) / (1 - P)
            assert p == pytest.approx((1 - re) / (1 - big_p), rel=1e-5)
    assert seen == {1.0, -1.0}


def test_refract_total_internal_reflection_draws_nothing():
    st = U4(1, 2, 3, 4)
    before = list(st)
    out = F3()
    s, c = math.sin(math.radians(60)), math.cos(math.radians(60))
    # inside glass (direction along +normal side means leaving): the ray travels against -normal
    p = L.oracle_refract(st, 1.5, 1.0, F3(s, 0, c), F3(0, 0, 1), out)
    assert p == 1.0
    assert list(st) == before  # refractive.rs:60-63 returns before the rng.gen()
    assert np.allclose(list(out), [s, 0, -c], atol=1e-6)


# ---------------------------------------------------------------- film / camera / tiles
def test_stratified_wavelengths_one_per_stratum():
    st = U4(4, 3, 2, 1)
    out = (C.c_float * 10)()
    for _ in range(200):
        L.oracle_sample_wavelengths(st, 380.0, 400.0, 10, out)
        strata = sorted(int((w - 380.0) // 40.0) for w in out)
        assert strata == list(range(10))


def test_wavelength_to_grain():
    assert L.oracle_wavelength_to_grain(380.0, 380.0, 400.0, 64) == 0
    assert L.oracle_wavelength_to_grain(386.25, 380.0, 400.0, 64) == 1  # (6.25 * 0.16) = 1.0
    assert L.oracle_wavelength_to_grain(779.99, 380.0, 400.0, 64) == 63
    assert L.oracle_wavelength_to_grain(300.0, 380.0, 400.0, 64) == 0  # negative saturates to 0 (`as usize`)


def test_to_pixel_corners_and_rejection():
    px, py = C.c_uint32(), C.c_uint32()
    assert L.oracle_to_pixel(1920, 1080, -1.0, -0.5625, C.byref(px), C.byref(py)) == 1 and (px.value, py.value) == (0, 0)
    assert L.oracle_to_pixel(1920, 1080, 0.0, 0.0, C.byref(px), C.byref(py)) == 1 and (px.value, py.value) == (960, 540)
    assert L.oracle_to_pixel(1920, 1080, 0.999, 0.562, C.byref(px), C.byref(py)) == 1 and (px.value, py.value) == (1919, 1079)
    assert L.oracle_to_pixel(1920, 1080, 0.0, 0.6, C.byref(px), C.byref(py)) == 0  # |y| > height/width
    assert L.oracle_to_pixel(1920, 1080, 1.0, 0.0, C.byref(px), C.byref(py)) == 0  # x == width -> get_pixel None
    assert L.oracle_to_pixel(100, 200, 0.49, 0.99, C.byref(px), C.byref(py)) == 1 and (px.value, py.value) == (99, 199)  # vertical


def test_to_view_area_normalises_by_the_longer_side():
    out = (C.c_float * 4)()
    L.oracle_to_view_area(0, 0, 32, 32, 1920, 1080, out)
    assert list(out) == [-1.0, -0.5625, f32(32 / 960), f32(32 / 960)]
    L.oracle_to_view_area(960, 540, 32, 24, 1920, 1080, out)
    assert list(out)[:2] == [0.0, 0.0] and out[3] == f32(24 / 960)


def test_tile_order_is_centre_out_and_complete():
    order = (C.c_uint32 * 64)()
    n = L.oracle_tile_order(256, 256, 32, order, 64)
    assert n == 64 and sorted(order) == list(range(64))
    centre = {27, 28, 35, 36}  # the four tiles around the image centre share the smallest |centre|^2
    assert set(order[:4]) == centre
    assert order[-1] in (0, 7, 56, 63)
    n = L.oracle_tile_order(1920, 1080, 32, order, 0)
    assert n == 60 * 34


def test_pinhole_camera_ray():
    from pyrite_amd import abi

    cam = abi.PyrCamera()
    for k, v in enumerate([1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 3, 4, 5, 1]):  # translation only
        cam.cam_to_world[k] = v
    cam.view_plane, cam.focus_distance, cam.aperture = 2.0, 1.0, 0.0
    st = U4(1, 2, 3, 4)
    before = list(st)
    r6 = F6()
    L.oracle_ray_towards(C.byref(cam), st, 0.0, 0.0, r6)
    assert list(r6) == [3.0, 4.0, 5.0, 0.0, 0.0, -1.0]  # looks down -Z (cameras.rs:81)
    assert list(st) == before  # no lens draws when aperture == 0
    L.oracle_ray_towards(C.byref(cam), st, 1.0, 1.0, r6)
    d = np.array([0.5, -0.5, -1.0]) / np.linalg.norm([0.5, -0.5, -1.0])  # (x/vp*fd, -y/vp*fd, -fd)
    assert np.allclose(list(r6)[3:], d, atol=1e-6)
    cam.aperture = 0.02
    L.oracle_ray_towards(C.byref(cam), st, 0.0, 0.0, r6)
    assert list(st) != before  # thin lens draws two numbers
    assert (r6[0] - 3) ** 2 + (r6[1] - 4) ** 2 <= 0.02 + 1e-6  # lens radius^2 = aperture * u


def test_trig_kernels_are_within_two_ulp():
    # sin / cos / acos are evaluated by fixed f32 polynomial kernels on both sides (oracle.cpp "Transcendentals");
    # they must stay within 2 ulp of the correctly rounded value over the ranges the renderer uses.
    def max_ulp(f, ref, xs):
        out = np.array([f(float(x)) for x in xs], dtype=np.float64)
        exact = ref(xs.astype(np.float64))
        ulp = np.spacing(np.abs(exact.astype(f32))).astype(np.float64)
        return float(np.max(np.abs(out - exact) / ulp))

    rng = np.random.RandomState(1)
    angles = rng.uniform(0, 2 * math.pi, 20000).astype(f32)
    assert max_ulp(L.oracle_sin32, np.sin, angles) < 2.0
    assert max_ulp(L.oracle_cos32, np.cos, angles) < 2.0
    assert max_ulp(L.oracle_acos32, np.arccos, rng.uniform(-1, 1, 20000).astype(f32)) < 2.0
    assert (L.oracle_sin32(0.0), L.oracle_cos32(0.0), L.oracle_acos32(1.0)) == (0.0, 1.0, 0.0)
    assert L.oracle_acos32(-1.0) == f32(math.pi) and L.oracle_acos32(0.0) == f32(math.pi / 2)
