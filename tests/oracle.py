"""ctypes wrapper around oracle/liboracle.so -- the CPU restatement of the reference used as the parity checker.
Test infrastructure only (see oracle/oracle.h): nothing under pyrite_amd/ imports this."""
import ctypes as C
import os
import subprocess

import numpy as np

from pyrite_amd import abi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB = os.path.join(ORACLE_DIR, "liboracle.so")
_lib = None

U4 = C.c_uint32 * 4
F3 = C.c_float * 3
F6 = C.c_float * 6


_lib_path = None


def use_library(path):
    """Load another build of the oracle (bench.py: oracle/liboracle_native.so) instead of the portable one. Before first use."""
    global _lib_path
    assert _lib is None, "the oracle library is already loaded"
    _lib_path = path


def lib():
    global _lib
    if _lib is None:
        src = os.path.join(ORACLE_DIR, "oracle.cpp")
        if _lib_path is None and (not os.path.exists(LIB) or (os.path.exists(src) and os.path.getmtime(src) > os.path.getmtime(LIB))):
            subprocess.check_call(["make", "-C", ORACLE_DIR, "-s"])
        L = C.CDLL(_lib_path or LIB)
        L.oracle_last_error.restype = C.c_char_p
        L.oracle_scene_create.argtypes = [C.POINTER(abi.PyrSceneDesc), C.POINTER(C.c_void_p)]
        L.oracle_scene_destroy.argtypes = [C.c_void_p]
        L.oracle_render_simple.argtypes = [C.c_void_p, C.POINTER(abi.PyrCamera), C.POINTER(abi.PyrFilmDesc), C.POINTER(abi.PyrRenderParams),
                                           C.c_void_p, C.c_int, C.POINTER(abi.PyrCounters)]
        L.oracle_intersect.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.POINTER(abi.PyrCounters)]
        L.oracle_bvh_num_nodes.argtypes = [C.c_void_p]
        L.oracle_bvh_num_nodes.restype = C.c_uint32
        L.oracle_bvh_node.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_float), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
        L.oracle_rng_seed.argtypes = [C.c_uint64, C.c_uint32, C.c_uint64, U4]
        L.oracle_rng_next_u32.argtypes = [U4]
        L.oracle_rng_next_u32.restype = C.c_uint32
        L.oracle_rng_gen_f32.argtypes = [U4]
        L.oracle_rng_gen_f32.restype = C.c_float
        L.oracle_rng_gen_range_f32.argtypes = [U4, C.c_float, C.c_float]
        L.oracle_rng_gen_range_f32.restype = C.c_float
        L.oracle_rng_gen_range_usize.argtypes = [U4, C.c_uint32]
        L.oracle_rng_gen_range_usize.restype = C.c_uint32
        L.oracle_rng_choose_index.argtypes = [U4, C.c_uint32]
        L.oracle_rng_choose_index.restype = C.c_uint32
        L.oracle_aabb_intersection_distance.argtypes = [F6, F6, C.POINTER(C.c_float)]
        L.oracle_schlick.argtypes = [C.c_float, C.c_float, F3, F3]
        L.oracle_schlick.restype = C.c_float
        L.oracle_fresnel.argtypes = [C.c_float, C.c_float, F3, F3]
        L.oracle_fresnel.restype = C.c_float
        L.oracle_ortho.argtypes = [F3, F3]
        L.oracle_sample_sphere.argtypes = [U4, F3]
        L.oracle_sample_hemisphere.argtypes = [U4, F3, F3]
        L.oracle_sample_cone.argtypes = [U4, F3, C.c_float, F3]
        L.oracle_solid_angle.argtypes = [C.c_float]
        L.oracle_solid_angle.restype = C.c_float
        for name in ("oracle_sin32", "oracle_cos32", "oracle_acos32"):
            getattr(L, name).argtypes = [C.c_float]
            getattr(L, name).restype = C.c_float
        L.oracle_blackbody.argtypes = [C.c_float, C.c_float]
        L.oracle_blackbody.restype = C.c_float
        L.oracle_triangle_intersect.argtypes = [F3, F3, F3, F6, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_float)]
        L.oracle_sphere_intersect.argtypes = [F3, C.c_float, F6, C.POINTER(C.c_float), F3]
        L.oracle_spectrum_get.argtypes = [C.c_uint32, C.c_float, C.c_float, C.c_void_p, C.c_uint32, C.c_float]
        L.oracle_spectrum_get.restype = C.c_float
        L.oracle_refract.argtypes = [U4, C.c_float, C.c_float, F3, F3, F3]
        L.oracle_refract.restype = C.c_float
        L.oracle_sample_wavelengths.argtypes = [U4, C.c_float, C.c_float, C.c_uint32, C.c_void_p]
        L.oracle_wavelength_to_grain.argtypes = [C.c_float, C.c_float, C.c_float, C.c_uint32]
        L.oracle_wavelength_to_grain.restype = C.c_uint32
        L.oracle_to_pixel.argtypes = [C.c_uint32, C.c_uint32, C.c_float, C.c_float, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
        L.oracle_to_view_area.argtypes = [C.c_uint32] * 6 + [C.c_float * 4]
        L.oracle_ray_towards.argtypes = [C.POINTER(abi.PyrCamera), U4, C.c_float, C.c_float, F6]
        L.oracle_tile_order.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint32]
        L.oracle_tile_order.restype = C.c_uint32
        L.oracle_run_program.argtypes = [C.c_void_p, C.c_uint32, C.c_float, F3, F3, C.c_float * 2, C.POINTER(C.c_int)]
        L.oracle_run_program.restype = C.c_float
        L.oracle_film_develop.argtypes = [C.POINTER(abi.PyrFilmDesc), C.c_void_p, C.POINTER(abi.PyrDevelopParams), C.c_void_p]
        L.oracle_texture_get.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_float, C.c_float, C.c_void_p]
        L.oracle_quat_from_cols.argtypes = [F3, F3, F3, C.c_float * 4]
        L.oracle_quat_rotate.argtypes = [C.c_float * 4, F3, F3]
        L.oracle_surface_data.argtypes = [C.c_void_p, F6, C.c_float, F3, C.c_float * 2, C.c_float * 4, F3]
        _lib = L
    return _lib


def film_develop(film, step_size=2.0, filter=None, white=None):
    """The oracle's restatement of main.rs:315-418 for a pyrite_amd Film -> uint8 [h, w, 3]."""
    from pyrite_amd.develop import develop_params

    p, keep = develop_params(film, step_size, filter, white)
    desc = film.desc()
    grains = np.ascontiguousarray(film.grains)
    out = np.zeros((film.height, film.width, 3), dtype=np.uint8)
    _check(lib().oracle_film_develop(C.byref(desc), grains.ctypes.data, C.byref(p), out.ctypes.data))
    del keep
    return out


def texture_get(texels, x, y):
    """texels: float32 [h, w] (mono) or [h, w, 4] (colour)."""
    texels = np.ascontiguousarray(texels, dtype=np.float32)
    channels = 1 if texels.ndim == 2 else texels.shape[2]
    out = np.zeros(channels, dtype=np.float32)
    lib().oracle_texture_get(channels, texels.shape[1], texels.shape[0], texels.ctypes.data, x, y, out.ctypes.data)
    return out


def quat_from_cols(c0, c1, c2):
    out = (C.c_float * 4)()
    lib().oracle_quat_from_cols(F3(*c0), F3(*c1), F3(*c2), out)
    return np.array(out[:], dtype=np.float32)


def quat_rotate(q, v):
    out = F3()
    lib().oracle_quat_rotate((C.c_float * 4)(*q), F3(*v), out)
    return np.array(out[:], dtype=np.float32)


class OracleError(RuntimeError):
    pass


def _check(status):
    if status != 0:
        raise OracleError("oracle error %d: %s" % (status, lib().oracle_last_error().decode()))


class OracleScene:
    def __init__(self, world):
        """`world` is a pyrite_amd.renderer.World (only its flattened description is used)."""
        self.world = world
        self.handle = C.c_void_p()
        _check(lib().oracle_scene_create(C.byref(world.desc), C.byref(self.handle)))

    def render(self, renderer, camera, film, threads=1, tile_range=None, film_rows=None, window=None, share=None):
        """Adds one render into `film.grains`, or -- with film_rows=(first_row, rows) -- into `window`, a float32
        [rows, width, bins, 2] array covering only those rows of the image `film` describes, or -- with a
        pyrite_amd.distributed.Share -- into `window` laid out as the share says (pixel rows or ringed tile blocks)."""
        params = renderer.params(0, tile_range, film_rows, share)
        desc = film.desc()
        counters = abi.PyrCounters()
        target = film.grains if window is None else window
        assert target.flags["C_CONTIGUOUS"] and target.dtype == np.float32
        if share is not None:
            assert window is not None and target.size == share.pixels(film.width) * film.bins * 2
        elif film_rows is not None:
            assert target.shape == (film_rows[1], film.width, film.bins, 2)
        _check(lib().oracle_render_simple(self.handle, C.byref(camera.c), C.byref(desc), C.byref(params), target.ctypes.data, threads,
                                          C.byref(counters)))
        return counters.as_dict()

    def intersect(self, rays):
        rays = np.ascontiguousarray(rays, dtype=np.float32).reshape(-1, 6)
        hits = np.zeros(len(rays), dtype=np.dtype([("distance", "<f4"), ("shape", "<u4"), ("u", "<f4"), ("v", "<f4")]))
        counters = abi.PyrCounters()
        _check(lib().oracle_intersect(self.handle, rays.ctypes.data, len(rays), hits.ctypes.data, C.byref(counters)))
        return hits, counters.as_dict()

    def surface_data(self, ray, wavelength=550.0):
        """(normal, texture, frame quaternion, shading normal after the normal map) at the ray's first hit, or None."""
        n, t, f, sn = F3(), (C.c_float * 2)(), (C.c_float * 4)(), F3()
        if not lib().oracle_surface_data(self.handle, F6(*[float(x) for x in ray]), wavelength, n, t, f, sn):
            return None
        return np.array(n[:], dtype=np.float32), np.array(t[:], dtype=np.float32), np.array(f[:], dtype=np.float32), np.array(sn[:], dtype=np.float32)

    def bvh_nodes(self):
        n = lib().oracle_bvh_num_nodes(self.handle)
        out = []
        for i in range(n):
            aabb = (C.c_float * 6)()
            size, item = C.c_uint32(), C.c_uint32()
            _check(lib().oracle_bvh_node(self.handle, i, aabb, C.byref(size), C.byref(item)))
            out.append((np.array(aabb[:], dtype=np.float32), size.value, item.value))
        return out

    def run_program(self, program, wavelength, normal=(0, 0, 1), incident=(0, 0, -1), texture=(0, 0)):
        used = C.c_int(0)
        v = lib().oracle_run_program(self.handle, program, wavelength, F3(*normal), F3(*incident), (C.c_float * 2)(*texture), C.byref(used))
        return float(v), bool(used.value)

    def close(self):
        if self.handle:
            lib().oracle_scene_destroy(self.handle)
            self.handle = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
