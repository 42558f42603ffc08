"""Host-side mirror of the reference's renderer seam.

    Renderer::render(&self, film, task_runner, on_status, camera, world, resources)   pyrite/src/renderer/mod.rs:77-111

`World` is the frozen scene (World::from_project + Resources), `Camera` the perspective camera
(cameras.rs:20-27), `Renderer` the parameter block (renderer/mod.rs:18-28). `Renderer.render` hands the call to
libpyrite_gpu.so -- the HIP kernels are the only implementation; nothing here computes radiance on the CPU."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import abi
from ._lib import check, lib
from .compiler import FlatScene, camera_from_project, renderer_from_project
from .film import Film


class World:
    """World::from_project (world.rs:39-271) result. Device scenes are created lazily, one per device."""

    def __init__(self, flat: FlatScene):
        self.flat = flat
        self._desc = flat.desc()
        self._scenes = {}

    @classmethod
    def from_project(cls, world, base_dir="."):
        return cls(FlatScene().add_world(world, base_dir))

    @property
    def desc(self):
        return self._desc

    def scene(self, device=0, slot=0):
        """The PyrScene on `device` (`slot` > 0: a further copy on the same device, for the one-GPU multi-rank test rig)."""
        key = device if slot == 0 else (device, slot)
        if key not in self._scenes:
            handle = C.c_void_p()
            check(lib().pyr_scene_create(C.byref(self._desc), int(device), C.byref(handle)))
            self._scenes[key] = handle
        return self._scenes[key]

    def bvh_info(self, device=0):
        info = abi.PyrBvhInfo()
        check(lib().pyr_scene_bvh_info(self.scene(device), C.byref(info)))
        return {name: int(getattr(info, name)) for name, _ in info._fields_}

    def intersect(self, rays, device=0, want_counters=False):
        """World::intersect (world.rs:273-299) for float32 rays [n,6] -> (structured hits, kernel ms, counters|None)."""
        rays = np.ascontiguousarray(rays, dtype=np.float32).reshape(-1, 6)
        hits = np.zeros(len(rays), dtype=np.dtype([("distance", "<f4"), ("shape", "<u4"), ("u", "<f4"), ("v", "<f4")]))
        ms = C.c_float(0)
        counters = abi.PyrCounters()
        check(lib().pyr_scene_intersect(self.scene(device), rays.ctypes.data, len(rays), hits.ctypes.data, C.byref(ms),
                                        C.byref(counters) if want_counters else None))
        return hits, float(ms.value), (counters.as_dict() if want_counters else None)

    def close(self):
        for handle in self._scenes.values():
            lib().pyr_scene_destroy(handle)
        self._scenes = {}

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Camera:
    def __init__(self, pyr_camera: abi.PyrCamera):
        self.c = pyr_camera

    @classmethod
    def from_project(cls, cam):
        return cls(camera_from_project(cam))


class Renderer:
    """Renderer (renderer/mod.rs:18-28) for Algorithm::Simple. `seed` has no reference counterpart (the reference
    seeds from OS entropy, simple.rs:26-28)."""

    def __init__(self, pixel_samples, bounces=8, light_samples=4, spectrum_samples=10, spectrum_bins=64, spectrum_span=(380.0, 780.0),
                 tile_size=32, seed=1):
        self.pixel_samples, self.bounces, self.light_samples = int(pixel_samples), int(bounces), int(light_samples)
        self.spectrum_samples, self.spectrum_bins, self.spectrum_span = int(spectrum_samples), int(spectrum_bins), tuple(spectrum_span)
        self.tile_size, self.seed = int(tile_size), int(seed)

    @classmethod
    def from_project(cls, r, seed=1):
        return cls(seed=seed, **renderer_from_project(r))

    def new_film(self, width, height):  # main.rs:190-195
        return Film(width, height, self.spectrum_bins, self.spectrum_span)

    def num_tiles(self, width, height):  # make_tiles, renderer/algorithm.rs:158-166
        ts = self.tile_size
        return ((width + ts - 1) // ts) * ((height + ts - 1) // ts)

    def params(self, flags=0, tile_range=None, film_rows=None, share=None):
        """PyrRenderParams of this renderer. `tile_range` / `film_rows` restrict the call to raster tiles [a, b) and to a window of
        pixel rows; `share` (pyrite_amd.distributed.Share) sets tiles, stride and film layout at once."""
        p = abi.PyrRenderParams()
        p.bounces, p.pixel_samples, p.light_samples = self.bounces, self.pixel_samples, self.light_samples
        p.spectrum_samples, p.tile_size, p.flags, p.seed = self.spectrum_samples, self.tile_size, flags, self.seed
        if tile_range is not None:
            p.tile_begin, p.tile_end = int(tile_range[0]), int(tile_range[1])
        if film_rows is not None:
            p.film_row_begin, p.film_row_count = int(film_rows[0]), int(film_rows[1])
        if share is not None:
            share.apply(p)
        return p

    def path_info(self, world: World, device=0):
        """Which kernel a render of `world` with this renderer would run (pyr_scene_path_info): a dict of PyrPathInfo's fields."""
        info, params = abi.PyrPathInfo(), self.params()
        check(lib().pyr_scene_path_info(world.scene(device), C.byref(params), C.byref(info)))
        return {name: int(getattr(info, name)) for name, _ in info._fields_ if name != "reserved"}

    def render(self, film: Film, camera: Camera, world: World, on_status=None, device=0, counters=False, tile_range=None, film_rows=None,
               window=None, share=None):
        """Blocking render into a host Film (adds to it). Returns the PyrCounters dict when counters=True.
        `tile_range` restricts the call to raster tiles [a, b); with film_rows=(first_row, rows) the exposures go to `window`,
        a float32 [rows, width, bins, 2] array covering only those rows of the image `film` describes."""
        params = self.params(abi.PYR_FLAG_COUNTERS if counters else 0, tile_range, film_rows, share)
        desc = film.desc()
        if window is not None:
            assert window.flags["C_CONTIGUOUS"] and window.dtype == np.float32
            if share is not None:
                assert window.size == share.pixels(film.width) * film.bins * 2
            else:
                assert window.shape == (film_rows[1], film.width, film.bins, 2)
            check(lib().pyr_render_simple(world.scene(device), C.byref(camera.c), C.byref(desc), C.byref(params), window.ctypes.data,
                                          C.cast(None, abi.PyrProgressFn), None))
            return self.counters(world, device) if counters else None
        if on_status is not None:
            cb = abi.PyrProgressFn(lambda user, percent, message: on_status(int(percent), message.decode()))
        else:
            cb = C.cast(None, abi.PyrProgressFn)
        grains = np.ascontiguousarray(film.grains)
        check(lib().pyr_render_simple(world.scene(device), C.byref(camera.c), C.byref(desc), C.byref(params), grains.ctypes.data, cb, None))
        if grains is not film.grains:
            film.grains[...] = grains
        if counters:
            out = abi.PyrCounters()
            check(lib().pyr_scene_counters(world.scene(device), C.byref(out)))
            return out.as_dict()
        return None

    def render_device(self, film_ptr, film_desc, camera: Camera, world: World, stream=0, device=0, flags=0, tile_range=None,
                      film_rows=None, share=None):
        """Asynchronous render into DEVICE memory (`film_ptr` = data_ptr of a float32 tensor laid out as the parameters say:
        [rows, w, bins, 2] pixel rows, or -- with a `share` of tile blocks -- [tiles, ts + 2, ts + 2, bins, 2])."""
        params = self.params(flags, tile_range, film_rows, share)
        check(lib().pyr_render_simple_device(world.scene(device), C.byref(camera.c), C.byref(film_desc), C.byref(params),
                                             C.c_void_p(film_ptr), C.c_void_p(stream)))

    def render_multi(self, film: Film, camera: Camera, world: World, devices, on_status=None):
        """pyr_render_simple_multi: one process driving `devices` (a list of device indices; a repeated index is the one-GPU
        test rig). Blocking; adds into the host Film."""
        handles = (C.c_void_p * len(devices))(*[world.scene(d, slot=i) for i, d in enumerate(devices)])
        params = self.params()
        desc = film.desc()
        cb = abi.PyrProgressFn(lambda user, percent, message: on_status(int(percent), message.decode())) if on_status else C.cast(None, abi.PyrProgressFn)
        grains = np.ascontiguousarray(film.grains)
        check(lib().pyr_render_simple_multi(handles, len(devices), C.byref(camera.c), C.byref(desc), C.byref(params), grains.ctypes.data, cb, None))
        if grains is not film.grains:
            film.grains[...] = grains

    def counters(self, world: World, device=0):
        out = abi.PyrCounters()
        check(lib().pyr_scene_counters(world.scene(device), C.byref(out)))
        return out.as_dict()
