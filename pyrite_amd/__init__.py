"""pyrite_amd -- MI355X-native drop-in for the hot path of Ogeon/pyrite's camera-to-light (`simple`) renderer.

Layout:
  project.py   the reference's project operator surface (lib.lua / project/mod.rs) as Python callables
  compiler.py  lowering of that tree: program compiler, material flattening, world flattening, OBJ ingest
  renderer.py  the Renderer::render seam, backed by csrc/libpyrite_gpu.so (HIP, gfx950) through the C ABI
  film.py      Film {acc, weight} grains
  scenes.py    the BASELINE.json configurations C1..C5 as project trees
  csrc/        HIP kernels + C-ABI host code (include/pyrite_gpu.h)
"""
from .film import Film  # noqa: F401
from .renderer import Camera, Renderer, World  # noqa: F401
