#!/usr/bin/env python3
"""Render a scene on the GPU and write the developed sRGB image: python tools/render_image.py C2 256 256 64 out.png"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pyrite_amd import scenes
from pyrite_amd.develop import develop, save_png
from pyrite_amd.project import blackbody

name, w, h, spp, out = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
builders = {"C1": scenes.c1_spheres, "C2": scenes.c2_cornell, "C3": scenes.c3_mesh_in_box, "spheres": scenes.spheres_example,
            "diamonds": scenes.diamonds_example, "lamps": scenes.lamps_example, "C5": lambda **kw: scenes.c3_mesh_in_box(glass=True, bounces=20, **kw)}
world, cam, r, film = scenes.build(builders[name](width=w, height=h, pixel_samples=spp), seed=1)
t = time.time(); r.render(film, cam, world); print("render %.2f s" % (time.time() - t))
white = blackbody(4000) if name in ("C1", "C2", "C3", "C5") else None  # cornell.lua:16
save_png(out, develop(film, white=white))
print("wrote", out)
