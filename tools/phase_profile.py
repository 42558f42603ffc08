#!/usr/bin/env python3
"""Developer tool: where the stage-scheduled render kernel spends its wave cycles.

Needs the -DPYR_PHASE_PROFILE build of the library (csrc/variants/lib_prof.so):
    PYRITE_GPU_LIB=pyrite_amd/csrc/variants/lib_prof.so python tools/phase_profile.py [C3|C2|C5] [w h spp]
Prints per phase: share of wave cycles, mean active lanes while the phase code runs, cycles per turn."""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pyrite_amd import _lib, scenes  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "C3"
w, h, spp = (int(a) for a in sys.argv[2:5]) if len(sys.argv) > 4 else (480, 270, 32)
if which == "C2":
    project = scenes.c2_cornell(w, h, spp)
elif which == "C5":
    project = scenes.c3_mesh_in_box(w, h, spp, glass=True, bounces=20)
else:
    project = scenes.c3_mesh_in_box(w, h, spp)
world, cam, r, film = scenes.build(project, seed=1)
lib = _lib.lib()
fn = lib.pyr_debug_phase_profile
fn.restype = C.c_int
fn.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
out = (C.c_ulonglong * 16)()
r.render(film, cam, world)  # warm up (BVH build, upload)
fn(out, 1)
t = time.time()
r.render(film, cam, world)
dt = time.time() - t
fn(out, 1)
if which == "C2" and os.environ.get("PYRITE_SCHEDULER", "sync") == "sync":  # the synchronous walk: lap timers of lane 0 of every wave
    names = ["start_sample", "extension traversal", "shade (surface, scatter, reflectance)", "light sample", "shadow traversal",
             "light accounting", "bounce tail / ended lanes", "expose"]
    total = float(sum(out[0:8])) or 1.0
    print("%s %dx%d x %d spp: %.3f s incl. film transfer" % (which, w, h, spp, dt))
    for name, c in zip(names, out[0:8]):
        print("%-40s %5.1f %% of wave cycles" % (name, 100.0 * c / total))
    sys.exit(0)
cyc, lanes, turns = list(out[0:4]), list(out[4:8]), list(out[8:12])
total = float(sum(cyc)) or 1.0
samples = w * h * spp
print("%s %dx%d x %d spp: %.3f s incl. film transfer, %.1f Msamples/s" % (which, w, h, spp, dt, samples / dt / 1e6))
for i, name in enumerate(["EXPOSE/NEW", "SHADE", "NEE", "TRAV"]):
    n = max(turns[i], 1)
    per_turn = cyc[i] / (n / (int(os.environ.get("PYRITE_SM_STEPS", "8")) if i == 3 else 1))
    print("%-11s %5.1f %% of wave cycles | mean active lanes %5.1f / 64 | %8.0f cycles per turn | %.2f turns per sample"
          % (name, 100.0 * cyc[i] / total, lanes[i] / n, per_turn, turns[i] * 64.0 / samples / (8 if i == 3 else 1)))
if any(out[12:16]):  # render_kernel_split: waves in two roles
    print("split scheduler: logic waves idle %.1f %% of their cycles, traversal waves idle %.1f %%" % (100.0 * out[12] / max(out[14], 1), 100.0 * out[13] / max(out[15], 1)))
    print("                 wave cycles logic %.3g, traversal %.3g" % (out[14], out[15]))
