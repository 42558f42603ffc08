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
elif which == "TEX":  # the reference's textures project (interpreter programs, textures, normal maps)
    project = scenes.textures_reference_example(os.path.join(ROOT, "tests", "golden", "textures"), w, h, spp)
elif which == "SPHERES":
    project = scenes.spheres_example(w, h, spp)
elif which == "C5":
    project = scenes.c3_mesh_in_box(w, h, spp, glass=True, bounces=20)
else:
    project = scenes.c3_mesh_in_box(w, h, spp)
world, cam, r, film = scenes.build(project, seed=1)
lib = _lib.lib()
fn = lib.pyr_debug_phase_profile32
fn.restype = C.c_int
fn.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
out = (C.c_ulonglong * 32)()
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
if out[14] and not out[15]:  # interpreter builds: contribute_pending behind the phases
    print("contribute (interpreter) %5.1f %% of wave cycles (not in the phases above) | mean lanes with something to apply %5.1f / 64 | %8.0f cycles per turn | %.2f turns per sample"
          % (100.0 * out[12] / (total + out[12]), out[13] / float(out[14]), out[12] / float(out[14]), out[14] * 64.0 / samples))
elif any(out[12:16]):  # render_kernel_px: waves in two roles
    logic, trav = float(max(out[14], 1)), float(max(out[15], 1))
    print("path exchange: traversal waves %.3g wave cycles: %.1f %% in traversal steps, %.1f %% exchanging paths, the rest waiting for work" % (trav, 100.0 * cyc[3] / trav, 100.0 * out[12] / trav))
    print("               logic waves     %.3g wave cycles: %.1f %% in EXPOSE / SHADE / NEE, %.1f %% exchanging paths, the rest waiting for work" % (logic, 100.0 * sum(cyc[0:3]) / logic, 100.0 * out[13] / logic))
    tt, lt = float(max(out[16], 1)), float(max(out[24], 1))
    print("               seats of a traversal wave when a turn starts (%.0f turns): walking %.1f, vacant %.1f, ray ended and no slot in the queue back %.1f" % (tt, out[17] / tt, out[18] / tt, out[19] / tt))
    print("               seats of a logic wave when a turn starts (%.0f turns): SHADE %.1f, NEE %.1f, EXPOSE / NEW %.1f, waiting for a slot of the traversal queue %.1f, vacant %.1f" % (
        lt, out[25] / lt, out[26] / lt, out[27] / lt, out[28] / lt, out[29] / lt))
