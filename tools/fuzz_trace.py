#!/usr/bin/env python3
"""Developer tool: python tools/fuzz_trace.py SEED X Y -- the oracle's paths through pixel (X, Y) of a fuzz scene
(ORACLE_DEBUG_PIXEL), and for every ray of them the closest hit of the oracle and of the GPU walk side by side: does a film
difference start at a hit (a tie, a missed primitive) or in the shading arithmetic?"""
import os, re, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))

seed, x, y = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
if os.environ.get("ORACLE_DEBUG_PIXEL") is None:  # the oracle reads the variable once: run again with it set, stderr kept
    env = dict(os.environ, ORACLE_DEBUG_PIXEL="%d,%d" % (x, y))
    out = subprocess.run([sys.executable, __file__] + sys.argv[1:], env=env, stderr=subprocess.PIPE, text=True)
    log = out.stderr
    sys.stdout.flush()
    open("/tmp/fuzz_trace_oracle.log", "w").write(log)
    sys.exit(out.returncode)

import oracle
from pyrite_amd import scenes
from pyrite_amd.renderer import World
from test_gpu_fuzz import random_project

project = random_project(1000 + seed)
world, cam, r, _ = scenes.build(project, seed=seed)
W, H = project["image"]["width"], project["image"]["height"]
# pass 1: the oracle's render with its stderr into a file we read back
saved = os.dup(2)
with open("/tmp/fuzz_trace_paths.log", "w") as f:
    os.dup2(f.fileno(), 2)
    cfilm = r.new_film(W, H)
    oracle.OracleScene(world).render(r, cam, cfilm, threads=1)
    os.dup2(saved, 2)
text = open("/tmp/fuzz_trace_paths.log").read()
print(text)
num = r"([-+0-9.einfa]+)"
rays, tags = [], []
for block in text.split("[oracle] tile")[1:]:
    head = block.split("\n")[0]
    prev = None
    cam_line = re.search(r"camera ray origin \(%s %s %s\) direction \(%s %s %s\)" % ((num,) * 6), block)
    if cam_line:
        rays.append([float(v) for v in cam_line.groups()])
        tags.append((head.strip()[:40], "camera"))
    for k, line in enumerate(l for l in block.split("\n")[1:] if l.strip().startswith("bounce")):
        pos = [float(v) for v in re.search(r"pos \(%s %s %s\)" % (num, num, num), line).groups()]
        inc = [float(v) for v in re.search(r"incident \(%s %s %s\)" % (num, num, num), line).groups()]
        if prev is not None:
            rays.append(prev + inc)
            tags.append((head.strip()[:40], k))
        prev = pos
if rays:
    rays = np.array(rays, dtype=np.float32)
    oh, _ = oracle.OracleScene(world).intersect(rays)
    gworld = World(world.flat)
    gh, _, _ = gworld.intersect(rays)
    for t, ray, a, b in zip(tags, rays, oh, gh):
        same = a.tobytes() == b.tobytes()
        print(t, "ray", ray, "\n   oracle", a, "\n   gpu   ", b, "" if same else "   <-- DIFFERENT")
