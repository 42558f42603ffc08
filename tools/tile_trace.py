#!/usr/bin/env python3
"""Developer tool (GPU box): python tools/tile_trace.py ROW COL SEED SPP [X Y] -- one tile of the full-size C3 image rendered by
the oracle and by the GPU: which pixels differ, and why? As tools/fuzz_trace.py does for a fuzz case: prints the oracle's paths
through the first differing pixel (ORACLE_DEBUG_PIXEL) and walks the camera ray, every extension ray and every unblocked shadow
ray of those paths through World::intersect on both sides. Two primitives met at the same f32 distance = a tie."""
import os, re, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
row, col, seed, spp = (int(a) for a in sys.argv[1:5])
pixel = (int(sys.argv[5]), int(sys.argv[6])) if len(sys.argv) > 6 else None
import oracle
from pyrite_amd import scenes
from test_gpu_parity import rel_l2

W, H = 1920, 1080
world, cam, r, _ = scenes.build(scenes.c3_mesh_in_box(W, H, spp), seed=seed)
tile = (W // 32) * row + col
sc = oracle.OracleScene(world)
if pixel is None:
    cfilm, gfilm = r.new_film(W, H), r.new_film(W, H)
    cc = sc.render(r, cam, cfilm, threads=8, tile_range=(tile, tile + 1))
    gc = r.render(gfilm, cam, world, tile_range=(tile, tile + 1), counters=True)
    print("counters equal:", all(gc[k] == cc[k] for k in ("samples", "extension_rays", "shadow_rays", "shaded_hits", "exposures")))
    e = rel_l2(gfilm, cfilm).reshape(H, W)
    wdiff = (gfilm.grains[..., 1] != cfilm.grains[..., 1]).any(axis=-1)
    bad = np.argwhere((e > 1e-5) | wdiff)
    print("tile (%d, %d) seed %d, %d spp: %d differing pixels:" % (row, col, seed, spp, len(bad)), [(int(x), int(y), float(e[y, x]), bool(wdiff[y, x])) for y, x in bad][:8], flush=True)
    if len(bad):
        y, x = bad[0]
        g, c = gfilm.grains[y, x], cfilm.grains[y, x]
        for b in np.nonzero((g[:, 0] != c[:, 0]) | (g[:, 1] != c[:, 1]))[0][:12]:
            print("   bin %d: gpu %.9g (weight %g)  oracle %.9g (weight %g)  ratio %.9g" % (b, g[b, 0], g[b, 1], c[b, 0], c[b, 1], g[b, 0] / c[b, 0] if c[b, 0] else float("nan")), flush=True)
        env = dict(os.environ, ORACLE_DEBUG_PIXEL="%d,%d" % (x, y))
        sys.exit(subprocess.run([sys.executable, __file__] + sys.argv[1:5] + [str(x), str(y)], env=env).returncode)
    sys.exit(0)
saved = os.dup(2)
with open("/tmp/tile_trace_paths.log", "w") as f:
    os.dup2(f.fileno(), 2)
    cfilm = r.new_film(W, H)
    sc.render(r, cam, cfilm, threads=1, tile_range=(tile, tile + 1))
    os.dup2(saved, 2)
text = open("/tmp/tile_trace_paths.log").read()
num = r"([-+0-9.einfa]+)"
rays, tags = [], []
for block in text.split("[oracle] tile")[1:]:
    head = block.split("\n")[0].strip()[:34]
    prev = None
    cam_line = re.search(r"camera ray origin \(%s %s %s\) direction \(%s %s %s\)" % ((num,) * 6), block)
    if cam_line:
        rays.append([float(v) for v in cam_line.groups()])
        tags.append((head, "camera"))
    for k, line in enumerate(l for l in block.split("\n")[1:] if l.strip().startswith("bounce")):
        pos = [float(v) for v in re.search(r"pos \(%s %s %s\)" % (num, num, num), line).groups()]
        inc = [float(v) for v in re.search(r"incident \(%s %s %s\)" % (num, num, num), line).groups()]
        if prev is not None and np.isfinite(prev).all():
            rays.append(prev + inc)
            tags.append((head, "bounce %d" % k))
        prev = pos
        for j, m in enumerate(re.finditer(r"\[color \d+ prob %s dir \(%s %s %s\)\]" % (num, num, num, num), line)):
            if np.isfinite(pos).all():
                rays.append(pos + [float(v) for v in m.groups()[1:]])
                tags.append((head, "bounce %d light %d (unblocked in the oracle)" % (k, j)))
print("pixel", pixel, ":", len(text.split("[oracle] tile")) - 1, "samples,", len(rays), "rays")
verdict = "no ray of the oracle's paths is answered differently: the difference is in the shading arithmetic (or in a blocked shadow ray)"
if rays:
    rays = np.array(rays, dtype=np.float32)
    oh, _ = sc.intersect(rays)
    gh, _, _ = world.intersect(rays)
    for t, ray, a, b in zip(tags, rays, oh, gh):
        if a.tobytes() != b.tobytes():
            tie = float(a[0]) == float(b[0])
            print(t, "ray", ray, "\n   oracle", a, "\n   gpu   ", b, "   <-- " + ("TIE: same f32 distance, another primitive" if tie else "DIFFERENT"))
            verdict = "tie" if tie else "DEFECT: the closest hit differs"
            break
print("verdict:", verdict)
if verdict.startswith("no ray"):
    print(text)
