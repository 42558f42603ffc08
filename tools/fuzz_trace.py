#!/usr/bin/env python3
"""Developer tool (GPU box): python tools/fuzz_trace.py scene|soup|mesh SEED [X Y] -- where does the GPU film of a fuzz case differ from
the oracle's, and why? Finds the differing pixels (or takes one), prints the oracle's paths through the first of them
(ORACLE_DEBUG_PIXEL) and walks the camera ray, every extension ray and every unblocked shadow ray of those paths through World::intersect on both sides:
a film difference that starts at two primitives met at the same f32 distance is a tie, anything else is a defect.
tools/fuzz_sched.py says which scheduler differs; for the synchronous walk of a big scene (two-child tree) give the pixel and
set PYRITE_WIDE_BVH=0 ORACLE_DEBUG_PIXEL=X,Y so that World::intersect walks that tree too."""
import os, re, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))

kind, seed = sys.argv[1], int(sys.argv[2])
pixel = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else None
if len(sys.argv) == 4 and sys.argv[3] == "all":  # every ray of every sample of the image (ORACLE_DEBUG_PIXEL=all): a difference the film does not show
    pixel = "all"                                # (a path counter gave it away): run as  ORACLE_DEBUG_PIXEL=all python tools/fuzz_trace.py KIND SEED all

import oracle
from pyrite_amd import scenes
from pyrite_amd.project import camera, transform, vector
from pyrite_amd.renderer import Camera, Renderer, World
from test_gpu_fuzz import random_project, random_soup
from test_gpu_parity import rel_l2


def build():
    if kind in ("scene", "mesh"):
        project = random_project(500000 + seed, knot=True) if kind == "mesh" else random_project(1000 + seed)
        world, cam, r, _ = scenes.build(project, seed=seed)
        return world, cam, r, project["image"]["width"], project["image"]["height"]
    world = World(random_soup(2000 + seed))
    r = Renderer(pixel_samples=3, bounces=6, light_samples=2, spectrum_samples=5, tile_size=16, seed=seed)
    cam = Camera.from_project(camera.perspective(fov=60, transform=transform.look_at(**{"from": vector(0, -9, 1), "to": vector(0, 0, 0), "up": vector(z=1)})))
    return world, cam, r, 40, 30


world, cam, r, W, H = build()
if pixel is None:  # pass 1: which pixels differ; then this script again for the first of them, with the oracle's printing switched on
    cfilm = r.new_film(W, H)
    oracle.OracleScene(world).render(r, cam, cfilm, threads=8)
    gfilm = r.new_film(W, H)
    r.render(gfilm, cam, World(world.flat))
    e = rel_l2(gfilm, cfilm).reshape(H, W)
    wdiff = (gfilm.grains[..., 1] != cfilm.grains[..., 1]).any(axis=-1)
    bad = np.argwhere((e > float(os.environ.get("FUZZ_TRACE_TOL", "1e-5"))) | wdiff)  # FUZZ_TRACE_TOL=1e-6: a difference the tests' tolerance hides (path counters gave it away)
    print("%s %d: %d x %d, %d differing pixels:" % (kind, seed, W, H, len(bad)), [(int(x), int(y), float(e[y, x]), bool(wdiff[y, x])) for y, x in bad][:8])
    sys.stdout.flush()
    if len(bad):
        y, x = bad[0]
        g, c = gfilm.grains[y, x], cfilm.grains[y, x]
        for b in np.nonzero((g[:, 0] != c[:, 0]) | (g[:, 1] != c[:, 1]))[0][:12]:
            print("   bin %d: gpu %.9g (weight %g)  oracle %.9g (weight %g)  ratio %.9g" % (b, g[b, 0], g[b, 1], c[b, 0], c[b, 1], g[b, 0] / c[b, 0] if c[b, 0] else float("nan")))
        sys.stdout.flush()
        env = dict(os.environ, ORACLE_DEBUG_PIXEL="%d,%d" % (x, y))
        sys.exit(subprocess.run([sys.executable, __file__, kind, str(seed), str(x), str(y)], env=env).returncode)
    sys.exit(0)

# pass 2: the oracle's render with its stderr (the paths through the pixel) in a file
saved = os.dup(2)
with open("/tmp/fuzz_trace_paths.log", "w") as f:
    os.dup2(f.fileno(), 2)
    cfilm = r.new_film(W, H)
    oracle.OracleScene(world).render(r, cam, cfilm, threads=1)
    os.dup2(saved, 2)
text = open("/tmp/fuzz_trace_paths.log").read()
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
            if np.isfinite(pos).all():  # the shadow rays the oracle found unblocked (blocked ones are not kept by trace_direct)
                rays.append(pos + [float(v) for v in m.groups()[1:]])
                tags.append((head, "bounce %d light %d (unblocked in the oracle)" % (k, j)))
# every shadow ray of the pixel's samples as the oracle traced it, blocked ones too (oracle.cpp trace_direct prints them)
shadow = re.findall(r"\[shadow\] origin \(%s %s %s\) direction \(%s %s %s\) hit (\d) distance %s limit %s blocked (\d)" % ((num,) * 6 + (num, num)), text)
for m in shadow:
    rays.append([float(v) for v in m[:6]])
    tags.append(("shadow ray", "oracle: hit %s at %s, limit %s, blocked %s" % (m[6], m[7], m[8], m[9])))
print("pixel", pixel, ":", len(text.split("[oracle] tile")) - 1, "samples,", len(rays), "rays")


def inconsistent_spheres(ray):
    """Spheres whose intersection routine (collision's, shapes/mod.rs:57-74, in f32 as both sides compute it) reports a hit NEARER than
    the entry of the sphere's own bounding box (math.rs:184-207): l.l - tca^2 has lost its digits (a ray passing thousands of radii
    away). The reference only tests a shape whose box is entered before the closest hit so far (spatial/bvh.rs:201-230), so whether
    such a hit counts depends on the ORDER in which its tree is walked -- the answer is the reference's tree's, not the scene's."""
    f32 = np.float32
    o, d = ray[:3].astype(f32), ray[3:].astype(f32)
    with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
        inv = (f32(1) / d).astype(f32)
        found = []
        for i, (cx, cy, cz, rad) in enumerate(np.asarray(world.flat.spheres, dtype=f32).reshape(-1, 4)):
            c = np.array([cx, cy, cz], dtype=f32)
            tmin, tmax = f32(-np.inf), f32(np.inf)
            for k in range(3):
                t1, t2 = f32(f32(c[k] - rad - o[k]) * inv[k]), f32(f32(c[k] + rad - o[k]) * inv[k])
                tmin, tmax = np.fmax(tmin, np.fmin(t1, t2)), np.fmin(tmax, np.fmax(t1, t2))
            if not (tmax >= tmin and tmax >= 0):
                continue
            l = (c - o).astype(f32)
            ll = f32(f32(f32(l[0] * l[0]) + f32(l[1] * l[1])) + f32(l[2] * l[2]))
            tca = f32(f32(f32(l[0] * d[0]) + f32(l[1] * d[1])) + f32(l[2] * d[2]))
            d2 = f32(ll - f32(tca * tca))
            r2 = f32(rad * rad)
            if tca < 0 or d2 > r2:
                continue
            near = f32(tca - np.sqrt(f32(r2 - d2)))
            if near < max(tmin, f32(0)):
                found.append("sphere %d: routine says %.9g (l.l %.9g - tca^2 = %.9g against r^2 %.9g), its box is entered at %.9g" % (i, near, ll, d2, r2, max(tmin, f32(0))))
    return found
verdict = "no ray of the oracle's paths is answered differently: the difference is in the shading arithmetic (or in a shadow ray)"
if rays:
    rays = np.array(rays, dtype=np.float32)
    oh, _ = oracle.OracleScene(world).intersect(rays)
    gh, _, _ = World(world.flat).intersect(rays)
    for t, ray, a, b in zip(tags, rays, oh, gh):
        if a.tobytes() != b.tobytes():
            tie = float(a[0]) == float(b[0])
            garbage = [] if tie else inconsistent_spheres(ray)
            print(t, "ray", ray, "\n   oracle", a, "\n   gpu   ", b, "   <-- " + ("TIE: same f32 distance, another primitive" if tie else "DIFFERENT"))
            for line in garbage:
                print("   " + line)
            verdict = "tie" if tie else ("order-dependent in the reference: a sphere hit nearer than its own box's entry" if garbage else "DEFECT: the closest hit differs")
            break
if verdict.startswith("no ray"):
    # The closest hits agree; the render's shadow test is an any-hit walk, though: a sphere hit without digits that lies inside the
    # blocking limit while the sphere's own box begins beyond the oracle's closest hit blocks here (the box is within the cut-off's margin) and
    # is skipped by the reference when it has met that closest hit first -- its tree's order again.
    for m in shadow:
        ray = np.array([float(v) for v in m[:6]], dtype=np.float32)
        limit, oracle_blocked = np.float32(float(m[8])), m[9] == "1"
        for line in inconsistent_spheres(ray):
            near = np.float32(float(re.search(r"routine says ([-+0-9.einfa]+)", line).group(1)))
            if near > np.float32(1e-4) and np.float32(near * near) < limit and not oracle_blocked:
                print("shadow ray", ray, "limit", limit, "oracle: hit at", m[7], "(not a blocker)\n   " + line + " -- inside the limit: an any-hit walk that enters the box counts it")
                verdict = "order-dependent in the reference: a sphere hit nearer than its own box's entry decides a shadow ray"
print("verdict:", verdict)
if verdict.startswith("no ray"):
    print(text)
