#!/usr/bin/env python3
"""Developer tool (GPU box): World::intersect on the full C3 mesh, GPU against the oracle, on millions of rays: bit-exact
(distance, shape, u, v) or a tie?   python tools/big_hit_check.py [million rays]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle
from pyrite_amd import scenes

millions = float(sys.argv[1]) if len(sys.argv) > 1 else 4.0
world, cam, r, _ = scenes.build(scenes.c3_mesh_in_box(96, 54, 1), seed=1)
world.scene(0)
osc = oracle.OracleScene(world)
total = ties = other = 0
t0 = time.time()
for batch in range(int(millions * 2)):  # half a million rays at a time
    rng = np.random.default_rng(1000 + batch)
    n = 500000
    o = rng.uniform([-55, 0, 0], [0, 55, 54], (n, 3))
    d = rng.normal(size=(n, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
    rays = np.concatenate([o, d], axis=1).astype(np.float32)
    oh, _ = osc.intersect(rays)
    gh, _, _ = world.intersect(rays)
    diff = np.nonzero((oh["distance"] != gh["distance"]) | (oh["shape"] != gh["shape"]) | (oh["u"] != gh["u"]) | (oh["v"] != gh["v"]))[0]
    tie = diff[(oh["distance"][diff] == gh["distance"][diff])]
    total += n; ties += len(tie); other += len(diff) - len(tie)
    for k in diff[:3]:
        if k not in tie: print("   NOT A TIE: ray", rays[k], "oracle", oh[k], "gpu", gh[k])
    print("%9d rays: %d ties, %d other differences (%.0f s)" % (total, ties, other, time.time() - t0), flush=True)
print("C3 mesh (819,212 triangles), %d incoherent rays: %d bit-exact, %d ties (same f32 distance, another triangle), %d other" % (total, total - ties - other, ties, other))
