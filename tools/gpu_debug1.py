import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle
from pyrite_amd import scenes
from pyrite_amd.project import renderer

for name, proj in (("C1", scenes.c1_spheres(width=64, height=64, pixel_samples=1)), ("C2", scenes.c2_cornell(width=64, height=64, pixel_samples=1))):
    for bounces, ls in ((1, 0), (1, 4), (2, 0), (2, 4), (8, 0), (8, 4)):
        proj["renderer"] = renderer.simple(pixel_samples=1, tile_size=1, bounces=bounces, light_samples=ls)
        world, cam, r, film = scenes.build(proj, seed=5)
        osc = oracle.OracleScene(world)
        of = r.new_film(64, 64)
        osc.render(r, cam, of, threads=8)
        r.render(film, cam, world)
        a, b = of.grains[..., 0], film.grains[..., 0]
        num = np.sqrt(((a - b) ** 2).sum(-1)); den = np.sqrt((a ** 2).sum(-1)) + 1e-6
        e = (num / den).reshape(-1)
        bad = np.argwhere(e > 1e-3).reshape(-1)
        print(name, "bounces", bounces, "ls", ls, "pixels differing >1e-3:", len(bad), "of", len(e), "median", np.median(e))
        if len(bad) and bounces == 1 and ls == 4:
            for k in bad[:5]:
                y, x = divmod(int(k), 64)
                print("   pixel", x, y, "oracle", a[y, x][a[y, x] != 0][:4], "gpu", b[y, x][b[y, x] != 0][:4])
