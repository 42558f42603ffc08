#!/usr/bin/env python3
"""Developer tool (GPU box): the sharded render of one rank -- with and without a forced one-rank RCCL communicator -- against the
plain render at several image sizes: do the gathered blocks arrive whole?   python tools/native_one_rank.py [spp]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from pyrite_amd import abi, scenes, distributed as pdist

spp = int(sys.argv[1]) if len(sys.argv) > 1 else 2
dev = torch.device("cuda", 0)
for W, H in ((640, 360), (960, 540), (1280, 720), (1600, 900), (1920, 1080), (2560, 1440)):
    project = scenes.c3_mesh_in_box(W, H, spp, segments=64, sides=64)
    world, cam, r, _ = scenes.build(project, seed=1)
    world.scene(0)
    desc = abi.PyrFilmDesc(W, H, r.spectrum_bins, r.spectrum_span[0], r.spectrum_span[1] - r.spectrum_span[0])
    stream = torch.cuda.current_stream(dev)
    plain = torch.zeros((H, W, r.spectrum_bins, 2), dtype=torch.float32, device=dev)
    r.render_device(plain.data_ptr(), desc, cam, world, stream=stream.cuda_stream, device=0)
    torch.cuda.synchronize(dev)
    tiles = ((W + 31) // 32) * ((H + 31) // 32)
    print("%d x %d: %d tiles = %.1f MB of blocks; plain weight %.0f" % (W, H, tiles, tiles * 34 * 34 * r.spectrum_bins * 8 / 1e6, float(plain[..., 1].sum(dtype=torch.float64))), flush=True)
    for forced in ("0", "1"):
        os.environ["PYRITE_FORCE_RCCL"] = forced
        comm = pdist.NativeSharded(0)
        film = torch.zeros_like(plain)
        comm.render(r, cam, world, desc, film, stream=stream.cuda_stream)
        torch.cuda.synchronize(dev)
        comm.status()
        same = bool(torch.equal(film[..., 1], plain[..., 1]))
        msg = "   PYRITE_FORCE_RCCL=%s (uses_rccl %s): weight %.0f, weights equal %s" % (forced, comm.uses_rccl, float(film[..., 1].sum(dtype=torch.float64)), same)
        if not same:
            rows = (film[..., 1] != plain[..., 1]).any(dim=2).any(dim=1).nonzero().flatten()
            msg += "; rows that differ: %d .. %d (%d rows)" % (int(rows[0]), int(rows[-1]), len(rows))
        print(msg, flush=True)
        comm.close()
    world.close()
