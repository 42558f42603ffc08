#!/usr/bin/env python3
"""Developer tool: the C3 scene with a mesh material that runs interpreter programs -- a fresnel mix of a mirror and an rgb()-coloured
diffuse coat (the coated ball of pyrite/test/spheres/spheres.lua with an rgb colour): what a textured production mesh looks like to
the kernels (the interpreter build of the stage scheduler on a scene that does not live in LDS). Times a 1920 x 1080 render for
the settings given as VAR=a,b arguments (every combination; a new scene per combination: PYRITE_HIT_TAPE is read at creation).
    python tools/bench_interp_mesh.py [spp] PYRITE_SM_LANES=16,32 PYRITE_HIT_TAPE=1,0"""
import itertools
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from pyrite_amd import abi, scenes  # noqa: E402
from pyrite_amd.project import fresnel, material, mix, rgb  # noqa: E402

args = sys.argv[1:]
spp = int(args.pop(0)) if args and args[0].isdigit() else 32
sweeps = [(a.split("=")[0], a.split("=")[1].split(",")) for a in args]
W, H = 1920, 1080
coat = {"surface": mix(material.mirror(color=1), material.diffuse(color=rgb(0.8, 0.45, 0.2)), fresnel(1.5))}
dev = torch.device("cuda", 0)
for combo in itertools.product(*[v for _, v in sweeps]) if sweeps else [()]:
    for (name, _), val in zip(sweeps, combo):
        os.environ[name] = val
    world, cam, r, _ = scenes.build(scenes.c3_mesh_in_box(W, H, spp, mesh_material=coat), seed=1)
    world.scene(0)
    film = torch.zeros((H, W, r.spectrum_bins, 2), dtype=torch.float32, device=dev)
    desc = abi.PyrFilmDesc(W, H, r.spectrum_bins, r.spectrum_span[0], r.spectrum_span[1] - r.spectrum_span[0])
    stream = torch.cuda.current_stream(dev)
    times = []
    for _ in range(4):
        film.zero_()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(stream)
        r.render_device(film.data_ptr(), desc, cam, world, stream=stream.cuda_stream, device=0)
        b.record(stream)
        torch.cuda.synchronize(dev)
        times.append(a.elapsed_time(b))
    ms = sorted(times[1:])[1]
    print("%-50s %9.2f ms %8.1f Msamples/s  weight %.6g" % (" ".join("%s=%s" % (n, v) for (n, _), v in zip(sweeps, combo)) or "defaults", ms, W * H * spp / ms / 1e3,
                                                            float(film[..., 1].sum(dtype=torch.float64))), flush=True)
    world.close()
