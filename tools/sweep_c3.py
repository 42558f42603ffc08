#!/usr/bin/env python3
"""Developer tool: time the C3 (or C5) render under different tuning variables, scene built once.

    python tools/sweep_c3.py [C3|C5] spp VAR=a,b,c [VAR2=...]     e.g.  PYRITE_LDS_STACK=4,8,12,40 PYRITE_SM_STEPS=4,8
Every combination is rendered PYRITE_SWEEP_RENDERS (6) times at 1920x1080 into a device film; the median of the later half is
reported (HIP events)."""
import itertools
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from pyrite_amd import abi, scenes  # noqa: E402

which, spp = sys.argv[1], int(sys.argv[2])
sweeps = [(a.split("=")[0], a.split("=")[1].split(",")) for a in sys.argv[3:]]
W, H = 1920, 1080
project = scenes.c3_mesh_in_box(W, H, spp, glass=(which == "C5"), bounces=20 if which == "C5" else None)
world, cam, r, _ = scenes.build(project, seed=1)
world.scene(0)
print("bvh", world.bvh_info(), flush=True)
dev = torch.device("cuda", 0)
film = torch.zeros((H, W, r.spectrum_bins, 2), dtype=torch.float32, device=dev)
desc = abi.PyrFilmDesc(W, H, r.spectrum_bins, r.spectrum_span[0], r.spectrum_span[1] - r.spectrum_span[0])
stream = torch.cuda.current_stream(dev)
for combo in itertools.product(*[vals for _, vals in sweeps]) if sweeps else [()]:
    for (name, _), val in zip(sweeps, combo):
        os.environ[name] = val
    # PYRITE_SWEEP_RENDERS renders, the median of the later half reported: the first render of a process runs while the clock
    # still ramps (2.07 -> 2.39 GHz over ~0.5 s, tools/clock_watch.py), and a 2 % difference between builds can be just that
    renders = int(os.environ.get("PYRITE_SWEEP_RENDERS", "6"))
    times = []
    for _ in range(renders):
        film.zero_()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(stream)
        r.render_device(film.data_ptr(), desc, cam, world, stream=stream.cuda_stream, device=0)
        b.record(stream)
        torch.cuda.synchronize(dev)
        times.append(a.elapsed_time(b))
    later = sorted(times[len(times) // 2:])
    ms = later[len(later) // 2]
    label = " ".join("%s=%s" % (n, v) for (n, _), v in zip(sweeps, combo))
    print("%-40s %9.2f ms  %7.1f Msamples/s  weight %.6g" % (label, ms, W * H * spp / ms / 1e3, float(film[..., 1].sum(dtype=torch.float64))), flush=True)
