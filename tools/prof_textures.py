"""Profiling target: the reference's own `textures` project (pyrite/test/textures/textures.lua: colour textures, normal maps,
a fresnel mirror / diffuse mix -- the program interpreter's path, render_kernel_sm<false, true, ...>) at the project's own
1024 x 512 x 400 spp, rendered three times into a device film; HIP events around each launch.
    python tools/prof_textures.py [spp]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from pyrite_amd import abi, scenes  # noqa: E402

spp = int(sys.argv[1]) if len(sys.argv) > 1 else 400
W, H = 1024, 512
project = scenes.textures_reference_example(os.path.join(ROOT, "tests", "golden", "textures"), W, H, spp)
world, cam, r, _ = scenes.build(project, seed=1)
world.scene(0)
dev = torch.device("cuda", 0)
film = torch.zeros((H, W, r.spectrum_bins, 2), dtype=torch.float32, device=dev)
desc = abi.PyrFilmDesc(W, H, r.spectrum_bins, r.spectrum_span[0], r.spectrum_span[1] - r.spectrum_span[0])
stream = torch.cuda.current_stream(dev)
best = None
for k in range(3):
    film.zero_()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(stream)
    r.render_device(film.data_ptr(), desc, cam, world, stream=stream.cuda_stream, device=0)
    b.record(stream)
    torch.cuda.synchronize(dev)
    ms = a.elapsed_time(b)
    best = ms if best is None else min(best, ms)
film.zero_()
r.render_device(film.data_ptr(), desc, cam, world, stream=stream.cuda_stream, device=0, flags=abi.PYR_FLAG_COUNTERS)
torch.cuda.synchronize(dev)
c = r.counters(world, 0)
samples = W * H * spp
nbytes = 32 * c["box_tests"] + 36 * c["triangle_tests"] + 16 * (c["sphere_tests"] + c["plane_tests"]) + 52 * c["shaded_hits"] + 16 * c["exposures"]
print("textures example %dx%d x %d spp (program interpreter, textures, normal maps): best of 3 %.2f ms = %.1f Msamples/s; algorithmic %.0f B per sample -> %.0f GB/s = %.3f of 8 TB/s; weight %.6g"
      % (W, H, spp, best, samples / best / 1e3, nbytes / samples, nbytes / best / 1e6, nbytes / best / 1e6 / 8000.0, float(film[..., 1].sum(dtype=torch.float64))))
