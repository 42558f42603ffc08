#!/usr/bin/env python3
"""Developer tool (GPU box): GPU clock and power while the C3 render runs in a loop -- is the kernel running at the clock the
arithmetic assumes?   python tools/clock_watch.py [seconds] [spp]"""
import os, subprocess, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from pyrite_amd import abi, scenes

seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 8.0
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 128
W, H = 1920, 1080
world, cam, r, _ = scenes.build(scenes.c3_mesh_in_box(W, H, spp), seed=1)
world.scene(0)
dev = torch.device("cuda", 0)
film = torch.zeros((H, W, r.spectrum_bins, 2), dtype=torch.float32, device=dev)
desc = abi.PyrFilmDesc(W, H, r.spectrum_bins, r.spectrum_span[0], r.spectrum_span[1] - r.spectrum_span[0])
stream = torch.cuda.current_stream(dev)
samples, stop = [], False


def watch():
    while not stop:
        try:
            out = subprocess.run(["/opt/rocm/bin/rocm-smi", "--showclocks", "--showpower", "--csv"], capture_output=True, text=True, timeout=5).stdout
            samples.append((time.time(), out.strip().replace("\n", " | ")))
        except Exception as e:  # noqa: BLE001
            samples.append((time.time(), "rocm-smi failed: %s" % e))
            break
        time.sleep(0.3)


print("idle:", subprocess.run(["/opt/rocm/bin/rocm-smi", "--showclocks", "--showpower", "--csv"], capture_output=True, text=True).stdout.strip().replace("\n", " | "), flush=True)
t = threading.Thread(target=watch)
t.start()
t0, n, times = time.time(), 0, []
while time.time() - t0 < seconds:
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(stream)
    r.render_device(film.data_ptr(), desc, cam, world, stream=stream.cuda_stream, device=0)
    b.record(stream)
    torch.cuda.synchronize(dev)
    times.append(a.elapsed_time(b))
    n += 1
stop = True
t.join()
print("%d renders, ms: first %.1f, median %.1f, last %.1f" % (n, times[0], sorted(times)[n // 2], times[-1]))
for ts, line in samples[:: max(1, len(samples) // 10)]:
    print("  +%.1f s  %s" % (ts - t0, line[:400]))
