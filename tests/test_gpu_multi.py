"""The multi-GPU entry points on the one-GPU box (reference: independent tiles, simple.rs:36-55, results
collected on one thread, renderer/mod.rs:181-183).

  * real RCCL: a ONE-rank communicator made by ncclCommInitRank (PYRITE_FORCE_RCCL=1) -- dlopen + every dlsym, the id
    hand-over, the status agreement (ncclAllReduce), the grouped self ncclSend / ncclRecv of the block buffer with its
    trailer, the stream ordering, the assembly from the gathered copy;
  * several ranks: RCCL refuses two ranks on one device, so the multi-rank flow runs against an in-process stand-in for
    librccl (tests/fake_rccl) in a process of its own -- including ranks that fail before and after the agreement and a message
    that never arrives, none of which may hang."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from pyrite_amd import scenes

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _librccl_mapped():
    with open("/proc/self/maps") as f:
        return any("librccl" in line for line in f)


def test_one_rank_rccl_communicator_gathers_through_send_and_recv(gpu_lib, monkeypatch):
    import torch

    from pyrite_amd import distributed as pdist

    monkeypatch.setenv("PYRITE_FORCE_RCCL", "1")
    world, cam, r, whole = scenes.build(scenes.c2_cornell(64, 48, 4), seed=6)
    r.tile_size = 16
    r.render(whole, cam, world)
    dev = torch.device("cuda", 0)
    stream = torch.cuda.current_stream(dev).cuda_stream
    comm = pdist.NativeSharded(0)
    assert comm.uses_rccl and _librccl_mapped()
    for _ in range(2):  # the second call reuses the communicator's buffers
        film = torch.zeros((48, 64, r.spectrum_bins, 2), dtype=torch.float32, device=dev)
        comm.render(r, cam, world, whole.desc(), film, stream=stream)
        torch.cuda.synchronize(dev)
        comm.status()
        got = film.cpu().numpy()
        assert np.array_equal(got[..., 1], whole.grains[..., 1])
        assert np.allclose(got, whole.grains, rtol=1e-5)
    comm.close()
    world.close()


def test_one_rank_rccl_gather_of_a_full_size_film_arrives_whole(gpu_lib, monkeypatch):
    """1920 x 1080 = 2040 ringed tile blocks = 1.2 GB from one rank. Round 3 found that RCCL 2.26.6 delivers about half of a
    single ncclSend / ncclRecv pair above 1 GiB and reports nothing (rows 544-1079 of the film stayed empty, the trailer word
    behind them with it): pyr_render_simple_sharded posts one message per 256 MiB since. Everything on the device: the films
    are compared there (1.06 GB each), and the launch-failure word must still arrive from behind the last block."""
    import torch

    from pyrite_amd import abi
    from pyrite_amd import distributed as pdist
    from pyrite_amd._lib import PyriteGpuError

    monkeypatch.setenv("PYRITE_FORCE_RCCL", "1")
    W, H = 1920, 1080
    world, cam, r, _ = scenes.build(scenes.c3_mesh_in_box(W, H, 1, segments=32, sides=32), seed=2)
    dev = torch.device("cuda", 0)
    stream = torch.cuda.current_stream(dev)
    desc = abi.PyrFilmDesc(W, H, r.spectrum_bins, r.spectrum_span[0], r.spectrum_span[1] - r.spectrum_span[0])
    plain = torch.zeros((H, W, r.spectrum_bins, 2), dtype=torch.float32, device=dev)
    r.render_device(plain.data_ptr(), desc, cam, world, stream=stream.cuda_stream, device=0)
    comm = pdist.NativeSharded(0)
    assert comm.uses_rccl
    film = torch.zeros_like(plain)
    comm.render(r, cam, world, desc, film, stream=stream.cuda_stream)
    torch.cuda.synchronize(dev)
    comm.status()
    assert float(plain[..., 1].sum(dtype=torch.float64)) >= 0.9999 * W * H * r.spectrum_samples
    assert torch.equal(film[..., 1], plain[..., 1]), "part of the film never arrived"
    assert torch.allclose(film[..., 0], plain[..., 0], rtol=1e-5, atol=0.0)
    monkeypatch.setenv("PYRITE_TEST_FAIL_RENDER_RANK", "0")  # the trailer sits behind 1.2 GB of blocks: it must arrive too
    film.zero_()
    comm.render(r, cam, world, desc, film, stream=stream.cuda_stream)
    torch.cuda.synchronize(dev)
    with pytest.raises(PyriteGpuError, match="could not launch"):
        comm.status()
    comm.close()
    world.close()
    del film, plain


def test_one_rank_rccl_communicator_reports_a_film_the_kernels_flagged(gpu_lib, monkeypatch):
    """The error path through the real library: the spectral tape shrunk under its bound (PYRITE_TEST_TAPE_OPS) makes the
    kernels set their overflow word; it travels in the trailer through ncclSend / ncclRecv and pyr_comm_status reports it."""
    import torch

    from pyrite_amd import distributed as pdist
    from pyrite_amd._lib import PyriteGpuError

    monkeypatch.setenv("PYRITE_FORCE_RCCL", "1")
    monkeypatch.setenv("PYRITE_SCHEDULER", "sm")
    world, cam, r, whole = scenes.build(scenes.c2_cornell(32, 32, 4), seed=1)
    r.render(whole, cam, world)
    dev = torch.device("cuda", 0)
    stream = torch.cuda.current_stream(dev).cuda_stream
    comm = pdist.NativeSharded(0)
    assert comm.uses_rccl
    film = torch.zeros((32, 32, r.spectrum_bins, 2), dtype=torch.float32, device=dev)
    monkeypatch.setenv("PYRITE_TEST_TAPE_OPS", "3")
    comm.render(r, cam, world, whole.desc(), film, stream=stream)
    torch.cuda.synchronize(dev)
    with pytest.raises(PyriteGpuError, match="rank 0.*spectral tape"):
        comm.status()
    monkeypatch.delenv("PYRITE_TEST_TAPE_OPS")
    film.zero_()
    comm.render(r, cam, world, whole.desc(), film, stream=stream)  # the communicator is alive, the word was cleared
    torch.cuda.synchronize(dev)
    comm.status()
    assert np.array_equal(film.cpu().numpy()[..., 1], whole.grains[..., 1])
    comm.close()
    world.close()


def test_multi_rank_flow_against_the_in_process_stand_in(gpu_lib):
    subprocess.check_call(["make", "-s", "-C", os.path.join(HERE, "fake_rccl")])
    env = dict(os.environ)
    env.pop("PYRITE_FORCE_RCCL", None)
    out = subprocess.run([sys.executable, os.path.join(HERE, "fake_rccl", "run_cases.py")], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    results = json.loads(out.stdout.strip().splitlines()[-1])
    cases = {k: v for k, v in results.items() if not k.endswith("_seconds")}
    assert len(cases) == 10  # eight small ones + the 8-rank plan at BASELINE's full size and a message lost in it
    failed = {k: v for k, v in cases.items() if v != "ok"}
    assert not failed, json.dumps(failed, indent=1) + out.stderr[-1500:]
