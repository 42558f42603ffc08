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
    out = subprocess.run([sys.executable, os.path.join(HERE, "fake_rccl", "run_cases.py")], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    results = json.loads(out.stdout.strip().splitlines()[-1])
    cases = {k: v for k, v in results.items() if not k.endswith("_seconds")}
    assert len(cases) == 8
    failed = {k: v for k, v in cases.items() if v != "ok"}
    assert not failed, json.dumps(failed, indent=1) + out.stderr[-1500:]
