"""The N>1 path on CPU: two gloo ranks shard the tiles, render their film windows (with the CPU oracle standing in for
the HIP kernels -- the sharding, windows and the single gather are what is under test) and rank 0 must end up with the
film a single process renders."""
import os
import socket
import sys
import tempfile

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from pyrite_amd import distributed as pdist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_partition_and_windows():
    assert pdist.partition_tiles(2040, 8) == [(255 * r, 255 * (r + 1)) for r in range(8)]
    assert pdist.partition_tiles(10, 4) == [(0, 3), (3, 6), (6, 8), (8, 10)]
    assert pdist.partition_tiles(2, 4) == [(0, 1), (1, 2), (2, 2), (2, 2)]
    # 1920x1080, 32-pixel tiles: 60 x 34 tiles; rank 1 of 8 starts mid-row 4 and ends mid-row 8
    (rng, (lo, rows)) = pdist.plan(1920, 1080, 32, 8)[1]
    assert rng == (255, 510) and lo == 4 * 32 - 1 and lo + rows == 9 * 32 + 1
    (rng, (lo, rows)) = pdist.plan(1920, 1080, 32, 8)[7]
    assert lo + rows == 1080  # clamped at the image edge
    assert pdist.plan(64, 64, 32, 1) == [((0, 4), (0, 64))]
    assert pdist.window_rows((2, 2), 2, 32, 64) == (0, 0)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world_size, port, out_path):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle
    from pyrite_amd import scenes

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world_size)
    try:
        world, cam, r, film = scenes.build(scenes.c2_cornell(40, 36, 2), seed=7)
        r.tile_size = 8  # 5 x 5 tiles (last row 4 pixels high): both ranks end mid-row
        sc = oracle.OracleScene(world)

        def render_window(tile_range, rows, window):
            view = window.numpy()[: rows[1]]
            assert view.flags["C_CONTIGUOUS"]
            sc.render(r, cam, film, threads=1, tile_range=tile_range, film_rows=rows, window=view)

        result = pdist.render_sharded(render_window, film.width, film.height, film.bins, r.tile_size, torch.device("cpu"))
        if rank == 0:
            np.save(out_path, result.numpy())
        else:
            assert result is None
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world_size", [2, 3])
def test_two_gloo_ranks_reproduce_the_single_process_film(world_size):
    import oracle
    from pyrite_amd import scenes

    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "film.npy")
        mp.spawn(_worker, args=(world_size, _free_port(), out), nprocs=world_size, join=True)
        sharded = np.load(out)
    world, cam, r, film = scenes.build(scenes.c2_cornell(40, 36, 2), seed=7)
    r.tile_size = 8
    oracle.OracleScene(world).render(r, cam, film, threads=1)
    assert np.array_equal(sharded[..., 1], film.grains[..., 1])  # every exposure arrived exactly once
    assert np.allclose(sharded, film.grains, rtol=1e-6, atol=1e-12)
    assert film.total_weight() == 40 * 36 * 2 * 10
