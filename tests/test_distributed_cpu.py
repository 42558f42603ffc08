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
    # 1920x1080, 32-pixel tiles: 60 x 34 tiles; contiguous: rank 1 of 8 starts mid-row 4 and ends mid-row 8
    ((rng, (lo, rows)),) = pdist.plan(1920, 1080, 32, 8, "contiguous")[1]
    assert rng == (255, 510) and lo == 4 * 32 - 1 and lo + rows == 9 * 32 + 1
    ((rng, (lo, rows)),) = pdist.plan(1920, 1080, 32, 8, "contiguous")[7]
    assert lo + rows == 1080  # clamped at the image edge
    assert pdist.plan(64, 64, 32, 1) == [[((0, 4), (0, 64))]]
    assert pdist.window_rows((2, 2), 2, 32, 64) == (0, 0)
    assert pdist.plan(64, 32, 32, 4, "contiguous")[3] == []  # two tiles, four ranks: the last two have nothing to do
    # cyclic (the default for more than one rank): tile rows dealt round-robin, cut in pieces when rows per rank are few
    shares = pdist.plan(1920, 1080, 32, 2)
    assert [len(s) for s in shares] == [17, 17] and shares[1][0] == ((60, 120), (31, 34)) and shares[0][0] == ((0, 60), (0, 33))
    assert shares[1][16] == ((33 * 60, 34 * 60), (33 * 32 - 1, 1080 - 33 * 32 + 1))  # the last, 24-pixel tile row
    shares = pdist.plan(1920, 1080, 32, 8)  # 34 rows on 8 ranks: half rows, 68 bands
    assert sorted(len(s) for s in shares) == [8, 8, 8, 8, 9, 9, 9, 9]
    assert shares[0][0] == ((0, 30), (0, 33)) and shares[1][0] == ((30, 60), (0, 33)) and shares[2][0] == ((60, 90), (31, 34))
    for n in (2, 3, 8):
        tiles = sorted(rng for share in pdist.plan(1920, 1080, 32, n) for rng, _ in share)
        assert tiles[0][0] == 0 and tiles[-1][1] == 2040 and all(a[1] == b[0] for a, b in zip(tiles, tiles[1:]))  # every tile once
    assert pdist.window_height(pdist.plan(1920, 1080, 32, 2)[0]) == 33 + 16 * 34


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world_size, port, out_path, sharding):
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
            view = window.numpy()
            assert view.flags["C_CONTIGUOUS"] and view.shape[0] == rows[1]
            sc.render(r, cam, film, threads=1, tile_range=tile_range, film_rows=rows, window=view)

        result = pdist.render_sharded(render_window, film.width, film.height, film.bins, r.tile_size, torch.device("cpu"), sharding=sharding)
        if rank == 0:
            np.save(out_path, result.numpy())
        else:
            assert result is None
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world_size,sharding", [(2, "cyclic"), (3, "cyclic"), (2, "contiguous"), (3, "contiguous")])
def test_gloo_ranks_reproduce_the_single_process_film(world_size, sharding):
    import oracle
    from pyrite_amd import scenes

    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "film.npy")
        mp.spawn(_worker, args=(world_size, _free_port(), out, sharding), nprocs=world_size, join=True)
        sharded = np.load(out)
    world, cam, r, film = scenes.build(scenes.c2_cornell(40, 36, 2), seed=7)
    r.tile_size = 8
    oracle.OracleScene(world).render(r, cam, film, threads=1)
    assert np.array_equal(sharded[..., 1], film.grains[..., 1])  # every exposure arrived exactly once
    assert np.allclose(sharded, film.grains, rtol=1e-6, atol=1e-12)
    assert film.total_weight() == 40 * 36 * 2 * 10
