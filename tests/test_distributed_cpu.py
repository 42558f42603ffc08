"""The N>1 path on CPU: gloo ranks shard the tiles, render their film buffers (with the CPU oracle standing in for the HIP
kernels -- the plan, the buffers, the single gather and the assembly are what is under test) and rank 0 must end up with the
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

from pyrite_amd import abi
from pyrite_amd import distributed as pdist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_partition_and_windows():
    assert pdist.partition_tiles(2040, 8) == [(255 * r, 255 * (r + 1)) for r in range(8)]
    assert pdist.partition_tiles(10, 4) == [(0, 3), (3, 6), (6, 8), (8, 10)]
    assert pdist.partition_tiles(2, 4) == [(0, 1), (1, 2), (2, 2), (2, 2)]
    # 1920x1080, 32-pixel tiles: 60 x 34 tiles; contiguous: rank 1 of 8 starts mid-row 4 and ends mid-row 8
    s = pdist.plan(1920, 1080, 32, 8, "contiguous")[1]
    assert (s.tile_begin, s.tile_end, s.tile_stride, s.layout) == (255, 510, 1, abi.PYR_FILM_ROWS)
    assert s.rows[0] == 4 * 32 - 1 and sum(s.rows) == 9 * 32 + 1
    assert sum(pdist.plan(1920, 1080, 32, 8, "contiguous")[7].rows) == 1080  # clamped at the image edge
    (s,) = pdist.plan(64, 64, 32, 1)
    assert (s.tile_begin, s.tile_end, s.rows, s.pixels(64)) == (0, 4, (0, 64), 64 * 64)
    assert pdist.window_rows((2, 2), 2, 32, 64) == (0, 0)
    assert pdist.plan(64, 32, 32, 4, "contiguous")[3].tile_count == 0  # two tiles, four ranks: the last two have nothing to do


def test_tile_plan_deals_every_tile_once():
    # the default for more than one rank: tiles dealt round-robin, one ringed block per tile
    for n in (2, 3, 8):
        shares = pdist.plan(1920, 1080, 32, n)
        assert all(s.layout == abi.PYR_FILM_TILE_BLOCKS and s.tile_stride == n for s in shares)
        tiles = sorted(t for s in shares for t in s.tiles())
        assert tiles == list(range(2040))
        assert max(s.tile_count for s in shares) - min(s.tile_count for s in shares) <= 1
    shares = pdist.plan(1920, 1080, 32, 8)
    assert [s.tile_count for s in shares] == [255] * 8
    assert shares[0].pixels(1920) == 255 * 34 * 34
    # the gather moves (34/32)^2 of the film, not the 2.1x of round 1's half-row bands
    assert sum(s.pixels(1920) for s in shares) / (1920 * 1080) < 1.14
    # more ranks than tiles
    shares = pdist.plan(64, 32, 32, 4)
    assert [s.tile_count for s in shares] == [1, 1, 0, 0]
    p = shares[1].apply(abi.PyrRenderParams())
    assert (p.tile_begin, p.tile_end, p.tile_stride, p.film_layout) == (1, 2, 4, abi.PYR_FILM_TILE_BLOCKS)


def test_blocks_assemble_matches_a_direct_scatter():
    """assemble_blocks_torch against a pixel-by-pixel restatement of the PYR_FILM_TILE_BLOCKS definition (pyrite_gpu.h)."""
    width, height, ts, bins = 21, 13, 8, 3  # 3 x 2 tiles, both edges cut
    rng = np.random.default_rng(5)
    for n, r in ((2, 1), (3, 0), (1, 0)):
        share = pdist.plan(width, height, ts, n, "tiles")[r]
        side = ts + 2
        blocks = rng.random((share.tile_count, side, side, bins, 2)).astype(np.float32)
        film = pdist.assemble_blocks_torch(torch.zeros((height, width, bins, 2)), torch.from_numpy(blocks).reshape(-1, bins, 2), share, ts).numpy()
        expect = np.zeros((height, width, bins, 2), dtype=np.float32)
        tiles_x = (width + ts - 1) // ts
        for k, tile in enumerate(share.tiles()):
            ty, tx = divmod(tile, tiles_x)
            for by in range(side):
                for bx in range(side):
                    x, y = tx * ts + bx - 1, ty * ts + by - 1
                    if 0 <= x < width and 0 <= y < height:
                        expect[y, x] += blocks[k, by, bx]
        assert np.array_equal(film, expect)


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

        def render_share(share, buffer):
            view = buffer.numpy()
            assert view.flags["C_CONTIGUOUS"] and view.shape[0] >= share.pixels(film.width)
            sc.render(r, cam, film, threads=1, share=share, window=view[:share.pixels(film.width)])

        result = pdist.render_sharded(render_share, film.width, film.height, film.bins, r.tile_size, torch.device("cpu"), sharding=sharding)
        if rank == 0:
            np.save(out_path, result.numpy())
        else:
            assert result is None
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world_size,sharding", [(2, "tiles"), (3, "tiles"), (2, "contiguous"), (3, "contiguous")])
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


def test_ring_catches_the_samples_a_tile_window_would_drop():
    """A share rendered into ringed blocks holds every exposure of its tiles: the block total equals samples x wavelengths
    even though a few samples land one pixel outside their tile (tile size 2 makes that common)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle
    from pyrite_amd import scenes

    world, cam, r, film = scenes.build(scenes.c2_cornell(24, 24, 64), seed=3)
    r.tile_size = 2
    share = pdist.plan(24, 24, 2, 3)[1]
    blocks = np.zeros((share.pixels(24), film.bins, 2), dtype=np.float32)
    c = oracle.OracleScene(world).render(r, cam, film, threads=2, share=share, window=blocks)
    assert c["samples"] == share.tile_count * 2 * 2 * 64
    assert blocks[..., 1].sum() == c["samples"] * r.spectrum_samples == c["exposures"]
