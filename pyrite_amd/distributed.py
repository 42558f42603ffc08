"""Multi-GPU sharding of one render: tiles are independent units (each has its own RNG streams and writes only its own
pixels, pyrite/src/renderer/simple.rs:41-47), the scene is replicated, so the image's tiles are split into bands of whole tile
rows dealt to the ranks (see `plan`) with no data-path collective, and the per-rank film windows are brought together by ONE gather.

Why windows carry a one-row halo: Film::expose recomputes the pixel from the view-plane position (film.rs:233-246) and
float rounding can put a sample that was drawn on a tile edge into the neighbouring pixel row (probability ~1e-6 per
sample). A band therefore covers pixel rows [first_tile_row*ts - 1, last_tile_row_end + 1) and rank 0 ADDS the gathered
windows into the film; overlapping rows (halos, and tile rows shared by two ranks) sum up exactly as they would in a
single-GPU film. With the per-(tile, iteration) RNG the N-GPU film equals the 1-GPU film up to fp32 add order.

The reference has no counterpart (single process, shared memory); the collective is `torch.distributed.gather`
(RCCL over xGMI with backend "nccl", gloo on CPU for tests)."""
from __future__ import annotations

import torch
import torch.distributed as dist


def tile_grid(width, height, tile_size):
    """make_tiles' grid, renderer/algorithm.rs:158-166."""
    return (width + tile_size - 1) // tile_size, (height + tile_size - 1) // tile_size


def partition_tiles(num_tiles, world_size):
    """Contiguous raster ranges [a, b), sizes differing by at most one tile."""
    base, extra = divmod(num_tiles, world_size)
    ranges, start = [], 0
    for r in range(world_size):
        n = base + (1 if r < extra else 0)
        ranges.append((start, start + n))
        start += n
    return ranges


def window_rows(tile_range, tiles_x, tile_size, height):
    """Pixel rows (first_row, count) a rank's film window must cover: its tile rows plus one halo row on either side."""
    a, b = tile_range
    if b <= a:
        return 0, 0
    first_tile_row, last_tile_row = a // tiles_x, (b - 1) // tiles_x
    lo = max(0, first_tile_row * tile_size - 1)
    hi = min(height, (last_tile_row + 1) * tile_size + 1)
    return lo, hi - lo


def plan(width, height, tile_size, world_size, sharding=None):
    """For every rank the list of BANDS it renders, identical on all ranks (no communication needed):
    [[(tile_range, (first_row, rows)), ...], ...]. A band is a contiguous raster tile range plus the pixel rows its film
    window must cover.

    "contiguous": one band per rank, equal tile counts. Cheapest (one launch, one halo) but only balanced when cost is
    uniform over the image: on C3 the rows that show the mesh cost 1.6x the others and the slowest of 8 ranks takes 1.37x
    the mean.  "cyclic": tile rows (cut into pieces when there are few rows per rank) are dealt round-robin, so
    every rank sees every part of the image; windows grow by the extra halos (34 / 32) and by the row pieces. Default: contiguous for one rank, cyclic otherwise; PYRITE_SHARDING=contiguous|cyclic overrides."""
    import os

    tiles_x, tiles_y = tile_grid(width, height, tile_size)
    sharding = sharding or os.environ.get("PYRITE_SHARDING") or ("contiguous" if world_size == 1 else "cyclic")
    if sharding == "contiguous":
        ranges = partition_tiles(tiles_x * tiles_y, world_size)
        return [[(rng, window_rows(rng, tiles_x, tile_size, height))] if rng[1] > rng[0] else [] for rng in ranges]
    if sharding != "cyclic":
        raise ValueError("unknown sharding %r" % (sharding,))
    # bands = tile rows, cut into `splits` pieces each when there are fewer than ~8 rows per rank (34 rows on 8 ranks would
    # leave two ranks with 5 rows against 4); a piece still needs the full-width rows of its tile row as window
    splits = max(1, min(tiles_x, -(-8 * world_size // tiles_y)))
    shares = [[] for _ in range(world_size)]
    k = 0
    for ty in range(tiles_y):
        for piece in range(splits):
            a, b = ty * tiles_x + tiles_x * piece // splits, ty * tiles_x + tiles_x * (piece + 1) // splits
            if b > a:
                shares[k % world_size].append(((a, b), window_rows((a, b), tiles_x, tile_size, height)))
                k += 1
    return shares


def window_height(share):
    """Rows of the stacked film window of one rank: its bands one after the other."""
    return sum(rows for _, (_, rows) in share)


def render_sharded(render_window, width, height, bins, tile_size, device, group=None, sharding=None):
    """Runs `render_window(tile_range, (first_row, rows), window_tensor)` for each band of this rank and gathers the film.

    `window_tensor` is a zeroed float32 [rows, width, bins, 2] view (on `device`) of the rank's stacked window. Returns the
    full film [height, width, bins, 2] on rank 0 and None elsewhere. Exactly one collective."""
    world_size = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    shares = plan(width, height, tile_size, world_size, sharding)
    max_rows = max(1, max(window_height(share) for share in shares))
    window = torch.zeros((max_rows, width, bins, 2), dtype=torch.float32, device=device)
    offset = 0
    for tile_range, (first_row, rows) in shares[rank]:
        render_window(tile_range, (first_row, rows), window[offset:offset + rows])
        offset += rows
    if world_size == 1:
        share = shares[0]
        if len(share) == 1 and share[0][1] == (0, height):
            return window  # the single band is the whole film
        return _assemble([window], shares, height)
    if window.is_cuda and dist.get_backend(group) == "gloo":
        # rehearsal on a box without RCCL peers (several ranks on one GPU): gloo gathers host tensors only
        staged = window.cpu()
        gathered = [torch.empty_like(staged) for _ in range(world_size)] if rank == 0 else None
        dist.gather(staged, gathered, dst=0, group=group)
        if rank != 0:
            return None
        return _assemble([g.to(window.device) for g in gathered], shares, height)
    gathered = [torch.empty_like(window) for _ in range(world_size)] if rank == 0 else None
    dist.gather(window, gathered, dst=0, group=group)
    if rank != 0:
        return None
    return _assemble(gathered, shares, height)


def _assemble(windows, shares, height):
    first = windows[0]
    film = torch.zeros((height,) + tuple(first.shape[1:]), dtype=first.dtype, device=first.device)
    for window, share in zip(windows, shares):
        offset = 0
        for _, (first_row, rows) in share:
            film[first_row:first_row + rows] += window[offset:offset + rows]
            offset += rows
    return film
