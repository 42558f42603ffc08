"""Multi-GPU sharding of one render: tiles are independent units (each has its own RNG streams and writes only its own
pixels, pyrite/src/renderer/simple.rs:41-47), the scene is replicated, so the image's tiles are split into one contiguous
raster range per rank with no data-path collective, and the per-rank film windows are brought together by ONE gather.

Why windows carry a one-row halo: Film::expose recomputes the pixel from the view-plane position (film.rs:233-246) and
float rounding can put a sample that was drawn on a tile edge into the neighbouring pixel row (probability ~1e-6 per
sample). A rank therefore owns pixel rows [first_tile_row*ts - 1, last_tile_row_end + 1) and rank 0 ADDS the gathered
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


def plan(width, height, tile_size, world_size):
    """[(tile_range, (first_row, rows))] for every rank, identical on all ranks (no communication needed)."""
    tiles_x, tiles_y = tile_grid(width, height, tile_size)
    ranges = partition_tiles(tiles_x * tiles_y, world_size)
    return [(rng, window_rows(rng, tiles_x, tile_size, height)) for rng in ranges]


def render_sharded(render_window, width, height, bins, tile_size, device, group=None):
    """Runs `render_window(tile_range, (first_row, rows), window_tensor)` for this rank's share and gathers the film.

    `window_tensor` is a zeroed float32 [max_rows, width, bins, 2] tensor on `device` whose first `rows` rows are the
    window. Returns the full film [height, width, bins, 2] on rank 0 and None elsewhere. Exactly one collective."""
    world_size = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    shares = plan(width, height, tile_size, world_size)
    max_rows = max(rows for _, (_, rows) in shares)
    tile_range, (first_row, rows) = shares[rank]
    window = torch.zeros((max_rows, width, bins, 2), dtype=torch.float32, device=device)
    if rows > 0:
        render_window(tile_range, (first_row, rows), window)
    if world_size == 1:
        return window[:height] if max_rows == height else _assemble([window], shares, height)
    gathered = [torch.empty_like(window) for _ in range(world_size)] if rank == 0 else None
    dist.gather(window, gathered, dst=0, group=group)
    if rank != 0:
        return None
    return _assemble(gathered, shares, height)


def _assemble(windows, shares, height):
    first = windows[0]
    film = torch.zeros((height,) + tuple(first.shape[1:]), dtype=first.dtype, device=first.device)
    for window, (_, (first_row, rows)) in zip(windows, shares):
        if rows:
            film[first_row:first_row + rows] += window[:rows]
    return film
