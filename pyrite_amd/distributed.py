"""Multi-GPU sharding of one render, one process per GPU.

Tiles are independent units (each has its own RNG streams and exposes only its own pixels, pyrite/src/renderer/simple.rs:41-47)
and the scene is replicated, so the image's tiles are dealt to the ranks with no data-path collective, every rank renders its
share in ONE launch into a private film buffer, and ONE gather brings the buffers to rank 0, which adds them into the film.
This is the host-side twin of include/pyrite_gpu.h's pyr_render_simple_sharded: the same plan, the same buffers, the
same assembly kernel; `NativeSharded` drives that entry point itself (RCCL send / recv inside the library), `render_sharded`
does the gather with torch.distributed (RCCL with backend "nccl", gloo on CPU for the tests).

Plans (`plan`):
  "tiles"       rank r of n renders the raster tiles r, r + n, r + 2n, ... into a buffer of ringed tile blocks
                (PYR_FILM_TILE_BLOCKS: (tile_size + 2)^2 pixels per tile). Every rank sees every part of the image, so the
                shares cost the same without measuring anything (C3 at 8 ranks: 255 tiles each; a share of contiguous rows
                costs up to 1.4x the mean there), and the gather moves (34/32)^2 = 1.13x the film.
  "contiguous"  one band of consecutive tiles per rank and the pixel rows it covers plus one halo row on either side
                (PYR_FILM_ROWS with a row window). Cheapest to gather; only balanced when cost is uniform over the image.
The ring / halo is there because Film::expose recomputes the pixel from the view-plane position (film.rs:233-246) and float
rounding can put a sample drawn on a tile edge into the neighbouring pixel (~1e-6 per sample); rank 0 ADDS the buffers, so
those samples land where a single-GPU render puts them. With the per-(tile, iteration) RNG the N-GPU film equals the 1-GPU
film up to fp32 add order."""
from __future__ import annotations

import ctypes as C
import os

import torch
import torch.distributed as dist

from . import abi


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
    """Pixel rows (first_row, count) a band's film window must cover: its tile rows plus one halo row on either side."""
    a, b = tile_range
    if b <= a:
        return 0, 0
    first_tile_row, last_tile_row = a // tiles_x, (b - 1) // tiles_x
    lo = max(0, first_tile_row * tile_size - 1)
    hi = min(height, (last_tile_row + 1) * tile_size + 1)
    return lo, hi - lo


class Share:
    """What one rank renders: the tiles tile_begin, tile_begin + tile_stride, ... < tile_end, and the layout of its buffer."""

    def __init__(self, tile_begin, tile_end, tile_stride, layout, rows=(0, 0), tile_size=32):
        self.tile_begin, self.tile_end, self.tile_stride, self.layout = int(tile_begin), int(tile_end), max(1, int(tile_stride)), int(layout)
        self.rows, self.tile_size = (int(rows[0]), int(rows[1])), int(tile_size)

    @property
    def tile_count(self):
        return max(0, -(-(self.tile_end - self.tile_begin) // self.tile_stride))

    def tiles(self):
        return range(self.tile_begin, self.tile_end, self.tile_stride)

    def pixels(self, width):
        """Pixels of the share's buffer (the buffer is float32 [pixels, bins, 2])."""
        if self.layout == abi.PYR_FILM_TILE_BLOCKS:
            return self.tile_count * (self.tile_size + 2) ** 2
        return self.rows[1] * width

    def apply(self, params):
        """Writes the share into a PyrRenderParams."""
        params.tile_begin, params.tile_end, params.tile_stride = self.tile_begin, self.tile_end, self.tile_stride
        params.film_layout = self.layout
        params.film_row_begin, params.film_row_count = (0, 0) if self.layout == abi.PYR_FILM_TILE_BLOCKS else self.rows
        return params

    def __repr__(self):
        return "Share(tiles %d:%d:%d, %s)" % (self.tile_begin, self.tile_end, self.tile_stride,
                                               "blocks" if self.layout == abi.PYR_FILM_TILE_BLOCKS else "rows %d+%d" % self.rows)


def plan(width, height, tile_size, world_size, sharding=None):
    """The share of every rank, identical on all ranks (no communication needed). Default: the whole image as rows for one
    rank, "tiles" otherwise; PYRITE_SHARDING=tiles|contiguous overrides."""
    tiles_x, tiles_y = tile_grid(width, height, tile_size)
    total = tiles_x * tiles_y
    sharding = sharding or os.environ.get("PYRITE_SHARDING") or ("contiguous" if world_size == 1 else "tiles")
    if sharding == "contiguous":
        return [Share(a, b, 1, abi.PYR_FILM_ROWS, window_rows((a, b), tiles_x, tile_size, height), tile_size) for a, b in partition_tiles(total, world_size)]
    if sharding != "tiles":
        raise ValueError("unknown sharding %r" % (sharding,))
    return [Share(min(r, total), total, world_size, abi.PYR_FILM_TILE_BLOCKS, tile_size=tile_size) for r in range(world_size)]


def assemble_blocks_torch(film, blocks, share, tile_size):
    """Adds a share's ringed tile blocks into the whole-image film [height, width, bins, 2] with torch ops (any device): the
    reference form of pyr_film_blocks_assemble_device, and the one the CPU (gloo) rehearsals use."""
    height, width = film.shape[0], film.shape[1]
    tiles_x, _ = tile_grid(width, height, tile_size)
    side = tile_size + 2
    blocks = blocks.reshape(-1, side, side, film.shape[2], 2)
    for k, tile in enumerate(share.tiles()):
        ty, tx = divmod(tile, tiles_x)
        x0, y0 = tx * tile_size - 1, ty * tile_size - 1  # image position of the block's corner (the ring)
        xa, ya, xb, yb = max(x0, 0), max(y0, 0), min(x0 + side, width), min(y0 + side, height)
        film[ya:yb, xa:xb] += blocks[k, ya - y0:yb - y0, xa - x0:xb - x0]
    return film


def assemble(film, buffer, share, tile_size):
    """Adds one rank's buffer into the film (rank 0's side of the gather)."""
    if share.tile_count == 0:
        return film
    if share.layout == abi.PYR_FILM_TILE_BLOCKS:
        n = share.pixels(film.shape[1])
        if film.is_cuda:
            from ._lib import check, lib

            desc = abi.PyrFilmDesc(film.shape[1], film.shape[0], film.shape[2], 0.0, 1.0)
            params = abi.PyrRenderParams()
            params.tile_size, params.pixel_samples = tile_size, 1
            share.apply(params)
            check(lib().pyr_film_blocks_assemble_device(C.byref(desc), C.byref(params), C.c_void_p(buffer.data_ptr()), C.c_void_p(film.data_ptr()),
                                                        film.device.index or 0, C.c_void_p(torch.cuda.current_stream(film.device).cuda_stream)))
            return film
        return assemble_blocks_torch(film, buffer[:n], share, tile_size)
    first_row, rows = share.rows
    film[first_row:first_row + rows] += buffer[:rows * film.shape[1]].reshape(rows, film.shape[1], film.shape[2], 2)
    return film


def render_sharded(render_share, width, height, bins, tile_size, device, group=None, sharding=None):
    """Runs `render_share(share, buffer)` once for this rank's share and gathers the film.

    `buffer` is a zeroed float32 [pixels, bins, 2] tensor on `device` laid out as the share says (Share.layout). Returns the
    full film [height, width, bins, 2] on rank 0 and None elsewhere. One gather (per 256 MiB of a share)."""
    world_size = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    shares = plan(width, height, tile_size, world_size, sharding)
    mine = shares[rank]
    if world_size == 1 and mine.layout == abi.PYR_FILM_ROWS and mine.rows == (0, height):
        film = torch.zeros((height, width, bins, 2), dtype=torch.float32, device=device)
        render_share(mine, film.view(-1, bins, 2))
        return film  # the single share is the whole film
    pixels = max(1, max(s.pixels(width) for s in shares))  # gather wants equal sizes
    buffer = torch.zeros((pixels, bins, 2), dtype=torch.float32, device=device)
    if mine.tile_count:
        render_share(mine, buffer)
    if world_size == 1:
        gathered = [buffer]
    elif buffer.is_cuda and dist.get_backend(group) == "gloo":
        # rehearsal on a box without RCCL peers (several ranks on one GPU): gloo gathers host tensors only
        staged = buffer.cpu()
        gathered = [torch.empty_like(staged) for _ in range(world_size)] if rank == 0 else None
        dist.gather(staged, gathered, dst=0, group=group)
        if rank == 0:
            gathered = [g.to(buffer.device) for g in gathered]
    else:
        gathered = [torch.empty_like(buffer) for _ in range(world_size)] if rank == 0 else None
        # at most 256 MiB per message: RCCL 2.26.6 was seen to deliver half of a single send / receive pair above 1 GiB without
        # an error (multi.cpp, tests/test_gpu_multi.py); C3's shares are below that from two ranks on, other films need not be
        flat, limit = buffer.view(-1), 1 << 26
        for at in range(0, flat.numel(), limit):
            parts = [g.view(-1)[at:at + limit] for g in gathered] if rank == 0 else None
            dist.gather(flat[at:at + limit], parts, dst=0, group=group)
    if rank != 0:
        return None
    film = torch.zeros((height, width, bins, 2), dtype=torch.float32, device=buffer.device)
    for g, share in zip(gathered, shares):
        assemble(film, g, share, tile_size)
    return film


class NativeSharded:
    """pyr_comm_* + pyr_render_simple_sharded: the render, the RCCL gather and the assembly all inside libpyrite_gpu.so.
    The communicator id travels over the torch.distributed group that is already up (any backend). Construction is
    collective: rank 0 broadcasts (ok, id) -- if it could not obtain an id every rank raises, none is left waiting -- and
    every rank must then attempt pyr_comm_create (ncclCommInitRank is itself collective)."""

    def __init__(self, device_index, group=None):
        from ._lib import PyriteGpuError, check, lib

        self._lib, self._check = lib(), check
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.world_size = dist.get_world_size(group) if dist.is_initialized() else 1
        self.device_index = int(device_index)
        self.handle = C.c_void_p()
        ident = (C.c_uint8 * abi.PYR_COMM_ID_BYTES)()
        forced = os.environ.get("PYRITE_FORCE_RCCL") == "1"
        if self.world_size > 1:
            box = [None]
            if self.rank == 0:
                rc = self._lib.pyr_comm_unique_id(ident)
                box = [(rc, bytes(ident) if rc == 0 else self._lib.pyr_last_error().decode())]
            dist.broadcast_object_list(box, src=0, group=group)
            rc, payload = box[0]
            if rc != 0:
                raise PyriteGpuError(rc, "rank 0 could not obtain a communicator id: %s" % payload)
            ident = (C.c_uint8 * abi.PYR_COMM_ID_BYTES).from_buffer_copy(payload)
        elif forced:
            check(self._lib.pyr_comm_unique_id(ident))
        check(self._lib.pyr_comm_create(ident, self.rank, self.world_size, self.device_index, C.byref(self.handle)))

    @property
    def uses_rccl(self):
        return bool(self._lib.pyr_comm_uses_rccl(self.handle))

    def status(self):
        """After the stream of the last render has been waited for: raises if some rank flagged its film invalid (rank 0
        sees every rank's verdict, the others their own) or the communicator is dead."""
        self._check(self._lib.pyr_comm_status(self.handle))

    def render(self, renderer, camera, world, film_desc, film_tensor, stream=0):
        """Adds one sharded render into `film_tensor` (rank 0: float32 [height, width, bins, 2] on this rank's GPU; None elsewhere)."""
        params = renderer.params()
        ptr = C.c_void_p(film_tensor.data_ptr()) if film_tensor is not None else None
        self._check(self._lib.pyr_render_simple_sharded(self.handle, world.scene(self.device_index), C.byref(camera.c), C.byref(film_desc), C.byref(params),
                                                        ptr, C.c_void_p(stream)))

    def close(self):
        if self.handle:
            self._lib.pyr_comm_destroy(self.handle)
            self.handle = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
