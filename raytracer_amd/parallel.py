"""Image-tile sharding over ranks (one process per GPU) and the one exchange step.

The frame is cut into TILE x TILE pixel tiles numbered row-major; rank r of n renders the
tiles t with t % n == r (round-robin, for load balance: the reference's contiguous row
bands, scene/camera.rs:86-143, put all the expensive glass pixels on a few ranks).  Each
rank fills a compact tile buffer; `gather_tiles` brings the buffers to rank 0 with one
gather (RCCL over xGMI on GPUs, gloo in the CPU tests) where rt_detile_u8 (or, for tests
on the CPU, `detile_reference`) rebuilds the row-major image — the counterpart of
draw::canvas::stitch_canvases (canvas.rs:47-64).
"""
import numpy as np

from . import TILE, shard_tiles


def tiles_per_shard(hsize, vsize, nranks):
    """Common (maximum) tile count of the shards: gather needs equal-sized buffers."""
    return max(shard_tiles(hsize, vsize, r, nranks) for r in range(nranks))


def shard_pixel_coords(hsize, vsize, rank, nranks):
    """(k, iy, ix) -> (py, px) arrays of this rank's compact tile buffer, and a validity mask
    (tiles on the right/bottom edge are partial).  Host-side description of the layout
    rt_render_device writes; used by tests and by host code that inspects shards."""
    tiles_x = (hsize + TILE - 1) // TILE
    nt = shard_tiles(hsize, vsize, rank, nranks)
    k = np.arange(nt)
    tile = rank + k * nranks
    ty, tx = tile // tiles_x, tile % tiles_x
    iy, ix = np.meshgrid(np.arange(TILE), np.arange(TILE), indexing="ij")
    py = ty[:, None, None] * TILE + iy[None]
    px = tx[:, None, None] * TILE + ix[None]
    return py, px, (py < vsize) & (px < hsize)


def gather_tiles(tiles, rank, nranks, dist, dst=0, out=None):
    """One gather of equal-sized per-rank tile buffers to `dst`.  Returns the list of buffers
    on dst (index = source rank), None elsewhere.  `out` (on dst): the buffers to receive into,
    e.g. the rows of one contiguous (nranks, n) buffer that rt_detile_u8 then reads; new
    buffers are made when it is None."""
    import torch
    if rank == dst:
        if out is None:
            out = [torch.empty_like(tiles) for _ in range(nranks)]
        if len(out) != nranks or any(o.numel() != tiles.numel() for o in out):
            raise ValueError("gather_tiles: need one receive buffer of the shard's size per rank")
    else:
        out = None
    dist.gather(tiles, out, dst=dst)
    return out


def detile_reference(shards, hsize, vsize):
    """NumPy statement of rt_detile_u8 (tests only): shards[r] is rank r's compact buffer of
    shape (tiles, TILE, TILE, C)."""
    n = len(shards)
    c = shards[0].shape[-1]
    img = np.zeros((vsize, hsize, c), dtype=shards[0].dtype)
    for r, buf in enumerate(shards):
        py, px, ok = shard_pixel_coords(hsize, vsize, r, n)
        nt = py.shape[0]
        img[py[ok], px[ok]] = buf[:nt][ok]
    return img
