"""The N > 1 decomposition on CPU: world_size-2 (and 3) gloo processes shard the tiles of a
frame, fill them with a per-pixel function, gather to rank 0 and stitch.  Checks the shard
arithmetic of the C ABI (rt_shard_tiles), the compact tile layout and the gather order that
bench.py relies on.  The render itself needs a GPU and is covered by
tests/test_gpu_parity.py::test_tile_shards_stitch_to_the_full_image."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def pixel_value(py, px):
    return np.stack([(py * 7 + px * 3) % 251, (py + 2 * px) % 241, (py * px) % 239], axis=-1).astype(np.uint8)


def _worker(rank, world, port, hs, vs, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import raytracer_amd as rt
    from raytracer_amd import parallel
    per = parallel.tiles_per_shard(hs, vs, world)
    py, px, ok = parallel.shard_pixel_coords(hs, vs, rank, world)
    buf = np.zeros((per, rt.TILE, rt.TILE, 3), dtype=np.uint8)
    buf[:py.shape[0]][ok] = pixel_value(py[ok], px[ok])
    t = torch.from_numpy(buf.reshape(-1))
    got = parallel.gather_tiles(t, rank, world, dist)
    # ... and as bench.py calls it: received straight into the rows of one contiguous buffer
    rows = torch.zeros((world, t.numel()), dtype=torch.uint8) if rank == 0 else None
    parallel.gather_tiles(t, rank, world, dist, out=list(rows.unbind(0)) if rank == 0 else None)
    if rank == 0:
        assert all(torch.equal(rows[r], got[r]) for r in range(world))
        shards = [g.numpy().reshape(per, rt.TILE, rt.TILE, 3) for g in got]
        img = parallel.detile_reference(shards, hs, vs)
        yy, xx = np.meshgrid(np.arange(vs), np.arange(hs), indexing="ij")
        q.put(bool(np.array_equal(img, pixel_value(yy, xx))))
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("world,hs,vs", [(2, 250, 141), (3, 64, 64), (2, 33, 17)])
def test_shard_gather_stitch(rt, world, hs, vs):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, hs, vs, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert q.get(timeout=5) is True


def test_shard_counts_cover_every_tile_once(rt):
    from raytracer_amd import parallel
    for hs, vs in ((1920, 1080), (7680, 4320), (33, 17), (1, 1)):
        total = ((hs + rt.TILE - 1) // rt.TILE) * ((vs + rt.TILE - 1) // rt.TILE)
        for n in (1, 2, 4, 8):
            assert sum(rt.shard_tiles(hs, vs, r, n) for r in range(n)) == total
            seen = np.zeros((vs, hs), dtype=np.int32)
            for r in range(n):
                py, px, ok = parallel.shard_pixel_coords(hs, vs, r, n)
                if hs * vs <= 1920 * 1080:
                    np.add.at(seen, (py[ok], px[ok]), 1)
            if hs * vs <= 1920 * 1080:
                assert (seen == 1).all()
