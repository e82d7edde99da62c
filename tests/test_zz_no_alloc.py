"""rt_render_device_noalloc (include/rt_core.h): the wavefront frame with its ray queues in the
library's fixed device array.  Through the C ABI, against rt_render_device (the existing entry
point it is a variant of), bit for bit, on two calls in a row, the second with a smaller input;
then shards, the fast mode, other depths and the size limit."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def both(rt, w, hs, vs, depth=5, mode=0, tile_first=0, tile_step=1):
    import torch
    if tile_step == 1:
        n = hs * vs * 3
    else:
        n = rt.shard_tiles(hs, vs, tile_first, tile_step) * rt.TILE * rt.TILE * 3
    outs = []
    stream = torch.cuda.current_stream().cuda_stream
    for noalloc in (False, True):
        u8 = torch.full((n,), 7, dtype=torch.uint8, device="cuda")
        f64 = torch.full((n,), -1.0, dtype=torch.float64, device="cuda")
        if noalloc:
            w.render_device_noalloc(d_u8=u8.data_ptr(), d_f64=f64.data_ptr(), stream=stream, depth=depth,
                                    mode=mode, tile_first=tile_first, tile_step=tile_step)
        else:
            w.render_device(d_u8=u8.data_ptr(), d_f64=f64.data_ptr(), stream=stream, depth=depth, mode=mode,
                            pipeline=rt.PIPELINE_WAVEFRONT, tile_first=tile_first, tile_step=tile_step)
        torch.cuda.synchronize()
        outs.append((u8.cpu().numpy(), f64.cpu().numpy()))
    return outs


def test_noalloc_equals_render_device_twice_second_smaller(rt):
    # 400x300 = 13 x 10 tiles = 130 tiles: seven batches of 19 at depth 5, ragged edges
    big = rt.World.example("pawn", 400, 300).commit()
    small = rt.World.example("pawn", 97, 61).commit()
    for w, (hs, vs) in ((big, (400, 300)), (small, (97, 61))):
        (au8, af64), (bu8, bf64) = both(rt, w, hs, vs)
        assert np.array_equal(au8, bu8)
        assert np.array_equal(af64, bf64)
        assert len(np.unique(bu8)) > 20  # a picture, not a constant
    # and once more on the first scene: the array keeps the last call's contents
    (au8, af64), (bu8, bf64) = both(rt, big, 400, 300)
    assert np.array_equal(au8, bu8) and np.array_equal(af64, bf64)


def test_noalloc_equals_the_oracle(rt, O):
    import torch
    w = rt.World.example("cover", 160, 120).commit()
    u8 = torch.zeros((120, 160, 3), dtype=torch.uint8, device="cuda")
    w.render_device_noalloc(d_u8=u8.data_ptr())
    torch.cuda.synchronize()
    _, ou8, _ = O.Scene.example("cover", 160, 120).render()
    assert np.array_equal(u8.cpu().numpy(), ou8)


@pytest.mark.parametrize("depth", [1, 2, 8])
def test_noalloc_depths(rt, depth):
    w = rt.World.example("pawn", 128, 96).commit()
    (au8, af64), (bu8, bf64) = both(rt, w, 128, 96, depth=depth)
    assert np.array_equal(au8, bu8) and np.array_equal(af64, bf64)


def test_noalloc_shards_and_fast_mode(rt):
    w = rt.World.example("tea set", 250, 141).commit()
    for first in (0, 2):
        (au8, af64), (bu8, bf64) = both(rt, w, 250, 141, tile_first=first, tile_step=3)
        assert np.array_equal(au8, bu8) and np.array_equal(af64, bf64)
    (au8, _), (bu8, _) = both(rt, w, 250, 141, mode=rt.MODE_F32_FAST)
    assert np.array_equal(au8, bu8)


def test_noalloc_limits(rt):
    import torch
    assert rt.noalloc_max_tiles(5) >= 1 and rt.noalloc_max_tiles(8) >= 1 and rt.noalloc_max_tiles(9) == -1
    assert rt.noalloc_max_tiles(1) > rt.noalloc_max_tiles(5) > rt.noalloc_max_tiles(8)
    w = rt.World.example("cover", 64, 64).commit()
    u8 = torch.zeros(64 * 64 * 3, dtype=torch.uint8, device="cuda")
    with pytest.raises(rt.RtError, match="depth"):
        w.render_device_noalloc(d_u8=u8.data_ptr(), depth=0)
    with pytest.raises(rt.RtError, match="no output"):
        w.render_device_noalloc()
    # 4096 x 2080 = 128 x 65 = 8320 tiles > RT_NOALLOC_MAX_TILES: refused before anything is issued
    cam = w.camera_abi()
    cam.hsize, cam.vsize = 4096, 2080
    with pytest.raises(rt.RtError, match="maximum per call"):
        w.render_device_noalloc(d_u8=u8.data_ptr(), cam=cam)
