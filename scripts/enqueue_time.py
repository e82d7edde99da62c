"""Host time to enqueue one frame (no synchronisation) vs GPU time per frame, 1/8 shard."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import raytracer_amd as rt
w = rt.World.example("pawn", 1920, 1080).commit()
F = 3
streams = [torch.cuda.Stream() for _ in range(F)]
for n in (1, 8):
    per = max(rt.shard_tiles(1920, 1080, s, n) for s in range(n))
    bufs = [torch.zeros(max(per * rt.TILE * rt.TILE * 3, 1920 * 1080 * 3), dtype=torch.uint8, device="cuda") for _ in range(F)]
    for k in range(2 * F):
        w.render_device(d_u8=bufs[k % F].data_ptr(), stream=streams[k % F].cuda_stream, tile_first=0, tile_step=n)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    K = 60
    for k in range(K):
        w.render_device(d_u8=bufs[k % F].data_ptr(), stream=streams[k % F].cuda_stream, tile_first=0, tile_step=n)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"N={n}: enqueue {1e3*(t1-t0)/K:.3f} ms/frame, total {1e3*(t2-t0)/K:.3f} ms/frame")
