"""Time one rank's share of the frame for N = 1, 2, 4, 8 on a single GPU (no exchange step)."""
import os, sys, time
os.environ["GPU_MAX_HW_QUEUES"] = os.environ.get("GPU_MAX_HW_QUEUES", "8") if os.environ.get("KEEP_HW_QUEUES") else "8"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import raytracer_amd as rt
name, hs, vs = (sys.argv[1], int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else ("pawn", 1920, 1080)
w = rt.World.example(name, hs, vs).commit()
F = int(os.environ.get("INFLIGHT", "4"))
streams = [torch.cuda.Stream() for _ in range(F)]
for n in (1, 2, 4, 8):
    per = max(rt.shard_tiles(hs, vs, s, n) for s in range(n))
    bufs = [torch.zeros(per * rt.TILE * rt.TILE * 3 if n > 1 else hs * vs * 3, dtype=torch.uint8, device="cuda")
            for _ in range(F)]
    worst = 0.0
    times = []
    for r in range(n):
        for k in range(2 * F):
            w.render_device(d_u8=bufs[k % F].data_ptr(), stream=streams[k % F].cuda_stream, tile_first=r, tile_step=n)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(20):
            w.render_device(d_u8=bufs[k % F].data_ptr(), stream=streams[k % F].cuda_stream, tile_first=r, tile_step=n)
        torch.cuda.synchronize()
        times.append((time.perf_counter() - t0) / 20 * 1e3)
        worst = max(times)
    print(f"{name} {hs}x{vs} N={n}: slowest rank {worst:.3f} ms/frame; all ranks", " ".join(f"{t:.2f}" for t in times), flush=True)
