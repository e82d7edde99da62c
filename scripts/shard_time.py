"""Time one rank's share of the frame for N = 1, 2, 4, 8 on a single GPU (no exchange step)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import raytracer_amd as rt
name, hs, vs = (sys.argv[1], int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else ("pawn", 1920, 1080)
w = rt.World.example(name, hs, vs).commit()
stream = torch.cuda.current_stream().cuda_stream
for n in (1, 2, 4, 8):
    per = max(rt.shard_tiles(hs, vs, s, n) for s in range(n))
    buf = torch.zeros(per * rt.TILE * rt.TILE * 3 if n > 1 else hs * vs * 3, dtype=torch.uint8, device="cuda")
    worst = 0.0
    times = []
    for r in range(n):
        for _ in range(3):
            w.render_device(d_u8=buf.data_ptr(), stream=stream, tile_first=r, tile_step=n)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            w.render_device(d_u8=buf.data_ptr(), stream=stream, tile_first=r, tile_step=n)
        torch.cuda.synchronize()
        times.append((time.perf_counter() - t0) / 20 * 1e3)
        worst = max(times)
    print(f"{name} {hs}x{vs} N={n}: slowest rank {worst:.3f} ms/frame; all ranks", " ".join(f"{t:.2f}" for t in times), flush=True)
