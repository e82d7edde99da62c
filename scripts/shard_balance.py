"""How evenly the round-robin deal of 32x32 tiles splits the pawn frame: every shard k of N rendered
alone with F frames in flight (no exchange), and the rays it traces.  python scripts/shard_balance.py [N ...]
env: SIZE=1080|8k, F (default 8)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import raytracer_amd as rt
size = os.environ.get("SIZE", "1080")
hs, vs = (7680, 4320) if size == "8k" else (1920, 1080)
F = int(os.environ.get("F", "8"))
w = rt.World.example("pawn", hs, vs).commit()
streams = [torch.cuda.Stream() for _ in range(F)]
for n in [int(a) for a in sys.argv[1:]] or [2, 4, 8]:
    per = max(rt.shard_tiles(hs, vs, s, n) for s in range(n))
    bufs = [torch.zeros(per * rt.TILE * rt.TILE * 3, dtype=torch.uint8, device="cuda") for _ in range(F)]
    ms, rays = [], []
    for k in range(n):
        def step(j, stats=False):
            w.render_device(d_u8=bufs[j % F].data_ptr(), stream=streams[j % F].cuda_stream, tile_first=k, tile_step=n,
                            pipeline=rt.PIPELINE_WAVEFRONT, want_stats=stats)
        step(0, True)
        torch.cuda.synchronize()
        rays.append(w.stats().rays())
        for j in range(2 * F):
            step(j)
        torch.cuda.synchronize()
        steps = 100 if size != "8k" else 16
        t0 = time.perf_counter()
        for j in range(steps):
            step(j)
        torch.cuda.synchronize()
        ms.append((time.perf_counter() - t0) / steps * 1e3)
    print(f"{hs}x{vs} N={n}: ms per shard " + " ".join(f"{m:.3f}" for m in ms) + f" | slowest / mean {max(ms) / (sum(ms) / n):.3f}"
          f" | rays per shard (M) " + " ".join(f"{r / 1e6:.2f}" for r in rays) + f" | most / mean {max(rays) / (sum(rays) / n):.3f}", flush=True)
