"""What one rank of an N-GPU job does per frame, timed on one GPU: the pawn frame's tiles dealt to
N shards, shard 0 rendered with F frames in flight (no exchange), with each pipeline.
python scripts/shard_time.py [N ...]      env: F (frames in flight, default 8), SIZE=1080|8k"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import raytracer_amd as rt
size = os.environ.get("SIZE", "1080")
hs, vs = (7680, 4320) if size == "8k" else (1920, 1080)
F = int(os.environ.get("F", "8"))
w = rt.World.example("pawn", hs, vs).commit()
streams = [torch.cuda.Stream() for _ in range(F)]
PIPES = {"wavefront": rt.PIPELINE_WAVEFRONT, "megakernel": rt.PIPELINE_MEGAKERNEL, "auto": rt.PIPELINE_AUTO}
for n in [int(a) for a in sys.argv[1:]] or [1, 2, 4, 8]:
    per = max(rt.shard_tiles(hs, vs, s, n) for s in range(n))
    bufs = [torch.zeros(per * rt.TILE * rt.TILE * 3, dtype=torch.uint8, device="cuda") for _ in range(F)]
    row = []
    for name, pipe in PIPES.items():
        def step(k):
            w.render_device(d_u8=bufs[k % F].data_ptr(), stream=streams[k % F].cuda_stream, tile_first=0, tile_step=n,
                            pipeline=pipe)
        for k in range(3 * F):
            step(k)
        torch.cuda.synchronize()
        steps = 200 if size != "8k" else 24
        t0 = time.perf_counter()
        for k in range(steps):
            step(k)
        t_issue = time.perf_counter() - t0
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        row.append(f"{name} {dt / steps * 1e3:.3f} ms (host issue {t_issue / steps * 1e3:.3f})")
    print(f"{hs}x{vs} N={n}, shard 0, {F} in flight: " + "; ".join(row), flush=True)
