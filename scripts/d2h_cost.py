"""What a device-to-host copy of every frame costs the frames in flight (pawn 1080p)."""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import raytracer_amd as rt
w = rt.World.example("pawn", 1920, 1080).commit()
F, K = 4, 32
streams = [torch.cuda.Stream() for _ in range(F)]
img = [torch.zeros((1080, 1920, 3), dtype=torch.uint8, device="cuda") for _ in range(F)]
host = [torch.zeros((1080, 1920, 3), dtype=torch.uint8).pin_memory() for _ in range(F)]
for mode in ("no copy", "copy, no waits", "copy, wait before a slot is used again"):
    for rep in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(K):
            s = k % F
            if mode.startswith("copy, wait") and k >= F:
                streams[s].synchronize()
            w.render_device(d_u8=img[s].data_ptr(), stream=streams[s].cuda_stream)
            if mode != "no copy":
                with torch.cuda.stream(streams[s]):
                    host[s].copy_(img[s], non_blocking=True)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / K * 1e3
    print(f"{mode}: {dt:.3f} ms/frame", flush=True)
