"""Wall time of what a one-shot host sees: commit, the first frame (buffers are sized and the
queue lengths read back), the second frame (issued from the remembered plan)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import raytracer_amd as rt
torch.cuda.init(); torch.zeros(1, device="cuda"); torch.cuda.synchronize()
for name, hs, vs in [("pawn", 1920, 1080), ("tea set", 1920, 1080), ("cover", 400, 400)]:
    for bvh in (0, 1):
        t0 = time.perf_counter()
        w = rt.World.example(name, hs, vs)
        t1 = time.perf_counter()
        w.commit(bvh_builder=bvh)
        t2 = time.perf_counter()
        ts = []
        for k in range(3):
            a = time.perf_counter()
            w.render()
            ts.append((time.perf_counter() - a) * 1e3)
        print(f"{name} {hs}x{vs} builder {bvh}: scene build {1e3*(t1-t0):.1f} ms, commit {1e3*(t2-t1):.1f} ms, "
              f"frames (host buffers) {ts[0]:.2f} / {ts[1]:.2f} / {ts[2]:.2f} ms", flush=True)
