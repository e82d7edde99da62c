"""rt_render (host buffers: D2H included), one frame at a time."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracer_amd as rt
w = rt.World.example("pawn", 1920, 1080).commit()
for _ in range(8):
    w.render()
t0 = time.perf_counter()
n = 20
for _ in range(n):
    u8, _, _ = w.render()
dt = (time.perf_counter() - t0) / n * 1e3
print("rt_render pawn 1080p, host buffer: %.3f ms per frame, %.0f Mrays/s" % (dt, 13199884 / dt / 1e3))
