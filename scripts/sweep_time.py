"""Camera sweep over the committed pawn scene (rth_render_sequence): ms per frame, host images
included, for cameras the scene has never rendered and for the same cameras again."""
import math, os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import numpy as np
import raytracer_amd as rt

hs, vs = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1920, 1080)
w = rt.World.example("pawn", hs, vs)
base = w.camera_abi()
w.commit()


def turned(angle):
    rot = np.array([[math.cos(angle), 0, math.sin(angle), 0], [0, 1, 0, 0],
                    [-math.sin(angle), 0, math.cos(angle), 0], [0, 0, 0, 1]], dtype=np.float64)
    c = rt.CameraABI()
    C.memmove(C.byref(c), C.byref(base), C.sizeof(c))
    c.inv[:] = (rot @ np.array(list(base.inv), dtype=np.float64).reshape(4, 4)).reshape(16)
    return c


w.render_sequence([turned(0.0)] * 4, inflight=4)  # buffers of four streams sized
n, first = int(os.environ.get("N", "32")), 1
for inflight in (1, 4):
    cams = [turned(0.002 * (first + k)) for k in range(n)]
    first += n
    t0 = time.perf_counter(); w.render_sequence(cams, inflight=inflight); t1 = time.perf_counter()
    w.render_sequence(cams, inflight=inflight); t2 = time.perf_counter()
    print(f"pawn {hs}x{vs} inflight={inflight}: new cameras {1e3 * (t1 - t0) / n:.3f} ms/frame, "
          f"seen cameras {1e3 * (t2 - t1) / n:.3f} ms/frame; issued on estimates / rendered again so far",
          w.sweep_counters(), flush=True)
