import sys, os
sys.path[0] = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
import numpy as np
import raytracer_amd as rt
from oracle import oracle as O
w = rt.World.example("pawn", 64, 64).commit()
s = O.Scene.example("pawn", 64, 64).use_bvh()
rng = np.random.default_rng(77)
n = 4000
o = np.stack([rng.uniform(-3, 3, n), rng.uniform(0.2, 5, n), rng.uniform(-5, 3, n)], axis=1)
tgt = np.stack([rng.uniform(-0.8, 0.8, n), rng.uniform(0, 2.5, n), rng.uniform(-0.8, 0.8, n)], axis=1)
d = tgt - o
d /= np.linalg.norm(d, axis=1, keepdims=True)
out = w.trace_rays(o, d)
bad = 0
for i in range(n):
    oi = s.trace_info(o[i], d[i])
    if bool(out[i, 0]) != oi["hit"] or (oi["hit"] and out[i, 1] != oi["t"]):
        bad += 1
        if bad <= 8:
            print(i, "o", o[i].tolist(), "d", d[i].tolist(), "dev hit", out[i, 0], "t", out[i, 1], "order", out[i, 2],
                  "| oracle hit", oi["hit"], "t", oi["t"], "id", oi["shape_id"], "count", oi["count"])
print("bad", bad, "of", n)
