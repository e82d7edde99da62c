import sys, os
sys.path[0] = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
import numpy as np
import raytracer_amd as rt
w = rt.World.example("pawn", 64, 64).commit()
rng = np.random.default_rng(77)
n = 4000
o = np.stack([rng.uniform(-3, 3, n), rng.uniform(0.2, 5, n), rng.uniform(-5, 3, n)], axis=1)
tgt = np.stack([rng.uniform(-0.8, 0.8, n), rng.uniform(0, 2.5, n), rng.uniform(-0.8, 0.8, n)], axis=1)
d = tgt - o
d /= np.linalg.norm(d, axis=1, keepdims=True)
single = np.array([w.trace_rays(o[i:i+1], d[i:i+1])[0, 0] for i in range(256)])
print("single-ray hits in first 256:", int(single.sum()))
for m in (1, 2, 4, 8, 16, 32, 64, 65, 128, 256, 1024, 4000):
    out = w.trace_rays(o[:m], d[:m])
    print(m, "hits", int(out[:, 0].sum()), "expected", int(single[:min(m,256)].sum()), "depth1:", int(w.trace_rays(o[:m], d[:m], depth=1)[:, 0].sum()), "depth0:", int(w.trace_rays(o[:m], d[:m], depth=0)[:, 0].sum()))
