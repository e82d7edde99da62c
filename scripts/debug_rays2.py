import sys, os
sys.path[0] = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
import numpy as np
import raytracer_amd as rt
w = rt.World.example("pawn", 64, 64).commit()
o = np.array([1.7155938512549582, 4.044285213235007, -2.94034483629179])
d = np.array([-0.19679150491689157, -0.6390955635683606, 0.7435253621900092])
for n in (1, 2, 3, 64, 65, 128, 4000):
    out = w.trace_rays(np.tile(o, (n, 1)), np.tile(d, (n, 1)))
    print(n, "hits", int(out[:, 0].sum()), "t0", out[0, 1], "tlast", out[-1, 1])
o2 = np.array([[1.7155938512549582, 4.044285213235007, -2.94034483629179], [0.30703173480189117, 1.200383230325545, -1.4608182852688794]])
d2 = np.array([[-0.19679150491689157, -0.6390955635683606, 0.7435253621900092], [-0.2939167903649645, -0.4702698132390941, 0.8321413480278606]])
out = w.trace_rays(o2, d2)
print(out[:, :8])
out = w.trace_rays(o2[1:], d2[1:])
print(out[:, :8])
