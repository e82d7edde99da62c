"""Commit time of a scene, first and second time in one process (the first pays the load of the
library's code objects), for both BVH builders."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracer_amd as rt
for name, n in [("pawn", 0), ("synthetic", 1000000)]:
    for builder, bn in [(1, "device LBVH"), (0, "host SAH")]:
        for k in range(2):
            w = rt.World.example(name, 640, 360, synthetic_tris=n).commit(bvh_builder=builder)
            print(name, bn, "commit", k + 1, "%.1f ms" % w.info()["commit_ms"], "nodes", w.info()["n_bvh_nodes"], flush=True)
            del w
