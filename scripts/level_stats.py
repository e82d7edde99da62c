import sys, os
sys.path[0] = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
import numpy as np
import raytracer_amd as rt
w = rt.World.example("pawn", 1920, 1080).commit()
prev = None
for depth in (1, 2, 3, 4, 5):
    u8, _, st = w.render(depth=depth, want_stats=True, pipeline=1)
    d = st.as_dict()
    if prev:
        delta = {k: d[k] - prev[k] for k in ("rays", "nodes_visited", "tri_tests", "pred_queries", "shade_calls", "rays_shadow")}
    else:
        delta = {k: d[k] for k in ("rays", "nodes_visited", "tri_tests", "pred_queries", "shade_calls", "rays_shadow")}
    q = delta["rays"] + delta["pred_queries"]
    print("depth", depth, "kernel_ms %.2f" % d["kernel_ms"], "max_nodes", d["max_nodes_per_query"], "level delta:", delta, "nodes/query %.1f tri/query %.1f" % (delta["nodes_visited"] / q, delta["tri_tests"] / q))
    prev = d
