"""Lane utilisation of the node loop and the triangle loop (statistics build of the kernels)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracer_amd as rt
for name, hs, vs in [("pawn", 1920, 1080), ("tea set", 1920, 1080), ("pawn", 7680, 4320)]:
    w = rt.World.example(name, hs, vs).commit()
    for pipe, pn in [(rt.PIPELINE_WAVEFRONT, "wavefront"), (rt.PIPELINE_MEGAKERNEL, "megakernel")]:
        _, _, st = w.render(pipeline=pipe, want_stats=True)
        d = st.as_dict()
        print(name, hs, pn, "node lanes/step %.1f" % (d["nodes_visited"] / max(1, d["node_wave_steps"])),
              "tri lanes/step %.1f" % (d["tri_tests"] / max(1, d["tri_wave_steps"])),
              "nodes/query %.1f" % (d["nodes_visited"] / (d["rays"] + d["pred_queries"])),
              "tris/query %.1f" % (d["tri_tests"] / (d["rays"] + d["pred_queries"])), flush=True)
