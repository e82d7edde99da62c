"""Fast mode's bookkeeping: candidates of the f32 search and f64 evaluations per frame."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracer_amd as rt
for name, hs, vs, n in [("pawn", 1920, 1080, 0), ("tea set", 1920, 1080, 0), ("synthetic", 3840, 2160, 1000000)]:
    w = rt.World.example(name, hs, vs, synthetic_tris=n).commit()
    for mode, mn in [(rt.MODE_F64_PARITY, "parity"), (rt.MODE_F32_FAST, "fast")]:
        w.render(mode=mode)
        _, _, st = w.render(mode=mode, want_stats=True)
        d = st.as_dict()
        q = d["rays"] + d["pred_queries"]
        print(name, mn, "queries", q, "tri tests", d["tri_tests"], "tri steps", d["tri_wave_steps"], "candidates", d["fast_candidates"],
              "exact", d["fast_exact"], "exact/query %.3f" % (d["fast_exact"] / q), flush=True)
