"""Per recursion level of the wavefront frame: rays, node / triangle loop lane occupancy, from
the statistics build rendered at depth 1, 2, ... (level L = depth L+1 minus depth L)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracer_amd as rt
name, hs, vs = (sys.argv[1], int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else ("pawn", 1920, 1080)
w = rt.World.example(name, hs, vs, synthetic_tris=1000000 if name == "synthetic" else 0).commit()
prev = None
for depth in range(1, 6):
    w.render(depth=depth, pipeline=rt.PIPELINE_WAVEFRONT)
    _, _, st = w.render(depth=depth, pipeline=rt.PIPELINE_WAVEFRONT, want_stats=True)
    d = st.as_dict()
    cur = d
    if prev is not None:
        cur = {k: d[k] - prev[k] for k in d if isinstance(d[k], int)}
    calls = cur["rays_primary"] + cur["rays_reflect"] + cur["rays_refract"]
    print(f"level {depth - 1}: calls {calls} shadow {cur['rays_shadow']} pred {cur['pred_queries']} "
          f"node lanes/step {cur['nodes_visited'] / max(1, cur['node_wave_steps']):.1f} "
          f"tri lanes/step {cur['tri_tests'] / max(1, cur['tri_wave_steps']):.1f} "
          f"nodes/query {cur['nodes_visited'] / max(1, calls + cur['rays_shadow'] + cur['pred_queries']):.1f} "
          f"node steps/wave-call {cur['node_wave_steps'] / max(1, calls / 64):.1f} "
          f"tri steps/wave-call {cur['tri_wave_steps'] / max(1, calls / 64):.1f}  frame kernel_ms(stats build) {d['kernel_ms']:.3f}", flush=True)
    prev = d
