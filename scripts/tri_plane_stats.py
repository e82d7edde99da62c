"""Where a ray meets the plane of every triangle it is tested against (needs a build with
-DRT_TRI_PLANE_STATS: bash scripts/build_variant.sh plane -DRT_TRI_PLANE_STATS;
RT_LIB=variants/plane.so python scripts/tri_plane_stats.py): behind the origin, inside the query's
interval, beyond it — what a test on the plane alone could reject before the full test."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import raytracer_amd as rt
name, hs, vs = (sys.argv[1], int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else ("pawn", 1920, 1080)
w = rt.World.example(name, hs, vs, synthetic_tris=1000000 if name == "synthetic" else 0).commit()
w.render(); w.render()
out = (C.c_ulonglong * 8)()
w.L.rt_debug_tri_plane(None, 1)
w.render()
torch.cuda.synchronize()
w.L.rt_debug_tri_plane(out, 0)
for k, kind in enumerate(("closest-hit", "shadow")):
    n, behind, beyond, hits = out[4 * k:4 * k + 4]
    print(f"{name} {kind}: {n} tests, plane behind the origin {100.0 * behind / max(n, 1):.1f} %, beyond the interval "
          f"{100.0 * beyond / max(n, 1):.1f} %, inside {100.0 * (n - behind - beyond) / max(n, 1):.1f} %, hits {100.0 * hits / max(n, 1):.1f} %")
