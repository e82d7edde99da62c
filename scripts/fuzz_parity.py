"""Randomised parity campaign: random scenes (tests/scenes.py) on the GPU core against the oracle."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import raytracer_amd as rt
from oracle import oracle as O
import scenes

first, count = int(sys.argv[1]), int(sys.argv[2])
bad = 0
t0 = time.time()
for seed in range(first, first + count):
    rng = np.random.default_rng(seed)
    n_tris = int(rng.choice([5, 40, 300, 1500, 9000], p=[0.23, 0.23, 0.23, 0.23, 0.08]))  # 9000: beyond the device builder's large subtrees
    n_prims = int(rng.integers(0, 8))
    builder = int(rng.integers(0, 2))
    pipe = [rt.PIPELINE_WAVEFRONT, rt.PIPELINE_MEGAKERNEL][int(rng.integers(0, 2))]
    hs, vs = int(rng.integers(33, 130)), int(rng.integers(17, 90))
    w = rt.World(); scenes.random_scene(w, rt.material, rt, seed, n_prims=n_prims, n_tris=n_tris, hsize=hs, vsize=vs)
    s = O.Scene(); scenes.random_scene(s, O.material, rt, seed, n_prims=n_prims, n_tris=n_tris, hsize=hs, vsize=vs)
    w.commit(bvh_builder=builder)
    u8, f64, st = w.render(pipeline=pipe, want_f64=True, want_stats=True)
    s.use_bvh()
    of64, ou8, cnt = s.render()
    # the frame again: the kernels without counters, issued from the remembered queue lengths
    u8b, f64b, _ = w.render(pipeline=pipe, want_f64=True)
    ok = np.array_equal(u8, ou8) and np.abs(f64 - of64).max() <= 1e-11 and \
        (st.rays_primary, st.rays_shadow, st.rays_reflect, st.rays_refract) == (cnt["primary"], cnt["shadow"], cnt["reflect"], cnt["refract"]) and \
        np.array_equal(u8b, u8) and np.array_equal(f64b, f64)
    if not ok:
        bad += 1
        d = np.abs(u8.astype(int) - ou8.astype(int)).max(axis=-1)
        print("MISMATCH seed", seed, "tris", n_tris, "prims", n_prims, "builder", builder, "pipe", pipe, "pixels", int((d > 0).sum()), flush=True)
    if (seed - first) % 20 == 19:
        print("done", seed - first + 1, "scenes,", bad, "mismatches, %.0f s" % (time.time() - t0), flush=True)
print("total", count, "mismatches", bad)
sys.exit(1 if bad else 0)
