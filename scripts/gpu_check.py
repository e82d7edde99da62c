"""Quick GPU check: render the example scenes at their default sizes, compare with the
reference's golden images, print timings and counters."""
import sys, os, time, json
sys.path[0] = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
import numpy as np
import raytracer_amd as rt
from oracle import oracle as O

scenes = sys.argv[1].split(",") if len(sys.argv) > 1 else ["cover", "pawn", "tea set"]
PIPE = int(os.environ.get("RT_PIPE", "0"))
for name in scenes:
    t0 = time.time()
    w = rt.World.example(name)
    t1 = time.time()
    w.commit()
    t2 = time.time()
    print(name, "build %.2fs commit %.2fs" % (t1 - t0, t2 - t1), w.info(), flush=True)
    u8, f64, st = w.render(want_stats=True, pipeline=PIPE)
    print(" stats", st.as_dict(), flush=True)
    u8b, _, _ = w.render(pipeline=PIPE)
    u8c, _, st2 = w.render(want_stats=False, pipeline=PIPE)
    s = w.stats()
    print(" no-stats kernel_ms %.3f  Mrays/s %.1f" % (s.kernel_ms, st.rays() / s.kernel_ms / 1e3), flush=True)
    gold = O.golden_image(name)
    diff = np.abs(u8.astype(np.int32) - gold.astype(np.int32)).max(axis=2)
    print(" vs golden: exact %d / %d, off-by-one %d, worse %d" % ((diff == 0).sum(), diff.size, (diff == 1).sum(), (diff > 1).sum()), flush=True)
    assert np.array_equal(u8, u8b)
    if (diff > 0).any():
        ys, xs = np.nonzero(diff)
        print(" first mismatches:", [(int(x), int(y), u8[y, x].tolist(), gold[y, x].tolist()) for x, y in list(zip(xs, ys))[:10]])
