"""Pin the CPU oracle against the reference's committed renders.

    python oracle/pin_goldens.py [--full] [--samples N] [--scenes cover,pawn,"tea set"]

Default: N random pixels per scene (seeded) are traced by brute force (every ray
against every triangle, as the reference does) and compared with the golden
image.  --full renders every pixel (minutes of CPU time on 8 cores for the mesh
scenes) and writes a summary to oracle/PIN_RESULTS.json.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path[0] = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
from oracle import oracle as O  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--full", action="store_true")
    ap.add_argument("--bvh", action="store_true", help="use the oracle's culling mode")
    ap.add_argument("--samples", type=int, default=3000)
    ap.add_argument("--scenes", default="cover,pawn,tea set")
    ap.add_argument("--threads", type=int, default=os.cpu_count())
    args = ap.parse_args()
    results = {}
    for name in args.scenes.split(","):
        gold = O.golden_image(name)
        h, w, _ = gold.shape
        sc = O.Scene.example(name)
        if args.bvh:
            sc.use_bvh()
        t0 = time.time()
        if args.full:
            _, u8, cnt = sc.render(threads=args.threads, want_f64=False)
            diff = np.abs(u8.astype(np.int32) - gold.astype(np.int32)).max(axis=2)
            n = h * w
        else:
            rng = np.random.default_rng(12345)
            xs = rng.integers(0, w, args.samples)
            ys = rng.integers(0, h, args.samples)
            _, u8, cnt = sc.render_pixels(xs, ys, threads=args.threads)
            diff = np.abs(u8.astype(np.int32) - gold[ys, xs].astype(np.int32)).max(axis=1)
            n = args.samples
        dt = time.time() - t0
        res = dict(pixels=int(n), exact=int((diff == 0).sum()), off_by_one=int((diff == 1).sum()),
                   worse=int((diff > 1).sum()), seconds=round(dt, 1), mode="bvh" if args.bvh else "brute",
                   full=bool(args.full), counters=cnt)
        results[name] = res
        print(name, res, flush=True)
    if args.full:
        out = os.path.join(os.path.dirname(__file__), "PIN_RESULTS.json")
        old = json.load(open(out)) if os.path.exists(out) else {}
        for k, v in results.items():
            old[k + (":bvh" if args.bvh else ":brute")] = v
        json.dump(old, open(out, "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
