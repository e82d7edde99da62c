"""Decode the reference's committed renders (tests/golden/*.png, copies of the
reference's renders/*.png) into raw RGB arrays stored as compressed .npz, so
that tests do not depend on an image library.  Run from the repo root:

    python tests/golden/make_rgb_npz.py
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from oracle.oracle import GOLDEN, REPO, read_png_rgb8  # noqa: E402

for name, (fn, w, h) in GOLDEN.items():
    path = os.path.join(REPO, "tests", "golden", fn)
    img = read_png_rgb8(path)
    assert img.shape == (h, w, 3), img.shape
    try:
        from PIL import Image
        ref = np.asarray(Image.open(path).convert("RGB"))
        assert np.array_equal(ref, img), "decoder disagrees with PIL"
        print(name, "decoder matches PIL")
    except ImportError:
        pass
    out = path.replace(".png", ".rgb.npz")
    np.savez_compressed(out, rgb=img)
    print(name, img.shape, "->", out, os.path.getsize(out), "bytes")
