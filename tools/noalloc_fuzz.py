"""lc_canny_noalloc on random images of random sizes (1 x 1 up to 4096 x 2304), one after the other on the same GPU's fixed working
array, against the oracle's Canny: what a call leaves in the array must never reach the next.
usage: python tools/noalloc_fuzz.py [CASES [SEED]]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "image-processing_amd"), os.path.join(ROOT, "tests")]
import torch  # noqa: E402,F401  (before liblanecv.so)
from lanecv import _abi  # noqa: E402
import hiprt  # noqa: E402
from oracle import lane_oracle as oracle  # noqa: E402

oracle.build()
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
ctx = _abi.context()
cap = 4096 * 2304
d_in, d_out = hiprt.DeviceBuffer(cap), hiprt.DeviceBuffer(cap)
bad = 0
t_last = time.time()
for k in range(cases):
    w = int(rng.choice([1, 2, 3, 5, 16, 63, 64, 65, 331, 640, 1280, 1920, 4096]))
    h = int(rng.choice([1, 2, 3, 7, 64, 65, 203, 360, 720, 1080, 2304]))
    kind = int(rng.integers(0, 4))
    if kind == 0:
        img = rng.integers(0, 256, (h, w), dtype=np.uint8)
    elif kind == 1:
        yy, xx = np.mgrid[0:h, 0:w]
        img = np.clip(90 + 60 * np.sin(xx / 29.0) * np.cos(yy / 17.0) + 90 * (np.abs(xx - yy * 0.7 - w / 3) < 3) + rng.normal(0, 4, (h, w)), 0, 255).astype(np.uint8)
    elif kind == 2:
        img = np.full((h, w), int(rng.integers(0, 256)), np.uint8)
        img[:, w // 2:] = int(rng.integers(0, 256))          # one long vertical edge: hysteresis over many tiles
        img[h // 3:, w // 2:] //= 2
    else:
        img = (rng.integers(0, 2, (h, w), dtype=np.uint8) * 255).astype(np.uint8)
    lo, hi = float(rng.integers(0, 300)), float(rng.integers(0, 500))
    d_in.upload(img)
    d_out.upload(np.full((h, w), 7, np.uint8))
    _abi.check(_abi.lib().lc_canny_noalloc(ctx, d_in.ptr, w, h, lo, hi, d_out.ptr, None))
    hiprt.synchronize()
    got = d_out.download((h, w))
    if not np.array_equal(got, oracle.canny(img, lo, hi)):
        bad += 1
        print("MISMATCH case %d: %dx%d kind %d thresholds %g %g" % (k, w, h, kind, lo, hi), flush=True)
    if time.time() - t_last > 20 or k == cases - 1:
        t_last = time.time()
        print("case %d of %d, %d mismatches so far" % (k + 1, cases, bad), flush=True)
d_in.free()
d_out.free()
print("fuzz done: %d cases, %d mismatches" % (cases, bad))
sys.exit(1 if bad else 0)
