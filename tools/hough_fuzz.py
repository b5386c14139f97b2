"""Random edge maps through lanecv.HoughLinesP against the oracle: sizes up to 1080p, point densities from sparse to dense
enough for every form of the kernel (all in LDS, mask in device memory, points in device memory, global-memory kernel),
random lines, rho, theta, threshold, minimum length and gap.   usage: python tools/hough_fuzz.py [CASES [SEED]]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "image-processing_amd")]
import torch  # noqa: E402,F401  (before liblanecv.so)
import lanecv as cv2  # noqa: E402
from oracle import lane_oracle as oracle  # noqa: E402

oracle.build()
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad = 0
t_last = time.time()
for k in range(cases):
    w = int(rng.choice([64, 200, 641, 960, 1280, 1920]))
    h = int(rng.choice([48, 120, 359, 540, 720, 1080]))
    dens = float(10 ** rng.uniform(-3.2, -1.1))
    e = (rng.random((h, w)) < dens).astype(np.uint8) * 255
    if rng.random() < 0.5:                                   # points in a band of rows only (dense mask words)
        y0 = int(rng.integers(0, h))
        e[:y0] = 0
    for _ in range(int(rng.integers(0, 6))):                 # lines, some with holes
        x = np.arange(int(rng.integers(0, w // 2)), int(rng.integers(w // 2, w)))
        y = (rng.uniform(0, h) + rng.uniform(-1.2, 1.2) * (x - w / 2)).astype(int)
        ok = (y >= 0) & (y < h) & (rng.random(len(x)) < rng.uniform(0.6, 1.0))
        e[y[ok], x[ok]] = 255
    rho = float(rng.choice([1, 2, 3.5, 6]))
    theta = np.pi / float(rng.choice([60, 45, 64, 90]))
    n = int(np.count_nonzero(e))
    thr = int(rng.integers(20, 260))
    minlen, gap = int(rng.integers(0, 80)), int(rng.integers(0, 40))
    got = cv2.HoughLinesP(e, rho, theta, thr, minLineLength=minlen, maxLineGap=gap)
    ref = oracle.hough_lines_p(e, rho, theta, thr, minlen, gap, cap=1 << 17)
    same = (got is None and ref is None) or (got is not None and ref is not None and np.array_equal(got, ref))
    if not same:
        bad += 1
        print("MISMATCH case %d: %dx%d points %d rho %g theta pi/%g thr %d minlen %d gap %d" % (k, w, h, n, rho, np.pi / theta, thr, minlen, gap), flush=True)
    if time.time() - t_last > 20 or k == cases - 1:
        t_last = time.time()
        print("case %d of %d, last: %dx%d %d points, %d segments; %d mismatches so far" % (k + 1, cases, w, h, n, 0 if ref is None else len(ref), bad), flush=True)
print("fuzz done: %d cases, %d mismatches" % (cases, bad))
sys.exit(1 if bad else 0)
