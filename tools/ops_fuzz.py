"""The cv2-shaped calls one by one on random images of random sizes (1 x 1 up to a few hundred pixels a side, odd widths,
views with strides) against the oracle: cvtColor (gray both ways, HSV), GaussianBlur, Canny with random thresholds, fillPoly of
random polygons, rectangle, bitwise_and / bitwise_or with and without masks, inRange, addWeighted, resize, thick lines and
polylines, the full-vote accumulator.   usage: python tools/ops_fuzz.py [CASES [SEED]]"""
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
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad = {}


def image(h, w, c=None):
    """Random content of a random kind, sometimes as a view with a row stride into a wider array."""
    shape = (h, w) if c is None else (h, w, c)
    kind = int(rng.integers(0, 4))
    if kind == 0:
        a = rng.integers(0, 256, shape, dtype=np.uint8)
    elif kind == 1:
        a = (rng.integers(0, 2, shape, dtype=np.uint8) * 255).astype(np.uint8)
    elif kind == 2:
        a = np.full(shape, int(rng.integers(0, 256)), np.uint8)
        a[h // 3:, w // 4:] = int(rng.integers(0, 256))
    else:
        yy, xx = np.mgrid[0:h, 0:w]
        g = ((xx * int(rng.integers(1, 9)) + yy * int(rng.integers(1, 9))) % 256).astype(np.uint8)
        a = g if c is None else np.repeat(g[:, :, None], c, axis=2)
    if rng.random() < 0.3:
        wide = np.zeros((h, w + int(rng.integers(1, 9))) + shape[2:], np.uint8)
        wide[:, :w] = a
        return wide[:, :w]
    return a


def check(name, got, ref):
    same = (got is None and ref is None) or (got is not None and ref is not None and np.array_equal(np.asarray(got), np.asarray(ref)))
    if not same:
        bad[name] = bad.get(name, 0) + 1
        print("MISMATCH %s" % name, None if got is None else np.asarray(got).shape, flush=True)


t_last = time.time()
for k in range(cases):
    h, w = int(rng.choice([1, 2, 3, 5, 8, 17, 64, 100, 255, 360])), int(rng.choice([1, 2, 3, 4, 7, 16, 63, 130, 257, 641]))
    bgr, gray = image(h, w, 3), image(h, w)
    for code in (cv2.COLOR_BGR2GRAY, cv2.COLOR_RGB2GRAY):
        check("cvtColor gray %d %dx%d" % (code, w, h), cv2.cvtColor(bgr, code), oracle.cvt_gray(bgr, code))
    check("cvtColor hsv %dx%d" % (w, h), cv2.cvtColor(bgr, cv2.COLOR_BGR2HSV), oracle.cvt_hsv(bgr))
    check("GaussianBlur %dx%d" % (w, h), cv2.GaussianBlur(gray, (5, 5), 0), oracle.gaussian5(gray))
    t1 = float(rng.integers(0, 300)); t2 = float(rng.integers(0, 400))
    check("Canny %g %g %dx%d" % (t1, t2, w, h), cv2.Canny(gray, t1, t2), oracle.canny(gray, t1, t2))
    other, mask = image(h, w), (rng.integers(0, 3, (h, w)) * 100).astype(np.uint8)
    check("bitwise_and %dx%d" % (w, h), cv2.bitwise_and(gray, other), oracle.bitwise_and(gray, other))
    check("bitwise_and mask %dx%d" % (w, h), cv2.bitwise_and(gray, gray, mask=mask), oracle.bitwise_and(gray, gray, mask))
    check("bitwise_and 3ch mask %dx%d" % (w, h), cv2.bitwise_and(bgr, bgr, mask=mask), oracle.and3_mask(bgr, mask))
    check("bitwise_or %dx%d" % (w, h), cv2.bitwise_or(gray, other), oracle.bitwise_or(gray, other))
    lo = rng.integers(0, 200, 3); hi = lo + rng.integers(0, 120, 3)
    check("inRange %dx%d" % (w, h), cv2.inRange(bgr, lo, hi), oracle.in_range(bgr, lo, hi))
    al, be, ga = float(rng.uniform(0, 1.5)), float(rng.uniform(0, 1.5)), float(rng.uniform(-20, 20))
    check("addWeighted %dx%d" % (w, h), cv2.addWeighted(bgr, al, image(h, w, 3) if False else bgr[::-1].copy(), be, ga), oracle.add_weighted(bgr, al, bgr[::-1].copy(), be, ga))
    dw, dh = int(rng.integers(1, 400)), int(rng.integers(1, 300))
    check("resize %dx%d -> %dx%d" % (w, h, dw, dh), cv2.resize(bgr, (dw, dh)), oracle.resize_linear(bgr, (dw, dh)))
    npts = int(rng.integers(3, 9))
    pts = np.stack([rng.integers(-20, w + 20, npts), rng.integers(-20, h + 20, npts)], axis=1).astype(np.int32)[None]
    a, b = np.zeros((h, w), np.uint8), np.zeros((h, w), np.uint8)
    cv2.fillPoly(a, pts, 255)
    check("fillPoly %dx%d %d points" % (w, h, npts), a, oracle.fill_poly(b, pts, 255))
    p1, p2 = (int(rng.integers(-5, w + 5)), int(rng.integers(-5, h + 5))), (int(rng.integers(-5, w + 5)), int(rng.integers(-5, h + 5)))
    a = np.ascontiguousarray(image(h, w))
    b = a.copy()
    cv2.rectangle(a, p1, p2, 0, -1)
    check("rectangle %dx%d" % (w, h), a, oracle.rectangle_filled(b, p1, p2, 0))
    th = int(rng.integers(1, 12))
    col = tuple(int(v) for v in rng.integers(0, 256, 3))
    a = image(h, w, 3).copy(); b = a.copy()
    cv2.line(a, p1, p2, col, th)
    check("line thickness %d %dx%d" % (th, w, h), a, oracle.line(b, p1, p2, col, th))
    a = image(h, w, 3).copy(); b = a.copy()
    closed = bool(rng.integers(0, 2))
    cv2.polylines(a, [pts[0]], closed, col, th)
    check("polylines thickness %d closed %s %dx%d" % (th, closed, w, h), a, oracle.polylines(b, [pts[0]], closed, col, th))
    if w >= 4 and h >= 4:
        e = (rng.random((h, w)) < 0.05).astype(np.uint8) * 255
        rho, theta = float(rng.choice([1, 2, 6])), np.pi / float(rng.choice([60, 90, 180]))
        check("accumulator %dx%d" % (w, h), cv2.hough_accumulator(e, rho, theta), oracle.hough_accum(e, rho, theta))
    if time.time() - t_last > 20 or k == cases - 1:
        t_last = time.time()
        print("case %d of %d; mismatches so far: %s" % (k + 1, cases, bad or "none"), flush=True)
print("fuzz done: %d cases, %d mismatching checks" % (cases, sum(bad.values())))
sys.exit(1 if bad else 0)
