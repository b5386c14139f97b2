"""HoughLinesP of 1080p edge maps with uniformly scattered points and one line, one map per call, against the oracle:
where the LDS kernel, its mask-in-device-memory form and the general kernel take over (LANECV_NO_SCATTER_KERNEL=1 leaves
the middle one out).   usage: python tools/scatter_probe.py"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "image-processing_amd"))
sys.path.insert(0, ROOT)
import torch
import lanecv as cv2
from oracle import lane_oracle as oracle
oracle.build()
rng = np.random.default_rng(3)
W, H = 1920, 1080
for dens in (0.001, 0.002, 0.004, 0.006, 0.007, 0.008, 0.012, 0.02, 0.04, 0.1):
    e = (rng.random((H, W)) < dens).astype(np.uint8) * 255
    for x in range(100, 1800, 3):
        e[int(1000 - 0.45 * x) % H, x] = 255
    n = int(np.count_nonzero(e))
    cv2.HoughLinesP(e, 6, np.pi / 60, 160, minLineLength=40, maxLineGap=25)
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); r = cv2.HoughLinesP(e, 6, np.pi / 60, 160, minLineLength=40, maxLineGap=25); ts.append(time.perf_counter() - t0)
    ref = oracle.hough_lines_p(e, 6, np.pi / 60, 160, 40, 25, cap=65536)
    same = (r is None and ref is None) or (r is not None and ref is not None and np.array_equal(np.asarray(r).reshape(-1, 4), np.asarray(ref).reshape(-1, 4)))
    print("points %6d  HoughLinesP %.3f ms  segments %d  same as oracle: %s" % (n, float(np.median(ts)) * 1e3, 0 if r is None else len(r), same), flush=True)
