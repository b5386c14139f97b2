"""Latency of the cv2-shaped calls on ONE frame (the way the reference's image scripts use them), call by call and as
a one-frame pipeline, against the oracle on one host core.   usage: python tools/single_frame_probe.py [W H]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "image-processing_amd"))
sys.path.insert(0, ROOT)
import torch  # noqa: E402,F401
import lanecv as cv2  # noqa: E402
from lanecv import synth  # noqa: E402
from lanecv.pipeline import LanePipeline, Recipe, triangle_roi  # noqa: E402

W, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (960, 540)
img = synth.synth_road(W, H, seed=3, t=0)
verts = triangle_roi(W, H)


def script(frame):
    gray = cv2.cvtColor(frame, cv2.COLOR_RGB2GRAY)
    edges = cv2.Canny(gray, 100, 200)
    mask = np.zeros_like(edges)
    cv2.fillPoly(mask, verts, 255)
    masked = cv2.bitwise_and(edges, mask)
    return cv2.HoughLinesP(masked, rho=6, theta=np.pi / 60, threshold=160, lines=np.array([]), minLineLength=40, maxLineGap=25)


def timed(fn, n):
    fn()
    t = []
    for _ in range(n):
        t0 = time.perf_counter()
        fn()
        t.append(time.perf_counter() - t0)
    return round(float(np.median(t)) * 1e3, 3)


out = {"w": W, "h": H, "calls_ms": timed(lambda: script(img), 30)}
pipe = LanePipeline(W, H, Recipe.simple(W, H), max_batch=1, nslots=1, seg_capacity=256)
one = img[None]


def through_pipe():
    pipe.submit(0, one)
    return pipe.collect(0)


out["pipeline_one_frame_ms"] = timed(through_pipe, 30)
a, b = script(img), through_pipe()[0]
out["same_segments"] = bool((a is None and b is None) or (a is not None and b is not None and (a.reshape(-1, 4) == b.reshape(-1, 4)).all()))
try:
    from oracle import lane_oracle as lo
    lo.build()
    roi = np.zeros((H, W), np.uint8)
    lo.fill_poly(roi, verts, 255)
    out["oracle_one_core_ms"] = timed(lambda: lo.lane_frame(img, 7, False, 100, 200, roi, 6, np.pi / 60, 160, 40, 25, want_acc=False), 5)
except Exception as e:      # the probe also runs where the oracle is not built
    out["oracle_one_core_ms"] = repr(e)
print(json.dumps(out))
pipe.close()
