"""Steady-state throughput and stage times of one recipe of the batched pipeline (inputs resident in HBM).
usage: python tools/recipe_probe.py RECIPE [W H BATCH STEPS [SRC_W SRC_H]]   (RECIPE: simple, curve, f1, north_star,
hsv_gated; SRC_W SRC_H: size of the submitted frames when the pipeline resizes them first)"""
import json
import os

import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "image-processing_amd"))

name = sys.argv[1]
W, H, B, STEPS = (int(v) for v in (sys.argv[2:6] + ["1920", "1080", "64", "40"][len(sys.argv) - 2:]))
SW, SH = (int(sys.argv[6]), int(sys.argv[7])) if len(sys.argv) > 7 else (W, H)
NS = int(os.environ.get("PROBE_SLOTS", "3"))     # batches in flight
import torch  # noqa: E402  (before liblanecv.so: both must share one HIP runtime)
from lanecv import _abi, synth  # noqa: E402
from lanecv.pipeline import LanePipeline, Recipe  # noqa: E402

rec = getattr(Recipe, name)(W, H)
base = np.stack([synth.synth_road(SW, SH, seed=100, t=t) for t in range(min(B, 16))])
frames = np.concatenate([base] * ((B + len(base) - 1) // len(base)))[:B]
pipe = LanePipeline(W, H, rec, max_batch=B, nslots=NS, seg_capacity=64, src_size=(SW, SH))
dev = torch.from_numpy(frames).cuda()
torch.cuda.synchronize()
stage = {"front": _abi.STAGE_FRONT, "hyst": _abi.STAGE_HYST, "final": _abi.STAGE_FINAL, "points": _abi.STAGE_POINTS,
         "lines": _abi.STAGE_LINES, "total": _abi.STAGE_TOTAL}


def run(steps, acc=None):
    fl = []
    for k in range(steps):
        s = k % NS
        if len(fl) == NS:
            d = fl.pop(0)
            pipe.wait(d)
            if acc is not None:
                acc.append({n: pipe.stage_ms(d, i) for n, i in stage.items()})
        pipe.submit_device(s, dev.data_ptr(), B)
        fl.append(s)
    for d in fl:
        pipe.wait(d)


run(5)
st = []
torch.cuda.synchronize()
t0 = time.perf_counter()
run(STEPS, st)
dt = time.perf_counter() - t0
ex = []
for _ in range(5):
    pipe.submit_device(0, dev.data_ptr(), B)
    pipe.wait(0)
    ex.append({n: pipe.stage_ms(0, i) for n, i in stage.items()})
seg, cnt = pipe.wait(0) if False else (None, None)
print(json.dumps({"recipe": name, "w": W, "h": H, "src": [SW, SH], "batch": B, "slots": NS, "fps": round(STEPS * B / dt, 1),
                  "ms_per_batch": round(dt / STEPS * 1e3, 4),
                  "stage_ms_in_flight": {n: round(float(np.mean([s[n] for s in st])), 4) for n in stage},
                  "stage_ms_exclusive": {n: round(float(np.median([s[n] for s in ex])), 4) for n in stage}}))
pipe.close()
