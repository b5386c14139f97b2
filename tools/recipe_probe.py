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
TIMES = os.environ.get("PROBE_STAGE_TIMES", "1") != "0"   # read the stage times of every batch (host time per batch)
import torch  # noqa: E402  (before liblanecv.so: both must share one HIP runtime)
from lanecv import _abi, synth  # noqa: E402
from lanecv.pipeline import LanePipeline, Recipe  # noqa: E402

rec = getattr(Recipe, name)(W, H)
base = np.stack([synth.synth_road(SW, SH, seed=100, t=t) for t in range(min(B, 16))])
frames = np.concatenate([base] * ((B + len(base) - 1) // len(base)))[:B]
pipe = LanePipeline(W, H, rec, max_batch=B, nslots=NS, seg_capacity=64, src_size=(SW, SH), timing=TIMES)
dev = torch.from_numpy(frames).cuda()
torch.cuda.synchronize()
stage = {"front": _abi.STAGE_FRONT, "hyst": _abi.STAGE_HYST, "final": _abi.STAGE_FINAL, "points": _abi.STAGE_POINTS,
         "lines": _abi.STAGE_LINES, "total": _abi.STAGE_TOTAL}


host = {"submit": 0.0, "wait": 0.0}      # seconds of host time inside the two calls


def run(steps, acc=None):
    fl = []
    for k in range(steps):
        s = k % NS
        if len(fl) == NS:
            d = fl.pop(0)
            t = time.perf_counter()
            pipe.wait(d)
            host["wait"] += time.perf_counter() - t
            if acc is not None and TIMES:
                acc.append({n: pipe.stage_ms(d, i) for n, i in stage.items()})
        t = time.perf_counter()
        pipe.submit_device(s, dev.data_ptr(), B)
        host["submit"] += time.perf_counter() - t
        fl.append(s)
    for d in fl:
        pipe.wait(d)


run(2 * NS + 1)
st = []
torch.cuda.synchronize()
host["submit"] = host["wait"] = 0.0
t0 = time.perf_counter()
run(STEPS, st)
dt = time.perf_counter() - t0
host_us = {k: round(v / STEPS * 1e6, 1) for k, v in host.items()}
ex = []
pipe.set_timing(True)
for _ in range(5):
    pipe.submit_device(0, dev.data_ptr(), B)
    pipe.wait(0)
    ex.append({n: pipe.stage_ms(0, i) for n, i in stage.items()})
print(json.dumps({"recipe": name, "w": W, "h": H, "src": [SW, SH], "batch": B, "slots": NS, "fps": round(STEPS * B / dt, 1),
                  "ms_per_batch": round(dt / STEPS * 1e3, 4), "layout_groups_streams_levels": pipe.layout(),
                  "host_us_per_batch": host_us,
                  "stage_ms_in_flight": {n: round(float(np.mean([s[n] for s in st])), 4) for n in stage} if st else None,
                  "stage_ms_exclusive": {n: round(float(np.median([s[n] for s in ex])), 4) for n in stage}}))
pipe.close()
