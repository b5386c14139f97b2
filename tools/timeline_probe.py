"""GPU-side timeline of a pipelined run without a profiler attached: per batch, when its stages began and ended (HIP
events), as one line per batch; then per stream the share of the window it was busy.
usage: python tools/timeline_probe.py RECIPE W H BATCH SLOTS STEPS"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "image-processing_amd"))
name, W, H, B, NS, STEPS = sys.argv[1], *(int(v) for v in sys.argv[2:7])
import torch  # noqa: E402
from lanecv import synth  # noqa: E402
from lanecv.pipeline import LanePipeline, Recipe  # noqa: E402

rec = getattr(Recipe, name)(W, H)
base = np.stack([synth.synth_road(W, H, seed=100, t=t) for t in range(min(B, 16))])
frames = np.concatenate([base] * ((B + len(base) - 1) // len(base)))[:B]
pipe = LanePipeline(W, H, rec, max_batch=B, nslots=NS, seg_capacity=64, timing=True)
dev = torch.from_numpy(frames).cuda()
torch.cuda.synchronize()
rows = []
fl = []
t00 = time.perf_counter()
for k in range(STEPS):
    s = k % NS
    if len(fl) == NS:
        d, kk, ts = fl.pop(0)
        pipe.wait(d)
        rows.append((kk, d, ts, time.perf_counter() - t00, pipe.event_times(d)))
    ts = time.perf_counter() - t00
    pipe.submit_device(s, dev.data_ptr(), B)
    fl.append((s, k, ts))
for d, kk, ts in fl:
    pipe.wait(d)
    rows.append((kk, d, ts, time.perf_counter() - t00, pipe.event_times(d)))
names = "START UPLOADED FRONT HYST EDGE POINTS VOTE LINES END ESTART LSTART VSTART".split()
rows = rows[len(rows) // 2:]
t0 = min(r[4][9] for r in rows)
print("batch slot | host submit, wait-return (ms) | GPU: front [a,b] hyst-end final-end points-end | lines [a,b]")
for kk, d, ts, tw, ev in rows[:48]:
    e = {n: v - t0 for n, v in zip(names, ev)}
    print("%4d %3d | %7.3f %7.3f | %7.3f %7.3f %7.3f %7.3f %7.3f | %7.3f %7.3f" % (
        kk, d, ts * 1e3, tw * 1e3, e["ESTART"], e["FRONT"], e["HYST"], e["EDGE"], e["POINTS"], e["LSTART"], e["LINES"]))
pipe.close()
