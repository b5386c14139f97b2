"""What ONE frame with too many edge points for half a CU's LDS costs a running pipeline (1080p x 256, six slots): the gap in the
waits around the batch that holds it (batch 60 of 300) and the batches after it.  EVERY=k: such a frame every k-th batch;
LANECV_PPHT_FULL_AT_ONCE=1: whole-CU launches from the first such frame on.  (gc is off: Python's collector pauses a run of this
length once, for 35 ms.)   usage: python tools/dense_event_probe.py"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "image-processing_amd"))
import gc
import torch
gc.disable()
from lanecv import synth, _abi
from lanecv.pipeline import LanePipeline, Recipe
w, h, B, NS = 1920, 1080, 256, 6
rec = Recipe.north_star(w, h)
pipe = LanePipeline(w, h, rec, max_batch=B, nslots=NS, seg_capacity=256)
road = [synth.synth_road(w, h, seed=5, t=t) for t in range(16)]
noise = synth.adversarial_frames(w, h)["noise"]
rows = int(os.environ.get("ROWS", 40))
sc = road[0].copy(); sc[h - rows:] = noise[h - rows:]
plain = np.stack([road[i % 16] for i in range(B)])
dense = plain.copy(); dense[17] = sc
d_plain = torch.from_numpy(plain).cuda(); d_dense = torch.from_numpy(dense).cuda()
torch.cuda.synchronize()
every = int(os.environ.get("EVERY", 0))      # 0: one event at batch 60; k: a dense frame every k-th batch
N = 300
stamps = []
def run():
    inflight = []
    del stamps[:]
    for b in range(N):
        s = b % NS
        if len(inflight) == NS: pipe.wait(inflight.pop(0)); stamps.append(time.perf_counter())
        d = d_dense if ((every and b % every == 0) or (not every and b == 60 and not os.environ.get("NONE"))) else d_plain
        pipe.submit_device(s, d.data_ptr(), B); inflight.append(s)
    for s in inflight: pipe.wait(s)
run()
torch.cuda.synchronize(); t0 = time.perf_counter(); run(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
gaps = np.diff(np.array(stamps)) * 1e3
print("event gap %.2f ms; median of the 64 gaps after it %.3f ms, of the 40 before it %.3f ms" % (gaps[59], np.median(gaps[60:124]), np.median(gaps[15:55])))
print("gaps above 0.8 ms:", [(int(i), round(float(g), 2)) for i, g in enumerate(gaps) if g > 0.8], "sum of gaps %.1f" % gaps.sum())
print("wait-to-wait gaps (ms) around batch 60:", " ".join("%.2f" % g for g in gaps[50:80]), " median %.3f" % np.median(gaps))
pipe.submit_device(0, d_dense.data_ptr(), B); pipe.wait(0)
print("dense frame: %d points; %d batches of %d in %.1f ms = %.3f ms a batch (%s)" % (int(pipe.read(0, _abi.BUF_NPOINTS).max()), N, B, dt * 1e3, dt * 1e3 / N,
      "a dense frame every %d batches" % every if every else "one dense frame in batch 60"), flush=True)
pipe.close()
