"""Soak run: the same two alternating batches through three slots for many steps; every batch's segments, point
counts and (every 50th step) masked map and accumulator must equal the first result for that batch.
usage: python tools/soak.py [STEPS] [BATCH]"""
import os

import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "image-processing_amd"))
import torch  # noqa: E402,F401  (before liblanecv.so)
from lanecv import _abi, synth  # noqa: E402
from lanecv.pipeline import LanePipeline, Recipe  # noqa: E402

STEPS = int(sys.argv[1]) if len(sys.argv) > 1 else 600
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
W, H = 1920, 1080
rec = Recipe.north_star(W, H)
NS = int(os.environ.get("SOAK_SLOTS", "3"))
pipe = LanePipeline(W, H, rec, max_batch=B, nslots=NS, seg_capacity=128)
adv = synth.adversarial_frames(W, H)
kinds = [np.stack([synth.synth_road(W, H, seed=100, t=t) for t in range(B)]),
         np.stack([synth.synth_road(W, H, seed=7, t=3 * t, shadow=(t % 2 == 0)) if t % 8 else adv["black"] for t in range(B)])]
dev = [torch.from_numpy(k).cuda() for k in kinds]
torch.cuda.synchronize()
first = {}
inflight = []
bad = 0


def check(step, slot, kind):
    global bad
    seg, cnt = pipe.wait(slot)
    got = {"seg": seg.copy(), "cnt": cnt.copy(), "np": pipe.read(slot, _abi.BUF_NPOINTS)}
    if step % 50 == 0 or kind not in first:
        got["masked"] = pipe.read(slot, _abi.BUF_MASKED)
        got["acc"] = pipe.read(slot, _abi.BUF_ACC)
    if kind not in first:
        first[kind] = got
        return
    ref = first[kind]
    for k, v in got.items():
        a, b = v, ref[k]
        if k == "seg":
            ok = all((a[i, :got["cnt"][i]] == b[i, :ref["cnt"][i]]).all() for i in range(len(a)) if got["cnt"][i] == ref["cnt"][i])
        else:
            ok = a.shape == b.shape and (a == b).all()
        if not ok:
            bad += 1
            print("MISMATCH step %d slot %d kind %d field %s" % (step, slot, kind, k), flush=True)


for step in range(STEPS):
    slot, kind = step % NS, (step // 2) % 2
    if len(inflight) == NS:
        check(*inflight.pop(0))
    pipe.submit_device(slot, dev[kind].data_ptr(), B if step % 7 else max(1, B // 2))
    inflight.append((step, slot, kind if step % 7 else kind + 2))
    if step % 100 == 0:
        print("step", step, flush=True)
for it in inflight:
    check(*it)
print("soak done: %d steps, %d mismatches" % (STEPS, bad))
pipe.close()
sys.exit(1 if bad else 0)
