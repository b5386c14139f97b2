"""The F1 recipe (resize -> BGR2GRAY -> blur -> Canny(50,150) -> car mask -> trapezoid -> ordered non-zero list) through random
pipelines against the oracle: source sizes (the fused reduction by three, plain resizes up and down, none), slots, batch sizes
that change, frames from device, pinned and ordinary memory.   usage: python tools/f1_fuzz.py [PIPELINES [SEED]]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "image-processing_amd")]
import torch  # noqa: E402,F401  (before liblanecv.so)
from lanecv import _abi, synth  # noqa: E402
from lanecv.pipeline import LanePipeline, Recipe  # noqa: E402
from oracle import lane_oracle as oracle  # noqa: E402

oracle.build()
npipes = int(sys.argv[1]) if len(sys.argv) > 1 else 20
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
w, h = 640, 360
rec = Recipe.f1(w, h)
bad = frames_done = 0
t_last = time.time()
for k in range(npipes):
    rng = np.random.default_rng([seed0, k])
    src = [(1920, 1080), (1280, 720), (640, 360), (700, 500), (960, 540), (320, 180), (1921, 1081)][int(rng.integers(0, 7))]
    keep = bool(rng.random() < 0.3)
    nslots, max_batch = int(rng.integers(1, 5)), int(rng.integers(1, 7))
    pipe = LanePipeline(w, h, rec, max_batch=max_batch, nslots=nslots, src_size=src, keep_intermediates=keep)
    adv = synth.adversarial_frames(src[0], src[1], seed=int(rng.integers(1, 1000)))
    pool = [synth.synth_road(src[0], src[1], seed=int(rng.integers(1, 9999)), t=int(t)) for t in rng.integers(0, 300, 4)]
    pool += [adv["noise"], adv["black"], adv["ramps"], adv["checker3"]]
    ref = {}
    inflight = []

    def check(slot, frames):
        global bad, frames_done
        pipe.wait(slot)
        masked = pipe.read(slot, _abi.BUF_MASKED)
        nz = pipe.nonzero(slot)
        for i, f in enumerate(frames):
            if id(f) not in ref:
                fr = oracle.resize_linear(f, (w, h)) if src != (w, h) else f
                ref[id(f)] = oracle.lane_frame(fr, rec.color_code, rec.blur, rec.canny[0], rec.canny[1], pipe.roi, rec.rho, rec.theta, rec.threshold,
                                               rec.min_line_length, rec.max_line_gap, car_mask=rec.car_mask, want_acc=False, want_segments=False)["masked"]
            ys, xs = np.nonzero(ref[id(f)])
            ok = (masked[i] == ref[id(f)]).all() and np.array_equal(nz[i][0], ys) and np.array_equal(nz[i][1], xs)
            frames_done += 1
            if not ok:
                bad += 1
                print("MISMATCH pipeline %d (source %s, keep %s, %d slots, batch %d) frame %d of %d" % (k, src, keep, nslots, max_batch, i, len(frames)), flush=True)

    for b in range(int(rng.integers(4, 12))):
        slot = b % nslots
        if len(inflight) == nslots:
            check(*inflight.pop(0))
        n = int(rng.integers(1, max_batch + 1))
        frames = [pool[int(j)] for j in rng.integers(0, len(pool), n)]
        arr = np.stack(frames)
        how = int(rng.integers(0, 3))
        if how == 0:
            pipe.submit_device(slot, pipe.upload(slot, arr), n)
        elif how == 1:
            buf = pipe.pinned_frames(n)
            buf[:] = arr
            pipe.submit(slot, buf)
        else:
            pipe.submit(slot, arr)
        inflight.append((slot, frames))
    for it in inflight:
        check(*it)
    pipe.close()
    if time.time() - t_last > 20 or k == npipes - 1:
        t_last = time.time()
        print("pipeline %d of %d (source %s), %d frames checked, %d mismatches so far" % (k + 1, npipes, src, frames_done, bad), flush=True)
print("fuzz done: %d pipelines, %d frames, %d mismatches" % (npipes, frames_done, bad))
sys.exit(1 if bad else 0)
