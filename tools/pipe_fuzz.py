"""Random batches through random pipelines against the oracle: sizes, recipes (north-star, simple, curve, the colour-gated one;
triangle, narrowed triangle, trapezoid or no ROI; sometimes the F1 car mask; thresholds of Canny varied), slot counts (taken and collected in random order), the
pipeline's readiness to move a level out, batch sizes that change from batch to batch, frames from device memory, pinned and ordinary host memory, and a pool
of frames that provokes every rare path (faint edges across the ROI border, noise bands, all-black, checkerboards).  Masked
map, accumulator and segments of every frame must equal the oracle's.   usage: python tools/pipe_fuzz.py [PIPELINES [SEED [ONLY]]]
(ONLY: run just that pipeline of the sequence, e.g. one that reported a mismatch)"""
import dataclasses
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "image-processing_amd")]
import torch  # noqa: E402,F401  (before liblanecv.so)
from lanecv import _abi, synth  # noqa: E402
from lanecv.pipeline import LanePipeline, Recipe, narrow_triangle_roi, trapezoid_roi, f1_car_mask  # noqa: E402
from oracle import lane_oracle as oracle  # noqa: E402

oracle.build()
npipes = int(sys.argv[1]) if len(sys.argv) > 1 else 20
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
only = int(sys.argv[3]) if len(sys.argv) > 3 else -1
rng = None
bad = frames_done = 0


def pool_of_frames(w, h):
    adv = synth.adversarial_frames(w, h, seed=int(rng.integers(1, 1000)))
    seed = int(rng.integers(1, 10000))
    road = [synth.synth_road(w, h, seed=seed, t=int(t)) for t in rng.integers(0, 400, 5)]
    pool = list(road)
    for _ in range(3):                                      # a faint vertical edge made an edge only by its strong upper end
        x, ytop = int(rng.integers(w // 4, 3 * w // 4)), int(rng.integers(10, h // 2))
        g = np.full((h, w), 60, np.int32)
        g[:ytop, x:] += 120
        g[ytop:, x:] += int(rng.integers(40, 70))
        pool.append(np.repeat(g[:, :, None], 3, axis=2).astype(np.uint8))
    for _ in range(2):                                      # noise in a band of rows
        f = road[0].copy()
        y0, y1 = sorted(int(v) for v in rng.integers(0, h, 2))
        f[y0:y1 + 1] = adv["noise"][y0:y1 + 1]
        pool.append(f)
    f = road[1].copy()                                      # a bright box across the ROI border
    f[h // 3:2 * h // 3, w // 5:w // 3] = 250
    pool.append(f)
    pool += [adv["black"], adv["checker2"], adv["border_impulses"], adv["ramps"]]
    return pool


t_last = time.time()
for k in range(npipes):
    if only >= 0 and k != only:
        continue
    rng = np.random.default_rng([seed0, k])
    w, h = [(640, 360), (960, 540), (1280, 720), (1920, 1080), (644, 362)][int(rng.integers(0, 5))]
    if os.environ.get("PIPE_FUZZ_SIZE"):          # e.g. 3840x2160
        w, h = (int(v) for v in os.environ["PIPE_FUZZ_SIZE"].split("x"))
    rec = [Recipe.north_star, Recipe.simple, Recipe.curve, Recipe.hsv_gated][int(rng.integers(0, 4))](w, h)
    rec = dataclasses.replace(rec, accumulator=True)
    u = rng.random()
    if u < 0.15:
        rec = dataclasses.replace(rec, roi_vertices=None, threshold=int(rec.threshold * 1.5))
    elif u < 0.3:
        rec = dataclasses.replace(rec, roi_vertices=narrow_triangle_roi(w, h))
    elif u < 0.4:
        rec = dataclasses.replace(rec, roi_vertices=trapezoid_roi(w, h), threshold=int(rec.threshold * 1.5))
    if rng.random() < 0.15:
        rec = dataclasses.replace(rec, car_mask=f1_car_mask(w, h))
    if rng.random() < 0.3:
        lo = int(rng.integers(20, 150))
        rec = dataclasses.replace(rec, canny=(lo, lo + int(rng.integers(1, 200))))
    nslots, max_batch = int(rng.integers(1, 7)), int(rng.integers(1, 9))
    if w <= 1280 and rng.random() < 0.15:
        max_batch = int(rng.integers(9, 41))
    off = [None, "0", "2"][int(rng.integers(0, 3))]          # batches the pipeline stays a level out (None: the default, 256)
    if off is None:
        os.environ.pop("LANECV_SKIP_OFF_BATCHES", None)
    else:
        os.environ["LANECV_SKIP_OFF_BATCHES"] = off
    if os.environ.get("PIPE_FUZZ_BLUR"):         # diagnostics: overrides that leave the random sequence alone
        rec = dataclasses.replace(rec, blur=os.environ["PIPE_FUZZ_BLUR"] == "1")
    nslots = int(os.environ.get("PIPE_FUZZ_SLOTS", nslots))
    max_batch = max(max_batch, int(os.environ.get("PIPE_FUZZ_BATCH", 0)))
    force_how = int(os.environ.get("PIPE_FUZZ_HOW", -1))
    keep = bool(rng.random() < 0.2)                           # gray, blurred and pre-ROI edge maps kept and compared too
    pipe = LanePipeline(w, h, rec, max_batch=max_batch, nslots=nslots, seg_capacity=8192, keep_intermediates=keep,
                        timing=os.environ.get("PIPE_FUZZ_TIMING") == "1")
    pool = pool_of_frames(w, h)
    point_cap = (max(1024, min(w * h, w * h // 8 + 1024)) + 3) & ~3
    ref = {}
    inflight = []

    def check(slot, frames, b, how, idx):
        global bad, frames_done
        for f in frames:
            if id(f) not in ref:
                ref[id(f)] = oracle.lane_frame(f, rec.color_code, rec.blur, rec.canny[0], rec.canny[1], pipe.roi, rec.rho, rec.theta, rec.threshold,
                                               rec.min_line_length, rec.max_line_gap, car_mask=rec.car_mask, seg_cap=1 << 16, want_acc=True,
                                               want_segments=True, gate=rec.hsv_gate)
        # a frame with more edge points than the pipeline keeps per frame must fail its batch, and only then
        too_many = any(np.count_nonzero(ref[id(f)]["masked"]) > point_cap for f in frames)
        try:
            segs = pipe.collect(slot)
        except _abi.error as e:
            frames_done += len(frames)
            if not (too_many and e.code == _abi.LC_ECAP):
                bad += 1
                print("UNEXPECTED ERROR pipeline %d: %s" % (k, e), flush=True)
            return
        if too_many:
            bad += 1
            print("MISSING LC_ECAP pipeline %d" % k, flush=True)
            return
        masked, acc = pipe.read(slot, _abi.BUF_MASKED), pipe.read(slot, _abi.BUF_ACC)
        npts = pipe.read(slot, _abi.BUF_NPOINTS)
        inter = {}
        if keep:
            inter = {"gray": pipe.read(slot, _abi.BUF_GRAY), "edges": pipe.read(slot, _abi.BUF_EDGES)}
            if rec.blur:
                inter["blur"] = pipe.read(slot, _abi.BUF_BLUR)
        for i, f in enumerate(frames):
            r = ref[id(f)]
            for name, v in inter.items():
                if not (v[i] == r[name]).all():
                    bad += 1
                    print("MISMATCH pipeline %d: %s of frame %d" % (k, name, i), flush=True)
            if npts[i] != np.count_nonzero(r["masked"]):
                bad += 1
                print("MISMATCH pipeline %d: the point count of frame %d" % (k, i), flush=True)
            got = np.zeros((0, 4), np.int32) if segs[i] is None else segs[i].reshape(-1, 4)
            okm, oka = (masked[i] == r["masked"]).all(), (acc[i] == r["acc"]).all()
            oks = got.shape == r["segments"].shape and (got == r["segments"]).all()
            frames_done += 1
            if not (okm and oka and oks):
                bad += 1
                if os.environ.get("PIPE_FUZZ_DUMP"):       # the frame and both masked maps, for a look offline
                    np.savez_compressed(os.environ["PIPE_FUZZ_DUMP"], frame=f, got=masked[i], ref=r["masked"], batch=np.stack(frames))
                ys = np.nonzero((masked[i] != r["masked"]).any(axis=1))[0]
                print("MISMATCH pipeline %d (%dx%d, blur %s, roi %s, gate %s, car %s, canny %s, off %s, %d slots, batch %d) batch %d slot %d how %d frame %d of %d (pool %d): masked %s acc %s "
                      "segments %s; differing rows %s..%s (%d pixels), levels %s, redone %s"
                      % (k, w, h, rec.blur, None if rec.roi_vertices is None else rec.roi_vertices.tolist(), rec.hsv_gate is not None, rec.car_mask is not None,
                         rec.canny, off, nslots, max_batch, b, slot, how, i, len(frames), idx[i], okm, oka, oks,
                         ys.min() if len(ys) else -1, ys.max() if len(ys) else -1, int((masked[i] != r["masked"]).sum()), pipe.layout(), pipe.row_skip(slot)), flush=True)

    for b in range(int(rng.integers(6, 20))):
        # slots are taken and collected in random order: any free slot, sometimes a wait before the slots are all in flight
        # (a bank of several slots is then launched with what it holds), not always the oldest batch first
        while len(inflight) == nslots or (inflight and rng.random() < 0.2):
            check(*inflight.pop(int(rng.integers(0, len(inflight))) if rng.random() < 0.5 else 0))
        busy = {it[0] for it in inflight}
        free = [x for x in range(nslots) if x not in busy]
        slot = free[int(rng.integers(0, len(free)))]
        n = int(rng.integers(1, max_batch + 1))
        idx = [int(j) for j in rng.integers(0, len(pool) if rng.random() < 0.5 else 5, n)]
        frames = [pool[j] for j in idx]
        arr = np.stack(frames)
        how = int(rng.integers(0, 3))
        how = force_how if force_how >= 0 else how
        if how == 0:
            pipe.submit_device(slot, pipe.upload(slot, arr), n)
        elif how == 1:
            buf = pipe.pinned_frames(n)
            buf[:] = arr
            pipe.submit(slot, buf)
        else:
            pipe.submit(slot, arr)
        inflight.append((slot, frames, b, how, idx))
    for it in inflight:
        check(*it)
    pipe.close()
    if time.time() - t_last > 20 or k == npipes - 1:
        t_last = time.time()
        print("pipeline %d of %d (%dx%d), %d frames checked, %d mismatches so far" % (k + 1, npipes, w, h, frames_done, bad), flush=True)
print("fuzz done: %d pipelines, %d frames, %d mismatches" % (npipes, frames_done, bad))
sys.exit(1 if bad else 0)
