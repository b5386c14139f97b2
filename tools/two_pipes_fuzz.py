"""Two pipelines of different sizes and recipes alive at once in one process, their batches interleaved at random (slots taken
and collected in random order), every frame against the oracle.   usage: python tools/two_pipes_fuzz.py [SEED [REPEATS]]"""
import os, sys, dataclasses
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "image-processing_amd")]
import torch
from lanecv import _abi, synth
from lanecv.pipeline import LanePipeline, Recipe
from oracle import lane_oracle as oracle
oracle.build()
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
bad = done = 0
for rep in range(int(sys.argv[2]) if len(sys.argv) > 2 else 10):
    cfg = []
    for k in range(2):
        w, h = [(640, 360), (960, 540), (1280, 720), (1920, 1080)][int(rng.integers(0, 4))]
        rec = dataclasses.replace([Recipe.north_star, Recipe.simple, Recipe.hsv_gated][int(rng.integers(0, 3))](w, h), accumulator=True)
        ns, mb = int(rng.integers(1, 5)), int(rng.integers(1, 9))
        pipe = LanePipeline(w, h, rec, max_batch=mb, nslots=ns, seg_capacity=4096)
        adv = synth.adversarial_frames(w, h)
        pool = [synth.synth_road(w, h, seed=int(rng.integers(1, 999)), t=int(t)) for t in rng.integers(0, 300, 4)]
        g = np.full((h, w), 60, np.int32); g[:h // 3, w // 2 + 30:] += 120; g[h // 3:, w // 2 + 30:] += 55
        pool += [np.repeat(g[:, :, None], 3, axis=2).astype(np.uint8), adv["checker2"], adv["black"]]
        cfg.append(dict(w=w, h=h, rec=rec, ns=ns, mb=mb, pipe=pipe, pool=pool, ref={}, inflight=[]))
    def check(c, slot, frames):
        global bad, done
        pipe, rec = c["pipe"], c["rec"]
        segs = pipe.collect(slot)
        masked, acc = pipe.read(slot, _abi.BUF_MASKED), pipe.read(slot, _abi.BUF_ACC)
        for i, f in enumerate(frames):
            if id(f) not in c["ref"]:
                c["ref"][id(f)] = oracle.lane_frame(f, rec.color_code, rec.blur, rec.canny[0], rec.canny[1], pipe.roi, rec.rho, rec.theta, rec.threshold,
                                                    rec.min_line_length, rec.max_line_gap, seg_cap=1 << 16, want_acc=True, want_segments=True, gate=rec.hsv_gate)
            r = c["ref"][id(f)]
            got = np.zeros((0, 4), np.int32) if segs[i] is None else segs[i].reshape(-1, 4)
            ok = (masked[i] == r["masked"]).all() and (acc[i] == r["acc"]).all() and got.shape == r["segments"].shape and (got == r["segments"]).all()
            done += 1
            if not ok:
                bad += 1
                print("MISMATCH rep %d pipeline %dx%d frame %d" % (rep, c["w"], c["h"], i), flush=True)
    for b in range(int(rng.integers(10, 30))):
        c = cfg[int(rng.integers(0, 2))]
        while len(c["inflight"]) == c["ns"] or (c["inflight"] and rng.random() < 0.2):
            check(c, *c["inflight"].pop(0))
        busy = {it[0] for it in c["inflight"]}
        free = [x for x in range(c["ns"]) if x not in busy]
        slot = free[int(rng.integers(0, len(free)))]
        n = int(rng.integers(1, c["mb"] + 1))
        frames = [c["pool"][int(j)] for j in rng.integers(0, len(c["pool"]), n)]
        c["pipe"].submit(slot, np.stack(frames))
        c["inflight"].append((slot, frames))
    for c in cfg:
        for it in c["inflight"]:
            check(c, *it)
        c["pipe"].close()
    print("rep %d done, %d frames, %d mismatches" % (rep, done, bad), flush=True)
print("two pipelines at once: %d frames, %d mismatches" % (done, bad))
sys.exit(1 if bad else 0)
