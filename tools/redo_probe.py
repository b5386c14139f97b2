"""What the ROI scheme costs when frames need the second pass: the benchmark's pipeline (1080p, batches of 256 resident
in HBM, four slots, recipe north_star) with a given share of every batch replaced by frames the units near the ROI
cannot settle (a faint vertical edge that is made an edge only by its strong upper end, far above the ROI).  One JSON
line per share: frames/s, the frames of the last batch that were classified in full, and the pixels per frame the front
kernel classifies at the level the pipeline settled at.
usage: python tools/redo_probe.py [SHARE ...]   (default 0 0.02 0.1 0.5)"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "image-processing_amd"))
import torch  # noqa: E402
from lanecv import synth  # noqa: E402
from lanecv.pipeline import LanePipeline, Recipe  # noqa: E402

W, H, B, NS, STEPS = 1920, 1080, 256, 4, 120
if not os.environ.get("REDO_PROBE_GC"):    # Python's collector off (REDO_PROBE_GC=1 leaves it on): a full collection is a 35 ms
    import gc                              # pause, a third of a run of this length, and lands in the fourth or so pipeline of a process
    gc.disable()
shares = [float(v) for v in sys.argv[1:]] or [0.0, 0.02, 0.1, 0.5]
rec = Recipe.north_star(W, H)
road = np.stack([synth.synth_road(W, H, seed=100, t=t) for t in range(16)])


def faint_edge(x, ytop):
    g = np.full((H, W), 60, np.int32)
    g[:ytop, x:] += 120
    g[ytop:, x:] += 60
    return np.repeat(g[:, :, None], 3, axis=2).astype(np.uint8)


for share in shares:
    frames = np.concatenate([road] * (B // len(road)))[:B].copy()
    k = int(round(share * B))
    for i in range(k):
        frames[(i * B) // max(k, 1)] = faint_edge(W // 2 + 40 + 13 * (i % 40), 60 + 7 * (i % 30))
    pipe = LanePipeline(W, H, rec, max_batch=B, nslots=NS, seg_capacity=128)
    dev = torch.from_numpy(frames).cuda()
    torch.cuda.synchronize()

    def run(steps):
        fl = []
        for s in range(steps):
            if len(fl) == NS:
                pipe.wait(fl.pop(0))
            pipe.submit_device(s % NS, dev.data_ptr(), B)
            fl.append(s % NS)
        for d in fl:
            pipe.wait(d)

    run(3 * NS)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(STEPS)
    dt = time.perf_counter() - t0
    first, pixels, redone = pipe.roi_skip((STEPS - 1) % NS)
    print(json.dumps({"share_of_frames_needing_the_second_pass": share, "frames_replaced_per_batch": k,
                      "fps": round(STEPS * B / dt, 1), "ms_per_batch": round(dt / STEPS * 1e3, 4),
                      "pixels_classified_per_frame_last_batch": pixels, "whole_frame": W * H,
                      "frames_classified_in_full_last_batch": redone,
                      "streams_sharing_a_queue": pipe.shared_queues(), "layout": pipe.layout()}))
    pipe.close()
    del dev
