"""What k_ppht_fast's instructions go to: the bench batch (256 x 1080p) through pipelines that differ in one thing -- the
recipe as it is; without the full-vote accumulator; with a threshold no bin reaches (votes only, no line is ever walked).
Run under `rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES`: three launches of k_ppht_fast per
variant, in this order."""
import dataclasses
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "image-processing_amd"))
from lanecv import synth  # noqa: E402
from lanecv.pipeline import LanePipeline, Recipe  # noqa: E402
import multiprocessing as mp  # noqa: E402

W, H, B = 1920, 1080, 256


def gen(t):
    return synth.synth_road(W, H, 100, t)


pool = mp.get_context("fork").Pool(16)      # (closed and joined, never terminated: see bench.make_frames)
try:
    frames = np.stack(pool.map(gen, range(B), chunksize=4))
finally:
    pool.close()
    pool.join()
base = Recipe.north_star(W, H)
for name, rec in (("as it is", base), ("no accumulator", dataclasses.replace(base, accumulator=False)),
                  ("threshold nothing reaches", dataclasses.replace(base, threshold=100000))):
    pipe = LanePipeline(W, H, rec, max_batch=B, nslots=1, seg_capacity=128)
    ptr = pipe.upload(0, frames)
    for _ in range(3):
        pipe.submit_device(0, ptr, B)
        seg, cnt = pipe.wait(0)
    print(name, "segments", int(cnt.sum()), flush=True)
    pipe.close()
