import sys, os, time, dataclasses
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "image-processing_amd")]
import numpy as np
from lanecv import synth, _abi
from lanecv.pipeline import LanePipeline, Recipe
if os.environ.get("LANECV_SO"):
    _abi._SO = os.path.abspath(os.environ["LANECV_SO"])   # A/B runs of two builds in one call
batch = 256
import multiprocessing as mp
def gen(t): return synth.synth_road(1920, 1080, 100, t)
pool = mp.get_context("fork").Pool(16)      # (closed and joined, never terminated: see bench.make_frames)
try:
    frames = np.stack(pool.map(gen, range(batch), chunksize=4))
finally:
    pool.close()
    pool.join()
for canny in [(100, 200), (100000, 200000)]:
    rec = dataclasses.replace(Recipe.north_star(1920, 1080), canny=canny, segments=False, accumulator=False)
    pipe = LanePipeline(1920, 1080, rec, max_batch=batch, nslots=1, seg_capacity=128, device=0, timing=True)
    ptr = pipe.upload(0, frames)
    ts = []
    for i in range(int(os.environ.get("FRONT_PROBE_ITERS", "6"))):
        pipe.submit_device(0, ptr, batch); pipe.wait(0)
        ts.append([pipe.stage_ms(0, s) for s in (_abi.STAGE_FRONT, _abi.STAGE_HYST, _abi.STAGE_FINAL)])
    ts = np.array(ts)
    print(canny, "median", np.median(ts, axis=0).round(4), "min", ts.min(axis=0).round(4), flush=True)
    pipe.close()
