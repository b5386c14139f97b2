"""python tools/size_probe.py <w> <h> <batch> <slots> <steps>: pipelined north_star throughput at another frame size,
with per-stage times of the last batch and how many frames took the general HoughLinesP kernel."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "image-processing_amd")]
import numpy as np
from lanecv import _abi, synth
from lanecv.pipeline import LanePipeline, Recipe
w, h, batch, nslots, steps = (int(a) for a in sys.argv[1:6])
import multiprocessing as mp
def gen(t): return synth.synth_road(w, h, 100, t)
pool = mp.get_context("fork").Pool(16)      # (closed and joined, never terminated: see bench.make_frames)
try:
    frames = np.stack(pool.map(gen, range(batch), chunksize=2))
finally:
    pool.close()
    pool.join()
rec = Recipe.north_star(w, h)
pipe = LanePipeline(w, h, rec, max_batch=batch, nslots=nslots, seg_capacity=256, device=0, timing=True)
ptr = pipe.upload(0, frames)
for s in range(nslots):
    pipe.submit_device(s, ptr, batch); pipe.wait(s)
t0 = time.perf_counter(); infl = []
for k in range(steps):
    s = k % nslots
    if len(infl) == nslots: pipe.wait(infl.pop(0))
    pipe.submit_device(s, ptr, batch); infl.append(s)
for s in infl: r = pipe.wait(s)
dt = time.perf_counter() - t0
names = {"front": _abi.STAGE_FRONT, "hyst": _abi.STAGE_HYST, "final": _abi.STAGE_FINAL, "points": _abi.STAGE_POINTS,
         "vote": _abi.STAGE_VOTE, "lines": _abi.STAGE_LINES}
print({n: round(pipe.stage_ms(0, i), 3) for n, i in names.items()})
npts = pipe.read(0, _abi.BUF_NPOINTS)
print("points per frame: min %d max %d; segments frame 0: %d" % (npts.min(), npts.max(), int(r[1][0])))
print("DONE %dx%d batch %d: ms/step %.3f fps %.0f" % (w, h, batch, dt / steps * 1e3, steps * batch / dt))
