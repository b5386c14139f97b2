import sys, os, faulthandler, time
faulthandler.dump_traceback_later(50, exit=True)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "image-processing_amd")]
import numpy as np
use_torch = sys.argv[1] == "torch"
batch = int(sys.argv[2]); nslots = int(sys.argv[3]); steps = int(sys.argv[4])
from lanecv import _abi
if os.environ.get('LANECV_SO'): _abi._SO = os.path.abspath(os.environ['LANECV_SO'])
from lanecv import synth
frames = np.stack([synth.synth_road(1920, 1080, 100, t) for t in range(batch)])
if use_torch:
    import torch
    torch.cuda.set_device(0)
from lanecv.pipeline import LanePipeline, Recipe
rec = Recipe.north_star(1920, 1080)
pipe = LanePipeline(1920, 1080, rec, max_batch=batch, nslots=nslots, seg_capacity=128, device=0)
print("pipe ok", flush=True)
if use_torch:
    dev = torch.from_numpy(frames).cuda(); torch.cuda.synchronize(); ptr = dev.data_ptr()
    sub = lambda s: pipe.submit_device(s, ptr, batch)
else:
    sub = lambda s: pipe.submit(s, frames)
quiet = batch >= 128
if quiet:
    import builtins
    _p = print
    print = lambda *a, **k: None
    if not use_torch:
        ptr = pipe.upload(0, frames)
        sub = lambda s: pipe.submit_device(s, ptr, batch)
    for s_ in range(nslots):
        sub(s_); pipe.wait(s_)
t0 = time.perf_counter()
infl = []
for k in range(steps):
    s = k % nslots
    if len(infl) == nslots:
        w = infl.pop(0); r = pipe.wait(w); print("waited", w, int(r[1][0]), flush=True)
    sub(s); infl.append(s); print("submitted", k, flush=True)
for w in infl:
    r = pipe.wait(w); print("waited", w, int(r[1][0]), flush=True)
dt = time.perf_counter() - t0
if quiet: print = _p
print("DONE ms/step %.3f fps %.0f" % (dt / steps * 1e3, steps * batch / dt), flush=True)
