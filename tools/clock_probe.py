"""The GPU's clock while HoughLinesP runs: per frame, the kernel's duration in s_memtime ticks against its duration in
ticks of the constant 100 MHz counter (s_memrealtime), alone and with batches in flight.  Needs the -DLC_STAMPS build at
tools/liblanecv_stamps.so (tools/README.md).   usage: python tools/clock_probe.py [SLOTS [BATCHES]]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "image-processing_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from lanecv import _abi  # noqa: E402
_abi._SO = os.path.join(ROOT, "tools", "liblanecv_stamps.so")
from lanecv import synth  # noqa: E402
from lanecv.pipeline import LanePipeline, Recipe  # noqa: E402

W, H, B = 1920, 1080, 256
NS = int(sys.argv[1]) if len(sys.argv) > 1 else 6
N = int(sys.argv[2]) if len(sys.argv) > 2 else 40
pipe = LanePipeline(W, H, Recipe.north_star(W, H), max_batch=B, nslots=NS, seg_capacity=64, timing=True)
base = np.stack([synth.synth_road(W, H, seed=0, t=t) for t in range(16)])
dev = torch.from_numpy(np.concatenate([base] * (B // 16))).cuda()
torch.cuda.synchronize()
fl = []
for k in range(N):
    s = k % NS
    if len(fl) == NS:
        pipe.wait(fl.pop(0))
    pipe.submit_device(s, dev.data_ptr(), B)
    fl.append(s)
    if k == N - NS:          # read while batches are in flight: the stamps of the launches that just finished
        cyc = (C.c_longlong * (2 * B))()
        real = (C.c_longlong * B)()
        _abi.lib().lc_debug_frame_cycles(cyc, B)
        _abi.lib().lc_debug_frame_real(real, B)
for d in fl:
    pipe.wait(d)
    last = d
    if d == fl[0]:
        lines_ms = pipe.stage_ms(d, _abi.STAGE_LINES)      # (a batch that ran with the others in flight)
c = np.array(list(cyc), dtype=np.float64).reshape(B, 2)[:, 1]
r = np.array(list(real), dtype=np.float64)
ok = r > 0
print("slots %d: whole kernel per frame: s_memtime ticks median %.0f max %.0f; 100 MHz ticks median %.0f max %.0f (= %.3f / %.3f ms); "
      "s_memtime ticks per microsecond: median %.1f min %.1f max %.1f; lines stage %.3f ms"
      % (NS, np.median(c[ok]), c[ok].max(), np.median(r[ok]), r[ok].max(), np.median(r[ok]) / 1e5, r[ok].max() / 1e5,
         np.median(c[ok] / r[ok] * 100), (c[ok] / r[ok] * 100).min(), (c[ok] / r[ok] * 100).max(), lines_ms))
pipe.close()
