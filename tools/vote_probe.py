"""LDS-atomic throughput of the stand-alone Hough vote (k_vote) on 1080p frames against the probe's peak rates:
the `vote` object of bench.py on its own, for rocprofv3 PMC passes over k_vote and k_lds_atomic_probe.
usage: python tools/vote_probe.py [BATCH]"""
import json
import os

import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "image-processing_amd"))

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
import torch  # noqa: E402  (before liblanecv.so: both must share one HIP runtime)
import bench  # noqa: E402
from lanecv import synth  # noqa: E402

base = np.stack([synth.synth_road(bench.W, bench.H, seed=100, t=t) for t in range(min(B, 16))])
frames = np.concatenate([base] * ((B + len(base) - 1) // len(base)))[:B]
dev = torch.from_numpy(frames).cuda()
torch.cuda.synchronize()
print(json.dumps(bench.vote_rates(dev.data_ptr(), B, 0)))
