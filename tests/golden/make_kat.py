"""Writes tests/golden/kat.json: known answers derived by hand-checkable integer
arithmetic (Python ints and fractions only -- neither the C oracle, the numpy
restatement nor the HIP library is involved).  These are NOT cv2 outputs: no
cv2 exists in this environment and the reference ships no fixtures.

Run:  python tests/golden/make_kat.py
"""
import json
import math
import os
import struct

kat = {}

# --- gray: 15-bit weights, +2^14, >>15 -------------------------------------
W = {"B": 3735, "G": 19235, "R": 9798}
g = lambda b, gg, r: (b * W["B"] + gg * W["G"] + r * W["R"] + (1 << 14)) >> 15
kat["gray_bgr2gray"] = [[[255, 0, 0], g(255, 0, 0)], [[0, 255, 0], g(0, 255, 0)],
                        [[0, 0, 255], g(0, 0, 255)], [[255, 255, 255], g(255, 255, 255)],
                        [[12, 200, 77], g(12, 200, 77)], [[1, 1, 1], g(1, 1, 1)]]
assert [v for _, v in kat["gray_bgr2gray"][:4]] == [29, 150, 76, 255]

# --- blur: 255 impulse -> outer product of [1 4 6 4 1], (S+128)>>8 ----------
b = [1, 4, 6, 4, 1]
kat["blur_impulse_center"] = [[(255 * bi * bj + 128) >> 8 for bj in b] for bi in b]
# impulse at the (0,0) corner of a >=5x5 image: output (y,x), y,x in 0..2, sees the
# source pixel through tap (2-y, 2-x) only (reflect-101 never maps onto index 0)
kat["blur_impulse_corner"] = [[(255 * b[2 - y] * b[2 - x] + 128) >> 8 for x in range(3)] for y in range(3)]
# impulse at (row 0, col 1): output (0,0) reaches column 1 through taps x-1 (reflected)
# and x+1, and row 0 through the centre tap only
kat["blur_impulse_at_0_1"] = {"out_0_0": (255 * b[2] * (b[1] + b[3]) + 128) >> 8}

# --- cv::RNG(2^64-1) multiply-with-carry ------------------------------------
state, outs = (1 << 64) - 1, []
for _ in range(8):
    state = ((state & 0xFFFFFFFF) * 4164903690 + (state >> 32)) & ((1 << 64) - 1)
    outs.append(state & 0xFFFFFFFF)
assert outs[0] == (1 << 32) - 4164903691
kat["rng_first8"] = outs

# --- Hough geometry for the reference's (rho=6, theta=pi/60) ----------------
f32 = lambda v: struct.unpack("f", struct.pack("f", v))[0]
theta = f32(math.pi / 60)
kat["hough_geometry"] = []
for (w, h) in ((960, 540), (1280, 720), (1920, 1080), (3840, 2160), (640, 360)):
    q = f32(((w + h) * 2 + 1) / f32(6.0))
    nr = int(q + 0.5) if abs(q - round(q)) != 0.5 else 2 * round(q / 2)
    kat["hough_geometry"].append({"w": w, "h": h, "numangle": 60, "numrho": nr})
kat["theta_f32"] = theta
kat["irho_f32"] = f32(1.0 / f32(6.0))
# a vertical line x=c of L pixels: n=0 row gets L votes at round(c/6) + (numrho-1)//2
kat["hough_vertical"] = {"w": 960, "h": 540, "x": 120, "y0": 100, "y1": 400,
                         "n": 0, "bin": 120 // 6 + (500 - 1) // 2, "votes": 300}
kat["hough_horizontal"] = {"w": 960, "h": 540, "y": 120, "x0": 100, "x1": 400,
                           "n": 30, "bin": 120 // 6 + (500 - 1) // 2, "votes": 300}

# --- triangle ROI mask of the reference at 960x540 --------------------------
# vertices (0,540),(480,270),(960,540); both slanted edges leave the image at
# y=540 and are clipped to y=539: (1,539) and (959,539); edge slope in 16.16:
dxl = -((479 << 16) // 269)      # trunc toward zero of (1-480)<<16 / (539-270)
dxr = (479 << 16) // 269
rows = {}
for y in (270, 271, 300, 400, 539):
    xl = ((480 << 16) + (y - 270) * dxl) >> 16
    xr = ((480 << 16) + (y - 270) * dxr) >> 16
    rows[str(y)] = [xl, xr]
kat["triangle_960x540_span"] = rows   # scanline span; the outline may add pixels beside it

here = os.path.dirname(os.path.abspath(__file__))
with open(os.path.join(here, "kat.json"), "w") as f:
    json.dump(kat, f, indent=1, sort_keys=True)
print("wrote kat.json")
