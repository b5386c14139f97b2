"""Known-answer tests that pin oracle/lane_oracle.c to hand-derived values
(tests/golden/kat.json, written by tests/golden/make_kat.py with integer
arithmetic only).  No cv2 output exists for this path: parity unpinned."""
import json
import os

import numpy as np

KAT = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "kat.json")))


def test_gray_kat(oracle):
    for bgr, want in KAT["gray_bgr2gray"]:
        px = np.array([[bgr]], np.uint8)
        assert oracle.cvt_gray(px, oracle.COLOR_BGR2GRAY)[0, 0] == want
        assert oracle.cvt_gray(px[..., ::-1], oracle.COLOR_RGB2GRAY)[0, 0] == want
    # the reference's RGB2GRAY-on-BGR mix-up: a pure blue BGR pixel reads as red's weight
    assert oracle.cvt_gray(np.array([[[255, 0, 0]]], np.uint8), oracle.COLOR_RGB2GRAY)[0, 0] == 76


def test_blur_kat(oracle):
    im = np.zeros((9, 11), np.uint8)
    im[4, 5] = 255
    assert oracle.gaussian5(im)[2:7, 3:8].tolist() == KAT["blur_impulse_center"]
    im = np.zeros((7, 7), np.uint8)
    im[0, 0] = 255
    assert oracle.gaussian5(im)[:3, :3].tolist() == KAT["blur_impulse_corner"]
    im = np.zeros((7, 7), np.uint8)
    im[0, 1] = 255
    assert oracle.gaussian5(im)[0, 0] == KAT["blur_impulse_at_0_1"]["out_0_0"]
    for v in (0, 1, 77, 254, 255):
        assert (oracle.gaussian5(np.full((6, 9), v, np.uint8)) == v).all()


def test_sobel_canny_kat(oracle):
    s = np.zeros((8, 10), np.uint8)
    s[:, 5:] = 255
    dx, dy = oracle.sobel3(s)
    assert (dx[:, 4] == 1020).all() and (dx[:, 5] == 1020).all() and (dy == 0).all()
    assert (dx[:, :4] == 0).all() and (dx[:, 6:] == 0).all()
    e = oracle.canny(s, 100, 200)
    # '>' towards x-1 and '>=' towards x+1 keeps the left of the two equal columns
    assert (e[:, 4] == 255).all() and e.sum() == 255 * 8
    # strict thresholds: magnitude == high is weak (and dies alone), == low is nothing
    assert oracle.canny(s, 100, 1020).sum() == 0
    assert oracle.canny(s, 1020, 2000).sum() == 0
    assert oracle.canny(s, 1019.9, 1019.9).sum() == 255 * 8   # floor(1019.9) = 1019 < 1020
    # thresholds given in the wrong order are swapped
    assert (oracle.canny(s, 200, 100) == e).all()
    assert oracle.canny(np.full((5, 5), 9, np.uint8), 0, 0).sum() == 0


def test_hysteresis_kat(oracle):
    # a horizontal edge whose contrast drops along x: strong part drags the weak part along,
    # an isolated weak stretch stays dark
    im = np.zeros((12, 40), np.uint8)
    im[6:, :10] = 200      # strong: |dy| = 800
    im[6:, 10:20] = 40     # weak:   |dy| = 160
    im[6:, 28:38] = 40     # weak, not connected to anything strong
    e, cls = oracle.canny(im, 100, 300, return_cls=True)
    row = 5 if e[5].any() else 6
    assert (e[row, 1:9] == 255).all()
    assert (e[row, 11:19] == 255).all()      # promoted through the chain
    assert (e[:, 24:] == 0).all()            # isolated weak edge dropped
    assert (cls[row, 29:37] == 0).all()      # ... although it was a weak candidate


def test_rng_kat(oracle):
    assert oracle.rng_stream(8).tolist() == KAT["rng_first8"]


def test_hough_geometry_kat(oracle):
    for g in KAT["hough_geometry"]:
        na, nr, tab = oracle.hough_geometry(g["w"], g["h"], 6, np.pi / 60)
        assert (na, nr) == (g["numangle"], g["numrho"])
        assert tab[0] == np.float32(KAT["irho_f32"]) and tab[1] == 0


def test_hough_line_kat(oracle):
    k = KAT["hough_vertical"]
    im = np.zeros((k["h"], k["w"]), np.uint8)
    im[k["y0"]:k["y1"], k["x"]] = 255
    acc = oracle.hough_accum(im, 6, np.pi / 60)
    assert acc[k["n"], k["bin"]] == k["votes"] and acc[k["n"]].sum() == k["votes"]
    assert (acc.sum(axis=1) == k["votes"]).all()           # one vote per pixel per angle
    lines = oracle.hough_lines_p(im, 6, np.pi / 60, 160, 40, 25)
    assert lines.shape == (1, 1, 4)
    x1, y1, x2, y2 = lines[0, 0]
    assert x1 == x2 == k["x"] and {y1, y2} == {k["y0"], k["y1"] - 1}
    k = KAT["hough_horizontal"]
    im = np.zeros((k["h"], k["w"]), np.uint8)
    im[k["y"], k["x0"]:k["x1"]] = 255
    acc = oracle.hough_accum(im, 6, np.pi / 60)
    assert acc[k["n"], k["bin"]] == k["votes"]
    assert oracle.hough_lines_p(np.zeros((20, 20), np.uint8), 6, np.pi / 60, 10, 5, 5) is None


def test_triangle_mask_kat(oracle):
    m = np.zeros((540, 960), np.uint8)
    oracle.fill_poly(m, [np.array([(0, 540), (480, 270), (960, 540)])], 255)
    assert not m[:270].any()
    assert np.flatnonzero(m[270]).tolist() == [480]
    for y, (xl, xr) in KAT["triangle_960x540_span"].items():
        nz = np.flatnonzero(m[int(y)])
        assert (m[int(y), xl:xr + 1] == 255).all()
        # the Bresenham outline may add at most a pixel or two beside the span
        assert xl - 2 <= nz[0] <= xl and xr <= nz[-1] <= xr + 2
    assert abs(int(m.sum()) // 255 - 960 * 540 // 4) < 1000


def test_fill_rect_equals_slice(oracle):
    m = np.zeros((50, 70), np.uint8)
    oracle.fill_poly(m, [np.array([(10, 5), (40, 5), (40, 30), (10, 30)])], 255)
    want = np.zeros_like(m)
    want[5:31, 10:41] = 255
    assert (m == want).all()
    r = np.full((50, 70), 255, np.uint8)
    oracle.rectangle_filled(r, (60, 40), (70, 50), 0)     # corners beyond the image are clipped
    want = np.full_like(r, 255)
    want[40:, 60:] = 0
    assert (r == want).all()
