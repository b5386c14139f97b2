"""Cross-checks oracle/lane_oracle.c against tests/np_restatement.py, a second
restatement written in a different style (vectorised numpy/scipy, closed forms,
connected-component labelling).  Catches transcription errors in either; it
cannot prove agreement with cv2 (absent here: parity unpinned)."""
import numpy as np
import pytest
from hypothesis import given, settings, strategies as st

import np_restatement as ref
from lanecv import synth


def _frames(w, h):
    fr = dict(synth.adversarial_frames(w, h))
    fr["road"] = synth.synth_road(w, h, seed=5, t=3)
    return fr


SIZES = [(64, 48), (641, 359), (33, 7), (5, 5), (3, 3), (1, 9), (9, 1), (2, 2), (1, 1)]


@pytest.mark.parametrize("w,h", SIZES)
def test_gray_blur_sobel_canny(oracle, w, h):
    for name, f in _frames(w, h).items():
        for code in (6, 7):
            g = oracle.cvt_gray(f, code)
            assert (g == ref.gray(f, code)).all(), name
        b = oracle.gaussian5(g)
        assert (b == ref.gaussian5(g)).all(), name
        dx, dy = oracle.sobel3(b)
        rdx, rdy = ref.sobel(b)
        assert (dx == rdx).all() and (dy == rdy).all(), name
        for src in (g, b):
            for lo, hi in ((100, 200), (50, 150), (0, 0), (20, 600)):
                e, cls = oracle.canny(src, lo, hi, return_cls=True)
                rcls = ref.canny_classes(src, lo, hi)
                assert (cls == rcls).all(), (name, lo, hi)
                assert (e == ref.hysteresis(rcls)).all(), (name, lo, hi)


def test_canny_1080p_road(oracle):
    f = synth.synth_road(1920, 1080, seed=3, t=0)
    g = oracle.gaussian5(oracle.cvt_gray(f, 6))
    assert (oracle.canny(g, 50, 150) == ref.canny(g, 50, 150)).all()


@settings(max_examples=60, deadline=None)
@given(st.integers(1, 40), st.integers(1, 40), st.integers(0, 2**31 - 1),
       st.integers(0, 400), st.integers(0, 800))
def test_canny_property(w, h, seed, lo, hi):
    from oracle import lane_oracle as oracle
    img = np.random.default_rng(seed).integers(0, 256, (h, w), dtype=np.uint8)
    img = ref.gaussian5(img) if seed & 1 else img
    e = oracle.canny(img, lo, hi)
    assert (e == ref.canny(img, lo, hi)).all()
    assert set(np.unique(e)) <= {0, 255}


POLYS = [
    [(0, 540), (480, 270), (960, 540)],                       # the reference's triangle
    [(96, 540), (480, 270), (864, 540)],                      # narrowed triangle (_6)
    [(0, 540), (960, 540), (960, 216), (0, 216)],             # F1 trapezoid
    [(115, 540), (115, 216), (307, 216), (336, 270), (336, 259), (624, 259),
     (624, 270), (652, 216), (844, 216), (844, 540)],         # F1 car silhouette
    [(-50, 100), (300, -40), (1200, 300), (500, 700)],        # leaves the image on all sides
    [(10, 10), (200, 300), (10, 300), (200, 10)],             # self-intersecting bow tie
    [(5, 5)], [(5, 5), (100, 50)],                            # degenerate
]


@pytest.mark.parametrize("poly", POLYS)
def test_fill_poly_matches_closed_form(oracle, poly):
    a = np.zeros((540, 960), np.uint8)
    b = np.zeros((540, 960), np.uint8)
    oracle.fill_poly(a, [np.array(poly)], 255)
    ref.fill_poly(b, [np.array(poly)], 255)
    assert (a == b).all()


@settings(max_examples=80, deadline=None)
@given(st.lists(st.tuples(st.integers(-30, 130), st.integers(-30, 110)), min_size=3, max_size=7),
       st.integers(20, 100), st.integers(20, 80))
def test_fill_poly_property(pts, w, h):
    from oracle import lane_oracle as oracle
    a = np.zeros((h, w), np.uint8)
    b = np.zeros((h, w), np.uint8)
    oracle.fill_poly(a, [np.array(pts)], 255)
    ref.fill_poly(b, [np.array(pts)], 255)
    assert (a == b).all()


def test_two_contours_even_odd(oracle):
    outer = [(5, 5), (90, 5), (90, 70), (5, 70)]
    inner = [(30, 20), (60, 20), (60, 50), (30, 50)]
    a = np.zeros((80, 100), np.uint8)
    b = np.zeros((80, 100), np.uint8)
    oracle.fill_poly(a, [np.array(outer), np.array(inner)], 255)
    ref.fill_poly(b, [np.array(outer), np.array(inner)], 255)
    assert (a == b).all()
    assert a[35, 45] == 0 and a[10, 10] == 255     # hole by the even-odd rule


def test_hough_accum_and_table(oracle):
    for (w, h) in ((960, 540), (641, 359), (64, 48)):
        na, nr, tab = oracle.hough_geometry(w, h, 6, np.pi / 60)
        rna, rnr, cos, sin = ref.hough_table(w, h, 6, np.pi / 60)
        assert (na, nr) == (rna, rnr)
        assert (tab[0::2] == cos).all() and (tab[1::2] == sin).all()
        f = synth.synth_road(w, h, seed=2, t=1)
        e = oracle.canny(oracle.cvt_gray(f, 7), 100, 200)
        acc = oracle.hough_accum(e, 6, np.pi / 60)
        assert (acc == ref.hough_accum(e, 6, np.pi / 60)).all()
        assert (acc.sum(axis=1) == np.count_nonzero(e)).all()
    # other rho/theta than the reference's
    e = (np.random.default_rng(0).random((90, 120)) < 0.05).astype(np.uint8) * 255
    for rho, theta in ((1, np.pi / 180), (2.5, 0.1), (6, np.pi / 60)):
        assert (oracle.hough_accum(e, rho, theta) == ref.hough_accum(e, rho, theta)).all()


@pytest.mark.parametrize("thr,min_len,gap", [(160, 40, 25), (130, 40, 150), (30, 10, 3)])
def test_hough_lines_p_small(oracle, thr, min_len, gap):
    w, h = 320, 180
    f = synth.synth_road(w, h, seed=11, t=4)
    e = oracle.canny(oracle.cvt_gray(f, 7), 100, 200)
    roi = np.zeros((h, w), np.uint8)
    oracle.fill_poly(roi, [np.array([(0, h), (w // 2, h // 2), (w, h)])], 255)
    m = oracle.bitwise_and(e, roi)
    thr = min(thr, 60)
    lines, acc = oracle.hough_lines_p(m, 6, np.pi / 60, thr, min_len, gap, return_acc=True)
    rl, racc = ref.hough_lines_p(m, 6, np.pi / 60, thr, min_len, gap)
    got = np.zeros((0, 4), np.int32) if lines is None else lines.reshape(-1, 4)
    assert got.shape == rl.shape and (got == rl).all()
    assert (acc == racc).all()
    assert len(got) > 0


def test_hough_lines_p_random_dots(oracle):
    rng = np.random.default_rng(4)
    for k in range(3):
        e = (rng.random((60, 80)) < 0.08).astype(np.uint8) * 255
        e[10 + k, 5:70] = 255
        e[5:55, 20 + 3 * k] = 255
        lines, acc = oracle.hough_lines_p(e, 1, np.pi / 180, 25, 15, 2, return_acc=True)
        rl, racc = ref.hough_lines_p(e, 1, np.pi / 180, 25, 15, 2)
        assert (lines.reshape(-1, 4) == rl).all() and (acc == racc).all()


def test_segment_invariants(oracle):
    w, h = 960, 540
    f = synth.synth_road(w, h, seed=1, t=0)
    e = oracle.canny(oracle.cvt_gray(f, 7), 100, 200)
    roi = np.zeros((h, w), np.uint8)
    oracle.fill_poly(roi, [np.array([(0, h), (w // 2, h // 2), (w, h)])], 255)
    m = oracle.bitwise_and(e, roi)
    l1 = oracle.hough_lines_p(m, 6, np.pi / 60, 160, 40, 25)
    l2 = oracle.hough_lines_p(m, 6, np.pi / 60, 160, 40, 25)
    assert l1 is not None and (l1 == l2).all()
    for x1, y1, x2, y2 in l1.reshape(-1, 4):
        assert m[y1, x1] and m[y2, x2]
        assert max(abs(x2 - x1), abs(y2 - y1)) >= 40
