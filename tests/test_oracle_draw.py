"""The oracle's thick cv2.line / cv2.polylines (a from-memory restatement of OpenCV's drawing.cpp: PARITY UNPINNED)
against hand-derived known answers and an independent evaluation from closed forms."""
import numpy as np
import pytest

import np_restatement as npr


def test_horizontal_and_vertical_lines_have_closed_form_spans(oracle):
    """By ThickLine's arithmetic the body of a thickness-t line is the quad p +- dp with |dp| = t / 2 pixels, plus 1/2
    for odd t, rows / columns rounded inclusively; the caps are circles of radius (t + 1) // 2."""
    for t, half, radius in ((2, 1, 1), (3, 2, 2), (5, 3, 3), (8, 4, 4), (10, 5, 5)):
        img = np.zeros((60, 80), np.uint8)
        oracle.line(img, (20, 30), (60, 30), 255, t)
        body = np.zeros_like(img)
        body[30 - half:30 + half + 1, 20:61] = 255
        assert (img[:, 20:61] == body[:, 20:61]).all(), t                     # between the end points: the body alone
        assert img[30, 20 - radius] == 255 and img[30, 60 + radius] == 255 and img[30, 20 - radius - 1] == 0, t
        assert not img[:30 - half].any() and not img[30 + half + 1:].any(), t
        ver = np.zeros((80, 60), np.uint8)
        oracle.line(ver, (30, 20), (30, 60), 255, t)
        assert (ver == img.T).all(), t                                        # the same figure, transposed


def test_thickness_one_is_the_thin_line(oracle):
    a = np.zeros((40, 50), np.uint8)
    oracle.line(a, (3, 4), (45, 30), 200, 1)
    b = np.zeros((40, 50), np.uint8)
    oracle.fill_poly(b, [np.array([(3, 4), (45, 30)])], 200)                  # a two-point "polygon" is its outline
    assert (a == b).all()


def test_three_channel_colour_and_polyline_caps(oracle):
    img = np.zeros((50, 70, 3), np.uint8)
    pts = np.array([[10, 10], [40, 12], [60, 40]], np.int32)
    oracle.polylines(img, [pts], False, (0, 255, 7), 5)
    on = img.any(axis=2)
    assert (img[on] == (0, 255, 7)).all()
    single = np.zeros_like(img)
    oracle.line(single, pts[0], pts[1], (0, 255, 7), 5)
    oracle.line(single, pts[1], pts[2], (0, 255, 7), 5)
    assert (single == img).all()                  # an open polyline = its segments, each with both caps
    closed = np.zeros_like(img)
    oracle.polylines(closed, [pts], True, (0, 255, 7), 5)
    oracle.line(single, pts[2], pts[0], (0, 255, 7), 5)
    assert (single == closed).all()


@pytest.mark.parametrize("seed", range(6))
def test_thick_lines_match_the_closed_form_evaluation(oracle, seed):
    rng = np.random.default_rng(seed)
    w, h = 90, 70
    for _ in range(40):
        t = int(rng.integers(2, 12))
        m = t + 3                                                            # keep the quad inside: no clipping involved
        p0 = (int(rng.integers(m, w - m)), int(rng.integers(m, h - m)))
        p1 = (int(rng.integers(m, w - m)), int(rng.integers(m, h - m)))
        img = np.zeros((h, w), np.uint8)
        oracle.line(img, p0, p1, 255, t)
        want = np.zeros((h, w), np.uint8)
        for x, y in npr.thick_line_pixels(w, h, p0, p1, t):
            want[y, x] = 255
        assert (img == want).all(), (p0, p1, t)


def test_lines_leaving_the_image_are_clipped(oracle):
    img = np.zeros((40, 40), np.uint8)
    oracle.line(img, (-30, 20), (70, 20), 255, 5)
    assert img[17:24].all() and not img[:17].any() and not img[24:].any()
    img[:] = 0
    oracle.line(img, (-50, -50), (-10, -20), 255, 8)                         # wholly outside
    assert not img.any()
    img[:] = 0
    oracle.line(img, (10, 10), (10, 10), 255, 5)                             # zero length: the two caps only
    ys, xs = np.nonzero(img)
    assert ys.min() == 7 and ys.max() == 13 and xs.min() == 7 and xs.max() == 13
