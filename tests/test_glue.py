"""Post-Hough glue (`lanecv.glue`) against a per-segment loop written the way the lane scripts do it
(`Simple Lane Detection with OpenCV_2(images).py:94-153`, `_4(mp4).py:120-178`, `_5(mp4).py:86-138`).
Host-only: no GPU, no oracle."""
import math
import warnings

import numpy as np
import pytest

from lanecv import glue


def _loop_split(lines, limit):
    lx, ly, rx, ry = [], [], [], []
    if lines is not None:
        for line in lines:
            for x1, y1, x2, y2 in line:
                with warnings.catch_warnings():
                    warnings.simplefilter("ignore")
                    slope = (y2 - y1) / (x2 - x1)
                if math.fabs(slope) < limit:
                    continue
                if slope <= 0:
                    lx.extend([x1, x2]); ly.extend([y1, y2])
                else:
                    rx.extend([x1, x2]); ry.extend([y1, y2])
    return lx, ly, rx, ry


def _segments(rng, n, w=960, h=540):
    seg = np.stack([rng.integers(0, w, n), rng.integers(h // 2, h, n), rng.integers(0, w, n), rng.integers(h // 2, h, n)], axis=1)
    return seg.astype(np.int32).reshape(-1, 1, 4)


def test_split_matches_the_loop_including_vertical_and_degenerate_segments():
    rng = np.random.default_rng(5)
    lines = _segments(rng, 200)
    lines[3, 0] = [100, 300, 100, 500]      # vertical, downwards: +inf -> right
    lines[4, 0] = [100, 500, 100, 300]      # vertical, upwards: -inf -> left
    lines[5, 0] = [250, 400, 250, 400]      # zero length: nan -> kept, right
    lines[6, 0] = [10, 300, 400, 310]       # nearly horizontal: dropped
    for limit in (0.5, 0.3):
        got = glue.split_by_slope(lines, limit)
        want = _loop_split(lines, limit)
        for g, w in zip(got, want):
            assert list(g) == [int(v) for v in w]
    assert all(len(a) == 0 for a in glue.split_by_slope(None))
    lx, ly, rx, ry = glue.split_by_slope(lines[3:6], 0.5)
    assert list(lx) == [100, 100] and list(ly) == [500, 300]
    assert list(rx) == [100, 100, 250, 250] and list(ry) == [300, 500, 400, 400]


def test_lane_lines_of_the_image_script():
    rng = np.random.default_rng(6)
    h = 540
    for trial in range(20):
        lines = _segments(rng, int(rng.integers(1, 40)))
        lx, ly, rx, ry = _loop_split(lines, 0.5)
        min_y, max_y = int(h * (3 / 5)), int(h)
        want = []
        for xs, ys in ((lx, ly), (rx, ry)):
            if xs and ys:
                with warnings.catch_warnings():
                    warnings.simplefilter("ignore")
                    poly = np.poly1d(np.polyfit(ys, xs, deg=1))
                want.append([int(poly(max_y)), max_y, int(poly(min_y)), min_y])
            else:
                want.append([0, max_y, 0, min_y])
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            assert glue.lane_lines(lines, h) == want
    assert glue.lane_lines(None, h) == [[0, 540, 0, 324], [0, 540, 0, 324]]
    # two clean lanes: the fitted lines pass through them
    lanes = np.array([[[200, 540, 400, 340]], [[760, 540, 560, 340]]], np.int32)
    left, right = glue.lane_lines(lanes, 540)     # int() truncates, so an exact 200.0 may come out as 199
    assert abs(left[0] - 200) <= 1 and abs(left[2] - 416) <= 1 and left[1::2] == [540, 324]
    assert abs(right[0] - 760) <= 1 and abs(right[2] - 544) <= 1 and right[1::2] == [540, 324]


def test_line_smoother_keeps_the_last_line_and_truncates():
    sm = glue.LineSmoother(alpha=0.1)
    lanes = np.array([[[200, 540, 400, 340]], [[760, 540, 560, 340]]], np.int32)
    assert sm.update(None, 540) == []
    first = sm.update(lanes, 540)
    assert first == glue.lane_lines(lanes, 540)  # the first line of a side is taken as it is
    shifted = lanes + np.array([37, 0, 37, 0], np.int32)
    cur = glue.lane_lines(shifted, 540)
    second = sm.update(shifted, 540)
    assert second == [[int(p * 0.9 + c * 0.1) for p, c in zip(first[k], cur[k])] for k in range(2)]
    only_right = shifted[1:]
    third = sm.update(only_right, 540)
    assert third[0] == second[0]                 # no left segments: the left line stays
    assert third[1] == [int(p * 0.9 + c * 0.1) for p, c in zip(second[1], cur[1])]


def test_curve_smoother_blends_coefficients():
    rng = np.random.default_rng(8)
    sm = glue.CurveSmoother(alpha=0.1)
    assert sm.update(None) == (None, None)
    prev = [None, None]
    for frame in range(5):
        lines = _segments(rng, 30)
        lx, ly, rx, ry = _loop_split(lines, 0.3)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            cur = [np.polyfit(ly, lx, 2) if lx else None, np.polyfit(ry, rx, 2) if rx else None]
            got = sm.update(lines)
        for side in range(2):
            if cur[side] is not None:
                prev[side] = cur[side] if prev[side] is None else prev[side] * 0.9 + cur[side] * 0.1
            assert (got[side] is None) == (prev[side] is None)
            if prev[side] is not None:
                assert np.array_equal(got[side], prev[side])


def test_pixel_curve_tracker_follows_the_f1_script():
    """`LaneTracker.fit_poly` restated line by line (F1_vo_Lane_Tracking.py:42-91) against the class."""
    from lanecv.glue import PixelCurveTracker, split_at_midpoint
    rng = np.random.default_rng(5)
    h, w = 360, 640
    trk = PixelCurveTracker()
    prev_l = prev_r = None
    for frame in range(4):
        n = [300, 40, 500, 0][frame]                     # frame 1: too few pixels, frame 3: none at all
        ys = np.sort(rng.integers(150, h, n))
        xs = np.where(rng.random(n) < 0.5, 200 + (ys - 150) // 3, 450 - (ys - 150) // 4) + rng.integers(-2, 3, n)
        lx, ly, rx, ry = split_at_midpoint(ys, xs, w)
        assert (lx < w // 2).all() and (rx >= w // 2).all() and len(lx) + len(rx) == n
        got_l, got_r = trk.fit_poly((h, w), lx, ly, rx, ry)
        ploty = np.linspace(0, h - 1, h)
        want = []
        for x_, y_, prev in ((lx, ly, prev_l), (rx, ry, prev_r)):
            fit = prev
            if len(y_) > 50:
                fit = np.polyfit(y_, x_, 2)
                if prev is not None:
                    fit = 0.7 * prev + (1 - 0.7) * fit
            want.append(fit)
        prev_l, prev_r = want
        for got, fit in ((got_l, prev_l), (got_r, prev_r)):
            if fit is None:
                assert got is None
            else:
                fx = fit[0] * ploty ** 2 + fit[1] * ploty + fit[2]
                assert got.dtype == np.int32 and got.shape == (1, h, 2)
                assert (got[0, :, 0] == fx.astype(np.int32)).all() and (got[0, :, 1] == np.arange(h)).all()
    assert prev_l is not None and prev_r is not None


def test_lane_kalman_stabiliser_follows_the_reference_recurrences():
    """`1000_Lane_Stabilizer.py:69-96`: noisy (slope, intercept) measurements of a straight lane through the
    stabiliser, against an independent float64 evaluation of the predict / correct recurrences (tolerance 1e-3
    relative: the filter itself runs in float32, as cv2's does)."""
    from lanecv.glue import KalmanFilter, LaneKalmanFilter
    rng = np.random.default_rng(11)
    frames = 100
    ms = 0.5 + rng.normal(0, 0.1, frames)
    bs = 100 + rng.normal(0, 5, frames)
    lane = LaneKalmanFilter()
    lane.kf.statePost = np.array([[ms[0]], [bs[0]], [0], [0]], dtype=np.float32)
    A = np.array([[1, 0, 1, 0], [0, 1, 0, 1], [0, 0, 1, 0], [0, 0, 0, 1]], float)
    Hm = np.array([[1, 0, 0, 0], [0, 1, 0, 0]], float)
    Q, R, P = np.eye(4) * 1e-4, np.eye(2) * 0.1, np.eye(4)
    x = np.array([[ms[0]], [bs[0]], [0], [0]], float)
    out = []
    for i in range(frames):
        pre = lane.predict()
        x, P = A @ x, A @ P @ A.T + Q
        assert pre.dtype == np.float32 and pre.shape == (4, 1)
        np.testing.assert_allclose(pre, x, rtol=1e-3, atol=1e-3)
        est = lane.correct(ms[i], bs[i])
        K = P @ Hm.T @ np.linalg.inv(Hm @ P @ Hm.T + R)
        x = x + K @ (np.array([[ms[i]], [bs[i]]]) - Hm @ x)
        P = P - K @ Hm @ P
        assert est.dtype == np.float32 and est.shape == (4, 1)
        np.testing.assert_allclose(est, x, rtol=1e-3, atol=1e-3)
        np.testing.assert_allclose(lane.kf.errorCovPost, P, rtol=2e-3, atol=1e-5)
        out.append(est[:2, 0].copy())
    out = np.array(out)
    # what the script plots: the filtered track is far smoother than the measurements and settles on the lane
    assert np.abs(np.diff(out[20:, 0])).mean() < 0.25 * np.abs(np.diff(ms[20:])).mean()
    assert abs(out[-30:, 0].mean() - 0.5) < 0.05 and abs(out[-30:, 1].mean() - 100) < 2.5
    # a frame without a measurement: predict alone carries the state on (cv2 copies the prediction to statePost)
    before = lane.kf.statePost.copy()
    lane.predict()
    np.testing.assert_allclose(lane.kf.statePost, A.astype(np.float32) @ before, rtol=1e-6)
    with pytest.raises(ValueError):
        KalmanFilter(0, 2)
