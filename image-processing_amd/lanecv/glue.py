"""Host-side glue behind HoughLinesP: what the lane scripts do with the segments.

Pure NumPy, no GPU work: a frame's segment list is a few hundred bytes.  Each function follows one of the
reference's post-Hough blocks and keeps its observable quirks:

* a segment's slope is ``(y2 - y1) / (x2 - x1)`` on NumPy integers, so a vertical segment has slope +-inf
  (or nan for a zero-length one) instead of raising; ``abs(slope) < limit`` drops near-horizontal segments,
  ``slope <= 0`` sends a segment to the left lane, anything else -- +inf and nan included -- to the right
  (`Simple Lane Detection with OpenCV_2(images).py:101-113`);
* both end points of every kept segment enter a least-squares fit of x as a polynomial in y (`np.polyfit`,
  the routine the scripts call), evaluated at the bottom row and at 3/5 of the height and truncated with
  ``int()`` (`..._2(images).py:116-143`);
* the video scripts smooth over frames with alpha = 0.1: `..._4(mp4).py:144-178` averages the four integers of
  each lane line (truncating again), `..._5(mp4).py:108-138` the three coefficients of a degree-2 fit;
* the F1 tracker works on edge pixels, not segments (`F1_vo_Lane_Tracking.py:114-132`, `:42-91`): the non-zero
  pixels of the masked edge map split at the middle column, a degree-2 fit x(y) per side when more than 50
  pixels are there, blended with the previous fit as ``0.7 * previous + 0.3 * new``, and sampled at every row
  into the int32 polyline the script draws.
"""
import numpy as np

__all__ = ["split_by_slope", "lane_lines", "LineSmoother", "CurveSmoother", "split_at_midpoint", "PixelCurveTracker"]


def split_by_slope(lines, min_abs_slope=0.5):
    """lines: what HoughLinesP returns -- ``(N, 1, 4)`` int32 or None.  Returns four 1-D int arrays
    ``(left_x, left_y, right_x, right_y)`` holding both end points of the kept segments, in segment order."""
    empty = np.zeros(0, dtype=np.int64)
    if lines is None:
        return empty, empty, empty, empty
    seg = np.asarray(lines).reshape(-1, 4).astype(np.int64)
    x1, y1, x2, y2 = seg[:, 0], seg[:, 1], seg[:, 2], seg[:, 3]
    with np.errstate(divide="ignore", invalid="ignore"):
        slope = (y2 - y1) / (x2 - x1)
    keep = ~(np.abs(slope) < min_abs_slope)          # nan compares False, as in the scripts: it stays
    left = keep & (slope <= 0)
    right = keep & ~(slope <= 0)

    def both_ends(a, b, sel):
        return np.stack([a[sel], b[sel]], axis=1).reshape(-1)

    return both_ends(x1, x2, left), both_ends(y1, y2, left), both_ends(x1, x2, right), both_ends(y1, y2, right)


def _fit_line(xs, ys, max_y, min_y):
    """[x at max_y, max_y, x at min_y, min_y] of the degree-1 fit x(y), or None without points."""
    if len(xs) == 0:
        return None
    poly = np.poly1d(np.polyfit(ys, xs, deg=1))
    return [int(poly(max_y)), max_y, int(poly(min_y)), min_y]


def lane_lines(lines, height, min_abs_slope=0.5, min_y_frac=3 / 5):
    """The two lane lines of `..._2(images).py:116-153`: ``[[xl_start, max_y, xl_end, min_y], [xr_start, max_y,
    xr_end, min_y]]`` with max_y = height and min_y = int(height * min_y_frac) -- 3 / 5 in most scripts, 0.65 in
    `Simple Lane Detection with OpenCV_6(mp4).py:141`; a side without segments gives x = 0."""
    lx, ly, rx, ry = split_by_slope(lines, min_abs_slope)
    min_y, max_y = int(height * min_y_frac), int(height)
    left = _fit_line(lx, ly, max_y, min_y) or [0, max_y, 0, min_y]
    right = _fit_line(rx, ry, max_y, min_y) or [0, max_y, 0, min_y]
    return [left, right]


class LineSmoother:
    """Frame-to-frame average of the two lane lines (`..._4(mp4).py:144-178`): the first line seen on a side is
    taken as is, later ones enter as ``int(prev * (1 - alpha) + curr * alpha)`` per integer; a frame without
    segments on a side leaves that side's line unchanged."""

    def __init__(self, alpha=0.1, min_abs_slope=0.3, min_y_frac=3 / 5):
        """min_y_frac: the lines are drawn up to row int(height * min_y_frac) -- 3 / 5 in `..._4(mp4).py`, 0.65 in
        `Simple Lane Detection with OpenCV_6(mp4).py:141`."""
        self.alpha, self.min_abs_slope, self.min_y_frac = alpha, min_abs_slope, min_y_frac
        self.left = None
        self.right = None

    def _blend(self, prev, curr):
        if curr is None:
            return prev
        if prev is None:
            return curr
        a = self.alpha
        return [int(p * (1 - a) + c * a) for p, c in zip(prev, curr)]

    def update(self, lines, height):
        """Returns the list of lines to draw (0, 1 or 2 entries, left first)."""
        lx, ly, rx, ry = split_by_slope(lines, self.min_abs_slope)
        min_y, max_y = int(height * self.min_y_frac), int(height)
        self.left = self._blend(self.left, _fit_line(lx, ly, max_y, min_y))
        self.right = self._blend(self.right, _fit_line(rx, ry, max_y, min_y))
        return [ln for ln in (self.left, self.right) if ln is not None]


class CurveSmoother:
    """Degree-2 fit x(y) per side with the coefficients averaged over frames (`..._5(mp4).py:108-138`).  A fit
    that NumPy refuses (the scripts swallow the exception) counts as no fit for that frame."""

    def __init__(self, alpha=0.1, min_abs_slope=0.3):
        self.alpha, self.min_abs_slope = alpha, min_abs_slope
        self.left = None
        self.right = None

    @staticmethod
    def _fit(xs, ys):
        if len(xs) == 0:
            return None
        try:
            return np.polyfit(ys, xs, 2)
        except Exception:
            return None

    def _blend(self, prev, curr):
        if curr is None:
            return prev
        if prev is None:
            return curr
        return prev * (1 - self.alpha) + curr * self.alpha

    def update(self, lines):
        """Returns ``(left_coefficients, right_coefficients)``; a side never seen yet is None."""
        lx, ly, rx, ry = split_by_slope(lines, self.min_abs_slope)
        self.left = self._blend(self.left, self._fit(lx, ly))
        self.right = self._blend(self.right, self._fit(rx, ry))
        return self.left, self.right


def split_at_midpoint(nonzeroy, nonzerox, width):
    """`F1_vo_Lane_Tracking.py:116-129`: ``(leftx, lefty, rightx, righty)`` with left = columns below
    ``width // 2``; the pixel order (row-major) is kept."""
    nonzeroy, nonzerox = np.asarray(nonzeroy), np.asarray(nonzerox)
    midpoint = width // 2
    left = nonzerox < midpoint
    right = nonzerox >= midpoint
    return nonzerox[left], nonzeroy[left], nonzerox[right], nonzeroy[right]


class PixelCurveTracker:
    """`LaneTracker.fit_poly` (`F1_vo_Lane_Tracking.py:33-91`): per side a degree-2 fit x(y) through the edge
    pixels with memory.  A side with at most `min_points` pixels, or a fit that raises TypeError, reuses the
    previous coefficients; a side never fitted yields None."""

    def __init__(self, alpha=0.7, min_points=50):
        self.alpha, self.min_points = alpha, min_points
        self.prev_left_fit = None
        self.prev_right_fit = None

    def _side(self, xs, ys, prev):
        if len(ys) > self.min_points:
            try:
                fit = np.polyfit(ys, xs, 2)
                if prev is not None:
                    fit = self.alpha * prev + (1 - self.alpha) * fit
                return fit, fit
            except TypeError:
                return prev, prev
        return prev, prev

    @staticmethod
    def _curve(fit, ploty):
        if fit is None:
            return None
        fitx = fit[0] * ploty ** 2 + fit[1] * ploty + fit[2]
        return np.array([np.transpose(np.vstack([fitx, ploty]))]).astype(np.int32)

    def fit_poly(self, shape, leftx, lefty, rightx, righty):
        """Returns ``(left_curve, right_curve)``: int32 arrays of shape (1, height, 2) holding (x, y) per row,
        or None for a side without a fit so far."""
        ploty = np.linspace(0, shape[0] - 1, shape[0])
        left_fit, self.prev_left_fit = self._side(leftx, lefty, self.prev_left_fit)
        right_fit, self.prev_right_fit = self._side(rightx, righty, self.prev_right_fit)
        return self._curve(left_fit, ploty), self._curve(right_fit, ploty)


class KalmanFilter:
    """The linear Kalman filter of ``cv2.KalmanFilter(dynamParams, measureParams, controlParams)`` as the lane
    stabiliser uses it (`1000_Lane_Stabilizer.py:33-63`): the same attribute names, float32 state, and the
    predict / correct recurrences OpenCV documents --

        predict:  x' = A x (+ B u),  P' = A P A^T + Q               (and x, P take these values until corrected)
        correct:  K = P' H^T (H P' H^T + R)^-1,  x = x' + K (z - H x'),  P = P' - K H P'

    Host-side glue on a 4-vector per lane: nothing here touches the GPU.  cv2 solves for K with an SVD in
    float32; this class solves the same small system in float32 with numpy, so the last bits may differ."""

    def __init__(self, dynamParams, measureParams, controlParams=0, dtype=np.float32):
        dp, mp, cp = int(dynamParams), int(measureParams), int(controlParams)
        if dp <= 0 or mp <= 0 or cp < 0:
            raise ValueError("KalmanFilter: dynamParams and measureParams must be positive")
        self._dtype = dtype
        self.statePre = np.zeros((dp, 1), dtype)
        self.statePost = np.zeros((dp, 1), dtype)
        self.transitionMatrix = np.eye(dp, dtype=dtype)
        self.processNoiseCov = np.eye(dp, dtype=dtype)
        self.measurementMatrix = np.zeros((mp, dp), dtype)
        self.measurementNoiseCov = np.eye(mp, dtype=dtype)
        self.errorCovPre = np.zeros((dp, dp), dtype)
        self.errorCovPost = np.zeros((dp, dp), dtype)
        self.gain = np.zeros((dp, mp), dtype)
        self.controlMatrix = np.zeros((dp, cp), dtype) if cp > 0 else None

    def _m(self, a):
        return np.asarray(a, dtype=self._dtype)

    def predict(self, control=None):
        A = self._m(self.transitionMatrix)
        self.statePre = A @ self._m(self.statePost)
        if control is not None and self.controlMatrix is not None:
            self.statePre = self.statePre + self._m(self.controlMatrix) @ self._m(control).reshape(-1, 1)
        self.errorCovPre = (A @ self._m(self.errorCovPost)) @ A.T + self._m(self.processNoiseCov)
        # cv2 copies the prediction into the posterior, so that a frame without a measurement keeps predicting
        self.statePost = self.statePre.copy()
        self.errorCovPost = self.errorCovPre.copy()
        return self.statePre

    def correct(self, measurement):
        H = self._m(self.measurementMatrix)
        z = self._m(measurement).reshape(-1, 1)
        HP = H @ self._m(self.errorCovPre)                       # H P'
        S = HP @ H.T + self._m(self.measurementNoiseCov)         # innovation covariance
        self.gain = np.linalg.solve(S, HP).T.astype(self._dtype)  # K = (S^-1 H P')^T, S and P' symmetric
        self.statePost = self._m(self.statePre) + self.gain @ (z - H @ self._m(self.statePre))
        self.errorCovPost = self._m(self.errorCovPre) - self.gain @ HP
        return self.statePost


class LaneKalmanFilter:
    """`1000_Lane_Stabilizer.py:28-63`: a lane as (slope m, intercept b) with their per-frame changes, constant
    velocity model, process noise 1e-4, measurement noise 0.1, unit initial covariance."""

    def __init__(self):
        kf = KalmanFilter(4, 2, 0)
        kf.transitionMatrix = np.array([[1, 0, 1, 0], [0, 1, 0, 1], [0, 0, 1, 0], [0, 0, 0, 1]], np.float32)
        kf.measurementMatrix = np.array([[1, 0, 0, 0], [0, 1, 0, 0]], np.float32)
        kf.processNoiseCov = np.eye(4, dtype=np.float32) * 1e-4
        kf.measurementNoiseCov = np.eye(2, dtype=np.float32) * 0.1
        kf.errorCovPost = np.eye(4, dtype=np.float32)
        self.kf = kf

    def predict(self):
        return self.kf.predict()

    def correct(self, m, b):
        return self.kf.correct(np.array([[np.float32(m)], [np.float32(b)]]))
