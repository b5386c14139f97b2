"""Batched, device-resident lane pipeline (the throughput path).

A `Recipe` names one of the reference's per-frame call sequences; `LanePipeline`
configures the HIP pipeline for it once (ROI mask rasterised once per shape instead
of per frame, as the scripts do at `Simple Lane Detection with OpenCV_2(images).py:14-23`)
and then processes batches of independent frames in `nslots` slots (buffer sets) that overlap on the GPU:

    pipe = LanePipeline(1920, 1080, Recipe.north_star(1920, 1080), max_batch=256, nslots=3)
    pipe.submit(0, frames)            # (n, h, w, 3) uint8, pinned or pageable host memory
    segs = pipe.collect(0)            # list of (k, 4) int32 arrays, one per frame
"""
import ctypes as C
from dataclasses import dataclass, field
from typing import Optional

import numpy as np

from . import _abi
from ._abi import error


def triangle_roi(w, h):
    """Vertices of the Simple Lane Detection scripts' ROI (`..._2(images).py:62-66`, cast to int32 at :77-80)."""
    return np.array([[(0, h), (w / 2, h / 2), (w, h)]], np.int32)


def narrow_triangle_roi(w, h):
    """`Simple Lane Detection with OpenCV_6(mp4).py:54-58`."""
    return np.array([[(int(w * 0.1), h), (w / 2, h / 2), (int(w * 0.9), h)]], np.int32)


def trapezoid_roi(w, h):
    """`F1_vo_Lane_Tracking.py:108-110`."""
    return np.array([[(0, h), (w, h), (w, int(h * 0.4)), (0, int(h * 0.4))]], np.int32)


def f1_car_mask(w, h):
    """`MonocularVO.get_mask` (`F1_vo_Lane_Tracking.py:158-182`) through this module's own fillPoly/rectangle."""
    import lanecv as cv2
    mask = np.full((h, w), 255, dtype=np.uint8)
    pts = np.array([
        [int(w * 0.12), h], [int(w * 0.12), int(h * 0.40)], [int(w * 0.32), int(h * 0.40)],
        [int(w * 0.35), int(h * 0.50)], [int(w * 0.35), int(h * 0.48)], [int(w * 0.65), int(h * 0.48)],
        [int(w * 0.65), int(h * 0.50)], [int(w * 0.68), int(h * 0.40)], [int(w * 0.88), int(h * 0.40)],
        [int(w * 0.88), h]], np.int32)
    cv2.fillPoly(mask, [pts], 0)
    ui_h = int(h * 0.35)
    cv2.rectangle(mask, (0, h - ui_h), (int(w * 0.12), h), 0, -1)
    cv2.rectangle(mask, (int(w * 0.88), h - ui_h), (w, h), 0, -1)
    return mask


@dataclass
class Recipe:
    color_code: int = 6
    blur: bool = False
    canny: tuple = (100, 200)
    roi_vertices: Optional[np.ndarray] = None     # int32 (1, N, 2) as handed to fillPoly, or None
    car_mask: Optional[np.ndarray] = None         # (h, w) uint8, applied as bitwise_and(e, e, mask=car_mask)
    rho: float = 6
    theta: float = np.pi / 60
    threshold: int = 160
    min_line_length: float = 40
    max_line_gap: float = 25
    segments: bool = True
    accumulator: bool = False
    # colour gate in front of the gray conversion: (lo0, hi0, lo1, hi1), each an (H, S, V) triple of the two
    # inRange calls whose masks are ORed, or None
    hsv_gate: Optional[tuple] = None
    points: bool = False      # keep the ordered edge-point list (what masked_edges.nonzero() enumerates)

    @staticmethod
    def simple(w, h, threshold=160, max_line_gap=25):
        """`Simple Lane Detection with OpenCV_2(images).py:54-92` (RGB2GRAY on BGR data, no blur)."""
        return Recipe(color_code=7, blur=False, canny=(100, 200), roi_vertices=triangle_roi(w, h),
                      threshold=threshold, max_line_gap=max_line_gap)

    @staticmethod
    def curve(w, h):
        """`Simple Lane Detection with OpenCV_5(mp4).py:66-84` (threshold 130, gap 150)."""
        return Recipe.simple(w, h, threshold=130, max_line_gap=150)

    @staticmethod
    def hsv_gated(w, h):
        """`Simple Lane Detection with OpenCV_6(mp4).py:54-113`: white and yellow HSV boxes gate the frame,
        BGR2GRAY, Canny(100,200), narrowed triangle, HoughLinesP(6, pi/60, 130, 40, 150)."""
        return Recipe(color_code=6, blur=False, canny=(100, 200), roi_vertices=narrow_triangle_roi(w, h),
                      threshold=130, max_line_gap=150,
                      hsv_gate=((0, 0, 200), (180, 30, 255), (15, 100, 100), (35, 255, 255)))

    @staticmethod
    def f1(w, h):
        """`F1_vo_Lane_Tracking.py:97-112`: blur, Canny(50,150), car mask, trapezoid; no Hough."""
        return Recipe(color_code=6, blur=True, canny=(50, 150), roi_vertices=trapezoid_roi(w, h),
                      car_mask=f1_car_mask(w, h), segments=False, accumulator=False, points=True)

    @staticmethod
    def north_star(w, h):
        """BGR2GRAY -> blur -> Canny(100,200) -> triangle ROI -> accumulator -> HoughLinesP(6, pi/60, 160, 40, 25)."""
        return Recipe(color_code=6, blur=True, canny=(100, 200), roi_vertices=triangle_roi(w, h), accumulator=True)


class LanePipeline:
    def __init__(self, width, height, recipe, max_batch=32, nslots=2, seg_capacity=128,
                 keep_intermediates=False, device=None, src_size=None, timing=False):
        """src_size = (width, height) of the submitted frames when they differ from the pipeline's size: the
        pipeline then starts with ``cv2.resize(frame, (width, height))`` (`F1_vo_Lane_Tracking.py:185`).
        timing: record the HIP events `stage_ms` reads (off by default: they are extra packets on every batch)."""
        import lanecv as cv2
        self.w, self.h, self.recipe = int(width), int(height), recipe
        self.sw, self.sh = (int(src_size[0]), int(src_size[1])) if src_size is not None else (self.w, self.h)
        self.max_batch, self.nslots, self.seg_capacity = int(max_batch), int(nslots), int(seg_capacity)
        self.keep = bool(keep_intermediates)
        self.roi = None
        if recipe.roi_vertices is not None:
            self.roi = np.zeros((self.h, self.w), np.uint8)
            cv2.fillPoly(self.roi, recipe.roi_vertices, 255)
        car = None
        if recipe.car_mask is not None:
            car = np.ascontiguousarray(recipe.car_mask, dtype=np.uint8)
            if car.shape != (self.h, self.w):
                raise error("car_mask shape %s differs from the frame size" % (car.shape,))
        p = _abi.PipeParams(
            width=self.w, height=self.h, max_batch=self.max_batch, nslots=self.nslots,
            color_code=int(recipe.color_code), blur=int(bool(recipe.blur)),
            canny_lo=float(recipe.canny[0]), canny_hi=float(recipe.canny[1]),
            roi_mask=self.roi.ctypes.data if self.roi is not None else None,
            car_mask=car.ctypes.data if car is not None else None,
            rho=float(recipe.rho), theta=float(recipe.theta), threshold=int(recipe.threshold),
            min_line_length=float(recipe.min_line_length), max_line_gap=float(recipe.max_line_gap),
            seg_capacity=self.seg_capacity, want_accumulator=int(bool(recipe.accumulator)),
            want_segments=int(bool(recipe.segments)), keep_intermediates=int(self.keep))
        p.want_points = int(bool(recipe.points))
        p.src_width, p.src_height = self.sw, self.sh
        if recipe.hsv_gate is not None:
            flat = [int(v) for box in recipe.hsv_gate for v in box]
            if len(flat) != 12:
                raise error("hsv_gate: expected (lo0, hi0, lo1, hi1) with three values each")
            p.hsv_gate = 1
            p.hsv_bounds = (C.c_int * 12)(*flat)
        self._h = C.c_void_p()
        self._ctx = _abi.context(device)
        _abi.check(_abi.lib().lc_pipe_create(self._ctx, C.byref(p), C.byref(self._h)))
        self._n = [0] * self.nslots
        if timing:
            self.set_timing(True)
        self._keepalive = [None] * self.nslots     # host frames of the batch in flight, per slot (the copy is asynchronous)
        self._pinned = []                          # page-locked blocks handed out by pinned_frames
        na, nr = C.c_int(), C.c_int()
        _abi.check(_abi.lib().lc_hough_geometry(self.w, self.h, float(recipe.rho), float(recipe.theta),
                                                C.byref(na), C.byref(nr)))
        self.numangle, self.numrho = na.value, nr.value

    def close(self):
        """Destroys the pipeline (which waits for its streams) and frees the page-locked blocks of `pinned_frames`:
        arrays obtained from it must not be used afterwards."""
        h, self._h = getattr(self, "_h", None), None
        if h and _abi is not None and getattr(_abi, "_lib", None) is not None:
            _abi._lib.lc_pipe_destroy(h)
            for p in getattr(self, "_pinned", []):
                _abi._lib.lc_host_free(C.c_void_p(p))
            self._pinned = []
            self._keepalive = [None] * self.nslots

    __del__ = close

    def pinned_frames(self, n=None):
        """A page-locked (n, h, w, 3) uint8 array for overlapped uploads; `close` frees it."""
        n = self.max_batch if n is None else n
        p = C.c_void_p()
        nbytes = n * self.sh * self.sw * 3
        _abi.check(_abi.lib().lc_host_alloc_pinned(nbytes, C.byref(p)))
        self._pinned.append(p.value)
        buf = (C.c_uint8 * nbytes).from_address(p.value)
        return np.frombuffer(buf, np.uint8).reshape(n, self.sh, self.sw, 3)

    def submit(self, slot, frames):
        frames = np.asarray(frames)
        if frames.ndim == 3:
            frames = frames[None]
        if frames.dtype != np.uint8 or frames.shape[1:] != (self.sh, self.sw, 3):
            raise error("submit: expected uint8 frames of shape (n, %d, %d, 3), got %s %s"
                        % (self.sh, self.sw, frames.dtype, frames.shape))
        frames = np.ascontiguousarray(frames)
        _abi.check(_abi.lib().lc_pipe_submit_host(self._h, slot, _abi.ptr(frames), len(frames)))
        self._n[slot] = len(frames)
        # the copy runs asynchronously on the upload stream: the array (possibly a temporary made above) lives until
        # the slot has been waited for
        self._keepalive[slot] = frames

    def upload(self, slot, frames):
        """Copies frames into the slot's own HBM input buffer (synchronous) and returns its device pointer;
        `submit_device(slot, ptr, n)` then runs with the input already resident."""
        frames = np.ascontiguousarray(frames, dtype=np.uint8)
        if frames.shape[1:] != (self.sh, self.sw, 3):
            raise error("upload: expected frames of shape (n, %d, %d, 3), got %s" % (self.sh, self.sw, frames.shape))
        _abi.check(_abi.lib().lc_pipe_upload(self._h, slot, _abi.ptr(frames), len(frames)))
        p = C.c_void_p()
        _abi.check(_abi.lib().lc_pipe_input_device_ptr(self._h, slot, C.byref(p)))
        return p.value

    def submit_device(self, slot, dev_ptr, nframes):
        """Frames already in HBM (e.g. a torch uint8 tensor's data_ptr()), [n][h][w][3] dense."""
        _abi.check(_abi.lib().lc_pipe_submit_device(self._h, slot, C.c_void_p(int(dev_ptr)), int(nframes)))
        self._n[slot] = int(nframes)

    def wait(self, slot):
        """Blocks until the slot is finished; returns (seg [n, cap, 4] int32, count [n] int32) or (None, None)."""
        n = self._n[slot]
        if self.recipe.segments:
            seg = np.empty((n, self.seg_capacity, 4), np.int32)
            cnt = np.empty(n, np.int32)
            try:
                _abi.check(_abi.lib().lc_pipe_wait(self._h, slot, _abi.ptr(seg), _abi.ptr(cnt)))
            finally:
                self._keepalive[slot] = None
            return seg, cnt
        try:
            _abi.check(_abi.lib().lc_pipe_wait(self._h, slot, None, None))
        finally:
            self._keepalive[slot] = None
        return None, None

    def collect(self, slot):
        """Per frame what cv2.HoughLinesP returns: an (N, 1, 4) int32 array or None."""
        seg, cnt = self.wait(slot)
        if seg is None:
            return None
        out = []
        for i in range(len(cnt)):
            if cnt[i] > self.seg_capacity:
                raise error("frame %d has %d segments, seg_capacity is %d" % (i, cnt[i], self.seg_capacity), _abi.LC_ECAP)
            out.append(seg[i, :cnt[i]].reshape(-1, 1, 4).copy() if cnt[i] else None)
        return out

    def read(self, slot, what):
        n = self._n[slot]
        if what == _abi.BUF_ACC:
            out = np.empty((n, self.numangle, self.numrho), np.int32)
        elif what == _abi.BUF_NPOINTS:
            out = np.empty(n, np.int32)
        elif what == _abi.BUF_RESIZED:
            out = np.empty((n, self.h, self.w, 3), np.uint8)
        else:
            out = np.empty((n, self.h, self.w), np.uint8)
        _abi.check(_abi.lib().lc_pipe_read(self._h, slot, what, _abi.ptr(out), out.nbytes))
        return out

    def nonzero(self, slot):
        """Per frame ``(ys, xs)`` of the masked edge map's non-zero pixels in row-major order -- the arrays
        ``masked_edges.nonzero()`` returns at `F1_vo_Lane_Tracking.py:117-119` -- from the device's point list."""
        counts = self.read(slot, _abi.BUF_NPOINTS)
        out = []
        for i, c in enumerate(counts):
            xy = np.empty(max(int(c), 1), np.uint32)
            n = C.c_int()
            _abi.check(_abi.lib().lc_pipe_read_points(self._h, slot, i, _abi.ptr(xy), len(xy), C.byref(n)))
            xy = xy[:n.value]
            out.append(((xy >> 16).astype(np.int64), (xy & 0xFFFF).astype(np.int64)))
        return out

    def hyst_rounds(self, slot):
        """(max over frames, total) cross-tile hysteresis closure rounds of the slot's last batch."""
        a, b = C.c_int(), C.c_int()
        _abi.check(_abi.lib().lc_pipe_hyst_rounds(self._h, slot, C.byref(a), C.byref(b)))
        return a.value, b.value

    def row_skip(self, slot=0):
        """(first row of which the edge kernel classifies anything, frames of the slot's last batch that needed the whole
        frame after all)."""
        first, _, redone = self.roi_skip(slot)
        return first, redone

    def roi_skip(self, slot=0):
        """(first row classified, pixels per frame the edge kernel classifies, frames of the slot's last batch redone in full)."""
        a, b, c = C.c_int(), C.c_longlong(), C.c_int()
        _abi.check(_abi.lib().lc_pipe_roi_skip(self._h, slot, C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value

    def shared_queues(self):
        """Streams of the pipeline that found no hardware queue of their own (0 = full overlap)."""
        n = C.c_int()
        _abi.check(_abi.lib().lc_pipe_shared_queues(self._h, C.byref(n)))
        return n.value

    def set_timing(self, on):
        """Batches submitted from now on record (or no longer record) the events `stage_ms` reads."""
        _abi.check(_abi.lib().lc_pipe_set_timing(self._h, int(bool(on))))

    def event_times(self, slot):
        """ms since the pipeline's creation at which the slot's last batch passed each of its twelve events
        (negative: not recorded); needs timing."""
        out = (C.c_float * 12)()
        _abi.check(_abi.lib().lc_pipe_event_times(self._h, slot, out))
        return list(out)

    def rows_uploaded(self, slot):
        """First row of every frame of the slot's last batch that `submit` copied to the GPU (0: whole frames)."""
        v = C.c_int()
        _abi.check(_abi.lib().lc_pipe_rows_uploaded(self._h, slot, C.byref(v)))
        return v.value

    def bytes_uploaded(self, slot):
        """Bytes per frame of the slot's last batch that `submit` copied to the GPU (0 for device submits)."""
        v = C.c_longlong()
        _abi.check(_abi.lib().lc_pipe_bytes_uploaded(self._h, slot, C.byref(v)))
        return v.value

    def layout(self):
        """(slots per HoughLinesP launch, HoughLinesP streams, levels of the ROI scheme)."""
        a, b, c = C.c_int(), C.c_int(), C.c_int()
        _abi.check(_abi.lib().lc_pipe_layout(self._h, C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value

    def stage_ms(self, slot, stage):
        ms = C.c_float()
        _abi.check(_abi.lib().lc_pipe_stage_ms(self._h, slot, stage, C.byref(ms)))
        return ms.value
