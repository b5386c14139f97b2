"""lanecv -- the cv2 subset the reference's lane scripts call, on MI355X.

`import lanecv as cv2` keeps `pipeline(image)` of the `Simple Lane Detection with
OpenCV_*.py` scripts and `LaneTracker.detect_and_draw` of `F1_vo_Lane_Tracking.py`
source-compatible for these calls:

    cvtColor (COLOR_BGR2GRAY / COLOR_RGB2GRAY / COLOR_BGR2HSV / COLOR_RGB2HSV),
    resize (INTER_LINEAR, explicit dsize), GaussianBlur ((5,5), sigma 0), Canny (aperture 3, L1), fillPoly (LINE_8, shift 0),
    rectangle (filled), line / polylines (LINE_8, any thickness), inRange (8UC3, integer bounds), bitwise_or, bitwise_and (optional mask),
    HoughLinesP, addWeighted

Same positional/keyword signatures and return conventions as cv2 (fillPoly and
rectangle work in place and return the image; HoughLinesP returns an (N,1,4)
int32 array or None).  Arguments outside the supported domain raise
`lanecv.error` -- nothing silently differs and nothing falls back to the CPU.
Every call runs HIP kernels through the C ABI of liblanecv.so; the batched,
device-resident path is `lanecv.pipeline.LanePipeline`.
"""
import ctypes as _C

import numpy as _np

from . import _abi
from ._abi import error
from .glue import KalmanFilter   # host-side, as cv2.KalmanFilter is (`1000_Lane_Stabilizer.py:33`)

__all__ = ["cvtColor", "resize", "GaussianBlur", "Canny", "fillPoly", "rectangle", "line", "polylines", "inRange", "bitwise_or",
           "bitwise_and", "HoughLinesP", "addWeighted", "hough_accumulator", "KalmanFilter", "error"]

COLOR_BGR2GRAY = 6
COLOR_RGB2GRAY = 7
COLOR_BGR2HSV = 40
COLOR_RGB2HSV = 41
INTER_LINEAR = 1
BORDER_REPLICATE = 1
BORDER_REFLECT_101 = 4
BORDER_REFLECT101 = 4
BORDER_DEFAULT = 4
LINE_4 = 4
LINE_8 = 8
LINE_AA = 16
FILLED = -1


def _image(a, name, ndim, writable=False):
    if not isinstance(a, _np.ndarray):
        raise error("%s is not a numpy array" % name)
    if a.dtype != _np.uint8:
        raise error("%s: only 8-bit unsigned images are supported (got %s)" % (name, a.dtype), _abi.LC_EUNSUPPORTED)
    if a.ndim != ndim or a.size == 0:
        raise error("%s: expected a non-empty %d-dimensional image, got shape %s" % (name, ndim, a.shape))
    if writable and not a.flags.writeable:
        raise error("%s is read-only" % name)
    return a


def _rows(a):
    """(array to pass, row stride in bytes): pixels of a row must be dense, rows may be strided."""
    if a.ndim == 2:
        dense = a.strides[1] == 1 and a.strides[0] >= a.shape[1]
    else:
        dense = a.strides[2] == 1 and a.strides[1] == a.shape[2] and a.strides[0] >= a.shape[1] * a.shape[2]
    if dense:
        return a, a.strides[0]
    c = _np.ascontiguousarray(a)
    return c, c.strides[0]


def _dst(dst, shape):
    if isinstance(dst, _np.ndarray) and dst.shape == shape and dst.dtype == _np.uint8 and dst.flags.c_contiguous \
            and dst.flags.writeable:
        return dst
    return _np.empty(shape, _np.uint8)


def cvtColor(src, code, dst=None, dstCn=0):
    """cv2.cvtColor for COLOR_BGR2GRAY / COLOR_RGB2GRAY (8UC3 -> 8UC1) and COLOR_BGR2HSV / COLOR_RGB2HSV
    (8UC3 -> 8UC3, hue in [0, 180)).
    Reference: Simple Lane Detection with OpenCV_2(images).py:69, F1_vo_Lane_Tracking.py:186,
    Simple Lane Detection with OpenCV_6(mp4).py:65."""
    if code not in (COLOR_BGR2GRAY, COLOR_RGB2GRAY, COLOR_BGR2HSV, COLOR_RGB2HSV):
        raise error("cvtColor: unsupported conversion code %r" % (code,), _abi.LC_EUNSUPPORTED)
    src = _image(src, "src", 3)
    if src.shape[2] != 3:
        raise error("cvtColor: expected 3 channels, got %d" % src.shape[2], _abi.LC_EUNSUPPORTED)
    s, stride = _rows(src)
    h, w = s.shape[:2]
    if code in (COLOR_BGR2HSV, COLOR_RGB2HSV):
        if dstCn not in (0, 3):
            raise error("cvtColor: dstCn must be 0 or 3 for an HSV destination")
        out = _dst(dst, (h, w, 3))
        _abi.check(_abi.lib().lc_cvt_hsv(_abi.context(), _abi.ptr(s), w, h, stride, int(code), _abi.ptr(out), 3 * w))
        return out
    if dstCn not in (0, 1):
        raise error("cvtColor: dstCn must be 0 or 1 for a gray destination")
    out = _dst(dst, (h, w))
    _abi.check(_abi.lib().lc_cvt_gray(_abi.context(), _abi.ptr(s), w, h, stride, int(code), _abi.ptr(out), w))
    return out


def resize(src, dsize, dst=None, fx=0, fy=0, interpolation=INTER_LINEAR):
    """cv2.resize(src, (width, height)) with INTER_LINEAR on an 8-bit image of 1 or 3 channels.
    Reference: F1_vo_Lane_Tracking.py:185."""
    if not isinstance(src, _np.ndarray) or src.ndim not in (2, 3):
        raise error("resize: src must be a 2- or 3-dimensional numpy array")
    src = _image(src, "src", src.ndim)
    cn = 1 if src.ndim == 2 else src.shape[2]
    if cn not in (1, 3):
        raise error("resize: 1 or 3 channels are supported, got %d" % cn, _abi.LC_EUNSUPPORTED)
    if interpolation != INTER_LINEAR:
        raise error("resize: only INTER_LINEAR is supported", _abi.LC_EUNSUPPORTED)
    if dsize is None or len(dsize) != 2 or int(dsize[0]) <= 0 or int(dsize[1]) <= 0 or fx or fy:
        raise error("resize: an explicit positive dsize (width, height) is required; fx / fy are not supported",
                    _abi.LC_EUNSUPPORTED)
    dw, dh = int(dsize[0]), int(dsize[1])
    s, stride = _rows(src)
    sh, sw = s.shape[:2]
    out = _dst(dst, (dh, dw) if src.ndim == 2 else (dh, dw, cn))
    _abi.check(_abi.lib().lc_resize_linear(_abi.context(), _abi.ptr(s), sw, sh, stride, cn, _abi.ptr(out), dw, dh, dw * cn))
    return out


def GaussianBlur(src, ksize, sigmaX, dst=None, sigmaY=0, borderType=BORDER_DEFAULT):
    """cv2.GaussianBlur(src, (5, 5), 0) on 8UC1.  Reference: F1_vo_Lane_Tracking.py:97."""
    src = _image(src, "src", 2)
    if tuple(ksize) != (5, 5) or sigmaX != 0 or sigmaY != 0:
        raise error("GaussianBlur: only ksize (5,5) with sigma 0 is supported", _abi.LC_EUNSUPPORTED)
    if borderType not in (BORDER_DEFAULT,):
        raise error("GaussianBlur: only BORDER_DEFAULT / BORDER_REFLECT_101 is supported", _abi.LC_EUNSUPPORTED)
    s, stride = _rows(src)
    h, w = s.shape
    out = _dst(dst, (h, w))
    _abi.check(_abi.lib().lc_gaussian5(_abi.context(), _abi.ptr(s), w, h, stride, _abi.ptr(out), w))
    return out


def Canny(image, threshold1, threshold2, edges=None, apertureSize=3, L2gradient=False):
    """cv2.Canny(image, threshold1, threshold2) on 8UC1, aperture 3, L1 gradient.
    Reference: Simple Lane Detection with OpenCV_2(images).py:72, F1_vo_Lane_Tracking.py:98."""
    image = _image(image, "image", 2)
    if apertureSize != 3 or L2gradient:
        raise error("Canny: only apertureSize=3, L2gradient=False is supported", _abi.LC_EUNSUPPORTED)
    s, stride = _rows(image)
    h, w = s.shape
    out = _dst(edges, (h, w))
    _abi.check(_abi.lib().lc_canny(_abi.context(), _abi.ptr(s), w, h, stride, float(threshold1), float(threshold2),
                                   _abi.ptr(out), w))
    return out


def _color(color):
    c = color[0] if isinstance(color, (tuple, list, _np.ndarray)) else color
    return int(min(255, max(0, _np.rint(float(c)))))


def _inplace(img, name):
    img = _image(img, name, 2, writable=True)
    if img.strides[1] != 1 or img.strides[0] < img.shape[1]:
        raise error("%s: rows of an in-place target must be dense" % name, _abi.LC_EUNSUPPORTED)
    return img


def fillPoly(img, pts, color, lineType=LINE_8, shift=0, offset=(0, 0)):
    """cv2.fillPoly on an 8UC1 image, in place (LINE_8, shift 0, no offset).  `pts` is a sequence
    (or 3-D array) of polygons; integer dtypes other than int32 are accepted and cast (cv2 itself
    insists on int32 -- the reference's trapezoid at F1_vo_Lane_Tracking.py:108-110 is int64 on Linux).
    Reference: Simple Lane Detection with OpenCV_2(images).py:20, F1_vo_Lane_Tracking.py:111, :175."""
    img = _inplace(img, "img")
    if lineType not in (LINE_8, 0) or shift != 0 or tuple(offset) != (0, 0):
        raise error("fillPoly: only lineType=LINE_8, shift=0, offset=(0,0) are supported", _abi.LC_EUNSUPPORTED)
    polys = []
    for p in pts:
        p = _np.asarray(p)
        if not _np.issubdtype(p.dtype, _np.integer):
            raise error("fillPoly: polygon vertices must be integers (cv2 asserts CV_32S), got %s" % p.dtype)
        if p.size and (p.ndim < 2 or p.shape[-1] != 2):
            raise error("fillPoly: each polygon must be an (N, 2) or (N, 1, 2) array, got shape %s" % (p.shape,))
        polys.append(_np.ascontiguousarray(p.reshape(-1, 2), dtype=_np.int32))
    flat = _np.ascontiguousarray(_np.concatenate(polys, axis=0)) if polys else _np.zeros((0, 2), _np.int32)
    npts = (_C.c_int * max(1, len(polys)))(*[len(p) for p in polys])
    h, w = img.shape
    _abi.check(_abi.lib().lc_fill_poly(_abi.context(), _abi.ptr(img), w, h, img.strides[0], _abi.ptr(flat),
                                       _C.cast(npts, _C.c_void_p), len(polys), _color(color)))
    return img


def rectangle(img, pt1, pt2, color, thickness=1, lineType=LINE_8, shift=0):
    """cv2.rectangle with thickness < 0 (filled) on an 8UC1 image, in place.
    Reference: F1_vo_Lane_Tracking.py:179-180."""
    img = _inplace(img, "img")
    if thickness >= 0:
        raise error("rectangle: only filled rectangles (thickness=-1) are supported", _abi.LC_EUNSUPPORTED)
    if shift != 0:
        raise error("rectangle: shift must be 0", _abi.LC_EUNSUPPORTED)
    h, w = img.shape
    _abi.check(_abi.lib().lc_rectangle_filled(_abi.context(), _abi.ptr(img), w, h, img.strides[0],
                                              int(pt1[0]), int(pt1[1]), int(pt2[0]), int(pt2[1]), _color(color)))
    return img


def _draw_target(img, name):
    if not isinstance(img, _np.ndarray) or img.ndim not in (2, 3):
        raise error("%s: img must be a 2- or 3-dimensional numpy array" % name)
    img = _image(img, "img", img.ndim, writable=True)
    cn = 1 if img.ndim == 2 else img.shape[2]
    if cn > 4:
        raise error("%s: at most 4 channels are supported, got %d" % (name, cn), _abi.LC_EUNSUPPORTED)
    dense = img.strides[-1] == 1 and (img.ndim == 2 or img.strides[1] == cn) and img.strides[0] >= img.shape[1] * cn
    if not dense:
        raise error("%s: pixels of a row of an in-place target must be dense" % name, _abi.LC_EUNSUPPORTED)
    return img, cn


def _color_n(color, cn):
    """cv2's Scalar: up to four numbers, missing ones 0; one value per channel, rounded and clamped."""
    c = list(color) if isinstance(color, (tuple, list, _np.ndarray)) else [color]
    c = [float(v) for v in c][:4] + [0.0] * 4
    return (_C.c_int * 4)(*[int(min(255, max(0, _np.rint(v)))) for v in c[:4]])


def _line_args(name, thickness, lineType, shift):
    if lineType not in (LINE_8, 0) or shift != 0:
        raise error("%s: only lineType=LINE_8 and shift=0 are supported" % name, _abi.LC_EUNSUPPORTED)
    if int(thickness) != thickness or thickness < 1:
        raise error("%s: thickness must be a positive integer, got %r" % (name, thickness))
    return int(thickness)


def line(img, pt1, pt2, color, thickness=1, lineType=LINE_8, shift=0):
    """cv2.line on an 8-bit image of 1..4 channels, in place (LINE_8, shift 0); returns img.
    Reference: Simple Lane Detection with OpenCV_2(images).py:47."""
    img, cn = _draw_target(img, "line")
    t = _line_args("line", thickness, lineType, shift)
    h, w = img.shape[:2]
    _abi.check(_abi.lib().lc_line(_abi.context(), _abi.ptr(img), w, h, img.strides[0], cn, int(pt1[0]), int(pt1[1]),
                                  int(pt2[0]), int(pt2[1]), _color_n(color, cn), t))
    return img


def polylines(img, pts, isClosed, color, thickness=1, lineType=LINE_8, shift=0):
    """cv2.polylines on an 8-bit image of 1..4 channels, in place; `pts` is a sequence (or 3-D array) of
    polygons of integer vertices, (N, 2) or (N, 1, 2) each; returns img.
    Reference: Simple Lane Detection with OpenCV_5(mp4).py:41, F1_vo_Lane_Tracking.py:138."""
    img, cn = _draw_target(img, "polylines")
    t = _line_args("polylines", thickness, lineType, shift)
    polys = []
    for p in pts:
        p = _np.asarray(p)
        if not _np.issubdtype(p.dtype, _np.integer):
            raise error("polylines: polygon vertices must be integers (cv2 asserts CV_32S), got %s" % p.dtype)
        if p.size and (p.ndim < 2 or p.shape[-1] != 2):
            raise error("polylines: each polygon must be an (N, 2) or (N, 1, 2) array, got shape %s" % (p.shape,))
        polys.append(_np.ascontiguousarray(p.reshape(-1, 2), dtype=_np.int32))
    flat = _np.ascontiguousarray(_np.concatenate(polys, axis=0)) if polys else _np.zeros((0, 2), _np.int32)
    npts = (_C.c_int * max(1, len(polys)))(*[len(p) for p in polys])
    h, w = img.shape[:2]
    _abi.check(_abi.lib().lc_polylines(_abi.context(), _abi.ptr(img), w, h, img.strides[0], cn, _abi.ptr(flat),
                                       _C.cast(npts, _C.c_void_p), len(polys), int(bool(isClosed)), _color_n(color, cn), t))
    return img


def _bounds3(b, name):
    a = _np.asarray(b)
    if a.size != 3:
        raise error("inRange: %s must hold three values for a 3-channel image, got shape %s" % (name, a.shape),
                    _abi.LC_EUNSUPPORTED)
    a = a.reshape(3)
    if not _np.all(a == _np.floor(a)):
        raise error("inRange: only integral bounds are supported", _abi.LC_EUNSUPPORTED)
    return (_C.c_int * 3)(*[int(max(-1, min(256, v))) for v in a])


def inRange(src, lowerb, upperb, dst=None):
    """cv2.inRange on an 8UC3 image with three integral bounds per side: 255 where every channel lies in
    [lowerb, upperb], else 0.  Reference: Simple Lane Detection with OpenCV_6(mp4).py:69-78."""
    src = _image(src, "src", 3)
    if src.shape[2] != 3:
        raise error("inRange: expected 3 channels, got %d" % src.shape[2], _abi.LC_EUNSUPPORTED)
    lo, hi = _bounds3(lowerb, "lowerb"), _bounds3(upperb, "upperb")
    s, stride = _rows(src)
    h, w = s.shape[:2]
    out = _dst(dst, (h, w))
    _abi.check(_abi.lib().lc_in_range3(_abi.context(), _abi.ptr(s), w, h, stride, lo, hi, _abi.ptr(out), w))
    return out


def bitwise_or(src1, src2, dst=None, mask=None):
    """cv2.bitwise_or on 8UC1 images (no mask).  Reference: Simple Lane Detection with OpenCV_6(mp4).py:81."""
    a = _image(src1, "src1", 2)
    b = _image(src2, "src2", 2)
    if a.shape != b.shape:
        raise error("bitwise_or: sizes of the inputs differ: %s vs %s" % (a.shape, b.shape))
    if mask is not None:
        raise error("bitwise_or: a mask is not supported", _abi.LC_EUNSUPPORTED)
    sa, ra = _rows(a)
    sb, rb = _rows(b)
    h, w = a.shape
    out = _dst(dst, (h, w))
    _abi.check(_abi.lib().lc_or(_abi.context(), _abi.ptr(sa), ra, _abi.ptr(sb), rb, _abi.ptr(out), w, w, h))
    return out


def _and3(src1, src2, dst, mask):
    """bitwise_and on 8UC3 images: the channels are ANDed as one (h, 3w) plane, then the 8UC1 mask blacks out
    whole pixels (`bitwise_and(image, image, mask=m)`, Simple Lane Detection with OpenCV_6(mp4).py:84)."""
    a = _image(src1, "src1", 3)
    b = _image(src2, "src2", 3)
    if a.shape != b.shape or a.shape[2] != 3:
        raise error("bitwise_and: expected two 3-channel images of one size, got %s and %s" % (a.shape, b.shape))
    h, w = a.shape[:2]
    a = _np.ascontiguousarray(a)
    if src1 is src2:
        both = a
    else:
        both = bitwise_and(a.reshape(h, 3 * w), _np.ascontiguousarray(b).reshape(h, 3 * w)).reshape(h, w, 3)
    if mask is None:
        if isinstance(dst, _np.ndarray) and dst.shape == both.shape and dst.dtype == _np.uint8:
            dst[...] = both
            return dst
        return both.copy() if both is a else both
    mask = _image(mask, "mask", 2)
    if mask.shape != (h, w):
        raise error("bitwise_and: mask size %s differs from the image size %s" % (mask.shape, (h, w)))
    sm, rm = _rows(mask)
    out = _dst(dst, (h, w, 3))
    _abi.check(_abi.lib().lc_and3_mask(_abi.context(), _abi.ptr(both), 3 * w, _abi.ptr(sm), rm, _abi.ptr(out), 3 * w, w, h))
    return out


def bitwise_and(src1, src2, dst=None, mask=None):
    """cv2.bitwise_and on 8UC1 (or 8UC3) images with an optional 8UC1 mask (dst is zero where mask is).
    Reference: Simple Lane Detection with OpenCV_2(images).py:23, F1_vo_Lane_Tracking.py:102, :112,
    Simple Lane Detection with OpenCV_6(mp4).py:84 (3 channels)."""
    if isinstance(src1, _np.ndarray) and src1.ndim == 3:
        return _and3(src1, src2, dst, mask)
    a = _image(src1, "src1", 2)
    b = _image(src2, "src2", 2)
    if a.shape != b.shape:
        raise error("bitwise_and: sizes of the inputs differ: %s vs %s" % (a.shape, b.shape))
    if mask is not None:
        mask = _image(mask, "mask", 2)
        if mask.shape != a.shape:
            raise error("bitwise_and: mask size %s differs from the image size %s" % (mask.shape, a.shape))
    sa, ra = _rows(a)
    sb, rb = _rows(b)
    sm, rm = _rows(mask) if mask is not None else (None, 0)
    h, w = a.shape
    out = _dst(dst, (h, w))
    _abi.check(_abi.lib().lc_and(_abi.context(), _abi.ptr(sa), ra, _abi.ptr(sb), rb,
                                 _abi.ptr(sm) if sm is not None else None, rm, _abi.ptr(out), w, w, h))
    return out


def HoughLinesP(image, rho, theta, threshold, lines=None, minLineLength=0, maxLineGap=0):
    """cv2.HoughLinesP: (N, 1, 4) int32 rows of (x1, y1, x2, y2) in OpenCV's detection order, or None.
    Reference: Simple Lane Detection with OpenCV_2(images).py:84-92, ..._5(mp4).py:76-84."""
    image = _image(image, "image", 2)
    s, stride = _rows(image)
    h, w = s.shape
    cap = 1024
    while True:
        out = _np.empty((cap, 4), _np.int32)
        n = _C.c_int()
        rc = _abi.lib().lc_hough_lines_p(_abi.context(), _abi.ptr(s), w, h, stride, float(rho), float(theta),
                                         int(threshold), float(minLineLength), float(maxLineGap),
                                         _abi.ptr(out), cap, _C.byref(n))
        if rc == _abi.LC_ECAP and n.value > cap:
            cap = n.value
            continue
        _abi.check(rc)
        break
    if n.value == 0:
        return None
    return out[:n.value].reshape(-1, 1, 4).copy()


def addWeighted(src1, alpha, src2, beta, gamma, dst=None, dtype=-1):
    """cv2.addWeighted on two uint8 images of one shape (the blend of the overlay the scripts draw).
    Reference: Simple Lane Detection with OpenCV_2(images).py:50, ..._5(mp4).py:50, F1_vo_Lane_Tracking.py:196."""
    if not isinstance(src1, _np.ndarray) or not isinstance(src2, _np.ndarray) or src1.ndim not in (2, 3):
        raise error("addWeighted: the sources must be 2- or 3-dimensional numpy arrays")
    a = _image(src1, "src1", src1.ndim)
    b = _image(src2, "src2", src1.ndim)
    if a.shape != b.shape:
        raise error("addWeighted: sizes of the inputs differ: %s vs %s" % (a.shape, b.shape))
    if dtype not in (-1, 0):
        raise error("addWeighted: only an 8-bit destination is supported", _abi.LC_EUNSUPPORTED)
    a, b = _np.ascontiguousarray(a), _np.ascontiguousarray(b)
    h = a.shape[0]
    row = a.size // h
    out = _dst(dst, a.shape)
    _abi.check(_abi.lib().lc_add_weighted(_abi.context(), _abi.ptr(a), row, float(alpha), _abi.ptr(b), row, float(beta),
                                          float(gamma), _abi.ptr(out), row, row, h))
    return out


def hough_accumulator(image, rho, theta):
    """The full-vote accumulator [numangle][numrho] (int32) HoughLinesP's voting rule gives for `image`."""
    image = _image(image, "image", 2)
    s, stride = _rows(image)
    h, w = s.shape
    na, nr = _C.c_int(), _C.c_int()
    _abi.check(_abi.lib().lc_hough_geometry(w, h, float(rho), float(theta), _C.byref(na), _C.byref(nr)))
    acc = _np.empty((na.value, nr.value), _np.int32)
    _abi.check(_abi.lib().lc_hough_accum(_abi.context(), _abi.ptr(s), w, h, stride, float(rho), float(theta),
                                         _abi.ptr(acc), acc.size, None, None))
    return acc
