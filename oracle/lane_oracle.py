"""ctypes front end of the CPU oracle (oracle/lane_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py, never by the shipped package.  PARITY UNPINNED
(see the header of lane_oracle.c): no cv2 and no reference fixtures exist for
this path, so the oracle is pinned by hand-derived known-answer tests only.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liblane_oracle.so")

COLOR_BGR2GRAY = 6
COLOR_RGB2GRAY = 7
COLOR_BGR2HSV = 40
COLOR_RGB2HSV = 41


class OracleError(Exception):
    pass


def build(force=False):
    src = os.path.join(_HERE, "lane_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "liblane_oracle.so"],
                              stdout=subprocess.DEVNULL)
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = C.CDLL(_SO)
        _lib.lo_rng_stream.restype = None
    return _lib


def _chk(rc, what):
    if rc != 0:
        raise OracleError("%s failed with code %d" % (what, rc))


def _u8(a):
    a = np.ascontiguousarray(a, dtype=np.uint8)
    return a, a.ctypes.data_as(C.c_void_p)


def cvt_gray(bgr, code=COLOR_BGR2GRAY):
    bgr, p = _u8(bgr)
    h, w, _ = bgr.shape
    out = np.empty((h, w), np.uint8)
    _chk(lib().lo_cvt_gray(p, w, h, C.c_long(3 * w), code, out.ctypes.data_as(C.c_void_p), C.c_long(w)), "cvt_gray")
    return out


def gaussian5(img):
    img, p = _u8(img)
    h, w = img.shape
    out = np.empty((h, w), np.uint8)
    _chk(lib().lo_gaussian5(p, w, h, C.c_long(w), out.ctypes.data_as(C.c_void_p), C.c_long(w)), "gaussian5")
    return out


def sobel3(img):
    img, p = _u8(img)
    h, w = img.shape
    dx = np.empty((h, w), np.int16)
    dy = np.empty((h, w), np.int16)
    _chk(lib().lo_sobel3(p, w, h, C.c_long(w), dx.ctypes.data_as(C.c_void_p), dy.ctypes.data_as(C.c_void_p)), "sobel3")
    return dx, dy


def canny(img, t1, t2, return_cls=False):
    img, p = _u8(img)
    h, w = img.shape
    out = np.empty((h, w), np.uint8)
    cls = np.empty((h, w), np.uint8) if return_cls else None
    _chk(lib().lo_canny(p, w, h, C.c_long(w), C.c_double(t1), C.c_double(t2),
                        out.ctypes.data_as(C.c_void_p), C.c_long(w),
                        cls.ctypes.data_as(C.c_void_p) if return_cls else None), "canny")
    return (out, cls) if return_cls else out


def bitwise_and(a, b, mask=None):
    a, pa = _u8(a)
    b, pb = _u8(b)
    h, w = a.shape
    out = np.empty((h, w), np.uint8)
    if mask is not None:
        mask, pm = _u8(mask)
    else:
        pm = None
    _chk(lib().lo_and(pa, C.c_long(w), pb, C.c_long(w), pm, C.c_long(w),
                      out.ctypes.data_as(C.c_void_p), C.c_long(w), w, h), "and")
    return out


def _i3(v):
    a = np.ascontiguousarray(np.rint(np.asarray(v, dtype=np.float64)).astype(np.int32).reshape(-1))
    return a, a.ctypes.data_as(C.c_void_p)


def cvt_hsv(bgr, code=COLOR_BGR2HSV):
    bgr, p = _u8(bgr)
    h, w, _ = bgr.shape
    out = np.empty((h, w, 3), np.uint8)
    _chk(lib().lo_cvt_hsv(p, w, h, C.c_long(3 * w), code, out.ctypes.data_as(C.c_void_p), C.c_long(3 * w)), "cvt_hsv")
    return out


def in_range(img3, lower, upper):
    img3, p = _u8(img3)
    h, w, _ = img3.shape
    lo, plo = _i3(lower)
    hi, phi = _i3(upper)
    out = np.empty((h, w), np.uint8)
    _chk(lib().lo_in_range3(p, w, h, C.c_long(3 * w), plo, phi, out.ctypes.data_as(C.c_void_p), C.c_long(w)), "in_range")
    return out


def bitwise_or(a, b):
    a, pa = _u8(a)
    b, pb = _u8(b)
    h, w = a.shape
    out = np.empty((h, w), np.uint8)
    _chk(lib().lo_or(pa, C.c_long(w), pb, C.c_long(w), out.ctypes.data_as(C.c_void_p), C.c_long(w), w, h), "or")
    return out


def and3_mask(img3, mask):
    """bitwise_and(img3, img3, mask=mask) into a fresh image."""
    img3, p = _u8(img3)
    mask, pm = _u8(mask)
    h, w, _ = img3.shape
    out = np.empty((h, w, 3), np.uint8)
    _chk(lib().lo_and3_mask(p, C.c_long(3 * w), pm, C.c_long(w), out.ctypes.data_as(C.c_void_p), C.c_long(3 * w), w, h), "and3_mask")
    return out


def gate_gray(bgr, gate, gray_code=COLOR_BGR2GRAY):
    """gate: ((lo0, hi0), (lo1, hi1)), each bound three H, S, V integers."""
    bgr, p = _u8(bgr)
    h, w, _ = bgr.shape
    g, pg = _i3(gate)
    out = np.empty((h, w), np.uint8)
    _chk(lib().lo_gate_gray(p, w, h, pg, gray_code, out.ctypes.data_as(C.c_void_p)), "gate_gray")
    return out


def add_weighted(a, alpha, b, beta, gamma):
    """cv2.addWeighted(a, alpha, b, beta, gamma) on uint8 arrays of one shape."""
    a, pa = _u8(a)
    b, pb = _u8(b)
    assert a.shape == b.shape
    out = np.empty_like(a)
    _chk(lib().lo_add_weighted(pa, C.c_double(alpha), pb, C.c_double(beta), C.c_double(gamma), out.ctypes.data_as(C.c_void_p),
                               C.c_size_t(a.size)), "add_weighted")
    return out


def resize_linear(img, dsize):
    """cv2.resize(img, dsize) with INTER_LINEAR on an 8-bit image of 1 or 3 channels; dsize = (width, height)."""
    img, p = _u8(img)
    sh, sw = img.shape[:2]
    cn = 1 if img.ndim == 2 else img.shape[2]
    dw, dh = int(dsize[0]), int(dsize[1])
    out = np.empty((dh, dw) if img.ndim == 2 else (dh, dw, cn), np.uint8)
    _chk(lib().lo_resize_linear(p, sw, sh, C.c_long(sw * cn), cn, out.ctypes.data_as(C.c_void_p), dw, dh, C.c_long(dw * cn)),
         "resize_linear")
    return out


def fill_poly(img, contours, color):
    """In place, like cv2.fillPoly.  contours: list of (N,2) integer arrays."""
    assert img.dtype == np.uint8 and img.ndim == 2 and img.flags.c_contiguous
    h, w = img.shape
    cs = [np.ascontiguousarray(np.asarray(c).reshape(-1, 2), dtype=np.int32) for c in contours]
    npts = (C.c_int * len(cs))(*[len(c) for c in cs])
    flat = np.ascontiguousarray(np.concatenate(cs, axis=0) if cs else np.zeros((0, 2), np.int32))
    _chk(lib().lo_fill_poly(img.ctypes.data_as(C.c_void_p), w, h, C.c_long(w),
                            flat.ctypes.data_as(C.c_void_p), npts, len(cs), int(color)), "fill_poly")
    return img


def rectangle_filled(img, pt1, pt2, color):
    assert img.dtype == np.uint8 and img.ndim == 2 and img.flags.c_contiguous
    h, w = img.shape
    _chk(lib().lo_rectangle_filled(img.ctypes.data_as(C.c_void_p), w, h, C.c_long(w),
                                   int(pt1[0]), int(pt1[1]), int(pt2[0]), int(pt2[1]), int(color)), "rectangle")
    return img


def _draw_target(img):
    assert img.dtype == np.uint8 and img.ndim in (2, 3) and img.flags.c_contiguous
    h, w = img.shape[:2]
    cn = 1 if img.ndim == 2 else img.shape[2]
    return h, w, cn


def _color(color, cn):
    c = list(color) if isinstance(color, (tuple, list, np.ndarray)) else [color]
    c = (c + [0, 0, 0, 0])[:max(cn, 1)]
    return (C.c_int * 4)(*([int(v) for v in c] + [0] * (4 - len(c))))


def line(img, pt1, pt2, color, thickness=1):
    """In place, like cv2.line (LINE_8, shift 0) on an 8-bit image of 1..4 channels."""
    h, w, cn = _draw_target(img)
    _chk(lib().lo_line(img.ctypes.data_as(C.c_void_p), w, h, C.c_long(w * cn), cn, int(pt1[0]), int(pt1[1]), int(pt2[0]),
                       int(pt2[1]), _color(color, cn), int(thickness)), "line")
    return img


def polylines(img, pts, is_closed, color, thickness=1):
    """In place, like cv2.polylines.  pts: list of (N,2) / (N,1,2) integer arrays."""
    h, w, cn = _draw_target(img)
    cs = [np.ascontiguousarray(np.asarray(c).reshape(-1, 2), dtype=np.int32) for c in pts]
    npts = (C.c_int * max(1, len(cs)))(*[len(c) for c in cs])
    flat = np.ascontiguousarray(np.concatenate(cs, axis=0) if cs else np.zeros((0, 2), np.int32))
    _chk(lib().lo_polylines(img.ctypes.data_as(C.c_void_p), w, h, C.c_long(w * cn), cn, flat.ctypes.data_as(C.c_void_p), npts,
                            len(cs), int(bool(is_closed)), _color(color, cn), int(thickness)), "polylines")
    return img


def hough_geometry(w, h, rho, theta):
    na, nr = C.c_int(), C.c_int()
    _chk(lib().lo_hough_geometry(w, h, C.c_double(rho), C.c_double(theta), C.byref(na), C.byref(nr), None, 0), "geometry")
    tab = np.empty(2 * na.value, np.float32)
    _chk(lib().lo_hough_geometry(w, h, C.c_double(rho), C.c_double(theta), None, None,
                                 tab.ctypes.data_as(C.c_void_p), tab.size), "geometry")
    return na.value, nr.value, tab


def rng_stream(n):
    out = np.empty(n, np.uint32)
    lib().lo_rng_stream(out.ctypes.data_as(C.c_void_p), n)
    return out


def hough_accum(img, rho, theta):
    img, p = _u8(img)
    h, w = img.shape
    na, nr, _ = hough_geometry(w, h, rho, theta)
    acc = np.empty((na, nr), np.int32)
    _chk(lib().lo_hough_accum(p, w, h, C.c_long(w), C.c_double(rho), C.c_double(theta),
                              acc.ctypes.data_as(C.c_void_p), acc.size, None, None), "hough_accum")
    return acc


def hough_lines_p(img, rho, theta, threshold, min_line_length=0, max_line_gap=0,
                  cap=4096, return_acc=False):
    """Returns an (N,1,4) int32 array or None, like cv2.HoughLinesP."""
    img, p = _u8(img)
    h, w = img.shape
    out = np.empty((cap, 4), np.int32)
    n = C.c_int()
    acc = None
    if return_acc:
        na, nr, _ = hough_geometry(w, h, rho, theta)
        acc = np.empty((na, nr), np.int32)
    _chk(lib().lo_hough_lines_p(p, w, h, C.c_long(w), C.c_double(rho), C.c_double(theta),
                                int(threshold), C.c_double(min_line_length), C.c_double(max_line_gap),
                                out.ctypes.data_as(C.c_void_p), cap, C.byref(n),
                                acc.ctypes.data_as(C.c_void_p) if return_acc else None,
                                acc.size if return_acc else 0), "hough_lines_p")
    lines = out[:n.value].reshape(-1, 1, 4).copy() if n.value else None
    return (lines, acc) if return_acc else lines


def lane_frame(bgr, code, blur, t1, t2, roi, rho, theta, threshold, min_len, max_gap,
               car_mask=None, seg_cap=4096, want_acc=True, want_segments=True, gate=None):
    """One frame through a recipe.  Returns a dict of every intermediate."""
    bgr, p = _u8(bgr)
    h, w, _ = bgr.shape
    na, nr, _ = hough_geometry(w, h, rho, theta)
    g = np.empty((h, w), np.uint8)
    b = np.empty((h, w), np.uint8)
    e = np.empty((h, w), np.uint8)
    m = np.empty((h, w), np.uint8)
    acc = np.empty((na, nr), np.int32) if want_acc else None
    seg = np.empty((seg_cap, 4), np.int32)
    n = C.c_int()
    roi_p = _u8(roi) if roi is not None else (None, None)
    car_p = _u8(car_mask) if car_mask is not None else (None, None)
    vp = lambda a: a.ctypes.data_as(C.c_void_p)
    gate_p = _i3(gate) if gate is not None else (None, None)
    _chk(lib().lo_lane_frame_gated(p, w, h, code, gate_p[1], int(bool(blur)), C.c_double(t1), C.c_double(t2),
                             car_p[1], roi_p[1], C.c_double(rho), C.c_double(theta), int(threshold),
                             C.c_double(min_len), C.c_double(max_gap),
                             vp(g), vp(b), vp(e), vp(m),
                             vp(acc) if want_acc else None, acc.size if want_acc else 0,
                             vp(seg), seg_cap, C.byref(n) if want_segments else None), "lane_frame")
    return {"gray": g, "blur": b, "edges": e, "masked": m, "acc": acc,
            "segments": seg[:n.value].copy() if want_segments else None}
