"""ctypes binding of liblanecv.so (include/lanecv.h).  There is no CPU path: if the
library or a HIP device is missing, calls raise `error` instead of falling back."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.environ.get("LANECV_SO") or os.path.join(_HERE, "liblanecv.so")   # LANECV_SO: another build of the same ABI (A/B runs)


class error(Exception):
    """Raised where cv2 would raise cv2.error (and for HIP failures).  The reference's
    `except Exception` handlers (Simple Lane Detection with OpenCV_3(mp4).py:182-187) catch it."""

    def __init__(self, msg, code=-1):
        super().__init__(msg)
        self.code = code


LC_OK, LC_EINVAL, LC_ENOMEM, LC_ECAP, LC_EHIP, LC_EUNSUPPORTED = 0, -1, -2, -3, -4, -5

BUF_GRAY, BUF_BLUR, BUF_EDGES, BUF_MASKED, BUF_ACC, BUF_NPOINTS, BUF_RESIZED = 1, 2, 3, 4, 5, 6, 7
STAGE_UPLOAD, STAGE_EDGE, STAGE_POINTS, STAGE_VOTE, STAGE_LINES, STAGE_TOTAL, STAGE_FRONT, STAGE_HYST, STAGE_FINAL = range(9)

# every symbol include/lanecv.h declares
SYMBOLS = [
    "lc_version", "lc_last_error", "lc_create", "lc_destroy", "lc_host_alloc_pinned", "lc_host_free",
    "lc_cvt_gray", "lc_cvt_hsv", "lc_in_range3", "lc_or", "lc_and3_mask", "lc_resize_linear", "lc_add_weighted", "lc_gaussian5", "lc_canny", "lc_canny_noalloc", "lc_fill_poly", "lc_rectangle_filled", "lc_line", "lc_polylines", "lc_and",
    "lc_hough_geometry", "lc_hough_accum", "lc_hough_lines_p", "lc_lds_atomic_rate",
    "lc_pipe_create", "lc_pipe_destroy", "lc_pipe_submit_host", "lc_pipe_submit_device", "lc_pipe_wait",
    "lc_pipe_read", "lc_pipe_read_points", "lc_pipe_stage_ms", "lc_pipe_input_device_ptr", "lc_pipe_upload", "lc_pipe_hyst_rounds", "lc_pipe_shared_queues", "lc_pipe_roi_skip", "lc_pipe_layout", "lc_pipe_set_timing", "lc_pipe_rows_uploaded", "lc_pipe_bytes_uploaded", "lc_pipe_event_times",
]


class PipeParams(C.Structure):
    _fields_ = [
        ("width", C.c_int), ("height", C.c_int), ("max_batch", C.c_int), ("nslots", C.c_int),
        ("color_code", C.c_int), ("blur", C.c_int),
        ("canny_lo", C.c_double), ("canny_hi", C.c_double),
        ("roi_mask", C.c_void_p), ("car_mask", C.c_void_p),
        ("rho", C.c_double), ("theta", C.c_double), ("threshold", C.c_int),
        ("min_line_length", C.c_double), ("max_line_gap", C.c_double),
        ("seg_capacity", C.c_int), ("want_accumulator", C.c_int), ("want_segments", C.c_int),
        ("keep_intermediates", C.c_int),
        ("hsv_gate", C.c_int), ("hsv_bounds", C.c_int * 12), ("want_points", C.c_int), ("src_width", C.c_int), ("src_height", C.c_int),
    ]


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            raise error("liblanecv.so is not built (run `make -C image-processing_amd/csrc`); "
                        "lanecv has no CPU fallback", LC_EHIP)
        L = C.CDLL(_SO)
        L.lc_last_error.restype = C.c_char_p
        vp, sz, i, d = C.c_void_p, C.c_size_t, C.c_int, C.c_double
        L.lc_create.argtypes = [i, C.POINTER(vp)]
        L.lc_destroy.argtypes = [vp]
        L.lc_host_alloc_pinned.argtypes = [sz, C.POINTER(vp)]
        L.lc_host_free.argtypes = [vp]
        L.lc_cvt_gray.argtypes = [vp, vp, i, i, sz, i, vp, sz]
        L.lc_cvt_hsv.argtypes = [vp, vp, i, i, sz, i, vp, sz]
        L.lc_in_range3.argtypes = [vp, vp, i, i, sz, C.POINTER(i), C.POINTER(i), vp, sz]
        L.lc_or.argtypes = [vp, vp, sz, vp, sz, vp, sz, i, i]
        L.lc_and3_mask.argtypes = [vp, vp, sz, vp, sz, vp, sz, i, i]
        L.lc_resize_linear.argtypes = [vp, vp, i, i, sz, i, vp, i, i, sz]
        L.lc_add_weighted.argtypes = [vp, vp, sz, d, vp, sz, d, d, vp, sz, i, i]
        L.lc_gaussian5.argtypes = [vp, vp, i, i, sz, vp, sz]
        L.lc_canny.argtypes = [vp, vp, i, i, sz, d, d, vp, sz]
        L.lc_canny_noalloc.argtypes = [vp, vp, i, i, d, d, vp, vp]
        L.lc_fill_poly.argtypes = [vp, vp, i, i, sz, vp, vp, i, i]
        L.lc_rectangle_filled.argtypes = [vp, vp, i, i, sz, i, i, i, i, i]
        L.lc_line.argtypes = [vp, vp, i, i, sz, i, i, i, i, i, C.POINTER(i), i]
        L.lc_polylines.argtypes = [vp, vp, i, i, sz, i, vp, vp, i, i, C.POINTER(i), i]
        L.lc_and.argtypes = [vp, vp, sz, vp, sz, vp, sz, vp, sz, i, i]
        L.lc_hough_geometry.argtypes = [i, i, d, d, C.POINTER(i), C.POINTER(i)]
        L.lc_hough_accum.argtypes = [vp, vp, i, i, sz, d, d, vp, i, C.POINTER(i), C.POINTER(i)]
        L.lc_hough_lines_p.argtypes = [vp, vp, i, i, sz, d, d, i, d, d, vp, i, C.POINTER(i)]
        L.lc_lds_atomic_rate.argtypes = [vp, i, i, C.POINTER(d)]
        L.lc_pipe_create.argtypes = [vp, C.POINTER(PipeParams), C.POINTER(vp)]
        L.lc_pipe_destroy.argtypes = [vp]
        L.lc_pipe_submit_host.argtypes = [vp, i, vp, i]
        L.lc_pipe_submit_device.argtypes = [vp, i, vp, i]
        L.lc_pipe_wait.argtypes = [vp, i, vp, vp]
        L.lc_pipe_read.argtypes = [vp, i, i, vp, sz]
        L.lc_pipe_read_points.argtypes = [vp, i, i, vp, i, C.POINTER(i)]
        L.lc_pipe_stage_ms.argtypes = [vp, i, i, C.POINTER(C.c_float)]
        L.lc_pipe_input_device_ptr.argtypes = [vp, i, C.POINTER(vp)]
        L.lc_pipe_upload.argtypes = [vp, i, vp, i]
        L.lc_pipe_shared_queues.argtypes = [vp, C.POINTER(i)]
        L.lc_pipe_set_timing.argtypes = [vp, i]
        L.lc_pipe_rows_uploaded.argtypes = [vp, i, C.POINTER(i)]
        L.lc_pipe_bytes_uploaded.argtypes = [vp, i, C.POINTER(C.c_longlong)]
        L.lc_pipe_event_times.argtypes = [vp, i, C.POINTER(C.c_float)]
        L.lc_pipe_layout.argtypes = [vp, C.POINTER(i), C.POINTER(i), C.POINTER(i)]
        L.lc_pipe_roi_skip.argtypes = [vp, i, C.POINTER(i), C.POINTER(C.c_longlong), C.POINTER(i)]
        L.lc_pipe_hyst_rounds.argtypes = [vp, i, C.POINTER(i), C.POINTER(i)]
        _lib = L
    return _lib


def check(rc):
    if rc != 0:
        msg = lib().lc_last_error()
        raise error((msg or b"").decode("utf-8", "replace") or "lanecv call failed (%d)" % rc, rc)


_ctx = {}


def context(device=None):
    """The per-device context of the calling process (created on first use)."""
    if device is None:
        device = int(os.environ.get("LANECV_DEVICE", "0"))
    c = _ctx.get(device)
    if c is None:
        h = C.c_void_p()
        check(lib().lc_create(device, C.byref(h)))
        c = _ctx[device] = h
    return c


def lds_atomic_rate(run=1, returning=False, device=None):
    """1e9 lane-level int32 LDS atomic adds per second this GPU sustains for the given address pattern
    (run = lanes sharing a word, 0 = pseudo-random words); the ceiling the vote kernels are reported against."""
    v = C.c_double()
    check(lib().lc_lds_atomic_rate(context(device), int(run), int(bool(returning)), C.byref(v)))
    return v.value


def ptr(a):
    return a.ctypes.data_as(C.c_void_p)
