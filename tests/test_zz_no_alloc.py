"""lc_canny_noalloc (the entry point for callers that cannot allocate) against lc_canny, through the C ABI:
two calls in a row on the same GPU's fixed working array, the second on a smaller image."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _scene(w, h, seed):
    """Gray road-like image with strong, weak and isolated-weak structure (so that hysteresis has work)."""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    img = 90 + 30 * np.sin(xx / 37.0) * np.cos(yy / 23.0)
    img += 80 * (np.abs(xx - (w / 2 + (yy - h / 2) * 0.8)) < 4)
    img += 35 * (np.abs(xx - (w / 2 - (yy - h / 2) * 0.6)) < 3)
    img += rng.normal(0, 5, (h, w))
    return np.clip(img, 0, 255).astype(np.uint8)


def _noalloc(ctx, img, lo, hi, d_in, d_out):
    from lanecv import _abi
    import hiprt
    h, w = img.shape
    d_in.upload(img)
    d_out.upload(np.full((h, w), 7, np.uint8))          # stale contents must be overwritten
    _abi.check(_abi.lib().lc_canny_noalloc(ctx, d_in.ptr, w, h, float(lo), float(hi), d_out.ptr, None))
    hiprt.synchronize()
    return d_out.download((h, w))


def test_noalloc_canny_matches_canny_twice(oracle):
    import lanecv as cv2
    from lanecv import _abi
    import hiprt
    ctx = _abi.context()
    big, small = _scene(1280, 720, 1), _scene(331, 203, 2)      # the second: smaller, and no multiple of anything
    d_in, d_out = hiprt.DeviceBuffer(big.size), hiprt.DeviceBuffer(big.size)
    try:
        for img, (lo, hi) in ((big, (100, 200)), (small, (50, 150)), (big, (50, 150))):
            want = cv2.Canny(img, lo, hi)
            got = _noalloc(ctx, img, lo, hi, d_in, d_out)
            assert want.any() and (want == 0).any()
            assert np.array_equal(got, want)
            assert np.array_equal(got, oracle.canny(img, lo, hi))
    finally:
        d_in.free()
        d_out.free()


def test_noalloc_canny_allocates_nothing_and_rejects_oversize():
    from lanecv import _abi
    import hiprt
    ctx = _abi.context()
    img = _scene(640, 360, 3)
    d_in, d_out = hiprt.DeviceBuffer(img.size), hiprt.DeviceBuffer(img.size)
    try:
        _noalloc(ctx, img, 100, 200, d_in, d_out)               # code objects loaded, kernels seen once
        before = hiprt.free_bytes()
        for _ in range(20):
            _noalloc(ctx, img, 100, 200, d_in, d_out)
        assert hiprt.free_bytes() == before
        L = _abi.lib()
        assert L.lc_canny_noalloc(ctx, d_in.ptr, 4097, 8, 100.0, 200.0, d_out.ptr, None) == _abi.LC_EUNSUPPORTED
        assert b"4096" in L.lc_last_error()
        assert L.lc_canny_noalloc(ctx, d_in.ptr, 8, 2305, 100.0, 200.0, d_out.ptr, None) == _abi.LC_EUNSUPPORTED
        assert L.lc_canny_noalloc(ctx, None, 8, 8, 100.0, 200.0, d_out.ptr, None) == _abi.LC_EINVAL
        assert L.lc_canny_noalloc(ctx, d_in.ptr, 8, 8, 100.0, 200.0, C.c_void_p(d_out.ptr.value + 4), None) == _abi.LC_EINVAL
    finally:
        d_in.free()
        d_out.free()
