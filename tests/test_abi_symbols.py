"""CPU-side checks of the drop-in boundary: the C-ABI library loads, exports every
symbol include/lanecv.h declares, and fails loudly (no CPU fallback) without a GPU."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_header_symbols_exported():
    from lanecv import _abi
    hdr = open(os.path.join(ROOT, "include", "lanecv.h")).read()
    declared = set(re.findall(r"\b(lc_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(_abi.SYMBOLS)
    L = _abi.lib()
    for s in declared:
        assert hasattr(L, s), s
    assert L.lc_version() >= 100


def test_geometry_needs_no_gpu():
    import ctypes as C
    import numpy as np
    from lanecv import _abi
    na, nr = C.c_int(), C.c_int()
    for (w, h, want) in ((960, 540, 500), (1280, 720, 667), (1920, 1080, 1000), (3840, 2160, 2000)):
        _abi.check(_abi.lib().lc_hough_geometry(w, h, 6.0, np.pi / 60, C.byref(na), C.byref(nr)))
        assert (na.value, nr.value) == (60, want)
    assert _abi.lib().lc_hough_geometry(10, 10, 0.0, 1.0, None, None) == _abi.LC_EINVAL
    assert b"rho" in _abi.lib().lc_last_error()


def test_no_cpu_fallback_without_gpu():
    import ctypes as C
    import numpy as np
    import lanecv
    from lanecv import _abi
    h = C.c_void_p()
    rc = _abi.lib().lc_create(0, C.byref(h))
    if rc == 0:
        _abi.lib().lc_destroy(h)
        pytest.skip("a GPU is present")
    assert rc == _abi.LC_EHIP
    with pytest.raises(lanecv.error):
        lanecv.Canny(np.zeros((8, 8), np.uint8), 100, 200)


def test_python_argument_checks_need_no_gpu():
    import numpy as np
    import lanecv as cv2
    g = np.zeros((10, 10), np.uint8)
    with pytest.raises(cv2.error):
        cv2.GaussianBlur(g, (3, 3), 0)
    with pytest.raises(cv2.error):
        cv2.cvtColor(np.zeros((4, 4, 3), np.uint8), 44)
    with pytest.raises(cv2.error):
        cv2.Canny(g, 1, 2, L2gradient=True)
    assert issubclass(cv2.error, Exception)
    assert (cv2.COLOR_BGR2GRAY, cv2.COLOR_RGB2GRAY, cv2.BORDER_DEFAULT, cv2.LINE_8) == (6, 7, 4, 8)
    assert (cv2.COLOR_BGR2HSV, cv2.COLOR_RGB2HSV) == (40, 41)
