"""GPU parity of the cv2-shaped per-op calls (through the C ABI) against the CPU
oracle: bit-exact for every image, accumulator and segment list."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SIZES = [(960, 540), (641, 359), (64, 48), (33, 7), (5, 5), (3, 3), (1, 9), (9, 1), (1, 1), (130, 70)]


@pytest.fixture(scope="module")
def cv2():
    import lanecv
    return lanecv


def _frames(w, h):
    from lanecv import synth
    fr = dict(synth.adversarial_frames(w, h))
    fr["road"] = synth.synth_road(w, h, seed=5, t=3)
    return fr


@pytest.mark.parametrize("w,h", SIZES)
def test_gray_blur_canny(cv2, oracle, w, h):
    for name, f in _frames(w, h).items():
        for code in (cv2.COLOR_BGR2GRAY, cv2.COLOR_RGB2GRAY):
            g = cv2.cvtColor(f, code)
            assert g.dtype == np.uint8 and g.shape == (h, w)
            assert (g == oracle.cvt_gray(f, code)).all(), (name, code)
        b = cv2.GaussianBlur(g, (5, 5), 0)
        assert (b == oracle.gaussian5(g)).all(), name
        for src in (g, b):
            for lo, hi in ((100, 200), (50, 150), (0, 0), (20, 600), (200, 100)):
                e = cv2.Canny(src, lo, hi)
                assert (e == oracle.canny(src, lo, hi)).all(), (name, lo, hi)


def test_canny_1080p_and_4k(cv2, oracle):
    from lanecv import synth
    for (w, h, seed) in ((1920, 1080, 3), (3840, 2160, 4)):
        g = oracle.gaussian5(oracle.cvt_gray(synth.synth_road(w, h, seed, 0), 6))
        assert (cv2.Canny(g, 50, 150) == oracle.canny(g, 50, 150)).all()
        assert (cv2.Canny(g, 100, 200) == oracle.canny(g, 100, 200)).all()


def test_canny_long_weak_chains(cv2, oracle):
    # a serpentine weak ridge that crosses many 64-pixel tiles with one strong seed at its end:
    # hysteresis must carry the seed through every tile border it crosses
    h, w = 300, 700
    im = np.zeros((h, w), np.float32)
    for k, y in enumerate(range(20, 280, 24)):
        im[y:y + 12, 10:w - 10] = 30
        x = w - 22 if k % 2 == 0 else 10
        im[y + 12:y + 24, x:x + 12] = 30
    im[20:32, 10:40] = 255
    img = im.astype(np.uint8)
    for lo, hi in ((20, 600), (50, 500), (100, 200)):
        assert (cv2.Canny(img, lo, hi) == oracle.canny(img, lo, hi)).all()
    rng = np.random.default_rng(1)
    noise = rng.integers(0, 256, (500, 900), dtype=np.uint8)
    sm = oracle.gaussian5(oracle.gaussian5(noise))
    for lo, hi in ((5, 60), (10, 40), (30, 90)):
        assert (cv2.Canny(sm, lo, hi) == oracle.canny(sm, lo, hi)).all()


def test_strided_and_dst(cv2, oracle):
    from lanecv import synth
    f = synth.synth_road(200, 100, 1, 0)
    big = np.zeros((100, 300, 3), np.uint8)
    big[:, :200] = f
    view = big[:, :200]                      # rows strided, pixels dense
    assert (cv2.cvtColor(view, cv2.COLOR_BGR2GRAY) == oracle.cvt_gray(f, 6)).all()
    g = oracle.cvt_gray(f, 6)
    assert (cv2.Canny(g[:, ::2], 100, 200) == oracle.canny(np.ascontiguousarray(g[:, ::2]), 100, 200)).all()
    dst = np.empty((100, 200), np.uint8)
    out = cv2.GaussianBlur(g, (5, 5), 0, dst)
    assert out is dst and (dst == oracle.gaussian5(g)).all()


POLYS = [
    [(0, 540), (480, 270), (960, 540)],
    [(96, 540), (480, 270), (864, 540)],
    [(0, 540), (960, 540), (960, 216), (0, 216)],
    [(115, 540), (115, 216), (307, 216), (336, 270), (336, 259), (624, 259),
     (624, 270), (652, 216), (844, 216), (844, 540)],
    [(-50, 100), (300, -40), (1200, 300), (500, 700)],
    [(10, 10), (200, 300), (10, 300), (200, 10)],
    [(5, 5)], [(5, 5), (100, 50)],
]


@pytest.mark.parametrize("poly", POLYS)
def test_fill_poly(cv2, oracle, poly):
    a = np.zeros((540, 960), np.uint8)
    b = np.zeros((540, 960), np.uint8)
    r = cv2.fillPoly(a, np.array([poly], np.int32), 255)
    assert r is a
    oracle.fill_poly(b, [np.array(poly)], 255)
    assert (a == b).all()
    # int64 vertices (numpy default on Linux) are accepted
    c = np.zeros((540, 960), np.uint8)
    cv2.fillPoly(c, np.array([poly]), 255)
    assert (c == b).all()


def test_fill_poly_random(cv2, oracle):
    rng = np.random.default_rng(3)
    for _ in range(40):
        w, h = int(rng.integers(20, 200)), int(rng.integers(20, 150))
        n = int(rng.integers(3, 8))
        pts = np.stack([rng.integers(-30, w + 30, n), rng.integers(-30, h + 30, n)], 1).astype(np.int32)
        a = np.zeros((h, w), np.uint8)
        b = np.zeros((h, w), np.uint8)
        cv2.fillPoly(a, [pts], 200)
        oracle.fill_poly(b, [pts], 200)
        assert (a == b).all(), pts.tolist()


def test_f1_masks(cv2, oracle):
    from lanecv.pipeline import f1_car_mask
    w, h = 640, 360
    m = f1_car_mask(w, h)
    ref = np.full((h, w), 255, np.uint8)
    pts = np.array([[int(w * 0.12), h], [int(w * 0.12), int(h * 0.40)], [int(w * 0.32), int(h * 0.40)],
                    [int(w * 0.35), int(h * 0.50)], [int(w * 0.35), int(h * 0.48)], [int(w * 0.65), int(h * 0.48)],
                    [int(w * 0.65), int(h * 0.50)], [int(w * 0.68), int(h * 0.40)], [int(w * 0.88), int(h * 0.40)],
                    [int(w * 0.88), h]], np.int32)
    oracle.fill_poly(ref, [pts], 0)
    ui_h = int(h * 0.35)
    oracle.rectangle_filled(ref, (0, h - ui_h), (int(w * 0.12), h), 0)
    oracle.rectangle_filled(ref, (int(w * 0.88), h - ui_h), (w, h), 0)
    assert (m == ref).all()


def test_bitwise_and(cv2, oracle):
    rng = np.random.default_rng(0)
    for (w, h) in ((960, 540), (37, 11), (1, 1)):
        a = rng.integers(0, 256, (h, w), dtype=np.uint8)
        b = rng.integers(0, 256, (h, w), dtype=np.uint8)
        m = (rng.random((h, w)) < 0.5).astype(np.uint8) * rng.integers(1, 256, (h, w), dtype=np.uint8)
        assert (cv2.bitwise_and(a, b) == (a & b)).all()
        assert (cv2.bitwise_and(a, a, mask=m) == oracle.bitwise_and(a, a, m)).all()
        assert (cv2.bitwise_and(a, b, mask=m) == np.where(m != 0, a & b, 0)).all()


def _masked(oracle, w, h, seed, t=0, code=7, blur=False, lo=100, hi=200):
    from lanecv import synth
    f = synth.synth_road(w, h, seed, t)
    g = oracle.cvt_gray(f, code)
    if blur:
        g = oracle.gaussian5(g)
    e = oracle.canny(g, lo, hi)
    roi = np.zeros((h, w), np.uint8)
    oracle.fill_poly(roi, [np.array([(0, h), (w // 2, h // 2), (w, h)])], 255)
    return oracle.bitwise_and(e, roi)


@pytest.mark.parametrize("w,h", [(960, 540), (1280, 720), (1920, 1080), (641, 359)])
def test_hough_accumulator(cv2, oracle, w, h):
    m = _masked(oracle, w, h, 2)
    acc = cv2.hough_accumulator(m, 6, np.pi / 60)
    ref = oracle.hough_accum(m, 6, np.pi / 60)
    assert acc.shape == ref.shape and (acc == ref).all()
    assert (acc.sum(axis=1) == np.count_nonzero(m)).all()
    e = (np.random.default_rng(0).random((90, 120)) < 0.2).astype(np.uint8) * 255
    for rho, theta in ((1, np.pi / 180), (2.5, 0.1)):
        assert (cv2.hough_accumulator(e, rho, theta) == oracle.hough_accum(e, rho, theta)).all()


@pytest.mark.parametrize("w,h,thr,gap", [(960, 540, 160, 25), (1280, 720, 130, 150), (1920, 1080, 160, 25),
                                         (641, 359, 60, 25), (320, 180, 30, 5)])
def test_hough_lines_p(cv2, oracle, w, h, thr, gap):
    for seed in (1, 2):
        m = _masked(oracle, w, h, seed)
        got = cv2.HoughLinesP(m, 6, np.pi / 60, thr, lines=np.array([]), minLineLength=40, maxLineGap=gap)
        ref = oracle.hough_lines_p(m, 6, np.pi / 60, thr, 40, gap)
        assert (got is None) == (ref is None)
        if ref is not None:
            assert got.shape == ref.shape and got.dtype == np.int32 and (got == ref).all()


def test_hough_lines_p_dense_and_empty(cv2, oracle):
    assert cv2.HoughLinesP(np.zeros((50, 60), np.uint8), 6, np.pi / 60, 10, minLineLength=5, maxLineGap=5) is None
    rng = np.random.default_rng(4)
    for k in range(3):
        e = (rng.random((60, 80)) < 0.08).astype(np.uint8) * 255
        e[10 + k, 5:70] = 255
        e[5:55, 20 + 3 * k] = 255
        got = cv2.HoughLinesP(e, 1, np.pi / 180, 25, minLineLength=15, maxLineGap=2)
        ref = oracle.hough_lines_p(e, 1, np.pi / 180, 25, 15, 2)
        assert (got == ref).all()
    full = np.full((40, 50), 255, np.uint8)
    got = cv2.HoughLinesP(full, 1, np.pi / 180, 20, minLineLength=10, maxLineGap=1)
    ref = oracle.hough_lines_p(full, 1, np.pi / 180, 20, 10, 1)
    assert got.shape == ref.shape and (got == ref).all()


def _same_lines(cv2, oracle, e, rho, theta, thr, minlen, gap):
    got = cv2.HoughLinesP(e, rho, theta, thr, minLineLength=minlen, maxLineGap=gap)
    ref = oracle.hough_lines_p(e, rho, theta, thr, minlen, gap)
    assert (got is None) == (ref is None), (e.shape, thr, minlen, gap)
    if ref is not None:
        assert got.shape == ref.shape and (got == ref).all(), (e.shape, thr, minlen, gap)
    return 0 if ref is None else len(ref)


def test_hough_lines_p_lds_path_edges(cv2, oracle):
    """Shapes that stress the all-in-LDS HoughLinesP kernel (at most 64 angles): degenerate point sets, every point
    starting a line, zero gap, lines longer than one fetch stretch, wide row spans, and frames too big for LDS."""
    th = np.pi / 60
    one = np.zeros((40, 70), np.uint8); one[17, 33] = 255
    _same_lines(cv2, oracle, one, 6, th, 1, 0, 0)
    row = np.zeros((40, 300), np.uint8); row[21, 3:290:2] = 255                 # all points in one row, gaps of 1
    assert _same_lines(cv2, oracle, row, 6, th, 20, 10, 1) > 0
    _same_lines(cv2, oracle, row, 6, th, 20, 10, 0)                             # gap 0: every hole ends the line
    far = np.zeros((400, 90), np.uint8); far[2, 5:80] = 255; far[397, 10:85] = 255   # two rows far apart
    assert _same_lines(cv2, oracle, far, 3, th, 30, 20, 3) == 2
    rng = np.random.default_rng(11)
    for k in range(3):                                                          # threshold 1: every alive point walks a line
        e = (rng.random((48, 64)) < 0.15).astype(np.uint8) * 255
        _same_lines(cv2, oracle, e, 2, th, 1, 3 * k, k)
    big = np.zeros((720, 1280), np.uint8)                                       # lines of 600+ steps, with holes
    for x in range(40, 1240):
        if x % 37 > 2:
            big[int(30 + 0.5 * x), x] = 255
            big[int(700 - 0.52 * x), x] = 255
    big[100:700, 640] = 255
    assert _same_lines(cv2, oracle, big, 6, th, 100, 40, 25) >= 3
    wide = np.zeros((64, 4096), np.uint8); wide[30, 10:4090] = 255; wide[10:50, 2048] = 255
    wide[(np.arange(4000) // 100 + 5) % 64, np.arange(4000) + 40] = 255
    _same_lines(cv2, oracle, wide, 6, th, 80, 40, 10)
    # an X of two long diagonals with lone points at the corners and mid-edges: every band of rows has another column
    # range, and the lone points vote in the first and last bins of the angles' accumulator rows
    xs = np.zeros((600, 800), np.uint8)
    t = np.arange(40, 560)
    xs[t, t + 100] = 255
    xs[t, 700 - t] = 255
    for y, x in ((0, 0), (0, 799), (599, 0), (599, 799), (300, 0), (300, 799), (0, 400), (599, 400), (17, 790), (580, 3)):
        xs[y, x] = 255
    for rho, thr in ((6, 100), (1, 60), (3.5, 80)):
        assert _same_lines(cv2, oracle, xs, rho, th, thr, 40, 10) >= 2
    tall = np.zeros((3000, 96), np.uint8)                                        # 2990 rows of points: bands of 64 rows
    tall[5:2995, 40] = 255; tall[5:2995:7, 90] = 255; tall[1500, 0:96:3] = 255
    assert _same_lines(cv2, oracle, tall, 6, th, 100, 40, 10) >= 1
    dense = (rng.random((360, 640)) < 0.03).astype(np.uint8) * 255              # ~6900 points: fits
    _same_lines(cv2, oracle, dense, 6, th, 60, 30, 4)
    denser = (rng.random((360, 640)) < 0.09).astype(np.uint8) * 255             # ~20700 points: still fits
    _same_lines(cv2, oracle, denser, 6, th, 150, 30, 4)
    densest = (rng.random((360, 640)) < 0.16).astype(np.uint8) * 255            # ~36900 points: the point list stays in device memory
    _same_lines(cv2, oracle, densest, 6, th, 260, 30, 4)
    # points scattered over a 1080p frame, nearly each in a mask word of its own, and lines through them: points and
    # accumulator fit the LDS, the mask words do not (the kernel's form that leaves the mask in device memory); at the
    # last density (~42 000 points) the points do not fit either (the form that leaves them in device memory too)
    for dens, thr in ((0.003, 120), (0.005, 160), (0.02, 220)):
        scat = (rng.random((1080, 1920)) < dens).astype(np.uint8) * 255
        xx = np.arange(100, 1800, 2)
        scat[(1000 - 0.45 * xx).astype(int), xx] = 255
        scat[(200 + 0.3 * xx).astype(int), xx] = 255
        scat[300:900, 960] = 255
        assert _same_lines(cv2, oracle, scat, 6, th, thr, 40, 25) >= 3
    # a 4K map: a bin-wide stripe across it holds more pixels than a 16-bit bin counts, so there the limit stays 32 759
    # points -- below it the kernel's forms, above it the global-memory kernel
    for dens in (0.003, 0.0045):
        big4k = (rng.random((2160, 3840)) < dens).astype(np.uint8) * 255
        xx = np.arange(200, 3600, 2)
        big4k[(2000 - 0.45 * xx).astype(int), xx] = 255
        assert _same_lines(cv2, oracle, big4k, 6, th, 300, 40, 25) >= 1


def test_errors(cv2):
    g = np.zeros((10, 10), np.uint8)
    with pytest.raises(Exception):
        cv2.cvtColor(g, cv2.COLOR_BGR2GRAY)                 # not 3-channel
    with pytest.raises(cv2.error):
        cv2.cvtColor(np.zeros((4, 4, 3), np.uint8), 44)     # unsupported code (BGR2Lab)
    with pytest.raises(cv2.error):
        cv2.GaussianBlur(g, (3, 3), 0)
    with pytest.raises(cv2.error):
        cv2.Canny(g, 1, 2, apertureSize=5)
    with pytest.raises(cv2.error):
        cv2.Canny(g.astype(np.float32), 1, 2)
    with pytest.raises(cv2.error):
        cv2.fillPoly(g, [np.array([[0.5, 1.0], [2.0, 3.0], [3.0, 1.0]])], 255)
    with pytest.raises(cv2.error):
        cv2.HoughLinesP(g, 0, np.pi / 60, 10)
    with pytest.raises(cv2.error):
        cv2.HoughLinesP(g, 1e-3, np.pi / 60, 10)                       # more distance bins than a row of LDS holds
    with pytest.raises(cv2.error):
        cv2.hough_accumulator(g, 1e-3, np.pi / 60)
    with pytest.raises(cv2.error):
        cv2.bitwise_and(g, np.zeros((5, 5), np.uint8))
    c = np.zeros((4, 4, 3), np.uint8)
    with pytest.raises(cv2.error):
        cv2.inRange(c, np.array([0.5, 0, 0]), np.array([9, 9, 9]))      # non-integral bound
    with pytest.raises(cv2.error):
        cv2.inRange(c, np.array([0, 0]), np.array([9, 9]))              # two bounds for three channels
    with pytest.raises(cv2.error):
        cv2.inRange(g, 0, 9)                                            # single-channel source
    with pytest.raises(cv2.error):
        cv2.bitwise_and(c, c, mask=np.zeros((5, 5), np.uint8))
    with pytest.raises(cv2.error):
        cv2.bitwise_or(g, np.zeros((5, 5), np.uint8))


def test_lds_atomic_rate_probe():
    """The measurement aid behind the vote figures: plausible rates, ordered as the address patterns suggest."""
    from lanecv import _abi
    own = _abi.lds_atomic_rate(1, False)
    one = _abi.lds_atomic_rate(64, False)
    assert own > 100.0 and one > 1.0          # 1e9 adds/s; a whole MI355X does thousands with every lane on its own word
    assert own >= one * 0.99
    with pytest.raises(_abi.error):
        _abi.lds_atomic_rate(65, False)


def test_size_switches_do_not_leak(cv2, oracle):
    """The per-op calls keep one working set per image size and rebuild it when the size changes: alternating two
    sizes must leave the GPU's free memory where it was."""
    import hiprt
    rng = np.random.default_rng(5)
    a = rng.integers(0, 256, (240, 320), dtype=np.uint8)
    b = rng.integers(0, 256, (300, 400), dtype=np.uint8)
    want_a, want_b = oracle.canny(a, 100, 200), oracle.canny(b, 100, 200)
    for _ in range(3):
        cv2.Canny(a, 100, 200)
        cv2.Canny(b, 100, 200)
    before = hiprt.free_bytes()
    for _ in range(100):
        assert np.array_equal(cv2.Canny(a, 100, 200), want_a)
        assert np.array_equal(cv2.Canny(b, 100, 200), want_b)
    after = hiprt.free_bytes()
    assert before - after < (8 << 20), "free memory fell by %d bytes over 200 size switches" % (before - after)


def test_point_capacity_overflow_is_reported_without_segments(oracle):
    """A frame with more edge points than the pipeline keeps per frame must fail loudly even when only the
    accumulator (no HoughLinesP) was asked for."""
    import lanecv
    from lanecv import synth
    from lanecv.pipeline import LanePipeline, Recipe
    w, h = 640, 360
    rec = Recipe(color_code=6, blur=False, canny=(20, 40), roi_vertices=None, accumulator=True, segments=False)
    pipe = LanePipeline(w, h, rec, max_batch=2, nslots=1)
    noise = synth.adversarial_frames(w, h)["noise"]
    assert np.count_nonzero(oracle.canny(oracle.cvt_gray(noise, 6), 20, 40)) > w * h // 8 + 1024
    calm = synth.adversarial_frames(w, h)["ramps"]
    assert np.count_nonzero(oracle.canny(oracle.cvt_gray(calm, 6), 20, 40)) < w * h // 8
    pipe.submit(0, np.stack([calm, noise]))
    with pytest.raises(lanecv.error) as e:
        pipe.wait(0)
    assert e.value.code == lanecv._abi.LC_ECAP
    pipe.submit(0, np.stack([calm]))     # the slot stays usable, and the next batch starts with a clean status
    pipe.wait(0)
    pipe.close()
