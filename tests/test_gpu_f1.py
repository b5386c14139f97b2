"""GPU parity of the F1 tracker's front and back ends (`F1_vo_Lane_Tracking.py:185`, `:114-132`): cv2.resize and
the ordered non-zero pixel list, per op and through the batched pipeline, against the CPU oracle, bit for bit.
cv2 itself is not installed here (parity unpinned)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def cv2():
    import lanecv
    return lanecv


@pytest.mark.parametrize("sw,sh,dw,dh", [(1920, 1080, 640, 360), (1280, 720, 640, 360), (641, 359, 640, 360),
                                         (100, 50, 333, 77), (37, 91, 64, 64), (5, 3, 2, 9), (1, 1, 4, 4), (640, 360, 640, 360)])
def test_resize(cv2, oracle, sw, sh, dw, dh):
    rng = np.random.default_rng(sw + 3 * dh)
    for shape in ((sh, sw, 3), (sh, sw)):
        img = rng.integers(0, 256, shape, dtype=np.uint8)
        got = cv2.resize(img, (dw, dh))
        assert got.shape == (dh, dw) + shape[2:] and got.dtype == np.uint8
        assert (got == oracle.resize_linear(img, (dw, dh))).all()
    wide = rng.integers(0, 256, (sh, sw + 5, 3), dtype=np.uint8)
    view = wide[:, 2:2 + sw]                                 # rows strided, pixels dense
    assert (cv2.resize(view, (dw, dh), interpolation=cv2.INTER_LINEAR) == oracle.resize_linear(np.ascontiguousarray(view), (dw, dh))).all()


def test_resize_errors(cv2):
    img = np.zeros((8, 8, 3), np.uint8)
    with pytest.raises(cv2.error):
        cv2.resize(img, None, fx=0.5, fy=0.5)
    with pytest.raises(cv2.error):
        cv2.resize(img, (4, 4), interpolation=3)
    with pytest.raises(cv2.error):
        cv2.resize(np.zeros((8, 8, 4), np.uint8), (4, 4))
    with pytest.raises(cv2.error):
        cv2.resize(img.astype(np.float32), (4, 4))


@pytest.mark.parametrize("src,keep", [((1920, 1080), False), ((1920, 1080), True), ((1280, 720), True), ((1280, 720), False),
                                      ((640, 360), False), ((700, 500), False)])
def test_f1_frame_to_curves(cv2, oracle, src, keep):
    """`process_frame` + `detect_and_draw` up to the fitted curves: resize -> BGR2GRAY -> blur -> Canny(50,150)
    -> car mask -> trapezoid -> nonzero -> mid-x split -> memory fit, pipeline against oracle + NumPy."""
    from lanecv import synth, _abi, glue
    from lanecv.pipeline import LanePipeline, Recipe
    w, h = 640, 360
    rec = Recipe.f1(w, h)
    # (1920 x 1080 without intermediates: the reduction by three is fused into the edge kernel, no resized frame exists)
    pipe = LanePipeline(w, h, rec, max_batch=4, nslots=2, src_size=src, keep_intermediates=keep)
    fused = src == (1920, 1080) and not keep
    frames = np.stack([synth.synth_road(src[0], src[1], seed=6, t=t) for t in range(3)] +
                      [synth.adversarial_frames(src[0], src[1])["noise"]])
    got_trk, want_trk = glue.PixelCurveTracker(), glue.PixelCurveTracker()
    for rep in range(2):                                     # the slot is reused: dirty tracking must hold
        batch = frames if rep == 0 else frames[::-1]
        pipe.submit(rep, batch)
        pipe.wait(rep)
        masked = pipe.read(rep, _abi.BUF_MASKED)
        if fused:
            with pytest.raises(cv2.error):
                pipe.read(rep, _abi.BUF_RESIZED)
        small = pipe.read(rep, _abi.BUF_RESIZED) if src != (w, h) and not fused else None
        nz = pipe.nonzero(rep)
        for i, f in enumerate(batch):
            fr = oracle.resize_linear(f, (w, h))
            assert small is None or (small[i] == fr).all(), i
            r = oracle.lane_frame(fr, rec.color_code, rec.blur, rec.canny[0], rec.canny[1], pipe.roi, rec.rho, rec.theta,
                                  rec.threshold, rec.min_line_length, rec.max_line_gap, car_mask=rec.car_mask,
                                  want_acc=False, want_segments=False)
            assert (masked[i] == r["masked"]).all(), i
            ys, xs = np.nonzero(r["masked"])
            assert (nz[i][0] == ys).all() and (nz[i][1] == xs).all(), i
            got = got_trk.fit_poly((h, w), *glue.split_at_midpoint(nz[i][0], nz[i][1], w))
            want = want_trk.fit_poly((h, w), *glue.split_at_midpoint(ys, xs, w))
            for g, x in zip(got, want):
                assert (g is None and x is None) or (g == x).all()
    pipe.close()


def test_points_of_a_hough_recipe(oracle):
    """The point list is also readable from a recipe with an accumulator (no HoughLinesP step to permute it)."""
    from lanecv import synth, _abi
    from lanecv.pipeline import LanePipeline, Recipe
    w, h = 960, 540
    rec = Recipe.simple(w, h)
    rec.segments, rec.accumulator = False, True
    pipe = LanePipeline(w, h, rec, max_batch=2, nslots=1)
    frames = np.stack([synth.synth_road(w, h, seed=3, t=t) for t in range(2)])
    pipe.submit(0, frames)
    pipe.wait(0)
    masked = pipe.read(0, _abi.BUF_MASKED)
    for i, (ys, xs) in enumerate(pipe.nonzero(0)):
        wy, wx = np.nonzero(masked[i])
        assert (ys == wy).all() and (xs == wx).all() and len(ys) > 0
    pipe.close()


def test_add_weighted(cv2, oracle):
    """cv2.addWeighted with the scripts' weight pairs and a few awkward ones, 1 and 3 channels."""
    rng = np.random.default_rng(3)
    for shape in ((360, 640, 3), (37, 53), (1, 1, 3), (5, 7, 3)):
        a = rng.integers(0, 256, shape, dtype=np.uint8)
        b = rng.integers(0, 256, shape, dtype=np.uint8)
        b[..., 0] = 0 if a.ndim == 3 else b[..., 0]
        for alpha, beta, gamma in ((0.8, 1.0, 0.0), (1.0, 0.8, 0.0), (1.0, 0.6, 0), (0.5, 0.5, 0.0), (1.0, 1.0, -300.0),
                                   (-1.0, 0.25, 127.5), (2.5, 0.0, 3.0)):
            got = cv2.addWeighted(a, alpha, b, beta, gamma)
            assert got.shape == a.shape and got.dtype == np.uint8
            assert (got == oracle.add_weighted(a, alpha, b, beta, gamma)).all(), (shape, alpha, beta, gamma)
    # every pair of byte values with the weights the scripts use
    a = np.repeat(np.arange(256, dtype=np.uint8)[:, None], 256, axis=1)
    b = a.T.copy()
    for alpha, beta in ((0.8, 1.0), (1.0, 0.8), (1.0, 0.6)):
        want = np.clip(np.floor(a.astype(np.float64) * alpha + b.astype(np.float64) * beta + 0.5), 0, 255).astype(np.uint8)
        assert (cv2.addWeighted(a, alpha, b, beta, 0.0) == want).all()     # no exact ties: plain rounding is unambiguous
    with pytest.raises(cv2.error):
        cv2.addWeighted(a, 1.0, b[:5], 1.0, 0.0)
