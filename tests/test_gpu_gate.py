"""GPU parity of the colour gate (cvtColor to HSV, inRange, bitwise_or, masked 3-channel bitwise_and) and of
the gated recipe of `Simple Lane Detection with OpenCV_6(mp4).py:54-113`, against the CPU oracle, bit for bit.
cv2 itself is not installed here: the oracle restates OpenCV's fixed-point HSV code (parity unpinned)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

WHITE_YELLOW = ((0, 0, 200), (180, 30, 255), (15, 100, 100), (35, 255, 255))


@pytest.fixture(scope="module")
def cv2():
    import lanecv
    return lanecv


def test_hsv_every_colour(cv2, oracle):
    """All 2^24 (b, g, r) triples in one 4096 x 4096 image."""
    v = np.arange(1 << 24, dtype=np.uint32)
    img = np.stack([(v & 255), (v >> 8) & 255, (v >> 16) & 255], axis=1).astype(np.uint8).reshape(4096, 4096, 3)
    got = cv2.cvtColor(img, cv2.COLOR_BGR2HSV)
    assert got.shape == img.shape and (got == oracle.cvt_hsv(img, oracle.COLOR_BGR2HSV)).all()
    sub = img[::7, ::5]
    assert (cv2.cvtColor(sub, cv2.COLOR_RGB2HSV) == oracle.cvt_hsv(np.ascontiguousarray(sub), oracle.COLOR_RGB2HSV)).all()
    assert int(got[..., 0].max()) < 180


@pytest.mark.parametrize("w,h", [(960, 540), (641, 359), (33, 7), (1, 1), (3, 5)])
def test_gate_ops(cv2, oracle, w, h):
    from lanecv import synth
    rng = np.random.default_rng(w * 31 + h)
    frames = [rng.integers(0, 256, (h, w, 3), dtype=np.uint8), synth.synth_road(w, h, seed=2, t=5)]
    bounds = [((0, 0, 200), (180, 30, 255)), ((15, 100, 100), (35, 255, 255)), ((-5, 10, 0), (300, 256, 255)),
              ((100, 0, 0), (50, 255, 255)), ((0, 0, 0), (255, 255, 255)), ((0, 0, 256), (180, 255, 300)),
              ((90, 128, 64), (90, 128, 255))]
    for f in frames:
        hsv = cv2.cvtColor(f, cv2.COLOR_BGR2HSV)
        assert (hsv == oracle.cvt_hsv(f)).all()
        masks = []
        for lo, hi in bounds:
            m = cv2.inRange(hsv, np.array(lo), np.array(hi))
            assert m.dtype == np.uint8 and (m == oracle.in_range(hsv, lo, hi)).all(), (lo, hi)
            masks.append(m)
        both = cv2.bitwise_or(masks[0], masks[1])
        assert (both == oracle.bitwise_or(masks[0], masks[1])).all()
        kept = cv2.bitwise_and(f, f, mask=both)
        assert kept.shape == f.shape and (kept == oracle.and3_mask(f, both)).all()
        g = rng.integers(0, 256, f.shape, dtype=np.uint8)
        assert (cv2.bitwise_and(f, g) == (f & g)).all()
        assert (cv2.bitwise_and(f, g, mask=masks[2]) == np.where(masks[2][..., None] != 0, f & g, 0)).all()


def test_script_call_sequence(cv2, oracle):
    """The gate exactly as `..._6(mp4).py:65-88` spells it, call by call, against the oracle's fused form."""
    from lanecv import synth
    w, h = 640, 360
    for image in (synth.synth_road(w, h, seed=11, t=2), synth.adversarial_frames(w, h)["noise"]):
        hsv = cv2.cvtColor(image, cv2.COLOR_BGR2HSV)
        mask_white = cv2.inRange(hsv, np.array([0, 0, 200]), np.array([180, 30, 255]))
        mask_yellow = cv2.inRange(hsv, np.array([15, 100, 100]), np.array([35, 255, 255]))
        combined_mask = cv2.bitwise_or(mask_white, mask_yellow)
        masked_image = cv2.bitwise_and(image, image, mask=combined_mask)
        gray_image = cv2.cvtColor(masked_image, cv2.COLOR_BGR2GRAY)
        assert (gray_image == oracle.gate_gray(image, WHITE_YELLOW)).all()
        assert combined_mask.any() and not combined_mask.all()


def _oracle_gated(oracle, f, rec, roi):
    return oracle.lane_frame(f, rec.color_code, rec.blur, rec.canny[0], rec.canny[1], roi, rec.rho, rec.theta,
                             rec.threshold, rec.min_line_length, rec.max_line_gap, want_acc=True, want_segments=True,
                             gate=rec.hsv_gate)


@pytest.mark.parametrize("w,h,keep", [(960, 540, True), (1280, 720, False), (641, 359, True), (641, 359, False),
                                      (130, 70, True), (1920, 1080, False)])
def test_gated_recipe_parity(oracle, w, h, keep):
    """Recipe.hsv_gated through the batched pipeline: the streaming front kernel fuses the gate (widths that
    are multiples of 4), other widths pass through a gated gray image."""
    from lanecv import synth, _abi
    from lanecv.pipeline import LanePipeline, Recipe
    rec = Recipe.hsv_gated(w, h)
    rec.accumulator = True
    if w < 200:
        rec.threshold, rec.min_line_length = 20, 10
    pipe = LanePipeline(w, h, rec, max_batch=4, nslots=2, seg_capacity=2048, keep_intermediates=keep)
    adv = synth.adversarial_frames(w, h)
    batches = [np.stack([synth.synth_road(w, h, seed=9, t=t) for t in range(3)]),
               np.stack([adv["noise"], adv["white"], adv["black"], synth.synth_road(w, h, seed=1, t=40, shadow=False)])]
    nseg = 0
    for slot, frames in enumerate(batches):
        pipe.submit(slot, frames)
        segs = pipe.collect(slot)
        masked, acc = pipe.read(slot, _abi.BUF_MASKED), pipe.read(slot, _abi.BUF_ACC)
        gray = pipe.read(slot, _abi.BUF_GRAY) if keep else None
        edges = pipe.read(slot, _abi.BUF_EDGES) if keep else None
        for i, f in enumerate(frames):
            r = _oracle_gated(oracle, f, rec, pipe.roi)
            if keep:
                assert (gray[i] == r["gray"]).all(), i
                assert (edges[i] == r["edges"]).all(), i
            assert (masked[i] == r["masked"]).all(), i
            assert (acc[i] == r["acc"]).all(), i
            got = np.zeros((0, 4), np.int32) if segs[i] is None else segs[i].reshape(-1, 4)
            assert got.shape == r["segments"].shape and (got == r["segments"]).all(), i
            nseg += len(got)
    assert nseg > 0
    pipe.close()


def test_gate_changes_the_result(oracle):
    """Same frames with and without the gate differ (the gate is really applied), and a gate that keeps
    everything equals no gate."""
    from lanecv import synth, _abi
    from lanecv.pipeline import LanePipeline, Recipe
    w, h = 960, 540
    frames = np.stack([synth.synth_road(w, h, seed=9, t=t) for t in range(2)])
    out = {}
    for name, gate in (("none", None), ("all", ((0, 0, 0), (255, 255, 255), (1, 1, 1), (0, 0, 0))),
                       ("script", WHITE_YELLOW)):
        rec = Recipe.hsv_gated(w, h)
        rec.hsv_gate = gate
        pipe = LanePipeline(w, h, rec, max_batch=2, nslots=1)
        pipe.submit(0, frames)
        pipe.wait(0)
        out[name] = pipe.read(0, _abi.BUF_MASKED)
        pipe.close()
    assert (out["none"] == out["all"]).all()
    assert (out["none"] != out["script"]).any()


GATES = [
    WHITE_YELLOW,                                                         # the script's numbers (one-sided form)
    ((0, 0, 180), (180, 60, 255), (10, 80, 80), (40, 255, 255)),         # same form, other sensitivities
    ((0, 0, 0), (179, 255, 255), (0, 0, 0), (149, 255, 255)),            # same form at its limits: keeps everything
    ((0, 0, 200), (180, 30, 255), (15, 100, 100), (150, 255, 255)),      # hue interval reaches the wrapped hues
    ((160, 50, 50), (179, 200, 220), (0, 10, 10), (5, 250, 250)),        # two-sided boxes, red hues on both ends
    ((0, 0, 200), (180, 30, 254), (15, 100, 100), (35, 255, 255)),       # an upper bound on V
    ((50, 0, 0), (40, 255, 255), (15, 100, 100), (35, 90, 255)),         # both boxes empty
]


@pytest.mark.parametrize("gi", range(len(GATES)))
def test_fused_gate_forms(oracle, gi):
    """The streaming kernel has a general gate and a cheaper form for boxes shaped like the script's; both must
    give the oracle's gated gray, with and without the intermediate images."""
    from lanecv import synth, _abi
    from lanecv.pipeline import LanePipeline, Recipe
    w, h = 640, 360
    gate = GATES[gi]
    rng = np.random.default_rng(gi)
    sat = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    sat[:, : w // 2, 0] = sat[:, : w // 2, 1]                              # left half: many b == g ties and low saturation
    frames = np.stack([synth.synth_road(w, h, seed=9, t=3), synth.adversarial_frames(w, h)["noise"], sat])
    masked = {}
    for keep in (True, False):
        rec = Recipe.hsv_gated(w, h)
        rec.hsv_gate, rec.segments = gate, False
        pipe = LanePipeline(w, h, rec, max_batch=3, nslots=1, keep_intermediates=keep)
        pipe.submit(0, frames)
        pipe.wait(0)
        masked[keep] = pipe.read(0, _abi.BUF_MASKED)
        if keep:
            gray = pipe.read(0, _abi.BUF_GRAY)
            for i, f in enumerate(frames):
                assert (gray[i] == oracle.gate_gray(f, gate)).all(), i
        pipe.close()
    assert (masked[True] == masked[False]).all()


@pytest.mark.parametrize("gi", [0, 3, 4, 5])
def test_fused_gate_every_colour(oracle, gi):
    """Every one of the 2^24 colours (one 4096 x 4096 frame) through the streaming kernel's gate: the gated gray
    must equal the oracle's for the script's boxes, a hue interval that reaches the wrapped hues, two-sided boxes
    and a bounded V."""
    from lanecv import _abi
    from lanecv.pipeline import LanePipeline, Recipe
    w = h = 4096
    c = np.arange(1 << 24, dtype=np.uint32)
    frame = np.stack([c & 255, (c >> 8) & 255, c >> 16], axis=1).astype(np.uint8).reshape(1, h, w, 3)
    rec = Recipe.hsv_gated(w, h)
    rec.hsv_gate, rec.segments, rec.roi_vertices = GATES[gi], False, None
    pipe = LanePipeline(w, h, rec, max_batch=1, nslots=1, keep_intermediates=True)
    pipe.submit(0, frame)
    pipe.wait(0)
    gray = pipe.read(0, _abi.BUF_GRAY)[0]
    pipe.close()
    want = oracle.gate_gray(frame[0], GATES[gi])
    bad = np.flatnonzero(gray.ravel() != want.ravel())
    assert bad.size == 0, (bad.size, [tuple(frame[0].reshape(-1, 3)[i]) for i in bad[:5]])
