"""GPU parity of the overlay's cv2.line / cv2.polylines (C ABI lc_line / lc_polylines) against the oracle's
restatement of OpenCV's thick lines, bit for bit, on the calls the reference's draw functions make."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def cv2():
    import lanecv
    return lanecv


@pytest.mark.parametrize("thickness", [1, 2, 3, 5, 8, 10, 31])
def test_line_random_segments(cv2, oracle, thickness):
    rng = np.random.default_rng(thickness)
    for shape in ((90, 130), (90, 130, 3), (57, 61, 4)):
        h, w = shape[:2]
        for _ in range(25):
            p0 = (int(rng.integers(-40, w + 40)), int(rng.integers(-40, h + 40)))
            p1 = (int(rng.integers(-40, w + 40)), int(rng.integers(-40, h + 40)))
            color = [int(v) for v in rng.integers(0, 256, 4)][:(1 if len(shape) == 2 else shape[2])]
            base = rng.integers(0, 256, shape, dtype=np.uint8)
            got, want = base.copy(), base.copy()
            assert cv2.line(got, p0, p1, color if len(color) > 1 else color[0], thickness) is got
            oracle.line(want, p0, p1, color, thickness)
            assert (got == want).all(), (shape, p0, p1, thickness)


def test_degenerate_and_axis_aligned_lines(cv2, oracle):
    for p0, p1 in (((20, 20), (20, 20)), ((5, 30), (70, 30)), ((40, 2), (40, 58)), ((-10, 59), (90, 59)), ((0, 0), (79, 59))):
        for t in (1, 2, 5, 8):
            got = np.zeros((60, 80, 3), np.uint8)
            want = np.zeros_like(got)
            cv2.line(got, p0, p1, (255, 0, 0), t)
            oracle.line(want, p0, p1, (255, 0, 0), t)
            assert (got == want).all(), (p0, p1, t)


def test_draw_lines_of_the_image_script(cv2, oracle):
    """`draw_lines` of Simple Lane Detection with OpenCV_2(images).py:27-51 as `pipeline` calls it (:146-153): the two
    lane lines, thickness 5, colour [255, 0, 0], onto a black 3-channel image that is then blended over the frame."""
    from lanecv import synth
    w, h = 960, 540
    frame = synth.synth_road(w, h, seed=2, t=5)
    lines = [[[211, 540, 423, 324]], [[843, 540, 571, 324]]]
    line_img = np.zeros((h, w, 3), np.uint8)
    ref_img = np.zeros((h, w, 3), np.uint8)
    for line in lines:
        for x1, y1, x2, y2 in line:
            cv2.line(line_img, (x1, y1), (x2, y2), [255, 0, 0], 5)
            oracle.line(ref_img, (x1, y1), (x2, y2), [255, 0, 0], 5)
    assert line_img.any() and (line_img == ref_img).all()
    out = cv2.addWeighted(frame, 0.8, line_img, 1.0, 0.0)
    assert (out == oracle.add_weighted(frame, 0.8, ref_img, 1.0, 0.0)).all()


@pytest.mark.parametrize("thickness,size", [(8, (1280, 720)), (5, (640, 360))])
def test_draw_curve_polylines(cv2, oracle, thickness, size):
    """`draw_curve` of Simple Lane Detection with OpenCV_5(mp4).py:21-52 (one vertex per image row, thickness 8,
    colour [0, 255, 0]) and the F1 tracker's curves (F1_vo_Lane_Tracking.py:137-140, thickness 5): the (1, N, 2) int32
    arrays the scripts build, curves that leave the image included."""
    w, h = size
    plot_y = np.linspace(0, h - 1, h)
    for fit in ((2.0e-4, -0.9, 0.62 * w), (-3.0e-4, 0.55, 0.40 * w), (1.5e-3, -1.9, 1.2 * w)):
        x = fit[0] * plot_y ** 2 + fit[1] * plot_y + fit[2]
        pts = np.array([np.transpose(np.vstack([x, plot_y]))]).astype(np.int32)
        got = np.zeros((h, w, 3), np.uint8)
        want = np.zeros_like(got)
        cv2.polylines(got, pts, False, [0, 255, 0], thickness)
        oracle.polylines(want, pts, False, [0, 255, 0], thickness)
        assert want.any() and (got == want).all(), fit
        got2 = np.zeros((h, w, 3), np.uint8)
        cv2.polylines(got2, [pts[0]], False, (0, 255, 0), thickness)          # the F1 script's list-of-arrays form
        assert (got2 == want).all()


def test_closed_polylines_and_several_contours(cv2, oracle):
    rng = np.random.default_rng(11)
    for closed in (False, True):
        for t in (1, 3, 6):
            contours = [rng.integers(-10, 110, (int(rng.integers(1, 7)), 2)).astype(np.int32) for _ in range(4)]
            got = np.zeros((100, 100), np.uint8)
            want = np.zeros_like(got)
            cv2.polylines(got, contours, closed, 255, t)
            oracle.polylines(want, contours, closed, 255, t)
            assert (got == want).all(), (closed, t)


def test_draw_argument_errors(cv2):
    img = np.zeros((10, 10, 3), np.uint8)
    with pytest.raises(cv2.error):
        cv2.line(img, (0, 0), (5, 5), (1, 2, 3), 0)
    with pytest.raises(cv2.error):
        cv2.line(img, (0, 0), (5, 5), (1, 2, 3), 2, lineType=cv2.LINE_AA)
    with pytest.raises(cv2.error):
        cv2.polylines(img, [np.zeros((3, 2), np.float32)], False, (1, 2, 3), 2)
    with pytest.raises(cv2.error):
        cv2.line(np.zeros((10, 10), np.float32), (0, 0), (5, 5), 1, 2)
