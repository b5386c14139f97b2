"""CPU oracle of cv2.resize (INTER_LINEAR, 8-bit): hand-derived known answers and an independent vectorised
restatement.  cv2 is not installed here, so this is what pins the oracle (parity unpinned)."""
import numpy as np
import pytest

from np_restatement import resize_linear_np


def test_known_answers(oracle):
    # two pixels stretched to four: weights (2048, 0), (1536, 512), (512, 1536), (2048 on the last) by hand:
    # H = 0, 255*512, 255*1536, 255*2048; D = ((2048 * (H >> 4)) >> 16) + 2 >> 2 = 0, 64, 191, 255
    src = np.array([[0, 255]], np.uint8)
    assert oracle.resize_linear(src, (4, 1)).tolist() == [[0, 64, 191, 255]]
    # a constant image stays constant for any size pair
    c = np.full((7, 5, 3), 200, np.uint8)
    for size in ((13, 9), (3, 2), (5, 7), (1, 1), (10, 14)):
        assert (oracle.resize_linear(c, size) == 200).all()
    rng = np.random.default_rng(1)
    # exact 2x2 decimation is the rounded mean of the four (OpenCV reroutes it to INTER_AREA)
    a = rng.integers(0, 256, (72, 128, 3), dtype=np.uint8)
    want = ((a[0::2, 0::2].astype(int) + a[0::2, 1::2] + a[1::2, 0::2] + a[1::2, 1::2] + 2) >> 2).astype(np.uint8)
    assert (oracle.resize_linear(a, (64, 36)) == want).all()
    # decimation by 3 samples the middle pixel of each 3x3 block exactly: (dx + 0.5) * 3 - 0.5 = 3 dx + 1
    b = rng.integers(0, 256, (108, 192, 3), dtype=np.uint8)
    assert (oracle.resize_linear(b, (64, 36)) == b[1::3, 1::3]).all()
    # same size: identity
    assert (oracle.resize_linear(b, (192, 108)) == b).all()


@pytest.mark.parametrize("sw,sh,dw,dh", [(1920, 1080, 640, 360), (1280, 720, 640, 360), (641, 359, 640, 360),
                                         (100, 50, 333, 77), (37, 91, 64, 64), (5, 3, 2, 9), (1, 1, 4, 4), (9, 1, 3, 1)])
def test_against_numpy_restatement(oracle, sw, sh, dw, dh):
    rng = np.random.default_rng(sw * 7 + dh)
    for shape in ((sh, sw, 3), (sh, sw)):
        img = rng.integers(0, 256, shape, dtype=np.uint8)
        got = oracle.resize_linear(img, (dw, dh))
        assert got.shape == (dh, dw) + shape[2:]
        assert (got == resize_linear_np(img, (dw, dh))).all()


def test_add_weighted_oracle(oracle):
    """addWeighted: every pair of byte values with the scripts' weight pairs against exact arithmetic (these
    weights never produce a .5 tie, so float rounding order cannot matter), plus saturation."""
    from fractions import Fraction
    a = np.repeat(np.arange(256, dtype=np.uint8)[:, None], 256, axis=1)
    b = a.T.copy()
    for alpha, beta, fa, fb in ((0.8, 1.0, Fraction(4, 5), 1), (1.0, 0.8, 1, Fraction(4, 5)), (1.0, 0.6, 1, Fraction(3, 5))):
        got = oracle.add_weighted(a, alpha, b, beta, 0.0)
        want = np.array([[min(255, int(x * fa + y * fb + Fraction(1, 2))) for y in range(256)] for x in range(256)], np.uint8)
        assert (got == want).all()
    assert oracle.add_weighted(np.array([[200]], np.uint8), 1.0, np.array([[100]], np.uint8), 1.0, 0.0)[0, 0] == 255
    assert oracle.add_weighted(np.array([[10]], np.uint8), 1.0, np.array([[100]], np.uint8), -1.0, 0.0)[0, 0] == 0
    assert oracle.add_weighted(np.array([[3]], np.uint8), 0.5, np.array([[0]], np.uint8), 0.0, 0.0)[0, 0] == 2    # 1.5 -> even
    assert oracle.add_weighted(np.array([[5]], np.uint8), 0.5, np.array([[0]], np.uint8), 0.0, 0.0)[0, 0] == 2    # 2.5 -> even
