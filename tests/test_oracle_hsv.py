"""CPU checks of the oracle's colour gate (BGR2HSV, inRange, bitwise_or, masked bitwise_and, gated gray):
hand-derived known answers and an independent integer restatement in NumPy.  PARITY UNPINNED like the rest of
the oracle: no cv2 here, the contract is OpenCV's 8U fixed-point HSV with a hue range of 180."""
from fractions import Fraction

import numpy as np


def _round_half_even(fr):
    q, r = divmod(fr.numerator, fr.denominator)
    twice = 2 * r
    if twice > fr.denominator or (twice == fr.denominator and q % 2 == 1):
        q += 1
    return q


def _np_hsv(bgr):
    """Vectorised restatement with tables built by exact rational rounding."""
    sdiv = np.array([0] + [_round_half_even(Fraction(255 << 12, i)) for i in range(1, 256)], np.int64)
    hdiv = np.array([0] + [_round_half_even(Fraction(180 << 12, 6 * i)) for i in range(1, 256)], np.int64)
    b, g, r = (bgr[..., k].astype(np.int64) for k in range(3))
    v = np.maximum(np.maximum(b, g), r)
    diff = v - np.minimum(np.minimum(b, g), r)
    s = (diff * sdiv[v] + 2048) >> 12
    h = np.where(v == r, g - b, np.where(v == g, b - r + 2 * diff, r - g + 4 * diff))
    h = (h * hdiv[diff] + 2048) >> 12          # NumPy's >> on negative int64 is arithmetic, as in C++
    h = np.where(h < 0, h + 180, h)
    return np.stack([h, s, v], axis=-1).astype(np.uint8)


def test_hsv_known_answers(oracle):
    px = np.array([[[255, 255, 255], [0, 255, 255], [0, 0, 255], [255, 0, 0], [0, 255, 0], [0, 0, 0],
                    [30, 60, 90], [200, 100, 50], [128, 128, 128], [10, 200, 180]]], np.uint8)
    want = np.array([[[0, 0, 255], [30, 255, 255], [0, 255, 255], [120, 255, 255], [60, 255, 255], [0, 0, 0],
                      [15, 170, 90], [110, 191, 200], [0, 0, 128], [33, 242, 200]]], np.uint8)
    assert (oracle.cvt_hsv(px) == want).all()
    # RGB2HSV reads the same bytes with red first
    assert (oracle.cvt_hsv(px[..., ::-1].copy(), 41) == want).all()


def test_hsv_matches_the_numpy_restatement(oracle):
    rng = np.random.default_rng(3)
    img = rng.integers(0, 256, (64, 97, 3), dtype=np.uint8)
    img[0, :32] = np.repeat(np.arange(32, dtype=np.uint8)[:, None] * 8, 3, axis=1)      # grays
    img[1, :64, 0] = 0; img[1, :64, 1] = np.arange(64) * 4; img[1, :64, 2] = 255          # red -> yellow ramp
    img[2, :64, 0] = np.arange(64) * 4; img[2, :64, 1] = 0; img[2, :64, 2] = 255          # red -> magenta ramp (negative hue)
    assert (oracle.cvt_hsv(img) == _np_hsv(img)).all()
    # every (v, diff) pair of the two tables at least once
    v, d = np.meshgrid(np.arange(256), np.arange(256), indexing="ij")
    ok = d <= v
    b = (v - d)[ok].astype(np.uint8); r = v[ok].astype(np.uint8); g = ((v[ok] + (v - d)[ok]) // 2).astype(np.uint8)
    tab = np.stack([b, g, r], axis=-1)[None]
    assert (oracle.cvt_hsv(tab) == _np_hsv(tab)).all()
    assert oracle.cvt_hsv(img)[..., 0].max() < 180


def test_gate_pieces_and_composition(oracle):
    rng = np.random.default_rng(4)
    img = rng.integers(0, 256, (40, 52, 3), dtype=np.uint8)
    img[5:12, 5:30] = (250, 252, 255)          # white paint
    img[20:28, 10:40] = (40, 200, 230)         # yellow paint
    gate = (((0, 0, 200), (180, 30, 255)), ((15, 100, 100), (35, 255, 255)))   # ..._6(mp4).py:69-75
    hsv = oracle.cvt_hsv(img)
    lo_w, hi_w = np.array(gate[0][0]), np.array(gate[0][1])
    m_white = oracle.in_range(hsv, *gate[0])
    assert (m_white == np.where(((hsv >= lo_w) & (hsv <= hi_w)).all(axis=-1), 255, 0)).all()
    m_yellow = oracle.in_range(hsv, *gate[1])
    both = oracle.bitwise_or(m_white, m_yellow)
    assert (both == (m_white | m_yellow)).all() and both[6, 6] == 255 and both[22, 12] == 255
    masked = oracle.and3_mask(img, both)
    assert (masked == img * (both[..., None] // 255)).all()
    gray = oracle.gate_gray(img, gate)
    assert (gray == oracle.cvt_gray(masked)).all()
    assert (gray[both == 0] == 0).all() and (gray[both == 255] == oracle.cvt_gray(img)[both == 255]).all()
    # the gated frame is the plain frame run on the masked image
    plain = oracle.lane_frame(masked, 6, False, 100, 200, None, 6, np.pi / 60, 20, 5, 5)
    gated = oracle.lane_frame(img, 6, False, 100, 200, None, 6, np.pi / 60, 20, 5, 5, gate=gate)
    for k in ("gray", "edges", "masked", "acc"):
        assert (plain[k] == gated[k]).all(), k
