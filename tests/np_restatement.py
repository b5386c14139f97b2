"""Second, independently written restatement of the path's arithmetic in
numpy/scipy (vectorised, closed forms where the C oracle iterates).  Used only
to cross-check oracle/lane_oracle.c for transcription errors; like the oracle
it is unpinned against a live cv2.
"""
import numpy as np
from scipy import ndimage


def gray(bgr, code):
    k = np.array([3735, 19235, 9798] if code == 6 else [9798, 19235, 3735], np.int64)
    return ((bgr.astype(np.int64) @ k + (1 << 14)) >> 15).astype(np.uint8)


def gaussian5(img):
    b = np.array([1, 4, 6, 4, 1], np.int64)
    s = ndimage.correlate(img.astype(np.int64), np.outer(b, b), mode="mirror")
    return ((s + 128) >> 8).astype(np.uint8)


def sobel(img):
    i = img.astype(np.int64)
    kx = np.array([[-1, 0, 1], [-2, 0, 2], [-1, 0, 1]], np.int64)
    dx = ndimage.correlate(i, kx, mode="nearest")
    dy = ndimage.correlate(i, kx.T, mode="nearest")
    return dx, dy


def canny_classes(img, t1, t2):
    """0 = weak candidate, 1 = none, 2 = strong (OpenCV's map encoding)."""
    lo, hi = sorted((int(np.floor(t1)), int(np.floor(t2))))
    dx, dy = sobel(img)
    mag = np.abs(dx) + np.abs(dy)
    mp = np.pad(mag, 1)
    h, w = mag.shape

    def nb(oy, ox):
        return mp[1 + oy:1 + oy + h, 1 + ox:1 + ox + w]

    ax, ay = np.abs(dx), np.abs(dy) << 15
    tg22 = ax * 13573
    tg67 = tg22 + (ax << 16)
    horiz = ay < tg22
    vert = ~horiz & (ay > tg67)
    diag = ~horiz & ~vert
    same = (dx ^ dy) >= 0          # s = +1: compare (y-1,x-1) and (y+1,x+1)
    keep = np.zeros_like(mag, bool)
    keep |= horiz & (mag > nb(0, -1)) & (mag >= nb(0, 1))
    keep |= vert & (mag > nb(-1, 0)) & (mag >= nb(1, 0))
    keep |= diag & same & (mag > nb(-1, -1)) & (mag > nb(1, 1))
    keep |= diag & ~same & (mag > nb(-1, 1)) & (mag > nb(1, -1))
    keep &= mag > lo
    cls = np.ones(mag.shape, np.uint8)
    cls[keep] = 0
    cls[keep & (mag > hi)] = 2
    return cls


def hysteresis(cls):
    cand = cls != 1
    lab, n = ndimage.label(cand, structure=np.ones((3, 3), int))
    good = np.zeros(n + 1, bool)
    good[np.unique(lab[cls == 2])] = True
    good[0] = False
    return np.where(good[lab], 255, 0).astype(np.uint8)


def canny(img, t1, t2):
    return hysteresis(canny_classes(img, t1, t2))


# ---- fillPoly: closed-form edges + closed-form Bresenham -------------------

def _clip_line(w, h, p1, p2):
    x1, y1 = p1
    x2, y2 = p2
    right, bottom = w - 1, h - 1
    c1 = (x1 < 0) + (x1 > right) * 2 + (y1 < 0) * 4 + (y1 > bottom) * 8
    c2 = (x2 < 0) + (x2 > right) * 2 + (y2 < 0) * 4 + (y2 > bottom) * 8
    if (c1 & c2) == 0 and (c1 | c2) != 0:
        if c1 & 12:
            a = 0 if c1 < 8 else bottom
            x1 += int(float(a - y1) * (x2 - x1) / (y2 - y1))
            y1 = a
            c1 = (x1 < 0) + (x1 > right) * 2
        if c2 & 12:
            a = 0 if c2 < 8 else bottom
            x2 += int(float(a - y2) * (x2 - x1) / (y2 - y1))
            y2 = a
            c2 = (x2 < 0) + (x2 > right) * 2
        if (c1 & c2) == 0 and (c1 | c2) != 0:
            if c1:
                a = 0 if c1 == 1 else right
                y1 += int(float(a - x1) * (y2 - y1) / (x2 - x1))
                x1 = a
                c1 = 0
            if c2:
                a = 0 if c2 == 1 else right
                y2 += int(float(a - x2) * (y2 - y1) / (x2 - x1))
                x2 = a
                c2 = 0
    return (c1 | c2) == 0, (x1, y1), (x2, y2)


def _line_pixels(w, h, p1, p2):
    inside = lambda p: 0 <= p[0] < w and 0 <= p[1] < h
    if not (inside(p1) and inside(p2)):
        ok, p1, p2 = _clip_line(w, h, p1, p2)
        if not ok:
            return np.zeros((0, 2), np.int64)
    dx, dy = p2[0] - p1[0], p2[1] - p1[1]
    if dx < 0:
        dx, dy, p1 = -dx, -dy, p2
    sy = 1
    if dy < 0:
        dy, sy = -dy, -1
    i = np.arange(max(dx, dy) + 1, dtype=np.int64)
    if dy > dx:   # y is the major axis
        minor = (2 * dx * i + dy - 1) // (2 * dy) if dy else i * 0
        return np.stack([p1[0] + minor, p1[1] + sy * i], 1)
    minor = (2 * dy * i + dx - 1) // (2 * dx) if dx else i * 0
    return np.stack([p1[0] + i, p1[1] + sy * minor], 1)


def _tdiv(a, b):
    q = abs(a) // abs(b)
    return q if (a >= 0) == (b > 0) else -q


def fill_poly(img, contours, color):
    h, w = img.shape
    edges = []
    for c in contours:
        c = [tuple(int(v) for v in p) for p in np.asarray(c).reshape(-1, 2)]
        for k in range(len(c)):
            p0, p1 = c[k - 1], c[k]
            px = _line_pixels(w, h, p0, p1)
            img[px[:, 1], px[:, 0]] = color
            if p0[1] == p1[1]:
                continue
            inside = lambda p: 0 <= p[0] < w and 0 <= p[1] < h
            a, b = (p0[0] << 16, p0[1]), (p1[0] << 16, p1[1])
            if inside(p0) and inside(p1):
                a, b = (a[0] + 32768, a[1]), (b[0] + 32768, b[1])
            else:
                _, t0, t1 = _clip_line(w, h, p0, p1)
                if t0[1] != t1[1]:
                    a, b = (t0[0] << 16, t0[1]), (t1[0] << 16, t1[1])
            dx = _tdiv(b[0] - a[0], b[1] - a[1])
            if p0[1] < p1[1]:
                edges.append((p0[1], p1[1], a[0] + (p0[1] - a[1]) * dx, dx))
            else:
                edges.append((p1[1], p0[1], b[0] + (p1[1] - b[1]) * dx, dx))
    if len(edges) < 2:
        return img
    for y in range(max(0, min(e[0] for e in edges)), min(h, max(e[1] for e in edges))):
        xs = sorted(e[2] + (y - e[0]) * e[3] for e in edges if e[0] <= y < e[1])
        for k in range(0, len(xs) - 1, 2):
            x1, x2 = xs[k] >> 16, xs[k + 1] >> 16
            if x1 < w and x2 >= 0:
                img[y, max(x1, 0):min(x2, w - 1) + 1] = color
    return img


# ---- Hough -----------------------------------------------------------------

def hough_table(w, h, rho, theta):
    rho, theta = np.float32(rho), np.float32(theta)
    irho = np.float32(1) / rho
    na = int(np.floor(np.pi / float(theta))) + 1
    if na > 1 and abs(np.pi - (na - 1) * float(theta)) < float(theta) / 2:
        na -= 1
    nr = int(np.rint(np.float32((w + h) * 2 + 1) / rho))
    n = np.arange(na, dtype=np.float64)
    cos = (np.cos(n * float(theta)) * float(irho)).astype(np.float32)
    sin = (np.sin(n * float(theta)) * float(irho)).astype(np.float32)
    return na, nr, cos, sin


def hough_accum(img, rho, theta):
    h, w = img.shape
    na, nr, cos, sin = hough_table(w, h, rho, theta)
    ys, xs = np.nonzero(img)
    xf, yf = xs.astype(np.float32), ys.astype(np.float32)
    acc = np.zeros((na, nr), np.int32)
    for n in range(na):
        r = np.rint(xf * cos[n] + yf * sin[n]).astype(np.int64) + (nr - 1) // 2
        acc[n] = np.bincount(r, minlength=nr)
    return acc


def rng_stream(n):
    state, out = (1 << 64) - 1, []
    for _ in range(n):
        state = ((state & 0xFFFFFFFF) * 4164903690 + (state >> 32)) & ((1 << 64) - 1)
        out.append(state & 0xFFFFFFFF)
    return out


def hough_lines_p(img, rho, theta, threshold, min_len, max_gap):
    """Pure-Python progressive probabilistic Hough; small images only."""
    h, w = img.shape
    na, nr, cos, sin = hough_table(w, h, rho, theta)
    min_len, max_gap = int(np.rint(min_len)), int(np.rint(max_gap))
    acc = np.zeros((na, nr), np.int64)
    mask = (img != 0)
    ys, xs = np.nonzero(img)
    nz = list(zip(xs.tolist(), ys.tolist()))
    state = (1 << 64) - 1
    lines = []
    f32 = np.float32

    def bins(x, y):
        return np.rint(f32(x) * cos + f32(y) * sin).astype(np.int64) + (nr - 1) // 2

    for count in range(len(nz), 0, -1):
        state = ((state & 0xFFFFFFFF) * 4164903690 + (state >> 32)) & ((1 << 64) - 1)
        idx = (state & 0xFFFFFFFF) % count
        x0, y0 = nz[idx]
        nz[idx] = nz[count - 1]
        if not mask[y0, x0]:
            continue
        r = bins(x0, y0)
        acc[np.arange(na), r] += 1
        vals = acc[np.arange(na), r]
        n_best = int(np.argmax(vals))          # first maximum wins ties
        if vals[n_best] < threshold:
            continue
        a, b = -sin[n_best], cos[n_best]
        if abs(a) > abs(b):
            xflag, dx0 = True, (1 if a > 0 else -1)
            dy0 = int(np.rint(b * f32(65536) / abs(a)))
            sx, sy = x0, (y0 << 16) + 32768
        else:
            xflag, dy0 = False, (1 if b > 0 else -1)
            dx0 = int(np.rint(a * f32(65536) / abs(b)))
            sx, sy = (x0 << 16) + 32768, y0
        ends = []
        for sgn in (1, -1):
            x, y, gap, end = sx, sy, 0, None
            while True:
                j1, i1 = (x, y >> 16) if xflag else (x >> 16, y)
                if not (0 <= j1 < w and 0 <= i1 < h):
                    break
                if mask[i1, j1]:
                    gap, end = 0, (j1, i1)
                else:
                    gap += 1
                    if gap > max_gap:
                        break
                x += sgn * dx0
                y += sgn * dy0
            ends.append(end)
        good = (abs(ends[1][0] - ends[0][0]) >= min_len or abs(ends[1][1] - ends[0][1]) >= min_len)
        for k, sgn in enumerate((1, -1)):
            x, y = sx, sy
            while True:
                j1, i1 = (x, y >> 16) if xflag else (x >> 16, y)
                if mask[i1, j1]:
                    if good:
                        acc[np.arange(na), bins(j1, i1)] -= 1
                    mask[i1, j1] = False
                if (j1, i1) == ends[k]:
                    break
                x += sgn * dx0
                y += sgn * dy0
        if good:
            lines.append((ends[0][0], ends[0][1], ends[1][0], ends[1][1]))
    return np.array(lines, np.int32).reshape(-1, 4), acc.astype(np.int32)


def resize_linear_np(img, dsize):
    """cv2.resize(img, dsize) INTER_LINEAR on 8-bit data, vectorised: float32 coordinates as OpenCV computes
    them, 11-bit weights, the `>> 4`, `>> 16`, `+ 2 >> 2` vertical pass; exact 2x2 decimation -> area mean."""
    a = np.asarray(img)
    sh, sw = a.shape[:2]
    dw, dh = int(dsize[0]), int(dsize[1])
    a3 = a.reshape(sh, sw, -1).astype(np.int64)
    sx_scale, sy_scale = 1.0 / (dw / sw), 1.0 / (dh / sh)
    if sx_scale == 2.0 and sy_scale == 2.0:
        out = (a3[0::2, 0::2][:dh, :dw] + a3[0::2, 1::2][:dh, :dw] + a3[1::2, 0::2][:dh, :dw] + a3[1::2, 1::2][:dh, :dw] + 2) >> 2
        return out.astype(np.uint8).reshape((dh, dw) + a.shape[2:])

    def axis(n_dst, n_src, scale, clamp_weights):
        f = ((np.arange(n_dst, dtype=np.float64) + 0.5) * scale - 0.5).astype(np.float32)
        s = np.floor(f).astype(np.int64)
        f = (f - s.astype(np.float32)).astype(np.float32)
        if clamp_weights:
            lo, hi = s < 0, s >= n_src - 1
            f = np.where(lo | hi, np.float32(0), f)
            s = np.where(lo, 0, np.where(hi, n_src - 1, s))
        w0 = np.rint((np.float32(1) - f) * np.float32(2048)).astype(np.int64)
        w1 = np.rint(f * np.float32(2048)).astype(np.int64)
        return s, w0, w1

    sx, a0, a1 = axis(dw, sw, sx_scale, True)
    sy, b0, b1 = axis(dh, sh, sy_scale, False)
    sx1 = np.minimum(sx + 1, sw - 1)
    r0, r1 = np.clip(sy, 0, sh - 1), np.clip(sy + 1, 0, sh - 1)
    h = a3[:, sx, :] * a0[None, :, None] + a3[:, sx1, :] * a1[None, :, None]       # (sh, dw, cn)
    h0, h1 = h[r0], h[r1]
    out = (((b0[:, None, None] * (h0 >> 4)) >> 16) + ((b1[:, None, None] * (h1 >> 4)) >> 16) + 2) >> 2
    return out.astype(np.uint8).reshape((dh, dw) + a.shape[2:])


# ---- thick lines (cv2.line / cv2.polylines with a thickness): an independent evaluation ---------------------------
# Structured differently from the oracle's scan loops: a pixel set per primitive from closed forms, united at the end.
def _circle_rows(radius):
    """{dy: half width} of OpenCV's filled midpoint circle (each step paints rows +-dy with half width dx and rows
    +-dx with half width dy)."""
    rows = {}
    err, dx, dy, plus, minus = 0, radius, 0, 1, 2 * radius - 1
    while dx >= dy:
        for r, hw in ((dy, dx), (-dy, dx), (dx, dy), (-dx, dy)):
            rows[r] = max(rows.get(r, -1), hw)
        dy += 1
        err += plus
        plus += 2
        if err > 0:
            err -= minus
            dx -= 1
            minus -= 2
    return rows


def _line2_pixels(w, h, p, q):
    """Pixels of a line between 16.16 fixed-point points, for lines that need no clipping (both ends inside)."""
    (x1, y1), (x2, y2) = p, q
    assert 0 <= x1 < (w << 16) and 0 <= x2 < (w << 16) and 0 <= y1 < (h << 16) and 0 <= y2 < (h << 16)
    out = {((x2 + 32768) >> 16, (y2 + 32768) >> 16)}
    ax, ay = abs(x2 - x1), abs(y2 - y1)
    tdiv = lambda a, b: abs(a) // b * (1 if a >= 0 else -1)            # C division truncates toward zero
    if ax > ay:
        if x2 < x1:
            x1, y1, x2, y2 = x2, y2, x1, y1
        step = tdiv((y2 - y1) * 65536, ax | 1)
        for i in range(((x2 - x1) >> 16) + 1):
            out.add((((x1 + 32768) >> 16) + i, (y1 + 32768 + i * step) >> 16))
    else:
        if y2 < y1:
            x1, y1, x2, y2 = x2, y2, x1, y1
        step = tdiv((x2 - x1) * 65536, ay | 1)
        for i in range(((y2 - y1) >> 16) + 1):
            out.add(((x1 + 32768 + i * step) >> 16, ((y1 + 32768) >> 16) + i))
    return out


def thick_line_pixels(w, h, p0, p1, thickness, caps=3):
    """Set of (x, y) a thick line paints, for lines whose quad lies inside the image with a margin (no clipping):
    outline of the quad p +- dp, rows between its corners filled between the two chains, round caps."""
    assert thickness >= 2
    X0, Y0, X1, Y1 = p0[0] << 16, p0[1] << 16, p1[0] << 16, p1[1] << 16
    t15 = thickness << 15
    pix = set()
    dx, dy = (X0 - X1) / 65536.0, (Y1 - Y0) / 65536.0
    rr = dx * dx + dy * dy
    if rr > 2.220446049250313e-16:
        r = (t15 + (thickness & 1) * 32768.0) / np.sqrt(rr)
        dpx, dpy = int(np.rint(dy * r)), int(np.rint(dx * r))
        quad = [(X0 + dpx, Y0 + dpy), (X0 - dpx, Y0 - dpy), (X1 - dpx, Y1 - dpy), (X1 + dpx, Y1 + dpy)]
        for k in range(4):
            pix |= _line2_pixels(w, h, quad[k - 1], quad[k])
        # fill: the chain to the right and to the left of the topmost vertex (first of the lowest y), each edge walked
        # from the rounded row of its upper end with slope ((xe - xs) * 2 + n) / (2 n), n = rows the edge spans
        rows_of = [(y + 32768) >> 16 for _, y in quad]
        imin = min(range(4), key=lambda k: (quad[k][1], k))
        ymin, ymax = rows_of[imin], max(rows_of)

        def chain(di):
            xs_at = {}
            idx0, y = imin, ymin
            for _ in range(4):
                idx = (idx0 + di) % 4
                ty = rows_of[idx]
                if ty > y:
                    xs, xe = quad[idx0][0], quad[idx][0]
                    num, den = (xe - xs) * 2 + (ty - y), 2 * (ty - y)
                    slope = abs(num) // den * (1 if num >= 0 else -1)
                    for yy in range(y, ty):
                        xs_at[yy] = xs + (yy - y) * slope
                    y = ty
                idx0 = idx
                if idx == imin:
                    break
            return xs_at

        a, b = chain(1), chain(3)
        for yy in range(ymin, ymax):
            if yy in a and yy in b:
                lo, hi = min(a[yy], b[yy]), max(a[yy], b[yy])
                for xx in range((lo + 32768) >> 16, ((hi + 32768) >> 16) + 1):
                    pix.add((xx, yy))
    radius = (t15 + 32768) >> 16
    rows = _circle_rows(radius)
    for k, (cx, cy) in enumerate(((p0[0], p0[1]), (p1[0], p1[1]))):
        if caps & (k + 1):
            for r, hw in rows.items():
                for xx in range(cx - hw, cx + hw + 1):
                    pix.add((xx, cy + r))
    return {(x, y) for x, y in pix if 0 <= x < w and 0 <= y < h}
