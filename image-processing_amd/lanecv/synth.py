"""Deterministic synthetic road-like BGR frames.

The reference ships no input images or videos (its scripts read whatever
*.jpg/*.png/*.mp4 sit in the working directory, e.g.
`Simple Lane Detection with OpenCV_2(images).py:157-167`), so benchmarks and
parity tests use frames drawn here: a sky gradient above the horizon, noisy
asphalt below it, two lane markings (white, solid; yellow, dashed) that
converge to a vanishing point which drifts sideways with the frame index, a
near-horizontal guard rail and an optional shadow band.
"""
import numpy as np


def _noise(rng, h, w):
    """Sum of four 3-bit uniforms minus its mean: sigma ~ 4.6, cheap to draw."""
    r = rng.integers(0, 1 << 12, size=(h, w), dtype=np.uint16)
    s = (r & 7) + ((r >> 3) & 7) + ((r >> 6) & 7) + ((r >> 9) & 7)
    return s.astype(np.int16) - 14


def synth_road(w, h, seed=0, t=0, shadow=True):
    """One uint8 BGR frame of shape (h, w, 3); same (w, h, seed, t) -> same bytes."""
    rng = np.random.Generator(np.random.PCG64(seed * 1_000_003 + t))
    yy = np.arange(h, dtype=np.float32)[:, None]
    xx = np.arange(w, dtype=np.float32)[None, :]
    hy = 0.5 * h                                  # horizon row
    vx = 0.5 * w + 0.03 * w * np.sin(2 * np.pi * t / 200.0)  # vanishing point x

    img = np.empty((h, w, 3), np.int16)
    # sky: vertical gradient, bluish
    sky = np.clip(200 - 80 * (yy / max(hy, 1)), 0, 255)
    img[..., 0] = np.broadcast_to(sky + 30, (h, w))
    img[..., 1] = np.broadcast_to(sky, (h, w))
    img[..., 2] = np.broadcast_to(sky - 20, (h, w))
    # asphalt below the horizon
    road = yy >= hy
    base = np.where(road, 90, 0).astype(np.int16)
    for c in range(3):
        img[..., c] = np.where(road, base, img[..., c])

    depth = np.clip((yy - hy) / max(h - hy, 1), 0, 1)       # 0 at horizon, 1 at bottom
    half_w = 1.0 + 0.012 * w * depth                         # marking half width
    # left marking: solid white, bottom at 18% of width
    xl = vx + (0.18 * w - vx) * depth
    left = road & (np.abs(xx - xl) <= half_w)
    # right marking: dashed yellow, bottom at 84% of width
    xr = vx + (0.84 * w - vx) * depth
    phase = (depth * 9.0 + t * 0.05) % 1.0
    right = road & (np.abs(xx - xr) <= half_w) & (phase < 0.6)
    lw = 235 + (_noise(rng, h, w) >> 1)
    for c, v in enumerate((40, 200, 230)):
        img[..., c] = np.where(right, v, img[..., c])
    for c in range(3):
        img[..., c] = np.where(left, lw, img[..., c])
    # guard rail: a bright, slightly tilted band on the left above the road edge
    gy = hy + 0.10 * h + 0.04 * (xx - 0.0) * (h / max(w, 1))
    rail = (np.abs(yy - gy) <= max(1.0, 0.004 * h)) & (xx < 0.30 * w)
    for c in range(3):
        img[..., c] = np.where(rail, 170, img[..., c])
    # shadow band across the road
    if shadow:
        sy = hy + (0.55 + 0.1 * np.sin(t / 37.0)) * (h - hy)
        band = (np.abs(yy - sy) <= 0.03 * h) & road
        for c in range(3):
            img[..., c] = np.where(band, (img[..., c] * 3) // 5, img[..., c])
    n = _noise(rng, h, w)
    img += n[..., None]
    return np.clip(img, 0, 255).astype(np.uint8)


def synth_clip(w, h, seed, nframes, t0=0):
    """(nframes, h, w, 3) uint8: consecutive frames of one clip."""
    out = np.empty((nframes, h, w, 3), np.uint8)
    for i in range(nframes):
        out[i] = synth_road(w, h, seed, t0 + i)
    return out


def adversarial_frames(w, h, seed=7):
    """Frames that stress borders, thresholds and dense edge maps."""
    rng = np.random.Generator(np.random.PCG64(seed))
    frames = {}
    frames["noise"] = rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8)
    frames["black"] = np.zeros((h, w, 3), np.uint8)
    frames["white"] = np.full((h, w, 3), 255, np.uint8)
    for p in (1, 2, 3):
        yy, xx = np.mgrid[0:h, 0:w]
        cb = ((((yy // p) + (xx // p)) & 1) * 255).astype(np.uint8)
        frames["checker%d" % p] = np.repeat(cb[..., None], 3, axis=2)
    imp = np.zeros((h, w, 3), np.uint8)
    imp[0, :, :] = 255
    imp[-1, ::2, :] = 255
    imp[::3, 0, :] = 255
    imp[:, -1, :] = 255
    frames["border_impulses"] = imp
    # smooth gradients + blocks: many weak edges chained to few strong ones
    g = (np.add.outer(np.arange(h) * 3, np.arange(w) * 2) % 256).astype(np.uint8)
    g[h // 4: h // 2, w // 4: w // 2] = 255
    frames["ramps"] = np.repeat(g[..., None], 3, axis=2)
    return frames
