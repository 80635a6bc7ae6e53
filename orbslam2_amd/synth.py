"""Seeded synthetic inputs for the ORB front-end.

No dataset is reachable from the build or the GPU box, so every test / bench
input is regenerated from a seed: a textured scene of rectangles and discs on
four fronto-parallel depth layers, rendered at 4x horizontal super-sampling so
the layer disparities {3.25, 11.5, 27.75, 61.0} px are exact, box-filtered down,
plus independent N(0, 3^2) sensor noise per camera.
"""
from __future__ import annotations

import numpy as np

LAYER_DISPARITY = (3.25, 11.5, 27.75, 61.0)
SS = 4  # horizontal super-sampling


def _paint(canvas, kind, x0, y0, a, b, grey, gx=0.0, gy=0.0):
    """Paint one shape (value = grey + gx*dx + gy*dy, a shading ramp that moves with the shape)."""
    h, w = canvas.shape
    if kind == 0:  # rectangle: a = width (hi-res), b = height
        xa, xb = max(x0, 0), min(x0 + a, w)
        ya, yb = max(y0, 0), min(y0 + b, h)
        if xa < xb and ya < yb:
            yy = (np.arange(ya, yb, dtype=np.float32)[:, None] - y0)
            xx = (np.arange(xa, xb, dtype=np.float32)[None, :] - x0) / SS
            canvas[ya:yb, xa:xb] = grey + gx * xx + gy * yy
    else:  # disc, radius a (low-res px); x is hi-res
        r = a
        ya, yb = max(y0 - r, 0), min(y0 + r + 1, h)
        xa, xb = max(x0 - r * SS, 0), min(x0 + r * SS + 1, w)
        if xa < xb and ya < yb:
            yy = np.arange(ya, yb, dtype=np.float32)[:, None] - y0
            xx = (np.arange(xa, xb, dtype=np.float32)[None, :] - x0) / SS
            m = (xx * xx + yy * yy) <= r * r
            canvas[ya:yb, xa:xb][m] = (grey + gx * xx + gy * yy)[m]


def _dense_texture(rng, wc, height, n_rect, n_disc):
    """One dense, corner-rich texture canvas (hi-res in x): smooth background + shaded shapes."""
    n = n_rect + n_disc
    kind = np.concatenate([np.zeros(n_rect, np.int32), np.ones(n_disc, np.int32)])
    x = rng.integers(-40 * SS, wc, n)
    y = rng.integers(-40, height, n)
    a = np.where(kind == 0, rng.integers(4, 41, n) * SS, rng.integers(2, 13, n))
    b = rng.integers(4, 41, n)
    grey = rng.integers(0, 256, n)
    gx = rng.uniform(-3.0, 3.0, n)
    gy = rng.uniform(-3.0, 3.0, n)
    order = rng.permutation(n)
    xs_hi = np.arange(wc, dtype=np.float32)[None, :] / SS
    ys_lo = np.arange(height, dtype=np.float32)[:, None]
    t = (128.0 + 50.0 * np.sin(xs_hi / 17.0) * np.cos(ys_lo / 11.0)).astype(np.float32)
    for i in order:
        _paint(t, kind[i], int(x[i]), int(y[i]), int(a[i]), int(b[i]), float(grey[i]), gx[i], gy[i])
    return np.clip(t, 0.0, 255.0)


def stereo_pair(width: int, height: int, seed: int = 1234, n_rect: int = 6000, n_disc: int = 3000,
                scale_shapes: bool = True, with_depth: bool = False, bf: float = 386.1448):
    """Returns (left, right[, depth]): uint8 images of size height x width (+ float32 depth, metres).

    Layer 0 (disparity 3.25 px) is a dense textured backdrop; layers 1..3 are a few large
    textured billboards at 11.5 / 27.75 / 61.0 px painted far -> near.
    """
    rng = np.random.Generator(np.random.PCG64(seed))
    if scale_shapes:
        # keep shape density constant relative to the KITTI frame the counts were chosen for
        f = (width * height) / float(1241 * 376)
        n_rect = max(8, int(n_rect * f))
        n_disc = max(4, int(n_disc * f))
    margin = (int(np.ceil(max(LAYER_DISPARITY))) + 64) * SS
    wc = width * SS + margin
    tex = _dense_texture(rng, wc, height, n_rect, n_disc)
    cl = np.empty((height, wc), np.float32)
    cr = np.empty((height, wc), np.float32)
    dl = np.zeros((height, wc), np.float32)
    for k, d in enumerate(LAYER_DISPARITY):
        tk = np.roll(tex, k * 1237 * SS, axis=1)
        if k & 1:
            tk = tk[::-1]
        sh = int(round(d * SS))
        if k == 0:
            mask = np.ones((height, wc), bool)
        else:
            mask = np.zeros((height, wc), bool)
            for _ in range(4):
                bw = int(rng.integers(width // 10, max(width // 4, width // 10 + 1))) * SS
                bh = int(rng.integers(height // 6, max(height // 2, height // 6 + 1)))
                bx = int(rng.integers(0, max(1, width * SS - bw // 2)))
                by = int(rng.integers(0, max(1, height - bh // 2)))
                mask[by:by + bh, bx:bx + bw] = True
        cl[mask] = tk[mask]
        dl[mask] = bf / d
        mr = np.roll(mask, -sh, axis=1)
        cr[mr] = np.roll(tk, -sh, axis=1)[mr]

    def down(c):
        return c[:, : width * SS].reshape(height, width, SS).mean(axis=2)

    left = down(cl) + rng.normal(0.0, 3.0, (height, width))
    right = down(cr) + rng.normal(0.0, 3.0, (height, width))
    left = np.clip(np.rint(left), 0, 255).astype(np.uint8)
    right = np.clip(np.rint(right), 0, 255).astype(np.uint8)
    if with_depth:
        depth = dl[:, 0: width * SS: SS].copy()
        holes = rng.random((height, width)) < 0.05
        depth[holes] = 0.0
        return left, right, depth
    return left, right


def mono_image(width: int, height: int, seed: int = 1234):
    return stereo_pair(width, height, seed)[0]
