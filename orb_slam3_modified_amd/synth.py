"""Deterministic synthetic camera streams (stand-ins for EuRoC / TUM-VI).

The datasets the reference is benchmarked on are not available offline, so
tests and bench.py use seeded synthetic frames with the properties the path needs:
  * plenty of FAST corners (>= 3x the per-level quota on the fine levels, so the quadtree's
    sorted-expansion branch src/ORBextractor.cc:689-753 runs),
  * a near-flat quarter of the image (cells where the iniThFAST pass finds nothing and the
    minThFAST retry of src/ORBextractor.cc:843-846 fires),
  * frame t is frame 0 translated by (1.5 t, 0.5 t) px (bilinear) plus fresh sensor noise.

Pure numpy; the same generator is used on the build container and on the GPU box.
"""
from __future__ import annotations

import numpy as np

DEFAULT_SEED = 20260925


def _value_noise(rng: np.random.Generator, h: int, w: int, cell: int, amp: float) -> np.ndarray:
    gh, gw = h // cell + 2, w // cell + 2
    g = rng.uniform(-1.0, 1.0, size=(gh, gw)).astype(np.float32)
    ys = np.arange(h, dtype=np.float32) / cell
    xs = np.arange(w, dtype=np.float32) / cell
    y0 = np.floor(ys).astype(np.int64)
    x0 = np.floor(xs).astype(np.int64)
    fy = (ys - y0)[:, None]
    fx = (xs - x0)[None, :]
    a = g[y0][:, x0]
    b = g[y0][:, x0 + 1]
    c = g[y0 + 1][:, x0]
    d = g[y0 + 1][:, x0 + 1]
    return amp * ((a * (1 - fx) + b * fx) * (1 - fy) + (c * (1 - fx) + d * fx) * fy)


def _draw_quads(rng: np.random.Generator, canvas: np.ndarray, n: int) -> None:
    h, w = canvas.shape
    for _ in range(n):
        cx, cy = rng.uniform(0, w), rng.uniform(0, h)
        sx, sy = rng.uniform(8, 80), rng.uniform(8, 80)
        th = rng.uniform(0, np.pi)
        val = rng.uniform(0, 255)
        r = int(np.ceil(0.75 * (sx + sy))) + 1
        x0, x1 = max(int(cx) - r, 0), min(int(cx) + r + 1, w)
        y0, y1 = max(int(cy) - r, 0), min(int(cy) + r + 1, h)
        if x0 >= x1 or y0 >= y1:
            continue
        yy, xx = np.mgrid[y0:y1, x0:x1]
        dx, dy = xx - cx, yy - cy
        u = dx * np.cos(th) + dy * np.sin(th)
        v = -dx * np.sin(th) + dy * np.cos(th)
        m = (np.abs(u) <= sx / 2) & (np.abs(v) <= sy / 2)
        canvas[y0:y1, x0:x1][m] = val


def make_canvas(height: int, width: int, seed: int = DEFAULT_SEED, max_shift: int = 0) -> np.ndarray:
    """Static float32 scene, larger than the frame by the camera travel."""
    rng = np.random.default_rng(seed)
    H = height + int(np.ceil(0.5 * max_shift)) + 2
    W = width + int(np.ceil(1.5 * max_shift)) + 2
    canvas = np.full((H, W), 118.0, dtype=np.float32)
    for cell, amp in ((64, 48.0), (32, 24.0), (16, 12.0), (8, 6.0)):
        canvas += _value_noise(rng, H, W, cell, amp)
    nquads = int(round(300 * (H * W) / (480.0 * 640.0)))
    _draw_quads(rng, canvas, nquads)
    # near-flat quarter of the scene (sigma ~1.5): forces the minThFAST retry
    fh, fw = H // 2, W // 2
    fy0, fx0 = int(rng.integers(0, H - fh)), int(rng.integers(0, W - fw))
    canvas[fy0:fy0 + fh, fx0:fx0 + fw] = 96.0 + rng.normal(0, 1.5, size=(fh, fw)).astype(np.float32)
    return canvas


def frame_from_canvas(canvas: np.ndarray, height: int, width: int, t: int, seed: int = DEFAULT_SEED) -> np.ndarray:
    ox, oy = 1.5 * t, 0.5 * t
    ix, iy = int(np.floor(ox)), int(np.floor(oy))
    fx, fy = np.float32(ox - ix), np.float32(oy - iy)
    a = canvas[iy:iy + height, ix:ix + width]
    b = canvas[iy:iy + height, ix + 1:ix + width + 1]
    c = canvas[iy + 1:iy + height + 1, ix:ix + width]
    d = canvas[iy + 1:iy + height + 1, ix + 1:ix + width + 1]
    img = (a * (1 - fx) + b * fx) * (1 - fy) + (c * (1 - fx) + d * fx) * fy
    rng = np.random.default_rng(seed + 1 + t)
    img = img + rng.normal(0, 2.0, size=img.shape).astype(np.float32)
    return np.clip(np.rint(img), 0, 255).astype(np.uint8)


def make_stream(n_frames: int, height: int = 480, width: int = 640, seed: int = DEFAULT_SEED) -> np.ndarray:
    """`n_frames` frames of one synthetic camera: uint8 array [n, height, width]."""
    canvas = make_canvas(height, width, seed, max_shift=max(n_frames - 1, 0))
    out = np.empty((n_frames, height, width), dtype=np.uint8)
    for t in range(n_frames):
        out[t] = frame_from_canvas(canvas, height, width, t, seed)
    return out


def make_camera_streams(n_cams: int, n_frames: int, height: int = 480, width: int = 640,
                        seed: int = DEFAULT_SEED) -> np.ndarray:
    """S-8cam: independent streams with seeds +1000*cam. uint8 [n_cams, n_frames, height, width]."""
    return np.stack([make_stream(n_frames, height, width, seed + 1000 * c) for c in range(n_cams)])
