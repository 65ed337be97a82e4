"""Independent cross-checks of the oracle's OpenCV-primitive restatements against the PUBLISHED definitions of the
algorithms, written from scratch in numpy / torch / scipy (none of which shares code with the oracle):

  * FAST-9/16 (Rosten & Drummond): corner <=> 9 contiguous circle pixels all brighter than v+t or all darker than v-t;
    OpenCV's score = the largest threshold at which the pixel is still a corner; 3x3 non-maximum suppression.
  * bilinear resize with half-pixel centres: the fixed-point result must be within 1 grey level of the float definition
    (torch.nn.functional.interpolate, align_corners=False).
  * 7x7 Gaussian, sigma 2, reflect-101 borders: within 1 grey level of scipy.ndimage float correlation ('mirror').
  * fastAtan2: within 0.3 degrees of atan2 (OpenCV's documented accuracy); cosf/sinf within 1 ulp of float64 libm.

These do not pin the oracle to OpenCV bit for bit (no OpenCV exists here; DESIGN.md "parity unpinned"); they pin it to
the algorithm definitions, which is what the restatement must at least satisfy."""
import math

import numpy as np
import pytest

from oracle import pyoracle as po

CIRCLE = [(0, 3), (1, 3), (2, 2), (3, 1), (3, 0), (3, -1), (2, -2), (1, -3), (0, -3), (-1, -3), (-2, -2), (-3, -1), (-3, 0),
          (-3, 1), (-2, 2), (-1, 3)]


def _is_corner(img, x, y, t):
    v = int(img[y, x])
    ring = [int(img[y + dy, x + dx]) for dx, dy in CIRCLE]
    for pol in (1, -1):
        flags = [(p - v) * pol > t for p in ring]
        ext = flags + flags
        run = best = 0
        for f in ext:
            run = run + 1 if f else 0
            best = max(best, run)
        if best >= 9:
            return True
    return False


def _naive_fast(img, t, nms=True):
    h, w = img.shape
    score = np.zeros((h, w), np.int32)
    for y in range(3, h - 3):
        for x in range(3, w - 3):
            if _is_corner(img, x, y, t):
                s = t
                while s < 255 and _is_corner(img, x, y, s + 1):
                    s += 1
                score[y, x] = s
    out = []
    for y in range(3, h - 3):
        for x in range(3, w - 3):
            s = score[y, x]
            if s == 0:
                continue
            if nms:
                nb = score[y - 1:y + 2, x - 1:x + 2].copy()
                nb[1, 1] = -1
                if not (s > nb.max()):
                    continue
            out.append((x, y, s))
    return out


@pytest.mark.parametrize("seed,t", [(0, 20), (1, 7), (2, 40), (3, 1)])
def test_fast_equals_definition(seed, t):
    rng = np.random.default_rng(seed)
    img = rng.integers(0, 256, (26, 31)).astype(np.uint8)
    img[5:18, 6:20] = rng.integers(100, 140, (13, 14))           # a smoother patch: fewer, stronger corners
    img[10:14, 10:14] = 250
    for nms in (True, False):
        got = po.fast(img, t, nms)
        want = _naive_fast(img, t, nms)
        if nms:   # order: row-major; response = the score
            assert [(int(k["x"]), int(k["y"]), int(k["response"])) for k in got] == [(x, y, int(s)) for x, y, s in want]
        else:     # without NMS OpenCV emits the corners without computing a score (response 0)
            assert [(int(k["x"]), int(k["y"])) for k in got] == [(x, y) for x, y, _ in want]


def test_resize_within_one_level_of_float_bilinear():
    import torch
    rng = np.random.default_rng(5)
    for (h, w, dh, dw) in [(480, 640, 400, 533), (400, 533, 333, 444), (134, 179, 112, 149), (97, 131, 64, 88)]:
        img = rng.integers(0, 256, (h, w)).astype(np.uint8)
        got = po.resize_linear(img, dw, dh).astype(np.float64)
        ref = torch.nn.functional.interpolate(torch.from_numpy(img)[None, None].double(), size=(dh, dw), mode="bilinear",
                                              align_corners=False, antialias=False)[0, 0].numpy()
        assert np.abs(got - ref).max() <= 1.0
        assert abs((got - ref).mean()) < 0.3                       # the two truncating shifts bias the result slightly downwards


def test_gaussian_blur_within_one_level_of_float_gaussian():
    from scipy import ndimage
    rng = np.random.default_rng(6)
    img = rng.integers(0, 256, (90, 123)).astype(np.uint8)
    x = np.arange(-3, 4)
    k = np.exp(-x * x / (2 * 2.0 * 2.0)); k /= k.sum()
    kq = po.gaussian_kernel7().astype(np.float64)
    assert kq.tolist() == [18, 34, 48, 56, 48, 34, 18] and kq.sum() == 256 and np.abs(kq - k * 256).max() < 1.0   # 8.8 quantisation
    got = po.gaussian_blur7(img).astype(np.float64)
    # exact separable correlation with the quantised kernel, reflect-101 ('mirror'), one final rounding
    refq = ndimage.correlate1d(ndimage.correlate1d(img.astype(np.float64), kq / 256, axis=1, mode="mirror"), kq / 256, axis=0, mode="mirror")
    assert np.abs(got - refq).max() <= 0.5 + 1e-9
    # and against the true float Gaussian: quantisation moves a pixel by at most a few levels, unbiased
    ref = ndimage.correlate1d(ndimage.correlate1d(img.astype(np.float64), k, axis=1, mode="mirror"), k, axis=0, mode="mirror")
    assert np.abs(got - ref).max() <= 2.0 and abs((got - ref).mean()) < 0.1


def test_fast_atan2_accuracy_and_trig_ulp():
    rng = np.random.default_rng(7)
    for _ in range(4000):
        y, x = (float(np.float32(v)) for v in rng.uniform(-3e6, 3e6, 2))
        a = po.fast_atan2(y, x)
        ref = math.degrees(math.atan2(y, x)) % 360.0
        d = abs(a - ref)
        assert min(d, 360 - d) < 0.3
    for deg in np.linspace(0, 360, 3001, dtype=np.float32):
        c, s = po.cos_sin_deg(float(deg))
        r = np.float64(np.float32(deg) * np.float32(math.pi / 180.0))
        for got, ref in ((c, math.cos(r)), (s, math.sin(r))):
            ulp = np.spacing(np.float32(abs(ref))) if ref != 0 else np.float32(1e-45)
            assert abs(np.float64(got) - ref) <= 1.0 * float(ulp) + 1e-12


def test_ic_angle_and_descriptor_follow_their_definitions():
    """IC_Angle = atan2(m01, m10) over the radius-15 disc; descriptor bit k = I(p_2k rotated) < I(p_2k+1 rotated)
    (Rublee et al.), evaluated with an independent numpy implementation on one keypoint of a real extraction."""
    from orb_slam3_modified_amd import synth
    img = synth.make_stream(1, 240, 320)[0]
    ora = po.OracleExtractor(300, 1.2, 1, 20, 7)
    kps, desc, _ = ora.extract(img, (0, 0))
    umax = [15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3]
    pat = po.pattern().reshape(256, 4).astype(np.int64)
    blurred = ora.level(0, blurred=True)
    for i in range(0, len(kps), max(1, len(kps) // 12)):
        x, y = int(kps["x"][i]), int(kps["y"][i])
        m10 = m01 = 0
        for v in range(-15, 16):
            for u in range(-umax[abs(v)], umax[abs(v)] + 1):
                m10 += u * int(img[y + v, x + u]); m01 += v * int(img[y + v, x + u])
        ref = math.degrees(math.atan2(m01, m10)) % 360.0
        d = abs(float(kps["angle"][i]) - ref)
        assert min(d, 360 - d) < 0.3
        ang = np.float32(kps["angle"][i]) * np.float32(math.pi / 180.0)
        a, b = np.float32(math.cos(ang)), np.float32(math.sin(ang))
        bits = []
        for k in range(256):
            x0, y0, x1, y1 = pat[k]
            def tap(px, py):
                ry = int(np.rint(np.float32(px) * b + np.float32(py) * a)); rx = int(np.rint(np.float32(px) * a - np.float32(py) * b))
                return int(blurred[y + ry, x + rx])
            bits.append(1 if tap(x0, y0) < tap(x1, y1) else 0)
        got = np.unpackbits(desc[i], bitorder="little")
        assert (got != np.array(bits)).sum() <= 2                  # float64-vs-float32 cos/sin may flip a rounding-edge tap
