"""ctypes binding of the CPU oracle (oracle/liborb_oracle.so).  TEST INFRASTRUCTURE ONLY:
imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg — never by the
product package `orb_slam3_modified_amd`."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liborb_oracle.so")

KP_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"), ("response", "<f4"),
                     ("octave", "<i4"), ("class_id", "<i4")])
assert KP_DTYPE.itemsize == 28


def build(force: bool = False) -> str:
    srcs = [os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith((".cpp", ".inc", "Makefile"))]
    stale = (not os.path.exists(_LIB_PATH)) or any(os.path.getmtime(s) > os.path.getmtime(_LIB_PATH) for s in srcs)
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _LIB_PATH


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = C.CDLL(_LIB_PATH)
        vp, ip, fp, u8p = C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_float), C.c_void_p
        L.orbo_create.restype = vp
        L.orbo_create.argtypes = [C.c_int, C.c_float, C.c_int, C.c_int, C.c_int]
        L.orbo_destroy.argtypes = [vp]
        L.orbo_extract.restype = C.c_int
        L.orbo_extract.argtypes = [vp, u8p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp, u8p, C.c_int, ip, ip]
        L.orbo_tables.argtypes = [vp, vp, vp, vp, vp, vp, vp]
        L.orbo_level_size.argtypes = [vp, C.c_int, ip, ip]
        L.orbo_level_copy.restype = C.c_int
        L.orbo_level_copy.argtypes = [vp, C.c_int, C.c_int, u8p, C.c_int]
        L.orbo_level_keypoints.restype = C.c_int
        L.orbo_level_keypoints.argtypes = [vp, C.c_int, C.c_int, vp, C.c_int]
        L.orbo_sorted_phase_count.argtypes = [vp]
        L.orbo_resize_linear.argtypes = [u8p, C.c_int, C.c_int, C.c_int, u8p, C.c_int, C.c_int, C.c_int]
        L.orbo_fast.restype = C.c_int
        L.orbo_fast.argtypes = [u8p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp, C.c_int]
        L.orbo_gaussian_blur7.argtypes = [u8p, C.c_int, C.c_int, C.c_int, u8p, C.c_int]
        L.orbo_gaussian_kernel7.argtypes = [vp]
        L.orbo_set_gauss_variant.restype = C.c_int
        L.orbo_set_gauss_variant.argtypes = [C.c_int, C.c_int]
        L.orbo_get_gauss_variant.argtypes = [ip, ip]
        L.orbo_set_gauss_tail.restype = C.c_int
        L.orbo_set_gauss_tail.argtypes = [C.c_int]
        L.orbo_get_gauss_tail.restype = C.c_int
        L.orbo_set_atan_fma.restype = C.c_int
        L.orbo_set_atan_fma.argtypes = [C.c_int]
        L.orbo_get_atan_fma.restype = C.c_int
        L.orbo_set_brief_fma.restype = C.c_int
        L.orbo_set_brief_fma.argtypes = [C.c_int]
        L.orbo_get_brief_fma.restype = C.c_int
        L.orbo_fast_atan2.restype = C.c_float
        L.orbo_fast_atan2.argtypes = [C.c_float, C.c_float]
        L.orbo_cos_sin_deg.argtypes = [C.c_float, fp, fp]
        L.orbo_distribute.restype = C.c_int
        L.orbo_distribute.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp, C.c_int]
        L.orbo_trig_hash.restype = C.c_uint64
        L.orbo_trig_hash.argtypes = [C.c_uint32, C.c_uint32]
        L.orbo_atan_hash.restype = C.c_uint64
        L.orbo_atan_hash.argtypes = [C.c_uint32, C.c_uint32]
        L.orbo_pattern.restype = C.POINTER(C.c_int8)
        L.orbo_brief_hash.restype = C.c_uint64
        L.orbo_brief_hash.argtypes = [C.c_uint32, C.c_uint32, C.POINTER(C.c_uint64)]
        L.orbo_rot_tap.argtypes = [C.c_int, C.c_int, C.c_float, C.c_float, ip, ip]
        L.orbo_rot_probe_hash.restype = C.c_uint64
        L.orbo_rot_probe_hash.argtypes = [C.c_uint32, C.POINTER(C.c_uint64)]
        _lib = L
    return _lib


def _ptr(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


def cvt_color_to_gray(img: np.ndarray, rgb: bool) -> np.ndarray:
    """cv::cvtColor(COLOR_RGB2GRAY / BGR2GRAY / RGBA2GRAY / BGRA2GRAY) for CV_8U as OpenCV 4.x computes it (recalled, like
    the other OpenCV primitives — parity unpinned): 15-bit fixed point, (R*9798 + G*19235 + B*3735 + 2^14) >> 15."""
    assert img.dtype == np.uint8 and img.ndim == 3 and img.shape[2] in (3, 4)
    a = img.astype(np.int64)
    r, g, b = (a[..., 0], a[..., 1], a[..., 2]) if rgb else (a[..., 2], a[..., 1], a[..., 0])
    return ((r * 9798 + g * 19235 + b * 3735 + (1 << 14)) >> 15).astype(np.uint8)


class OracleExtractor:
    """Mirror of ORB_SLAM3::ORBextractor (include/ORBextractor.h:49-83) over the oracle."""

    def __init__(self, nfeatures=1000, scale_factor=1.2, nlevels=8, ini_th=20, min_th=7):
        self.nfeatures, self.nlevels = nfeatures, nlevels
        self._h = lib().orbo_create(nfeatures, scale_factor, nlevels, ini_th, min_th)
        self.cap = nfeatures + 35 * nlevels + 64   # a level returns up to max(quota + 3, 4 x root nodes) keypoints (<= 8 roots)

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orbo_destroy(self._h)
            self._h = None

    def tables(self):
        n = self.nlevels
        sc, isc, s2, is2 = (np.zeros(n, np.float32) for _ in range(4))
        quota, umax = np.zeros(n, np.int32), np.zeros(16, np.int32)
        lib().orbo_tables(self._h, _ptr(sc), _ptr(isc), _ptr(s2), _ptr(is2), _ptr(quota), _ptr(umax))
        return dict(scale=sc, inv_scale=isc, sigma2=s2, inv_sigma2=is2, quota=quota, umax=umax)

    def extract(self, img: np.ndarray, lapping=(0, 0)):
        """Returns (keypoints[KP_DTYPE], descriptors[n,32] u8, monoIndex)."""
        assert img.dtype == np.uint8 and img.ndim == 2
        img = np.ascontiguousarray(img)
        kps = np.zeros(self.cap, KP_DTYPE)
        desc = np.zeros((self.cap, 32), np.uint8)
        n, mono = C.c_int(0), C.c_int(0)
        rc = lib().orbo_extract(self._h, _ptr(img), img.shape[0], img.shape[1], img.strides[0], int(lapping[0]),
                                int(lapping[1]), _ptr(kps), _ptr(desc), self.cap, C.byref(n), C.byref(mono))
        if rc != 0:
            raise RuntimeError(f"oracle extract rc={rc} n={n.value}")
        return kps[:n.value].copy(), desc[:n.value].copy(), mono.value

    def level(self, level: int, blurred: bool = False) -> np.ndarray:
        w, h = C.c_int(0), C.c_int(0)
        assert lib().orbo_level_size(self._h, level, C.byref(w), C.byref(h)) == 0
        out = np.zeros((h.value, w.value), np.uint8)
        if lib().orbo_level_copy(self._h, level, int(blurred), _ptr(out), w.value) != 0:
            raise RuntimeError("level not available")
        return out

    def level_keypoints(self, level: int, stage: int) -> np.ndarray:
        n = lib().orbo_level_keypoints(self._h, level, stage, None, 0)
        out = np.zeros(max(n, 1), KP_DTYPE)
        lib().orbo_level_keypoints(self._h, level, stage, _ptr(out), n)
        return out[:n]

    def sorted_phase_count(self) -> int:
        return lib().orbo_sorted_phase_count(self._h)


def resize_linear(src: np.ndarray, dw: int, dh: int) -> np.ndarray:
    src = np.ascontiguousarray(src)
    dst = np.zeros((dh, dw), np.uint8)
    lib().orbo_resize_linear(_ptr(src), src.shape[1], src.shape[0], src.strides[0], _ptr(dst), dw, dh, dw)
    return dst


def cv_resize(src: np.ndarray, dw: int, dh: int) -> np.ndarray:
    """cv::resize(src, dst, Size(dw, dh)) (INTER_LINEAR default; an exact 2 x 2 downscale becomes INTER_AREA), src/System.cc:441-446."""
    src = np.ascontiguousarray(src)
    dst = np.zeros((dh, dw), np.uint8)
    lib().orbo_cv_resize(_ptr(src), src.shape[1], src.shape[0], src.strides[0], _ptr(dst), dw, dh, dw)
    return dst


def fast(img: np.ndarray, threshold: int, nms: bool = True) -> np.ndarray:
    img = np.ascontiguousarray(img)
    cap = img.size
    out = np.zeros(max(cap, 1), KP_DTYPE)
    n = lib().orbo_fast(_ptr(img), img.shape[1], img.shape[0], img.strides[0], threshold, int(nms), _ptr(out), cap)
    return out[:n]


def gaussian_blur7(img: np.ndarray) -> np.ndarray:
    img = np.ascontiguousarray(img)
    dst = np.zeros_like(img)
    lib().orbo_gaussian_blur7(_ptr(img), img.shape[1], img.shape[0], img.strides[0], _ptr(dst), dst.strides[0])
    return dst


class opencv_variant:
    """with opencv_variant(gauss_kernel, gauss_round, gauss_tail, atan_fma): ... — which OpenCV build's cv::GaussianBlur / cv::fastAtan2 the
    oracle restates inside the block (process-wide switches of liborb_oracle.so, so also of everything compiled over oracle/ref_shims);
    the previous variant is restored on exit.  Same meaning as the product's orbx_set_option names (include/orbx.h, INTEGRATION.md
    section 6): kernel 0 = {18,34,48,56,...} (OpenCV >= 4.5.1), 1 = {18,34,49,55,...} (3.x .. 4.5.0); round 0 = half up, 1 = ties to even
    (<= 3.4.1 SSE2 column pass), 2 = floor (3.4.2 .. 4.5.0 SIMD column pass under the 257 kernel); tail V = the last w mod V columns round
    half up; atan_fma 1 = fastAtan2's polynomial contracted into FMAs (OpenCV's AVX2 dispatch copy); brief_fma 1 = the pattern rotation of
    src/ORBextractor.cc:118-120 contracted into FMAs (the reference itself built with -march=native)."""

    def __init__(self, gauss_kernel: int = 0, gauss_round: int = 0, gauss_tail: int = 0, atan_fma: int = 0, brief_fma: int = 0):
        self.want = (int(gauss_kernel), int(gauss_round), int(gauss_tail), int(atan_fma), int(brief_fma))

    @staticmethod
    def _set(v):
        L = lib()
        if (L.orbo_set_gauss_variant(v[0], v[1]) != 0 or L.orbo_set_gauss_tail(v[2]) != 0 or L.orbo_set_atan_fma(v[3]) != 0
                or L.orbo_set_brief_fma(v[4]) != 0):
            raise ValueError(f"no such OpenCV variant: {v}")

    def __enter__(self):
        k, r = C.c_int(0), C.c_int(0)
        lib().orbo_get_gauss_variant(C.byref(k), C.byref(r))
        self.prev = (k.value, r.value, lib().orbo_get_gauss_tail(), lib().orbo_get_atan_fma(), lib().orbo_get_brief_fma())
        try:
            self._set(self.want)
        except ValueError:
            self._set(self.prev)
            raise
        return self

    def __exit__(self, *exc):
        self._set(self.prev)
        return False

    def options(self) -> dict:
        """the same variant as orbx_set_option name -> value pairs"""
        return dict(gauss_kernel=self.want[0], gauss_round=self.want[1], gauss_tail=self.want[2], atan_fma=self.want[3], brief_fma=self.want[4])


# (gauss_kernel, gauss_round, gauss_tail, atan_fma, brief_fma); the first is the default.  The named ones are INTEGRATION.md section 6's rows.
OPENCV_VARIANTS = {
    "opencv>=4.5.1": (0, 0, 0, 0, 0),
    "opencv>=4.5.1, avx2 dispatch, reference -march=native": (0, 0, 0, 1, 1),
    "opencv3.4.2-4.5.0 scalar": (1, 0, 0, 0, 0),
    "opencv3.4.2-4.5.0 simd8": (1, 2, 8, 0, 0),
    "opencv3.4.2-4.5.0 simd16, avx2 dispatch, reference -march=native": (1, 2, 16, 1, 1),
    "opencv<=3.4.1 sse2": (1, 1, 4, 0, 0),
    "diffused kernel, ties to even": (0, 1, 0, 0, 1),
    "diffused kernel, floor, tail 32": (0, 2, 32, 0, 0),
}


def gaussian_kernel7() -> np.ndarray:
    k = np.zeros(7, np.int32)
    lib().orbo_gaussian_kernel7(_ptr(k))
    return k


def fast_atan2(y: float, x: float) -> float:
    return float(lib().orbo_fast_atan2(y, x))


def cos_sin_deg(angle_deg: float):
    a, b = C.c_float(0), C.c_float(0)
    lib().orbo_cos_sin_deg(angle_deg, C.byref(a), C.byref(b))
    return a.value, b.value


def trig_hash(first_bits: int, count: int) -> int:
    """64-bit digest of (cosf, sinf)(angle * pi/180) over `count` consecutive float bit patterns (see orbx_debug_trig_hash)."""
    return int(lib().orbo_trig_hash(first_bits, count))


def brief_hash(first_bits: int, count: int):
    """(digest, n_diff): 64-bit digest of the 512 rotated pattern points over `count` consecutive float bit patterns of the angle (see
    orbx_debug_brief_hash), and how many points round differently under the other brief_fma setting."""
    nd = C.c_uint64(0)
    h = lib().orbo_brief_hash(first_bits, count, C.byref(nd))
    return int(h), int(nd.value)


def rot_probe_hash(n: int):
    """(digest, n_diff) over the operand sequence of tests/support/contract_probe.cpp under the current brief_fma setting."""
    nd = C.c_uint64(0)
    h = lib().orbo_rot_probe_hash(n, C.byref(nd))
    return int(h), int(nd.value)


def rot_tap(x: int, y: int, a: float, b: float):
    ry, rx = C.c_int(0), C.c_int(0)
    lib().orbo_rot_tap(x, y, a, b, C.byref(ry), C.byref(rx))
    return ry.value, rx.value


def atan_hash(seed: int, count: int) -> int:
    """64-bit digest of fastAtan2 over `count` pseudo-random moment pairs (see orbx_debug_atan_hash)."""
    return int(lib().orbo_atan_hash(seed, count))


def distribute(cand: np.ndarray, minX, maxX, minY, maxY, N) -> np.ndarray:
    cand = np.ascontiguousarray(cand)
    out = np.zeros(len(cand) + 8, KP_DTYPE)
    n = lib().orbo_distribute(_ptr(cand), len(cand), minX, maxX, minY, maxY, N, _ptr(out), len(out))
    return out[:n]


def pattern() -> np.ndarray:
    p = lib().orbo_pattern()
    return np.ctypeslib.as_array(p, shape=(1024,)).copy()


# ---- matcher / bag-of-words oracle (oracle/match_oracle.cpp) ------------------------------------------------
def _mlib():
    L = lib()
    if not getattr(L, "_mo_bound", False):
        vp = C.c_void_p
        L.mo_hamming.restype = C.c_int
        L.mo_hamming.argtypes = [vp, vp]
        L.mo_nn_csr.argtypes = [vp, C.c_int, vp, vp, vp, C.c_int, vp, vp, vp, vp, vp]
        L.mo_knn2.argtypes = [vp, C.c_int, vp, C.c_int, vp, vp]
        L.mo_voc_load.restype = vp
        L.mo_voc_load.argtypes = [C.c_char_p]
        L.mo_voc_free.argtypes = [vp]
        L.mo_voc_descend.argtypes = [vp, vp, C.c_int, C.c_int, vp, vp, vp]
        L.mo_voc_transform.restype = C.c_int
        L.mo_voc_transform.argtypes = [vp, vp, C.c_int, C.c_int, vp, vp, vp, vp, C.POINTER(C.c_int)]
        L.mo_score_l1.restype = C.c_double
        L.mo_score_l1.argtypes = [vp, vp, C.c_int, vp, vp, C.c_int]
        f32 = C.c_float
        L.mo_features_in_area.restype = C.c_int
        L.mo_features_in_area.argtypes = [vp, C.c_int, f32, f32, f32, f32, vp, vp, vp, vp, vp, C.c_int, vp, vp, C.c_int]
        L.mo_search_for_initialization.restype = C.c_int
        L.mo_search_for_initialization.argtypes = [vp, vp, C.c_int, vp, vp, C.c_int, f32, f32, f32, f32, vp, C.c_int, f32, C.c_int, vp]
        L.mo_stereo_matches.restype = C.c_int
        L.mo_stereo_matches.argtypes = [vp, vp, C.c_int, vp, vp, C.c_int, vp, vp, vp, vp, vp, vp, vp, f32, f32, vp, vp]
        L._mo_bound = True
    return L


def hamming(a, b) -> int:
    a = np.ascontiguousarray(a, np.uint8); b = np.ascontiguousarray(b, np.uint8)
    return int(_mlib().mo_hamming(_ptr(a), _ptr(b)))


def nn_csr(q, t, row_ptr, cand, last_wins=False):
    q = np.ascontiguousarray(q, np.uint8); t = np.ascontiguousarray(t, np.uint8)
    rp = np.ascontiguousarray(row_ptr, np.int32); cd = np.ascontiguousarray(cand, np.int32)
    nq = len(q)
    bi, bd, si, sd = (np.zeros(max(nq, 1), np.int32) for _ in range(4))
    do = np.zeros(max(len(cd), 1), np.int32)
    _mlib().mo_nn_csr(_ptr(q), nq, _ptr(t), _ptr(rp), _ptr(cd), int(last_wins), _ptr(bi), _ptr(bd), _ptr(si), _ptr(sd), _ptr(do))
    return bi[:nq], bd[:nq], si[:nq], sd[:nq], do[:len(cd)]


def features_in_area(kps, bounds, qx, qy, qr, qmin, qmax):
    """Frame::GetFeaturesInArea (src/Frame.cc:657-723) over the frame grid of `kps` for a list of queries -> (row_ptr, cand)."""
    k = np.ascontiguousarray(kps)
    qx, qy, qr = (np.ascontiguousarray(v, np.float32) for v in (qx, qy, qr))
    qmin, qmax = (np.ascontiguousarray(v, np.int32) for v in (qmin, qmax))
    nq = len(qx)
    cap = max(len(k) * max(nq, 1), 1)
    rp = np.zeros(nq + 1, np.int32)
    cand = np.zeros(cap, np.int32)
    nnz = _mlib().mo_features_in_area(_ptr(k), len(k), *[float(b) for b in bounds], _ptr(qx), _ptr(qy), _ptr(qr), _ptr(qmin), _ptr(qmax),
                                      nq, _ptr(rp), _ptr(cand), cap)
    assert nnz >= 0
    return rp, cand[:nnz].copy()


def undistort_keypoints(kps, fx, fy, cx, cy, dist):
    """Frame::UndistortKeyPoints (src/Frame.cc:747-780) -> undistorted copy of the keypoints (only pt changes)."""
    k = np.ascontiguousarray(kps).copy()
    dist = np.ascontiguousarray(dist, np.float32)
    if len(k) == 0 or dist[0] == 0.0:
        return k
    xy = np.ascontiguousarray(np.stack([k["x"], k["y"]], 1), np.float32)
    out = np.zeros_like(xy)
    L = _mlib()
    L.mo_undistort_points.restype = None
    L.mo_undistort_points.argtypes = [C.c_void_p, C.c_int] + [C.c_float] * 4 + [C.c_void_p, C.c_int, C.c_void_p]
    L.mo_undistort_points(_ptr(xy), len(k), float(fx), float(fy), float(cx), float(cy), _ptr(dist), len(dist), _ptr(out))
    k["x"], k["y"] = out[:, 0], out[:, 1]
    return k


class _MoGrid(C.Structure):
    _fields_ = [("min_x", C.c_float), ("min_y", C.c_float), ("inv_w", C.c_float), ("inv_h", C.c_float),
                ("cell_start", C.c_void_p), ("cell_idx", C.c_void_p)]


def _mogrid(grid):
    cs = None if grid.get("cell_start") is None else np.ascontiguousarray(grid["cell_start"], np.int32)
    ci = None if cs is None else np.ascontiguousarray(np.concatenate([grid["cell_idx"], [0]]), np.int32)
    g = _MoGrid(float(grid["min_x"]), float(grid["min_y"]), float(grid["inv_w"]), float(grid["inv_h"]),
                None if cs is None else cs.ctypes.data, None if ci is None else ci.ctypes.data)
    return g, (cs, ci)


def assign_grid(kps, min_x, min_y, inv_w, inv_h):
    """Frame::AssignFeaturesToGrid / PosInGrid (src/Frame.cc:385-416, :725-735) -> (cell_start [64*48+1], cell_idx), numpy float32."""
    k = np.ascontiguousarray(kps)
    f32 = np.float32
    px = ((k["x"].astype(f32) - f32(min_x)).astype(f32) * f32(inv_w)).astype(f32)
    py = ((k["y"].astype(f32) - f32(min_y)).astype(f32) * f32(inv_h)).astype(f32)
    rnd = lambda v: np.where(v >= 0, np.floor(v.astype(np.float64) + 0.5), np.ceil(v.astype(np.float64) - 0.5)).astype(np.int64)  # C round()
    posx, posy = rnd(px), rnd(py)
    ok = (posx >= 0) & (posx < 64) & (posy >= 0) & (posy < 48)
    cell = np.where(ok, posx * 48 + posy, -1)
    order = np.argsort(cell, kind="stable")
    order = order[cell[order] >= 0]
    counts = np.bincount(cell[order], minlength=64 * 48)
    start = np.concatenate([[0], np.cumsum(counts)]).astype(np.int32)
    return start, order.astype(np.int32)


def window_search_grid(kps, desc, grid, qx, qy, qr, qmin, qmax, q_desc, kp_skip=None, kp_uright=None, q_xr=None):
    """The generic window primitive (mo_window_search_grid) -> dict like ORBmatcher.WindowSearchGrid."""
    k = np.ascontiguousarray(kps); d = np.ascontiguousarray(desc, np.uint8)
    qx, qy, qr = (np.ascontiguousarray(v, np.float32) for v in (qx, qy, qr))
    qmin, qmax = (np.ascontiguousarray(v, np.int32) for v in (qmin, qmax))
    qd = np.ascontiguousarray(q_desc, np.uint8)
    skip = None if kp_skip is None else np.ascontiguousarray(kp_skip, np.uint8)
    ur = None if kp_uright is None else np.ascontiguousarray(kp_uright, np.float32)
    xr = None if q_xr is None else np.ascontiguousarray(q_xr, np.float32)
    nq = len(qx)
    g, keep = _mogrid(grid)
    L = _mlib()
    vp = C.c_void_p
    L.mo_window_search_grid.restype = C.c_int
    L.mo_window_search_grid.argtypes = [vp, vp, C.c_int, vp] + [vp] * 9 + [C.c_int, vp, vp, vp, C.c_int, vp, vp, vp, vp]
    rp = np.zeros(nq + 1, np.int32)
    bi, bd, si, sd = (np.zeros(max(nq, 1), np.int32) for _ in range(4))
    cap = max(len(k) * max(nq, 1), 1)
    cand, dist = np.zeros(cap, np.int32), np.zeros(cap, np.int32)
    pp = lambda a: None if a is None else _ptr(a)
    nnz = L.mo_window_search_grid(_ptr(k), _ptr(d), len(k), C.addressof(g), pp(skip), pp(ur), _ptr(qx), _ptr(qy), _ptr(qr), _ptr(qmin),
                                  _ptr(qmax), _ptr(qd), pp(xr), nq, _ptr(rp), _ptr(cand), _ptr(dist), cap, _ptr(bi), _ptr(bd), _ptr(si), _ptr(sd))
    assert nnz >= 0
    return dict(row_ptr=rp, cand=cand[:nnz].copy(), dist=dist[:nnz].copy(), best_idx=bi[:nq], best_dist=bd[:nq], second_idx=si[:nq],
                second_dist=sd[:nq])


def window_nearest(kps, desc, grid, qx, qy, qr, qmin, qmax, q_desc, kp_uright=None, inv_level_sigma2=None, q_ur=None):
    """mo_window_nearest: arg-min with Fuse's optional reprojection gate (src/ORBmatcher.cc:1262-1309) -> (best_idx, best_dist)."""
    k = np.ascontiguousarray(kps); d = np.ascontiguousarray(desc, np.uint8)
    qx, qy, qr = (np.ascontiguousarray(v, np.float32) for v in (qx, qy, qr))
    qmin, qmax = (np.ascontiguousarray(v, np.int32) for v in (qmin, qmax))
    qd = np.ascontiguousarray(q_desc, np.uint8)
    ur = None if kp_uright is None else np.ascontiguousarray(kp_uright, np.float32)
    sig = None if inv_level_sigma2 is None else np.ascontiguousarray(inv_level_sigma2, np.float32)
    qu = None if q_ur is None else np.ascontiguousarray(q_ur, np.float32)
    nq = len(qx)
    g, keep = _mogrid(grid)
    L = _mlib()
    vp = C.c_void_p
    L.mo_window_nearest.restype = None
    L.mo_window_nearest.argtypes = [vp, vp, C.c_int, vp] + [vp] * 9 + [C.c_int, vp, vp]
    bi, bd = np.zeros(max(nq, 1), np.int32), np.zeros(max(nq, 1), np.int32)
    pp = lambda a: None if a is None else _ptr(a)
    L.mo_window_nearest(_ptr(k), _ptr(d), len(k), C.addressof(g), pp(ur), pp(sig), _ptr(qx), _ptr(qy), _ptr(qr), _ptr(qmin), _ptr(qmax), pp(qu),
                        _ptr(qd), nq, _ptr(bi), _ptr(bd))
    return bi[:nq], bd[:nq]


def search_for_initialization(kps1, desc1, kps2, desc2, bounds, prev_xy, window_size=100, nnratio=0.9, check_ori=True):
    """ORBmatcher::SearchForInitialization (src/ORBmatcher.cc:648-763) -> (nmatches, vnMatches12, updated vbPrevMatched)."""
    k1, k2 = np.ascontiguousarray(kps1), np.ascontiguousarray(kps2)
    d1, d2 = np.ascontiguousarray(desc1, np.uint8), np.ascontiguousarray(desc2, np.uint8)
    prev = np.ascontiguousarray(prev_xy, np.float32).copy()
    m12 = np.zeros(max(len(k1), 1), np.int32)
    n = _mlib().mo_search_for_initialization(_ptr(k1), _ptr(d1), len(k1), _ptr(k2), _ptr(d2), len(k2), *[float(b) for b in bounds],
                                             _ptr(prev), int(window_size), float(nnratio), int(check_ori), _ptr(m12))
    return int(n), m12[:len(k1)].copy(), prev


def search_by_projection(kps, desc, bounds, scale_factors, kp_obs, mp, th=1.0, nnratio=0.8, u_right=None):
    """ORBmatcher::SearchByProjection(Frame&, vector<MapPoint*>&, th, ...) (src/ORBmatcher.cc:43-141, Nleft == -1) on
    flattened state -> (nmatches, kp_match, updated kp_obs)."""
    k = np.ascontiguousarray(kps)
    d = np.ascontiguousarray(desc, np.uint8)
    obs = np.ascontiguousarray(kp_obs, np.int32).copy()
    sf = np.ascontiguousarray(scale_factors, np.float32)
    skip = np.ascontiguousarray(1 - np.asarray(mp["in_view"], np.uint8), np.uint8)
    px, py, vc = (np.ascontiguousarray(mp[key], np.float32) for key in ("proj_x", "proj_y", "view_cos"))
    pxr = np.ascontiguousarray(mp["proj_xr"] if mp.get("proj_xr") is not None else np.zeros(len(px)), np.float32)
    lvl, mobs = (np.ascontiguousarray(mp[key], np.int32) for key in ("level", "obs"))
    md = np.ascontiguousarray(mp["desc"], np.uint8)
    ur = None if u_right is None else np.ascontiguousarray(u_right, np.float32)
    match = np.zeros(max(len(k), 1), np.int32)
    L = _mlib()
    L.mo_search_by_projection.restype = C.c_int
    L.mo_search_by_projection.argtypes = [C.c_void_p] * 4 + [C.c_int] + [C.c_float] * 4 + [C.c_void_p] * 9 + [C.c_int, C.c_float, C.c_float,
                                                                                                          C.c_void_p]
    n = L.mo_search_by_projection(_ptr(k), _ptr(d), _ptr(ur) if ur is not None else None, _ptr(obs), len(k), *[float(b) for b in bounds],
                                  _ptr(sf), _ptr(skip), _ptr(px), _ptr(py), _ptr(pxr), _ptr(vc), _ptr(lvl), _ptr(md), _ptr(mobs),
                                  len(px), float(th), float(nnratio), _ptr(match))
    return int(n), match[:len(k)].copy(), obs


def search_by_projection_last(kps, desc, bounds, scale_factors, kp_obs, lp, th, direction=0, check_ori=True, u_right=None, mbf=0.0):
    """ORBmatcher::SearchByProjection(CurrentFrame, LastFrame, th, bMono) (src/ORBmatcher.cc:1676-1885, Nleft == -1) after
    the projection step, on flattened state -> (nmatches, kp_match, updated kp_obs)."""
    k = np.ascontiguousarray(kps)
    d = np.ascontiguousarray(desc, np.uint8)
    obs = np.ascontiguousarray(kp_obs, np.int32).copy()
    sf = np.ascontiguousarray(scale_factors, np.float32)
    valid = np.ascontiguousarray(lp["valid"], np.uint8)
    u, v, invz, ang = (np.ascontiguousarray(lp[key], np.float32) for key in ("u", "v", "invz", "angle"))
    octv, lobs = (np.ascontiguousarray(lp[key], np.int32) for key in ("octave", "obs"))
    ld = np.ascontiguousarray(lp["desc"], np.uint8)
    ur = None if u_right is None else np.ascontiguousarray(u_right, np.float32)
    match = np.zeros(max(len(k), 1), np.int32)
    L = _mlib()
    vp, f32 = C.c_void_p, C.c_float
    L.mo_search_by_projection_last.restype = C.c_int
    L.mo_search_by_projection_last.argtypes = [vp] * 4 + [C.c_int] + [f32] * 4 + [vp, f32] + [vp] * 8 + [C.c_int, f32, C.c_int, C.c_int, vp]
    n = L.mo_search_by_projection_last(_ptr(k), _ptr(d), _ptr(ur) if ur is not None else None, _ptr(obs), len(k), *[float(b) for b in bounds],
                                       _ptr(sf), float(mbf), _ptr(valid), _ptr(u), _ptr(v), _ptr(invz), _ptr(octv), _ptr(ang), _ptr(ld),
                                       _ptr(lobs), len(u), float(th), int(direction), int(check_ori), _ptr(match))
    return int(n), match[:len(k)].copy(), obs


def search_by_bow(kf_desc, kf_angle, kf_valid, kf_fv, f_desc, f_angle, f_fv, nnratio=0.7, check_ori=True):
    """ORBmatcher::SearchByBoW(KeyFrame*, Frame&, ...) (src/ORBmatcher.cc:223-425, Nleft == -1).  kf_fv / f_fv: feature vectors as
    dict node -> list of feature indices.  Returns (nmatches, match_kf [nf])."""
    def flat(fv):
        nodes, idx = [], []
        for k in sorted(fv):
            for i in fv[k]:
                nodes.append(k); idx.append(i)
        return np.array(nodes, np.uint32), np.array(idx, np.uint32)
    kd, fd = np.ascontiguousarray(kf_desc, np.uint8), np.ascontiguousarray(f_desc, np.uint8)
    ka, fa = np.ascontiguousarray(kf_angle, np.float32), np.ascontiguousarray(f_angle, np.float32)
    kv = np.ascontiguousarray(kf_valid, np.uint8)
    kn, ki = flat(kf_fv); fn, fi = flat(f_fv)
    match = np.zeros(max(len(fd), 1), np.int32)
    L = _mlib()
    vp, f32 = C.c_void_p, C.c_float
    L.mo_search_by_bow.restype = C.c_int
    L.mo_search_by_bow.argtypes = [vp, vp, vp, C.c_int, vp, vp, C.c_int, vp, vp, C.c_int, vp, vp, C.c_int, f32, C.c_int, vp]
    n = L.mo_search_by_bow(_ptr(kd), _ptr(ka), _ptr(kv), len(kd), _ptr(kn), _ptr(ki), len(kn), _ptr(fd), _ptr(fa), len(fd), _ptr(fn), _ptr(fi),
                           len(fn), float(nnratio), int(check_ori), _ptr(match))
    return int(n), match[:len(fd)].copy()


def stereo_matches(kpsL, descL, kpsR, descR, pyrL, pyrR, scale, inv_scale, mb, mbf):
    """Frame::ComputeStereoMatches (src/Frame.cc:811-981) -> (mvuRight, mvDepth, number of matches kept)."""
    kL, kR = np.ascontiguousarray(kpsL), np.ascontiguousarray(kpsR)
    dL, dR = np.ascontiguousarray(descL, np.uint8), np.ascontiguousarray(descR, np.uint8)
    pl = [np.ascontiguousarray(p) for p in pyrL]
    pr = [np.ascontiguousarray(p) for p in pyrR]
    n = len(pl)
    PL = (C.c_void_p * n)(*[p.ctypes.data for p in pl])
    PR = (C.c_void_p * n)(*[p.ctypes.data for p in pr])
    w = np.array([p.shape[1] for p in pl], np.int32); h = np.array([p.shape[0] for p in pl], np.int32)
    pitch = np.array([p.strides[0] for p in pl], np.int32)
    assert all(a.shape == b.shape for a, b in zip(pl, pr))
    sc, isc = np.ascontiguousarray(scale, np.float32), np.ascontiguousarray(inv_scale, np.float32)
    ur = np.zeros(max(len(kL), 1), np.float32); dp = np.zeros(max(len(kL), 1), np.float32)
    kept = _mlib().mo_stereo_matches(_ptr(kL), _ptr(dL), len(kL), _ptr(kR), _ptr(dR), len(kR), PL, PR, _ptr(w), _ptr(h), _ptr(pitch),
                                     _ptr(sc), _ptr(isc), float(mb), float(mbf), _ptr(ur), _ptr(dp))
    return ur[:len(kL)].copy(), dp[:len(kL)].copy(), int(kept)


def knn2(q, t):
    q = np.ascontiguousarray(q, np.uint8); t = np.ascontiguousarray(t, np.uint8)
    idx = np.zeros((max(len(q), 1), 2), np.int32); dist = np.zeros((max(len(q), 1), 2), np.int32)
    _mlib().mo_knn2(_ptr(q), len(q), _ptr(t), len(t), _ptr(idx), _ptr(dist))
    return idx[:len(q)], dist[:len(q)]


class OracleVocabulary:
    def __init__(self, path: str):
        self._h = _mlib().mo_voc_load(path.encode())
        if not self._h:
            raise RuntimeError("oracle vocabulary load failed")

    def __del__(self):
        if getattr(self, "_h", None):
            _mlib().mo_voc_free(self._h)
            self._h = None

    def descend(self, desc, levelsup=4):
        d = np.ascontiguousarray(desc, np.uint8).reshape(-1, 32)
        n = len(d)
        word, node, w = np.zeros(max(n, 1), np.uint32), np.zeros(max(n, 1), np.uint32), np.zeros(max(n, 1), np.float64)
        _mlib().mo_voc_descend(self._h, _ptr(d), n, levelsup, _ptr(word), _ptr(w), _ptr(node))
        return word[:n], w[:n], node[:n]

    def transform(self, desc, levelsup=4):
        d = np.ascontiguousarray(desc, np.uint8).reshape(-1, 32)
        n = len(d)
        ids, vals = np.zeros(max(n, 1), np.uint32), np.zeros(max(n, 1), np.float64)
        fvn, fvf, nfv = np.zeros(max(n, 1), np.uint32), np.zeros(max(n, 1), np.uint32), C.c_int(0)
        k = _mlib().mo_voc_transform(self._h, _ptr(d), n, levelsup, _ptr(ids), _ptr(vals), _ptr(fvn), _ptr(fvf), C.byref(nfv))
        fv = {}
        for a, b in zip(fvn[:nfv.value], fvf[:nfv.value]):
            fv.setdefault(int(a), []).append(int(b))
        return (ids[:k].copy(), vals[:k].copy()), fv


def score_l1(a, b) -> float:
    ia, va = np.ascontiguousarray(a[0], np.uint32), np.ascontiguousarray(a[1], np.float64)
    ib, vb = np.ascontiguousarray(b[0], np.uint32), np.ascontiguousarray(b[1], np.float64)
    return float(_mlib().mo_score_l1(_ptr(ia), _ptr(va), len(ia), _ptr(ib), _ptr(vb), len(ib)))


# ---- the REFERENCE's own DBoW2 code (oracle/_ref/libref_dbow2.so, built by oracle/ref_fragments.mk) ---------
_REF_PATH = os.path.join(_HERE, "_ref", "libref_dbow2.so")
_ref = None


def ref_available() -> bool:
    return os.path.exists(_REF_PATH)


def ref_lib():
    global _ref
    if _ref is None:
        R = C.CDLL(_REF_PATH)
        vp = C.c_void_p
        R.ref_forb_distance.restype = C.c_int
        R.ref_forb_distance.argtypes = [vp, vp]
        R.ref_voc_load.restype = vp
        R.ref_voc_load.argtypes = [C.c_char_p]
        R.ref_voc_free.argtypes = [vp]
        R.ref_voc_size.argtypes = [vp]
        R.ref_voc_transform.restype = C.c_int
        R.ref_voc_transform.argtypes = [vp, vp, C.c_int, C.c_int, vp, vp, vp, vp, C.POINTER(C.c_int)]
        R.ref_voc_score.restype = C.c_double
        R.ref_voc_score.argtypes = [vp, vp, vp, C.c_int, vp, vp, C.c_int]
        _ref = R
    return _ref


def ref_hamming(a, b) -> int:
    a = np.ascontiguousarray(a, np.uint8); b = np.ascontiguousarray(b, np.uint8)
    return int(ref_lib().ref_forb_distance(_ptr(a), _ptr(b)))


class RefVocabulary:
    """DBoW2::TemplatedVocabulary<cv::Mat, FORB> of the reference itself."""

    def __init__(self, path: str):
        self._h = ref_lib().ref_voc_load(path.encode())
        if not self._h:
            raise RuntimeError("reference vocabulary load failed")

    def __del__(self):
        if getattr(self, "_h", None):
            ref_lib().ref_voc_free(self._h)
            self._h = None

    def saveToTextFile(self, path: str) -> None:
        """The reference's own TemplatedVocabulary::saveToTextFile."""
        L = ref_lib()
        L.ref_voc_save.argtypes = [C.c_void_p, C.c_char_p]
        L.ref_voc_save(self._h, path.encode())

    def size(self) -> int:
        return ref_lib().ref_voc_size(self._h)

    def transform(self, desc, levelsup=4):
        d = np.ascontiguousarray(desc, np.uint8).reshape(-1, 32)
        n = len(d)
        ids, vals = np.zeros(max(n, 1), np.uint32), np.zeros(max(n, 1), np.float64)
        fvn, fvf, nfv = np.zeros(max(n, 1), np.uint32), np.zeros(max(n, 1), np.uint32), C.c_int(0)
        k = ref_lib().ref_voc_transform(self._h, _ptr(d), n, levelsup, _ptr(ids), _ptr(vals), _ptr(fvn), _ptr(fvf), C.byref(nfv))
        fv = {}
        for a, b in zip(fvn[:nfv.value], fvf[:nfv.value]):
            fv.setdefault(int(a), []).append(int(b))
        return (ids[:k].copy(), vals[:k].copy()), fv

    def score(self, a, b) -> float:
        ia, va = np.ascontiguousarray(a[0], np.uint32), np.ascontiguousarray(a[1], np.float64)
        ib, vb = np.ascontiguousarray(b[0], np.uint32), np.ascontiguousarray(b[1], np.float64)
        return float(ref_lib().ref_voc_score(self._h, _ptr(ia), _ptr(va), len(ia), _ptr(ib), _ptr(vb), len(ib)))


class OracleKeyFrameDatabase:
    """KeyFrameDatabase (src/KeyFrameDatabase.cc) with the reference's inverted file; `query` = the opening shared by the
    Detect* routines (share-a-word list in list order, common-word counts, thresholds, L1 scores)."""

    def __init__(self):
        L = _mlib()
        vp = C.c_void_p
        L.mo_kfdb_new.restype = vp
        L.mo_kfdb_free.argtypes = [vp]
        L.mo_kfdb_add.argtypes = [vp, C.c_long, vp, vp, C.c_int]
        L.mo_kfdb_erase.argtypes = [vp, C.c_long]
        L.mo_kfdb_query.restype = C.c_int
        L.mo_kfdb_query.argtypes = [vp, vp, vp, C.c_int, vp, C.c_int, C.c_int, vp, vp, vp, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        self._L, self._h, self._n = L, vp(L.mo_kfdb_new()), 0

    def __del__(self):
        try:
            if self._h:
                self._L.mo_kfdb_free(self._h)
                self._h = None
        except Exception:
            pass

    def add(self, kf_id, bow):
        ids = np.ascontiguousarray(bow[0], np.uint32); vals = np.ascontiguousarray(bow[1], np.float64)
        self._L.mo_kfdb_add(self._h, int(kf_id), _ptr(ids), _ptr(vals), len(ids))
        self._n += 1

    def erase(self, kf_id):
        self._L.mo_kfdb_erase(self._h, int(kf_id))

    def query(self, bow, exclude=(), min_words_floor=0):
        ids = np.ascontiguousarray(bow[0], np.uint32); vals = np.ascontiguousarray(bow[1], np.float64)
        ex = np.ascontiguousarray(list(exclude), np.int64)
        cap = max(self._n, 1)
        kf = np.zeros(cap, np.int64); words = np.zeros(cap, np.int32); score = np.zeros(cap, np.float64)
        mx, mn = C.c_int(0), C.c_int(0)
        k = self._L.mo_kfdb_query(self._h, _ptr(ids), _ptr(vals), len(ids), _ptr(ex), len(ex), int(min_words_floor), _ptr(kf), _ptr(words),
                                  _ptr(score), C.byref(mx), C.byref(mn))
        return dict(kf=kf[:k].copy(), words=words[:k].copy(), score=score[:k].copy(), max_common=mx.value, min_common=mn.value)


# ---- the REFERENCE's own ORBextractor (oracle/_ref/libref_orbextractor.so: src/ORBextractor.cc compiled where it lies) ----
_REF_EXT_PATH = os.path.join(_HERE, "_ref", "libref_orbextractor.so")
# the same file compiled the way the reference's CMakeLists.txt:10-13 does on an FMA machine (-O3 -mfma, the compiler's default
# -ffp-contract): what "the reference CPU path" is for a maintainer who builds with -march=native
_REF_EXT_FMA_PATH = os.path.join(_HERE, "_ref", "libref_orbextractor_fma.so")
_ref_ext = {}


def ref_extractor_available(fma: bool = False) -> bool:
    if fma:
        try:
            has = " fma " in open("/proc/cpuinfo").read()
        except OSError:
            has = False
        return has and os.path.exists(_REF_EXT_FMA_PATH)
    return os.path.exists(_REF_EXT_PATH)


def _ref_ext_lib(fma: bool = False):
    if fma not in _ref_ext:
        lib()   # liborb_oracle.so provides the five forwarded OpenCV primitives
        R = C.CDLL(_REF_EXT_FMA_PATH if fma else _REF_EXT_PATH)
        vp = C.c_void_p
        R.ref_ext_create.restype = vp
        R.ref_ext_create.argtypes = [C.c_int, C.c_float, C.c_int, C.c_int, C.c_int]
        R.ref_ext_destroy.argtypes = [vp]
        R.ref_ext_extract.restype = C.c_int
        R.ref_ext_extract.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, C.c_int, C.POINTER(C.c_int)]
        R.ref_ext_tables.argtypes = [vp, vp, vp, vp, vp]
        R.ref_ext_level.restype = C.c_int
        R.ref_ext_level.argtypes = [vp, C.c_int, C.c_int, vp, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        _ref_ext[fma] = R
    return _ref_ext[fma]


class RefExtractor:
    """ORB_SLAM3::ORBextractor — the reference's own class (include/ORBextractor.h:49-83)."""

    def __init__(self, nfeatures=1000, scale_factor=1.2, nlevels=8, ini_th=20, min_th=7, fma: bool = False):
        self.nfeatures, self.nlevels = nfeatures, nlevels
        self._R = _ref_ext_lib(fma)
        self._h = self._R.ref_ext_create(nfeatures, scale_factor, nlevels, ini_th, min_th)
        self.cap = nfeatures + 35 * nlevels + 64   # a level returns up to max(quota + 3, 4 x root nodes) keypoints (<= 8 roots)

    def __del__(self):
        if getattr(self, "_h", None):
            self._R.ref_ext_destroy(self._h)
            self._h = None

    def tables(self):
        n = self.nlevels
        sc, isc, s2, is2 = (np.zeros(n, np.float32) for _ in range(4))
        self._R.ref_ext_tables(self._h, _ptr(sc), _ptr(isc), _ptr(s2), _ptr(is2))
        return dict(scale=sc, inv_scale=isc, sigma2=s2, inv_sigma2=is2)

    def extract(self, img: np.ndarray, lapping=(0, 0)):
        """operator()(image, mask, keypoints, descriptors, vLappingArea) -> (keypoints, descriptors, return value)."""
        assert img.dtype == np.uint8 and img.ndim == 2 and img.strides[1] == 1
        kps = np.zeros(self.cap, KP_DTYPE)
        desc = np.zeros((self.cap, 32), np.uint8)
        n = C.c_int(0)
        mono = self._R.ref_ext_extract(self._h, img.ctypes.data, img.shape[0], img.shape[1], img.strides[0], int(lapping[0]),
                                              int(lapping[1]), _ptr(kps), _ptr(desc), self.cap, C.byref(n))
        if mono == -100000:
            raise RuntimeError(f"reference extractor returned {n.value} keypoints, more than the capacity {self.cap}")
        return kps[:n.value].copy(), desc[:n.value].copy(), mono

    def level(self, level: int, with_border: bool = False) -> np.ndarray:
        w, h = C.c_int(0), C.c_int(0)
        assert self._R.ref_ext_level(self._h, level, 0, None, C.byref(w), C.byref(h)) == 0
        b = 19 if with_border else 0
        out = np.zeros((h.value + 2 * b, w.value + 2 * b), np.uint8)
        assert self._R.ref_ext_level(self._h, level, int(with_border), _ptr(out), C.byref(w), C.byref(h)) == 0
        return out
