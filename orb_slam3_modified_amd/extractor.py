"""Host-side mirror of ORB_SLAM3::ORBextractor (reference include/ORBextractor.h:49-83) over the C ABI.

Same constructor arguments, same call semantics (`operator()` -> `__call__` returning monoIndex, keypoints in
the reference's output order, 32-byte descriptors), same getters, `mvImagePyramid` available after a call.
All arithmetic runs in the HIP kernels of liborbx.so; this file only marshals buffers.
"""
from __future__ import annotations

import ctypes as C
from typing import List, Sequence, Tuple

import numpy as np

from . import _lib
from ._lib import KP_DTYPE, check, lib, ptr


class ORBextractor:
    HARRIS_SCORE, FAST_SCORE = 0, 1  # include/ORBextractor.h:47

    def __init__(self, nfeatures: int, scaleFactor: float, nlevels: int, iniThFAST: int, minThFAST: int,
                 device_id: int = -1):
        self._L = lib()
        self._ctx = C.c_void_p(0)
        rc = self._L.orbx_create(C.byref(self._ctx), nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST, device_id)
        if rc != 0:
            raise _lib.OrbxError(rc, "orbx_create failed (no HIP device?)" if rc == _lib.ORBX_E_DEVICE else "bad arguments")
        self.nfeatures, self.nlevels, self.scaleFactor = nfeatures, nlevels, float(np.float32(scaleFactor))
        self.iniThFAST, self.minThFAST, self.device_id = iniThFAST, minThFAST, device_id
        self._ctor_scale = scaleFactor
        self.capacity = self._L.orbx_keypoint_capacity(self._ctx)
        n = nlevels
        self._scale, self._inv, self._s2, self._is2 = (np.zeros(n, np.float32) for _ in range(4))
        self._quota = np.zeros(n, np.int32)
        check(self._L.orbx_scale_tables(self._ctx, ptr(self._scale), ptr(self._inv), ptr(self._s2), ptr(self._is2),
                                        ptr(self._quota)), self._ctx)
        self._last_frames = 0

    def clone(self) -> "ORBextractor":
        """A second context with the same parameters (own stream, own buffers): contexts are independent, so two of them
        can work on two halves of a batch concurrently (replay lanes, stereo)."""
        c = ORBextractor(self.nfeatures, self._ctor_scale, self.nlevels, self.iniThFAST, self.minThFAST, self.device_id)
        for k, v in self.cpu_profile()[1].items():   # the five result-changing options travel with the clone (launch-shape knobs do not)
            c.set_option(k, v)
        return c

    def close(self):
        if getattr(self, "_ctx", None) and self._ctx.value:
            self._L.orbx_destroy(self._ctx)
            self._ctx = C.c_void_p(0)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- getters (include/ORBextractor.h:63-82) -------------------------------------------------------
    def GetLevels(self) -> int: return self.nlevels
    def GetScaleFactor(self) -> float: return self.scaleFactor
    def GetScaleFactors(self) -> np.ndarray: return self._scale.copy()
    def GetInverseScaleFactors(self) -> np.ndarray: return self._inv.copy()
    def GetScaleSigmaSquares(self) -> np.ndarray: return self._s2.copy()
    def GetInverseScaleSigmaSquares(self) -> np.ndarray: return self._is2.copy()
    def features_per_level(self) -> np.ndarray: return self._quota.copy()

    # ---- operator() (src/ORBextractor.cc:1086-1168) ----------------------------------------------------
    def __call__(self, image: np.ndarray, mask=None, vLappingArea: Sequence[int] = (0, 0)
                 ) -> Tuple[int, np.ndarray, np.ndarray]:
        """Returns (monoIndex, keypoints[KP_DTYPE], descriptors[n,32] uint8); (-1, empty, empty) for an empty image."""
        if image is None or image.size == 0:
            return -1, np.zeros(0, KP_DTYPE), np.zeros((0, 32), np.uint8)
        assert image.dtype == np.uint8 and image.ndim == 2, "CV_8UC1 expected (src/ORBextractor.cc:1094)"
        if image.strides[1] != 1:
            image = np.ascontiguousarray(image)
        kps = np.zeros(self.capacity, KP_DTYPE)
        desc = np.zeros((self.capacity, 32), np.uint8)
        n, mono = C.c_int(0), C.c_int(0)
        check(self._L.orbx_extract(self._ctx, ptr(image), image.shape[0], image.shape[1], image.strides[0],
                                   int(vLappingArea[0]), int(vLappingArea[1]), ptr(kps), ptr(desc), C.byref(n),
                                   C.byref(mono)), self._ctx)
        self._last_frames = 1
        return mono.value, kps[:n.value].copy(), desc[:n.value].copy()

    def extract_color(self, image: np.ndarray, rgb: bool, vLappingArea: Sequence[int] = (0, 0)):
        """Tracking::GrabImageMonocular's cvtColor (src/Tracking.cc:1572-1585) fused behind the upload: image is [H, W, 3|4]
        uint8, rgb = mbRGB.  Returns (monoIndex, keypoints, descriptors); the grey plane is pyramid level 0 afterwards."""
        assert image.dtype == np.uint8 and image.ndim == 3 and image.shape[2] in (3, 4)
        if image.strides[2] != 1 or image.strides[1] != image.shape[2]:
            image = np.ascontiguousarray(image)
        kps = np.zeros(self.capacity, KP_DTYPE)
        desc = np.zeros((self.capacity, 32), np.uint8)
        n, mono = C.c_int(0), C.c_int(0)
        check(self._L.orbx_extract_color(self._ctx, ptr(image), image.shape[0], image.shape[1], image.strides[0], image.shape[2], int(rgb),
                                         int(vLappingArea[0]), int(vLappingArea[1]), ptr(kps), ptr(desc), C.byref(n), C.byref(mono)),
              self._ctx)
        self._last_frames = 1
        return mono.value, kps[:n.value].copy(), desc[:n.value].copy()

    def extract_resized(self, image: np.ndarray, new_size: Tuple[int, int], vLappingArea: Sequence[int] = (0, 0)):
        """cv::resize(im, ., newImSize) (src/System.cc:441-446) fused behind the upload, then operator().  new_size = (rows, cols)."""
        img = np.ascontiguousarray(image)
        assert img.dtype == np.uint8 and img.ndim == 2
        kps = np.zeros(self.capacity, KP_DTYPE)
        desc = np.zeros((self.capacity, 32), np.uint8)
        n, mono = C.c_int(0), C.c_int(0)
        check(self._L.orbx_extract_resized(self._ctx, ptr(img), img.shape[0], img.shape[1], img.strides[0], int(new_size[0]), int(new_size[1]),
                                           int(vLappingArea[0]), int(vLappingArea[1]), ptr(kps), ptr(desc), C.byref(n), C.byref(mono)), self._ctx)
        return mono.value, kps[:n.value].copy(), desc[:n.value].copy()

    def publish_descriptors(self, desc: np.ndarray) -> None:
        """orbx_publish_descriptors: `desc` (C-contiguous [n, 32] uint8) holds the rows of this context's last single-frame extraction:
        ORBVocabulary.descend_published(desc) then finds the records the extraction graph computed (no device round trip)."""
        assert desc.dtype == np.uint8 and desc.flags["C_CONTIGUOUS"]
        check(self._L.orbx_publish_descriptors(self._ctx, ptr(desc), int(desc.shape[0])), self._ctx)

    def extract_batch(self, images: np.ndarray, vLappingArea: Sequence[int] = (0, 0)
                      ) -> List[Tuple[int, np.ndarray, np.ndarray]]:
        """Batch replay over host frames [B, H, W] (frames are independent: SURVEY.md §8(e))."""
        assert images.dtype == np.uint8 and images.ndim == 3
        images = np.ascontiguousarray(images)
        B, H, W = images.shape
        kps = np.zeros((B, self.capacity), KP_DTYPE)
        desc = np.zeros((B, self.capacity, 32), np.uint8)
        counts = np.zeros((B, 2), np.int32)
        check(self._L.orbx_extract_batch(self._ctx, ptr(images), B, H, W, images.strides[1], images.strides[0],
                                         int(vLappingArea[0]), int(vLappingArea[1]), ptr(kps), ptr(desc), ptr(counts)),
              self._ctx)
        self._last_frames = B
        return [(int(counts[f, 1]), kps[f, :counts[f, 0]].copy(), desc[f, :counts[f, 0]].copy()) for f in range(B)]

    def extract_batch_device(self, d_imgs: int, nframes: int, rows: int, cols: int, row_stride: int, frame_stride: int,
                             d_kps: int, d_desc: int, d_counts: int, vLappingArea: Sequence[int] = (0, 0),
                             stream: int = 0) -> None:
        """Device-resident batch: raw HBM addresses (e.g. torch `tensor.data_ptr()`); asynchronous on `stream`."""
        check(self._L.orbx_extract_batch_device(self._ctx, ptr(d_imgs), nframes, rows, cols, row_stride, frame_stride,
                                                int(vLappingArea[0]), int(vLappingArea[1]), ptr(d_kps), ptr(d_desc),
                                                ptr(d_counts), ptr(stream)), self._ctx)
        self._last_frames = nframes

    # ---- mvImagePyramid (include/ORBextractor.h:83) ---------------------------------------------------
    def pyramid_level(self, level: int, frame: int = 0) -> np.ndarray:
        w, h = C.c_int(0), C.c_int(0)
        check(self._L.orbx_pyramid_level(self._ctx, frame, level, None, 0, C.byref(w), C.byref(h)), self._ctx)
        out = np.zeros((h.value, w.value), np.uint8)
        check(self._L.orbx_pyramid_level(self._ctx, frame, level, ptr(out), w.value, C.byref(w), C.byref(h)), self._ctx)
        return out

    @property
    def mvImagePyramid(self) -> List[np.ndarray]:
        return [self.pyramid_level(l) for l in range(self.nlevels)]

    def debug_blur_level(self, level: int, frame: int = 0) -> np.ndarray:
        """The blurred copy of pyramid level `level` the descriptors sample (stage-level parity tests)."""
        w, h = C.c_int(0), C.c_int(0)
        check(self._L.orbx_pyramid_level(self._ctx, frame, level, None, 0, C.byref(w), C.byref(h)), self._ctx)
        out = np.zeros((h.value, w.value), np.uint8)
        check(self._L.orbx_debug_blur_level(self._ctx, frame, level, ptr(out), w.value), self._ctx)
        return out

    # ---- stage dumps / profiling ------------------------------------------------------------------------
    def debug_level_points(self, level: int, stage: int, frame: int = 0):
        n = check(self._L.orbx_debug_level_points(self._ctx, frame, level, stage, None, 0), self._ctx)
        buf = np.zeros(max(n, 1), np.uint32)
        check(self._L.orbx_debug_level_points(self._ctx, frame, level, stage, ptr(buf), n), self._ctx)
        buf = buf[:n]
        return (buf & 0xfff).astype(np.int32), ((buf >> 12) & 0xfff).astype(np.int32), (buf >> 24).astype(np.int32)

    def debug_trig(self, y, x=None):
        """Device fastAtan2 + cos/sin (test hook): returns (angle_deg, cos, sin) float32 arrays."""
        y = np.ascontiguousarray(y, np.float32)
        xx = None if x is None else np.ascontiguousarray(x, np.float32)
        n = len(y)
        ang, a, b = (np.zeros(max(n, 1), np.float32) for _ in range(3))
        check(self._L.orbx_debug_trig(self._ctx, ptr(y), ptr(xx), n, int(x is None), ptr(ang), ptr(a), ptr(b)), self._ctx)
        return ang[:n], a[:n], b[:n]

    def debug_gnu_sort(self, elems: np.ndarray, threads: int = 256) -> np.ndarray:
        """Test hook: the kernel's workgroup-parallel restatement of libstdc++'s std::sort on 64-bit (key << 32 | payload) words."""
        v = np.ascontiguousarray(elems, np.uint64).copy()
        check(self._L.orbx_debug_gnu_sort(self._ctx, ptr(v), len(v), int(threads)), self._ctx)
        return v

    def debug_trig_hash(self, first_bits: int, count: int) -> int:
        """Digest of the device (cos, sin)(angle * pi/180) over `count` consecutive float bit patterns (exhaustive test hook)."""
        h = C.c_uint64(0)
        check(self._L.orbx_debug_trig_hash(self._ctx, first_bits, count, C.byref(h)), self._ctx)
        return int(h.value)

    def debug_atan_hash(self, seed: int, count: int) -> int:
        """Digest of the device fastAtan2 over `count` pseudo-random moment pairs (test hook)."""
        h = C.c_uint64(0)
        check(self._L.orbx_debug_atan_hash(self._ctx, seed, count, C.byref(h)), self._ctx)
        return int(h.value)

    def debug_brief_hash(self, first_bits: int, count: int) -> int:
        """Digest of the device's rotated BRIEF pattern (all 512 points) over `count` consecutive float bit patterns of the angle (test hook)."""
        h = C.c_uint64(0)
        check(self._L.orbx_debug_brief_hash(self._ctx, first_bits, count, C.byref(h)), self._ctx)
        return int(h.value)

    def debug_calib_copy(self, d_src: int, d_dst: int, nbytes: int, width: int, stream: int = 0):
        """Known-traffic device copy (counter calibration, tools/pmc_traffic.py)."""
        check(self._L.orbx_debug_calib_copy(self._ctx, ptr(d_src), ptr(d_dst), nbytes, width, ptr(stream)), self._ctx)

    def UndistortKeyPoints(self, kps: np.ndarray, fx: float, fy: float, cx: float, cy: float, dist_coef) -> np.ndarray:
        """Frame::UndistortKeyPoints (src/Frame.cc:747-780) on the GPU: mvKeys -> mvKeysUn."""
        k = np.ascontiguousarray(kps)
        dc = np.ascontiguousarray(dist_coef, np.float32)
        out = np.zeros(max(len(k), 1), KP_DTYPE)
        check(self._L.orbx_undistort_keypoints(self._ctx, ptr(k), len(k), float(fx), float(fy), float(cx), float(cy), ptr(dc), len(dc), ptr(out)), self._ctx)
        return out[:len(k)]

    def undistort_keypoints_device(self, d_kps: int, d_counts: int, nframes: int, fx, fy, cx, cy, dist_coef, d_kps_un: int, stream: int = 0) -> None:
        dc = np.ascontiguousarray(dist_coef, np.float32)
        check(self._L.orbx_undistort_keypoints_device(self._ctx, ptr(d_kps), ptr(d_counts), int(nframes), self.capacity, float(fx), float(fy),
                                                      float(cx), float(cy), ptr(dc), len(dc), ptr(d_kps_un), ptr(stream)), self._ctx)

    def reserve(self, rows: int, cols: int, nframes: int) -> None:
        """Allocate the device buffers for batches of this shape now (orbx_reserve) rather than in the first call."""
        check(self._L.orbx_reserve(self._ctx, int(rows), int(cols), int(nframes)), self._ctx)

    def set_option(self, name: str, value: int) -> None:
        """Option of this context (orbx_set_option, include/orbx.h).  Scheduling / launch-shape knobs never change results; the five
        OpenCV-build options do, by design: gauss_kernel / gauss_round / gauss_tail (which release's 8-bit cv::GaussianBlur), atan_fma,
        brief_fma (FMA contraction in cv::fastAtan2 / in the reference's own pattern rotation) — INTEGRATION.md section 6."""
        check(self._L.orbx_set_option(self._ctx, name.encode(), int(value)), self._ctx)

    def set_cpu_profile(self, name: str, fma_build: int = 0) -> None:
        """orbx_set_cpu_profile: the five result-changing options by name — "opencv>=4.5.1" (default), "opencv-4.4" (= -avx2; also -sse,
        -avx512, -scalar), "opencv-3.2"; fma_build bit 0 = the reference built -march=native on an FMA machine, bit 1 = OpenCV's AVX2 fastAtan2."""
        check(self._L.orbx_set_cpu_profile(self._ctx, name.encode(), int(fma_build)), self._ctx)

    def cpu_profile(self):
        """(description string, {option: value}) of the ACTIVE set (orbx_get_cpu_profile)."""
        import ctypes as C
        buf = C.create_string_buffer(256)
        v = np.zeros(5, np.int32)
        check(self._L.orbx_get_cpu_profile(self._ctx, buf, 256, ptr(v)), self._ctx)
        return buf.value.decode(), dict(zip(("gauss_kernel", "gauss_round", "gauss_tail", "atan_fma", "brief_fma"), (int(x) for x in v)))

    @staticmethod
    def cpu_profiles():
        """{name: (description, option values with fma_build = 0)} of every named profile (host-only table: no device needed)."""
        L = _lib.lib()
        out = {}
        for i in range(64):
            nm = L.orbx_cpu_profile_name(i)
            if not nm:      # NULL ends the table
                break
            v = np.zeros(5, np.int32)
            check(L.orbx_cpu_profile_values(nm, 0, ptr(v)))
            out[nm.decode()] = (L.orbx_cpu_profile_description(nm).decode(), tuple(int(x) for x in v))
        return out

    def profile_enable(self, on: bool = True):
        check(self._L.orbx_profile_enable(self._ctx, int(on)), self._ctx)

    def profile_read(self):
        ms = np.zeros(_lib.NUM_KERNELS, np.float64)
        n = np.zeros(_lib.NUM_KERNELS, np.int64)
        check(self._L.orbx_profile_read(self._ctx, ptr(ms), ptr(n)), self._ctx)
        return {self._L.orbx_kernel_name(i).decode(): (float(ms[i]), int(n[i])) for i in range(_lib.NUM_KERNELS)}
