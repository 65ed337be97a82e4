"""Host-side mirror of the ORB_SLAM3::ORBmatcher primitives (reference include/ORBmatcher.h:36-103).

The reference's 12 Search*/Fuse routines share one inner pattern: candidate index list -> Hamming distances ->
best / second best with per-routine tie rule and thresholds.  The distance and arg-min part
runs on the GPU (orbx_nn_csr / orbx_knn2_allpairs); the pointer-rich, order-dependent bookkeeping that wraps it
(MapPoint observations, greedy un-matching) stays with the caller, exactly as in the reference.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from ._lib import check, lib, ptr


class ORBmatcher:
    TH_LOW, TH_HIGH, HISTO_LENGTH = 50, 100, 30  # src/ORBmatcher.cc:35-37

    def __init__(self, extractor_or_ctx, nnratio: float = 0.6, checkOri: bool = True):
        self._L = lib()
        self._ctx = getattr(extractor_or_ctx, "_ctx", extractor_or_ctx)
        self.mfNNratio, self.mbCheckOrientation = float(nnratio), bool(checkOri)

    @staticmethod
    def DescriptorDistance(a: np.ndarray, b: np.ndarray) -> int:
        """src/ORBmatcher.cc:2058-2074"""
        a = np.ascontiguousarray(a, np.uint8).reshape(32)
        b = np.ascontiguousarray(b, np.uint8).reshape(32)
        return int(lib().orbx_hamming(ptr(a), ptr(b)))

    def nn_csr(self, q_desc, t_desc, row_ptr, cand, last_wins: bool = False, want_dist: bool = False):
        q = np.ascontiguousarray(q_desc, np.uint8).reshape(-1, 32)
        t = np.ascontiguousarray(t_desc, np.uint8).reshape(-1, 32)
        rp = np.ascontiguousarray(row_ptr, np.int32)
        cd = np.ascontiguousarray(cand, np.int32)
        nq = len(q)
        bi, bd, si, sd = (np.zeros(max(nq, 1), np.int32) for _ in range(4))
        do = np.zeros(max(len(cd), 1), np.int32) if want_dist else None
        check(self._L.orbx_nn_csr(self._ctx, ptr(q), nq, ptr(t), len(t), ptr(rp), ptr(cd), int(last_wins), ptr(bi), ptr(bd),
                                  ptr(si), ptr(sd), ptr(do)), self._ctx)
        out = (bi[:nq], bd[:nq], si[:nq], sd[:nq])
        return out + (do[:len(cd)],) if want_dist else out

    def knn2(self, q_desc, t_desc):
        """cv::BFMatcher(NORM_HAMMING).knnMatch(k=2) (src/Frame.cc:43,1144) -> (idx[nq,2], dist[nq,2])."""
        q = np.ascontiguousarray(q_desc, np.uint8).reshape(-1, 32)
        t = np.ascontiguousarray(t_desc, np.uint8).reshape(-1, 32)
        idx = np.zeros((max(len(q), 1), 2), np.int32)
        dist = np.zeros((max(len(q), 1), 2), np.int32)
        check(self._L.orbx_knn2_allpairs(self._ctx, ptr(q), len(q), ptr(t), len(t), ptr(idx), ptr(dist)), self._ctx)
        return idx[:len(q)], dist[:len(q)]

    def GetFeaturesInArea(self, kps, bounds, qx, qy, qr, min_level, max_level):
        """Frame::GetFeaturesInArea (src/Frame.cc:657-723) for many queries over the frame grid of `kps`
        (Frame::AssignFeaturesToGrid, :385-416), built and queried on the GPU -> CSR (row_ptr, cand) in the
        reference's candidate order.  bounds = (mnMinX, mnMinY, mnMaxX, mnMaxY)."""
        k = np.ascontiguousarray(kps)
        qx, qy, qr = (np.ascontiguousarray(v, np.float32) for v in (qx, qy, qr))
        lo, hi = (np.ascontiguousarray(v, np.int32) for v in (min_level, max_level))
        nq = len(qx)
        rp = np.zeros(nq + 1, np.int32)
        cap = 1 << 16
        while True:
            cand = np.zeros(cap, np.int32)
            rc = self._L.orbx_features_in_area(self._ctx, ptr(k), len(k), *[float(b) for b in bounds], ptr(qx), ptr(qy), ptr(qr),
                                               ptr(lo), ptr(hi), nq, ptr(rp), ptr(cand), cap)
            if rc == -4 and cap < (1 << 28):   # ORBX_E_CAPACITY: grow and retry
                cap *= 8
                continue
            check(rc, self._ctx)
            return rp, cand[:rc].copy()

    def SearchForInitialization(self, F1, F2, vbPrevMatched, windowSize: int = 10):
        """src/ORBmatcher.cc:648-763.  F1 / F2: frame-like objects with `mvKeysUn` (KP_DTYPE array), `mDescriptors`
        ([n,32] uint8) and `bounds` (mnMinX, mnMinY, mnMaxX, mnMaxY).  Returns (nmatches, vnMatches12) and updates
        vbPrevMatched ([n1,2] float32) in place, like the reference."""
        k1, k2 = np.ascontiguousarray(F1.mvKeysUn), np.ascontiguousarray(F2.mvKeysUn)
        d1 = np.ascontiguousarray(F1.mDescriptors, np.uint8).reshape(-1, 32)
        d2 = np.ascontiguousarray(F2.mDescriptors, np.uint8).reshape(-1, 32)
        assert vbPrevMatched.dtype == np.float32 and vbPrevMatched.flags["C_CONTIGUOUS"] and vbPrevMatched.shape == (len(k1), 2)
        m12 = np.zeros(max(len(k1), 1), np.int32)
        n = C.c_int(0)
        check(self._L.orbx_search_for_initialization(self._ctx, ptr(k1), ptr(d1), len(k1), ptr(k2), ptr(d2), len(k2),
                                                     *[float(b) for b in F2.bounds], ptr(vbPrevMatched), int(windowSize),
                                                     self.mfNNratio, int(self.mbCheckOrientation), ptr(m12), C.byref(n)), self._ctx)
        return n.value, m12[:len(k1)].copy()

    @staticmethod
    def ComputeStereoMatches(left_extractor, right_extractor, kpsL, descL, kpsR, descR, mb: float, mbf: float):
        """Frame::ComputeStereoMatches (src/Frame.cc:811-981) on the device pyramids of the two extractors (each must
        have just extracted its image).  Returns (mvuRight, mvDepth, number of matches kept)."""
        kL, kR = np.ascontiguousarray(kpsL), np.ascontiguousarray(kpsR)
        dL = np.ascontiguousarray(descL, np.uint8).reshape(-1, 32)
        dR = np.ascontiguousarray(descR, np.uint8).reshape(-1, 32)
        ur = np.zeros(max(len(kL), 1), np.float32)
        dp = np.zeros(max(len(kL), 1), np.float32)
        n = C.c_int(0)
        check(lib().orbx_stereo_matches(left_extractor._ctx, right_extractor._ctx, ptr(kL), ptr(dL), len(kL), ptr(kR), ptr(dR), len(kR),
                                        float(mb), float(mbf), ptr(ur), ptr(dp), C.byref(n)), left_extractor._ctx)
        return ur[:len(kL)].copy(), dp[:len(kL)].copy(), n.value

    @staticmethod
    def ComputeThreeMaxima(histo_counts):
        """src/ORBmatcher.cc:2012-2053 on bin populations."""
        max1 = max2 = max3 = 0
        ind1 = ind2 = ind3 = -1
        for i, s in enumerate(histo_counts):
            s = int(s)
            if s > max1:
                max3, max2, max1 = max2, max1, s
                ind3, ind2, ind1 = ind2, ind1, i
            elif s > max2:
                max3, max2 = max2, s
                ind3, ind2 = ind2, i
            elif s > max3:
                max3, ind3 = s, i
        if max2 < 0.1 * float(max1):
            ind2 = ind3 = -1
        elif max3 < 0.1 * float(max1):
            ind3 = -1
        return ind1, ind2, ind3
