"""Host-side mirror of the ORB_SLAM3::ORBmatcher primitives (reference include/ORBmatcher.h:36-103).

The reference's 12 Search*/Fuse routines share one inner pattern: candidate index list -> Hamming distances ->
best / second best with per-routine tie rule and thresholds.  The distance and arg-min part
runs on the GPU (orbx_nn_csr / orbx_knn2_allpairs); the pointer-rich, order-dependent bookkeeping that wraps it
(MapPoint observations, greedy un-matching) stays with the caller, exactly as in the reference.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from ._lib import OrbxGrid, check, lib, ptr


class ORBmatcher:
    TH_LOW, TH_HIGH, HISTO_LENGTH = 50, 100, 30  # src/ORBmatcher.cc:35-37

    def __init__(self, extractor_or_ctx, nnratio: float = 0.6, checkOri: bool = True):
        self._L = lib()
        self._owner = extractor_or_ctx   # keeps the extractor (and with it the orbx context) alive for as long as this object uses it
        self._ctx = getattr(extractor_or_ctx, "_ctx", extractor_or_ctx)
        self.mfNNratio, self.mbCheckOrientation = float(nnratio), bool(checkOri)

    def set_option(self, name: str, value: int) -> None:
        """Scheduling knob of the underlying context (orbx_set_option); never changes results."""
        check(self._L.orbx_set_option(self._ctx, name.encode(), int(value)), self._ctx)

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

    def nn_groups(self, q_desc, q_group, t_desc, group_ptr, group_cand, max_dist: int, pool_cap: int):
        """orbx_nn_groups: every query against its group's whole candidate list, only candidates within max_dist reported.
        Returns (q_off, q_cnt, entries[(idx, dist)]); raises OrbxError(ORBX_E_CAPACITY) when pool_cap entries do not suffice."""
        q = np.ascontiguousarray(q_desc, np.uint8).reshape(-1, 32)
        t = np.ascontiguousarray(t_desc, np.uint8).reshape(-1, 32)
        qg = np.ascontiguousarray(q_group, np.int32)
        gp = np.ascontiguousarray(group_ptr, np.int32)
        gc = np.ascontiguousarray(group_cand, np.int32)
        nq = len(q)
        off, cnt = np.zeros(max(nq, 1), np.int32), np.zeros(max(nq, 1), np.int32)
        ent = np.zeros((max(pool_cap, 1), 2), np.int32)
        n = C.c_int(0)
        check(self._L.orbx_nn_groups(self._ctx, ptr(q), ptr(qg), nq, ptr(t), len(t), ptr(gp), ptr(gc), len(gp) - 1, int(max_dist), ptr(off), ptr(cnt),
                                     ptr(ent), int(pool_cap), C.byref(n)), self._ctx)
        return off[:nq], cnt[:nq], ent[:n.value]

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

    def WindowSearch(self, kps, desc, bounds, qx, qy, qr, min_level, max_level, q_desc, kp_skip=None, kp_uright=None, q_xr=None):
        """Window query + gates + every candidate's Hamming distance + best / second per query in one device pass
        (orbx_window_search) -> dict(row_ptr, cand, dist, best_idx, best_dist, second_idx, second_dist)."""
        k = np.ascontiguousarray(kps)
        d = np.ascontiguousarray(desc, np.uint8).reshape(-1, 32)
        qx, qy, qr = (np.ascontiguousarray(v, np.float32) for v in (qx, qy, qr))
        lo, hi = (np.ascontiguousarray(v, np.int32) for v in (min_level, max_level))
        qd = np.ascontiguousarray(q_desc, np.uint8).reshape(-1, 32)
        nq = len(qx)
        skip = None if kp_skip is None else np.ascontiguousarray(kp_skip, np.uint8)
        ur = None if kp_uright is None else np.ascontiguousarray(kp_uright, np.float32)
        xr = None if q_xr is None else np.ascontiguousarray(q_xr, np.float32)
        rp = np.zeros(nq + 1, np.int32)
        bi, bd, si, sd = (np.zeros(max(nq, 1), np.int32) for _ in range(4))
        cap = 1 << 16
        while True:
            cand, dist = np.zeros(cap, np.int32), np.zeros(cap, np.int32)
            rc = self._L.orbx_window_search(self._ctx, ptr(k), ptr(d), len(k), *[float(b) for b in bounds], ptr(skip), ptr(ur),
                                            ptr(qx), ptr(qy), ptr(qr), ptr(lo), ptr(hi), ptr(qd), ptr(xr), nq, ptr(rp), ptr(cand),
                                            ptr(dist), cap, ptr(bi), ptr(bd), ptr(si), ptr(sd))
            if rc == -4 and cap < (1 << 28):
                cap *= 8
                continue
            check(rc, self._ctx)
            return dict(row_ptr=rp, cand=cand[:rc].copy(), dist=dist[:rc].copy(), best_idx=bi[:nq], best_dist=bd[:nq],
                        second_idx=si[:nq], second_dist=sd[:nq])

    @staticmethod
    def _grid(grid):
        """grid: dict(min_x, min_y, inv_w, inv_h[, cell_start, cell_idx]) -> (OrbxGrid, keep-alive arrays)."""
        cs = None if grid.get("cell_start") is None else np.ascontiguousarray(grid["cell_start"], np.int32)
        ci = None if cs is None else np.ascontiguousarray(grid["cell_idx"], np.int32)
        g = OrbxGrid(float(grid["min_x"]), float(grid["min_y"]), float(grid["inv_w"]), float(grid["inv_h"]),
                     None if cs is None else cs.ctypes.data, None if ci is None or len(ci) == 0 else ci.ctypes.data)
        if cs is not None and len(ci) == 0:
            ci = np.zeros(1, np.int32)
            g.cell_idx = ci.ctypes.data
        return g, (cs, ci)

    def WindowSearchGrid(self, kps, desc, grid, qx, qy, qr, min_level, max_level, q_desc, kp_skip=None, kp_uright=None, q_xr=None,
                         want_lists=True):
        """orbx_window_search_grid: orbx_window_search over a caller-held grid (Frame / KeyFrame::GetFeaturesInArea)."""
        k = np.ascontiguousarray(kps)
        d = np.ascontiguousarray(desc, np.uint8).reshape(-1, 32)
        qx, qy, qr = (np.ascontiguousarray(v, np.float32) for v in (qx, qy, qr))
        lo, hi = (np.ascontiguousarray(v, np.int32) for v in (min_level, max_level))
        qd = np.ascontiguousarray(q_desc, np.uint8).reshape(-1, 32)
        nq = len(qx)
        skip = None if kp_skip is None else np.ascontiguousarray(kp_skip, np.uint8)
        ur = None if kp_uright is None else np.ascontiguousarray(kp_uright, np.float32)
        xr = None if q_xr is None else np.ascontiguousarray(q_xr, np.float32)
        g, keep = self._grid(grid)
        rp = np.zeros(nq + 1, np.int32)
        bi, bd, si, sd = (np.zeros(max(nq, 1), np.int32) for _ in range(4))
        cap = 1 << 12
        while True:
            cand, dist = (np.zeros(cap, np.int32), np.zeros(cap, np.int32)) if want_lists else (None, None)
            rc = self._L.orbx_window_search_grid(self._ctx, ptr(k), ptr(d), len(k), C.byref(g), ptr(skip), ptr(ur), ptr(qx), ptr(qy), ptr(qr),
                                                 ptr(lo), ptr(hi), ptr(qd), ptr(xr), nq, ptr(rp), ptr(cand), ptr(dist), cap if want_lists else 0,
                                                 ptr(bi), ptr(bd), ptr(si), ptr(sd))
            if rc == -4 and want_lists:   # ORBX_E_CAPACITY: row_ptr[nq] is the capacity needed
                cap = int(rp[nq]) + 16
                continue
            check(rc, self._ctx)
            out = dict(row_ptr=rp, best_idx=bi[:nq], best_dist=bd[:nq], second_idx=si[:nq], second_dist=sd[:nq])
            if want_lists:
                out.update(cand=cand[:rc].copy(), dist=dist[:rc].copy())
            return out

    def Target(self, kps, desc, grid, kp_uright=None, inv_level_sigma2=None) -> "SearchTarget":
        """orbx_target_create: keypoints, descriptors and grid of a Frame / KeyFrame resident in HBM for repeated searches."""
        return SearchTarget(self, kps, desc, grid, kp_uright, inv_level_sigma2)

    def WindowNearest(self, kps, desc, grid, qx, qy, qr, min_level, max_level, q_desc, kp_uright=None, inv_level_sigma2=None, q_ur=None):
        """orbx_window_nearest: the device arg-min of Fuse x2 / SearchBySim3 (optionally with Fuse's reprojection gate)."""
        k = np.ascontiguousarray(kps)
        d = np.ascontiguousarray(desc, np.uint8).reshape(-1, 32)
        qx, qy, qr = (np.ascontiguousarray(v, np.float32) for v in (qx, qy, qr))
        lo, hi = (np.ascontiguousarray(v, np.int32) for v in (min_level, max_level))
        qd = np.ascontiguousarray(q_desc, np.uint8).reshape(-1, 32)
        nq = len(qx)
        ur = None if kp_uright is None else np.ascontiguousarray(kp_uright, np.float32)
        sig = None if inv_level_sigma2 is None else np.ascontiguousarray(inv_level_sigma2, np.float32)
        qu = None if q_ur is None else np.ascontiguousarray(q_ur, np.float32)
        g, keep = self._grid(grid)
        bi, bd = np.zeros(max(nq, 1), np.int32), np.zeros(max(nq, 1), np.int32)
        check(self._L.orbx_window_nearest(self._ctx, ptr(k), ptr(d), len(k), C.byref(g), ptr(ur), ptr(sig), 0 if sig is None else len(sig), ptr(qx),
                                          ptr(qy), ptr(qr), ptr(lo), ptr(hi), ptr(qu), ptr(qd), nq, ptr(bi), ptr(bd)), self._ctx)
        return bi[:nq], bd[:nq]

    def SearchByProjection(self, F, mp, th: float = 1.0):
        """ORBmatcher::SearchByProjection(Frame&, const vector<MapPoint*>&, th, ...) (src/ORBmatcher.cc:43-141), F.Nleft == -1.
        F: frame-like with mvKeysUn, mDescriptors, bounds, mvScaleFactors, mvuRight (or None), kp_obs (int32 [n], updated in
        place: Observations() of the bound map point, -1 = none).  mp: dict of arrays in_view, proj_x, proj_y, proj_xr (or
        None), view_cos, level, desc, obs.  Returns (nmatches, kp_match [n])."""
        k = np.ascontiguousarray(F.mvKeysUn)
        d = np.ascontiguousarray(F.mDescriptors, np.uint8).reshape(-1, 32)
        ur = None if getattr(F, "mvuRight", None) is None else np.ascontiguousarray(F.mvuRight, np.float32)
        assert F.kp_obs.dtype == np.int32 and F.kp_obs.flags["C_CONTIGUOUS"] and len(F.kp_obs) == len(k)
        sf = np.ascontiguousarray(F.mvScaleFactors, np.float32)
        inv = np.ascontiguousarray(mp["in_view"], np.uint8)
        px, py, vc = (np.ascontiguousarray(mp[key], np.float32) for key in ("proj_x", "proj_y", "view_cos"))
        pxr = None if mp.get("proj_xr") is None else np.ascontiguousarray(mp["proj_xr"], np.float32)
        lvl, obs = (np.ascontiguousarray(mp[key], np.int32) for key in ("level", "obs"))
        md = np.ascontiguousarray(mp["desc"], np.uint8).reshape(-1, 32)
        match = np.zeros(max(len(k), 1), np.int32)
        n = C.c_int(0)
        check(self._L.orbx_search_by_projection(self._ctx, ptr(k), ptr(d), ptr(ur), ptr(F.kp_obs), len(k), *[float(b) for b in F.bounds],
                                                ptr(sf), len(sf), ptr(inv), ptr(px), ptr(py), ptr(pxr), ptr(vc), ptr(lvl), ptr(md),
                                                ptr(obs), len(inv), float(th), self.mfNNratio, ptr(match), C.byref(n)), self._ctx)
        return n.value, match[:len(k)].copy()

    def SearchByProjectionLastFrame(self, F, lp, th: float, direction: int = 0, mbf: float = 0.0):
        """ORBmatcher::SearchByProjection(CurrentFrame, LastFrame, th, bMono) (src/ORBmatcher.cc:1676-1885), Nleft == -1, after
        the caller's projection step.  F as in SearchByProjection; lp: dict of arrays valid, u, v, invz, octave, angle, desc, obs
        (one entry per last-frame keypoint).  Returns (nmatches, kp_match [n]: -1 untouched / i bound / -2 cleared)."""
        k = np.ascontiguousarray(F.mvKeysUn)
        d = np.ascontiguousarray(F.mDescriptors, np.uint8).reshape(-1, 32)
        ur = None if getattr(F, "mvuRight", None) is None else np.ascontiguousarray(F.mvuRight, np.float32)
        assert F.kp_obs.dtype == np.int32 and F.kp_obs.flags["C_CONTIGUOUS"] and len(F.kp_obs) == len(k)
        sf = np.ascontiguousarray(F.mvScaleFactors, np.float32)
        valid = np.ascontiguousarray(lp["valid"], np.uint8)
        u, v, invz, ang = (np.ascontiguousarray(lp[key], np.float32) for key in ("u", "v", "invz", "angle"))
        octv, obs = (np.ascontiguousarray(lp[key], np.int32) for key in ("octave", "obs"))
        ld = np.ascontiguousarray(lp["desc"], np.uint8).reshape(-1, 32)
        match = np.zeros(max(len(k), 1), np.int32)
        n = C.c_int(0)
        check(self._L.orbx_search_by_projection_last(self._ctx, ptr(k), ptr(d), ptr(ur), ptr(F.kp_obs), len(k), *[float(b) for b in F.bounds],
                                                     ptr(sf), len(sf), float(mbf), ptr(valid), ptr(u), ptr(v), ptr(invz), ptr(octv), ptr(ang),
                                                     ptr(ld), ptr(obs), len(valid), float(th), int(direction), int(self.mbCheckOrientation),
                                                     ptr(match), C.byref(n)), self._ctx)
        return n.value, match[:len(k)].copy()

    def SearchByBoW(self, kf_desc, kf_angle, kf_valid, kf_fv, f_desc, f_angle, f_fv):
        """ORBmatcher::SearchByBoW(KeyFrame*, Frame&, vpMapPointMatches) (src/ORBmatcher.cc:223-425), single-camera.
        kf_fv / f_fv: DBoW2::FeatureVector as dict node -> list of feature indices.  Returns (nmatches, match_kf [nf])."""
        def csr(fv):
            nodes = sorted(fv)
            ptr_, idx = [0], []
            for k in nodes:
                idx.extend(fv[k]); ptr_.append(len(idx))
            return np.array(nodes, np.uint32), np.array(ptr_, np.int32), np.array(idx, np.uint32)
        kd = np.ascontiguousarray(kf_desc, np.uint8).reshape(-1, 32); fd = np.ascontiguousarray(f_desc, np.uint8).reshape(-1, 32)
        ka, fa = np.ascontiguousarray(kf_angle, np.float32), np.ascontiguousarray(f_angle, np.float32)
        kv = np.ascontiguousarray(kf_valid, np.uint8)
        kn, kp, ki = csr(kf_fv); fn, fp, fi = csr(f_fv)
        match = np.zeros(max(len(fd), 1), np.int32)
        n = C.c_int(0)
        check(self._L.orbx_search_by_bow(self._ctx, ptr(kd), ptr(ka), ptr(kv), len(kd), ptr(kn), ptr(kp), ptr(ki), len(kn), ptr(fd), ptr(fa),
                                         len(fd), ptr(fn), ptr(fp), ptr(fi), len(fn), self.mfNNratio, int(self.mbCheckOrientation),
                                         ptr(match), C.byref(n)), self._ctx)
        return n.value, match[:len(fd)].copy()

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


class SearchTarget:
    """A search target resident in HBM (orbx_target_*): uploaded once, searched many times; results equal those of
    ORBmatcher.WindowSearchGrid / WindowNearest on the same arrays."""

    def __init__(self, matcher: "ORBmatcher", kps, desc, grid, kp_uright=None, inv_level_sigma2=None):
        self._m = matcher           # keeps the context alive
        self._L = matcher._L
        self._ctx = matcher._ctx
        k = np.ascontiguousarray(kps)
        d = np.ascontiguousarray(desc, np.uint8).reshape(-1, 32)
        ur = None if kp_uright is None else np.ascontiguousarray(kp_uright, np.float32)
        sig = None if inv_level_sigma2 is None else np.ascontiguousarray(inv_level_sigma2, np.float32)
        g, keep = matcher._grid(grid)
        h = C.c_void_p()
        check(self._L.orbx_target_create(self._ctx, ptr(k), ptr(d), len(k), C.byref(g), ptr(ur), ptr(sig), 0 if sig is None else len(sig), C.byref(h)),
              self._ctx)
        self._h = h

    def assign(self, kps, desc, grid, kp_uright=None, inv_level_sigma2=None):
        """orbx_target_assign: new contents in the same device block.  A failed assign leaves the target INVALID (searches raise
        OrbxError) until a later assign succeeds."""
        k = np.ascontiguousarray(kps)
        d = np.ascontiguousarray(desc, np.uint8).reshape(-1, 32)
        ur = None if kp_uright is None else np.ascontiguousarray(kp_uright, np.float32)
        sig = None if inv_level_sigma2 is None else np.ascontiguousarray(inv_level_sigma2, np.float32)
        g, keep = self._m._grid(grid)
        check(self._L.orbx_target_assign(self._ctx, self._h, ptr(k), ptr(d), len(k), C.byref(g), ptr(ur), ptr(sig), 0 if sig is None else len(sig)),
              self._ctx)

    def close(self):
        if getattr(self, "_h", None):
            self._L.orbx_target_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __len__(self):
        return check(int(self._L.orbx_target_size(self._h)), self._ctx)

    def search(self, qx, qy, qr, min_level, max_level, q_desc, kp_skip=None, q_xr=None, want_lists=True):
        qx, qy, qr = (np.ascontiguousarray(v, np.float32) for v in (qx, qy, qr))
        lo, hi = (np.ascontiguousarray(v, np.int32) for v in (min_level, max_level))
        qd = np.ascontiguousarray(q_desc, np.uint8).reshape(-1, 32)
        nq = len(qx)
        skip = None if kp_skip is None else np.ascontiguousarray(kp_skip, np.uint8)
        xr = None if q_xr is None else np.ascontiguousarray(q_xr, np.float32)
        rp = np.zeros(nq + 1, np.int32)
        bi, bd, si, sd = (np.zeros(max(nq, 1), np.int32) for _ in range(4))
        cap = 1 << 12
        while True:
            cand, dist = (np.zeros(cap, np.int32), np.zeros(cap, np.int32)) if want_lists else (None, None)
            rc = self._L.orbx_target_search(self._ctx, self._h, ptr(skip), ptr(qx), ptr(qy), ptr(qr), ptr(lo), ptr(hi), ptr(qd), ptr(xr), nq, ptr(rp),
                                            ptr(cand), ptr(dist), cap if want_lists else 0, ptr(bi), ptr(bd), ptr(si), ptr(sd))
            if rc == -4 and want_lists:   # ORBX_E_CAPACITY: row_ptr[nq] is the capacity needed
                cap = int(rp[nq]) + 16
                continue
            check(rc, self._ctx)
            out = dict(row_ptr=rp, best_idx=bi[:nq], best_dist=bd[:nq], second_idx=si[:nq], second_dist=sd[:nq])
            if want_lists:
                out.update(cand=cand[:rc].copy(), dist=dist[:rc].copy())
            return out

    def search_view(self, qx, qy, qr, min_level, max_level, q_desc, kp_skip=None, q_xr=None, copy=True):
        """orbx_target_search_view: the candidate lists read where the kernel left them (the call's pinned output, valid until the
        context's SECOND next view call — copied here unless copy=False, which hands out arrays over the blob itself).  Returns (spans [nq] structured: start, count, best_idx, best_dist, second_idx, second_dist;
        pool [total] structured: idx, dist)."""
        qx, qy, qr = (np.ascontiguousarray(v, np.float32) for v in (qx, qy, qr))
        lo, hi = (np.ascontiguousarray(v, np.int32) for v in (min_level, max_level))
        qd = np.ascontiguousarray(q_desc, np.uint8).reshape(-1, 32)
        nq = len(qx)
        skip = None if kp_skip is None else np.ascontiguousarray(kp_skip, np.uint8)
        xr = None if q_xr is None else np.ascontiguousarray(q_xr, np.float32)
        sp, pl = C.c_void_p(), C.c_void_p()
        total = check(int(self._L.orbx_target_search_view(self._ctx, self._h, ptr(skip), ptr(qx), ptr(qy), ptr(qr), ptr(lo), ptr(hi), ptr(qd), ptr(xr), nq,
                                                          C.byref(sp), C.byref(pl))), self._ctx)
        span_t = np.dtype([("start", "<i4"), ("count", "<i4"), ("best_idx", "<i4"), ("best_dist", "<i4"), ("second_idx", "<i4"), ("second_dist", "<i4"),
                           ("reserved0", "<i4"), ("reserved1", "<i4")])
        cand_t = np.dtype([("idx", "<i4"), ("dist", "<i4")])
        if nq == 0 or not sp.value:
            return np.zeros(0, span_t), np.zeros(0, cand_t)
        spans = np.frombuffer((C.c_char * (span_t.itemsize * nq)).from_address(sp.value), span_t)
        end = int((spans["start"] + spans["count"]).max()) if nq else 0
        pool = np.frombuffer((C.c_char * (cand_t.itemsize * max(end, 1))).from_address(pl.value), cand_t)[:end] if end and pl.value else np.zeros(0, cand_t)
        if copy:
            spans, pool = spans.copy(), pool.copy()
        assert int(spans["count"].sum()) == total
        return spans, pool

    def search_view_begin(self, qx, qy, qr, min_level, max_level, q_desc, kp_skip=None, q_xr=None):
        """orbx_target_search_view_begin: the call issued, nothing waited for.  Returns a ticket for search_view_end (it keeps the query arrays
        alive: the C ABI wants them valid and unchanged until _end)."""
        qx, qy, qr = (np.ascontiguousarray(v, np.float32) for v in (qx, qy, qr))
        lo, hi = (np.ascontiguousarray(v, np.int32) for v in (min_level, max_level))
        qd = np.ascontiguousarray(q_desc, np.uint8).reshape(-1, 32)
        skip = None if kp_skip is None else np.ascontiguousarray(kp_skip, np.uint8)
        xr = None if q_xr is None else np.ascontiguousarray(q_xr, np.float32)
        slot = check(int(self._L.orbx_target_search_view_begin(self._ctx, self._h, ptr(skip), ptr(qx), ptr(qy), ptr(qr), ptr(lo), ptr(hi), ptr(qd), ptr(xr), len(qx))),
                     self._ctx)
        return (slot, len(qx), (qx, qy, qr, lo, hi, qd, skip, xr))

    def search_view_cancel(self, ticket):
        """orbx_target_search_view_cancel: gives back the slot of a ticket that search_view_end will never see."""
        check(int(self._L.orbx_target_search_view_cancel(self._ctx, int(ticket[0]))), self._ctx)

    def search_view_end(self, ticket, copy=True):
        """orbx_target_search_view_end -> (spans, pool) as search_view returns them."""
        slot, nq, _keep = ticket
        sp, pl = C.c_void_p(), C.c_void_p()
        total = check(int(self._L.orbx_target_search_view_end(self._ctx, slot, C.byref(sp), C.byref(pl))), self._ctx)
        span_t = np.dtype([("start", "<i4"), ("count", "<i4"), ("best_idx", "<i4"), ("best_dist", "<i4"), ("second_idx", "<i4"), ("second_dist", "<i4"),
                           ("reserved0", "<i4"), ("reserved1", "<i4")])
        cand_t = np.dtype([("idx", "<i4"), ("dist", "<i4")])
        if nq == 0 or not sp.value:
            return np.zeros(0, span_t), np.zeros(0, cand_t)
        spans = np.frombuffer((C.c_char * (span_t.itemsize * nq)).from_address(sp.value), span_t)
        end = int((spans["start"] + spans["count"]).max())
        pool = np.frombuffer((C.c_char * (cand_t.itemsize * max(end, 1))).from_address(pl.value), cand_t)[:end] if end and pl.value else np.zeros(0, cand_t)
        assert int(spans["count"].sum()) == total
        return (spans.copy(), pool.copy()) if copy else (spans, pool)

    def nearest(self, qx, qy, qr, min_level, max_level, q_desc, q_ur=None):
        qx, qy, qr = (np.ascontiguousarray(v, np.float32) for v in (qx, qy, qr))
        lo, hi = (np.ascontiguousarray(v, np.int32) for v in (min_level, max_level))
        qd = np.ascontiguousarray(q_desc, np.uint8).reshape(-1, 32)
        nq = len(qx)
        qu = None if q_ur is None else np.ascontiguousarray(q_ur, np.float32)
        bi, bd = np.zeros(max(nq, 1), np.int32), np.zeros(max(nq, 1), np.int32)
        check(self._L.orbx_target_nearest(self._ctx, self._h, 0 if qu is None else 1, ptr(qx), ptr(qy), ptr(qr), ptr(lo), ptr(hi), ptr(qu), ptr(qd), nq,
                                          ptr(bi), ptr(bd)), self._ctx)
        return bi[:nq], bd[:nq]
