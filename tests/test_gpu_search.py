"""Frame grid queries and ORBmatcher::SearchForInitialization on the GPU against the oracle (src/Frame.cc:385-416,
:657-735; src/ORBmatcher.cc:648-763): identical candidate lists (order included) and identical matches."""
import numpy as np
import pytest

from oracle import pyoracle as po
from orb_slam3_modified_amd import ORBextractor, ORBmatcher, OrbxError, synth

pytestmark = pytest.mark.gpu


class F:  # the members of ORB_SLAM3::Frame the routine touches
    def __init__(self, kps, desc, bounds):
        self.mvKeysUn, self.mDescriptors, self.bounds = kps, desc, bounds


@pytest.fixture(scope="module")
def frames():
    gpu = ORBextractor(5000, 1.2, 8, 20, 7)          # mpIniORBextractor: 5 x nFeatures (src/Tracking.cc:603)
    out = [gpu(f, None, (0, 1000)) for f in synth.make_stream(3)]
    return gpu, [F(k, d, (0.0, 0.0, 640.0, 480.0)) for _, k, d in out]


def test_features_in_area_matches_reference_order(frames):
    gpu, fr = frames
    m = ORBmatcher(gpu)
    k2 = fr[1].mvKeysUn
    rng = np.random.default_rng(3)
    nq = 700
    qx = rng.uniform(-80, 720, nq).astype(np.float32); qy = rng.uniform(-60, 540, nq).astype(np.float32)
    qr = rng.choice([0.0, 3.0, 10.0, 15.5, 40.0, 100.0, 1000.0], nq).astype(np.float32)
    lo = rng.choice([-1, 0, 1, 3], nq).astype(np.int32); hi = rng.choice([-1, 0, 2, 7], nq).astype(np.int32)
    qx[:20] = k2["x"][:20]; qy[:20] = k2["y"][:20]          # exact hits: strict `<` on the window edge
    for bounds in ((0.0, 0.0, 640.0, 480.0), (-12.5, -7.25, 655.0, 490.5)):   # undistorted bounds may exceed the image
        rp, cand = m.GetFeaturesInArea(k2, bounds, qx, qy, qr, lo, hi)
        orp, ocand = po.features_in_area(k2, bounds, qx, qy, qr, lo, hi)
        assert np.array_equal(rp, orp) and np.array_equal(cand, ocand)
        assert rp[-1] > 10000
    rp, cand = m.GetFeaturesInArea(k2[:0], (0, 0, 640, 480), qx[:5], qy[:5], qr[:5], lo[:5], hi[:5])     # empty frame
    assert rp.tolist() == [0] * 6 and len(cand) == 0
    rp, cand = m.GetFeaturesInArea(k2, (0, 0, 640, 480), qx[:0], qy[:0], qr[:0], lo[:0], hi[:0])         # no query
    assert rp.tolist() == [0] and len(cand) == 0


@pytest.mark.parametrize("window,ratio,ori", [(100, 0.9, True), (100, 0.9, False), (30, 0.6, True), (200, 0.9, True)])
def test_search_for_initialization_equals_oracle(frames, window, ratio, ori):
    gpu, fr = frames
    m = ORBmatcher(gpu, ratio, ori)
    F1, F2, F3 = fr
    prev = np.stack([F1.mvKeysUn["x"], F1.mvKeysUn["y"]], 1).astype(np.float32).copy()     # src/Tracking.cc:2470-2472
    oprev = prev.copy()
    for Fb in (F2, F3):                       # chained calls: vbPrevMatched carries over (src/Tracking.cc:2495)
        n, m12 = m.SearchForInitialization(F1, Fb, prev, window)
        on, om12, oprev = po.search_for_initialization(F1.mvKeysUn, F1.mDescriptors, Fb.mvKeysUn, Fb.mDescriptors, Fb.bounds,
                                                       oprev, window, ratio, ori)
        assert n == on and np.array_equal(m12, om12) and prev.tobytes() == oprev.tobytes()
        assert n == (m12 >= 0).sum()
    assert n > 50


def test_search_for_initialization_degenerate(frames):
    gpu, fr = frames
    m = ORBmatcher(gpu, 0.9, True)
    F1 = fr[0]
    empty = F(F1.mvKeysUn[:0], F1.mDescriptors[:0], F1.bounds)
    prev = np.stack([F1.mvKeysUn["x"], F1.mvKeysUn["y"]], 1).astype(np.float32).copy()
    n, m12 = m.SearchForInitialization(F1, empty, prev, 100)
    assert n == 0 and (m12 == -1).all()
    # identical frames: every level-0 keypoint matches itself at distance 0
    n, m12 = m.SearchForInitialization(F1, F1, prev, 100)
    on, om12, _ = po.search_for_initialization(F1.mvKeysUn, F1.mDescriptors, F1.mvKeysUn, F1.mDescriptors, F1.bounds, prev, 100, 0.9, True)
    assert n == on and np.array_equal(m12, om12)
    lvl0 = F1.mvKeysUn["octave"] == 0
    assert (m12[~lvl0] == -1).all() and (m12[lvl0] >= 0).sum() > 0.9 * lvl0.sum()


@pytest.mark.parametrize("disp,noise", [(12, 2), (37, 0), (3, 4)])
def test_compute_stereo_matches_equals_oracle(disp, noise):
    """Frame::ComputeStereoMatches on the device pyramids of two extractor instances (stereo: src/Frame.cc:122-141)."""
    rng = np.random.default_rng(disp)
    L = synth.make_stream(1, 480, 752)[0]
    R = np.roll(L, -disp, axis=1).copy()
    if noise:
        R = np.clip(R.astype(np.int32) + rng.integers(-noise, noise + 1, R.shape), 0, 255).astype(np.uint8)
    exL, exR = ORBextractor(1200, 1.2, 8, 20, 7), ORBextractor(1200, 1.2, 8, 20, 7)
    _, kL, dL = exL(L, None, (0, 0))
    _, kR, dR = exR(R, None, (0, 0))
    mb, mbf = 0.11, 47.90639384423901          # EuRoC: baseline, baseline * fx (Examples/Stereo/EuRoC.yaml)
    ur, dp, kept = ORBmatcher.ComputeStereoMatches(exL, exR, kL, dL, kR, dR, mb, mbf)
    our, odp, okept = po.stereo_matches(kL, dL, kR, dR, exL.mvImagePyramid, exR.mvImagePyramid, exL.GetScaleFactors(),
                                        exL.GetInverseScaleFactors(), mb, mbf)
    assert kept == okept and ur.tobytes() == our.tobytes() and dp.tobytes() == odp.tobytes()
    m = ur >= 0
    assert m.sum() == kept and kept > 200
    assert abs(np.median((kL["x"] - ur)[m]) - disp) < 0.5


def test_compute_stereo_matches_degenerate():
    L = synth.make_stream(1, 480, 752)[0]
    exL, exR = ORBextractor(1200, 1.2, 8, 20, 7), ORBextractor(1200, 1.2, 8, 20, 7)
    _, kL, dL = exL(L, None, (0, 0))
    _, kR, dR = exR(np.ascontiguousarray(L[:, ::-1]), None, (0, 0))        # mirrored right image: (almost) nothing matches
    ur, dp, kept = ORBmatcher.ComputeStereoMatches(exL, exR, kL, dL, kR, dR, 0.11, 47.9)
    our, odp, okept = po.stereo_matches(kL, dL, kR, dR, exL.mvImagePyramid, exR.mvImagePyramid, exL.GetScaleFactors(),
                                        exL.GetInverseScaleFactors(), 0.11, 47.9)
    assert kept == okept and ur.tobytes() == our.tobytes() and dp.tobytes() == odp.tobytes()
    ur, dp, kept = ORBmatcher.ComputeStereoMatches(exL, exR, kL, dL, kR[:0], dR[:0], 0.11, 47.9)
    assert kept == 0 and (ur == -1).all() and (dp == -1).all()


def _map_points(Fa, Fb, rng, stereo):
    """Local-map stand-in: the keypoints of frame A seen as map points projected into frame B (small drift + noise)."""
    ka = Fa.mvKeysUn
    n = len(ka)
    mp = dict(
        in_view=(rng.random(n) < 0.9).astype(np.uint8),
        proj_x=(ka["x"] + 1.5 + rng.normal(0, 1.0, n)).astype(np.float32),
        proj_y=(ka["y"] + 0.5 + rng.normal(0, 1.0, n)).astype(np.float32),
        view_cos=rng.choice([0.9, 0.9979, 0.998, 0.9981, 1.0], n).astype(np.float32),
        level=np.clip(ka["octave"] + rng.integers(-1, 2, n), 0, 7).astype(np.int32),
        desc=Fa.mDescriptors.copy(),
        obs=rng.choice([0, 0, 1, 3, 7], n).astype(np.int32),
    )
    mp["proj_xr"] = (mp["proj_x"] - rng.uniform(0.5, 30, n)).astype(np.float32) if stereo else None
    return mp


@pytest.mark.parametrize("th,ratio,stereo", [(1.0, 0.8, False), (3.0, 0.8, False), (1.0, 0.8, True), (5.0, 0.6, True), (15.0, 0.9, False)])
def test_search_by_projection_equals_oracle(frames, th, ratio, stereo):
    gpu, fr = frames
    m = ORBmatcher(gpu, ratio, True)
    rng = np.random.default_rng(int(th * 10) + stereo)
    sf = gpu.GetScaleFactors()
    total = 0
    for Fa, Fb in ((fr[0], fr[1]), (fr[1], fr[2])):
        mp = _map_points(Fa, Fb, rng, stereo)
        nb = len(Fb.mvKeysUn)
        kp_obs = rng.choice([-1, -1, -1, 0, 2], nb).astype(np.int32)            # some keypoints are already bound
        u_right = np.where(rng.random(nb) < 0.6, Fb.mvKeysUn["x"] - rng.uniform(0.5, 30, nb), -1.0).astype(np.float32) if stereo else None
        Fb.mvScaleFactors, Fb.mvuRight, Fb.kp_obs = sf, u_right, kp_obs.copy()
        n, match = m.SearchByProjection(Fb, mp, th)
        on, omatch, oobs = po.search_by_projection(Fb.mvKeysUn, Fb.mDescriptors, Fb.bounds, sf, kp_obs, mp, th, ratio, u_right)
        assert n == on and np.array_equal(match, omatch) and np.array_equal(Fb.kp_obs, oobs)
        assert n >= (match >= 0).sum()       # a keypoint bound to an unobserved map point may be re-bound (counted twice)
        total += n
    assert total > 200


def test_window_search_gates_and_best_second(frames):
    gpu, fr = frames
    m = ORBmatcher(gpu)
    Fa, Fb = fr[0], fr[1]
    k, d = Fb.mvKeysUn, Fb.mDescriptors
    rng = np.random.default_rng(11)
    nq = min(len(Fa.mvKeysUn), 1500)
    qx, qy = Fa.mvKeysUn["x"][:nq].copy(), Fa.mvKeysUn["y"][:nq].copy()
    qr = rng.choice([4.0, 12.0, 30.0], nq).astype(np.float32)
    lo = np.maximum(Fa.mvKeysUn["octave"][:nq] - 1, 0).astype(np.int32); hi = (lo + 1).astype(np.int32)
    skip = (rng.random(len(k)) < 0.3).astype(np.uint8)
    ur = np.where(rng.random(len(k)) < 0.5, k["x"] - 5.0, -1.0).astype(np.float32)
    qxr = (qx - rng.uniform(0, 12, nq)).astype(np.float32)
    w = m.WindowSearch(k, d, Fb.bounds, qx, qy, qr, lo, hi, Fa.mDescriptors[:nq], skip, ur, qxr)
    # oracle: plain window lists, then the gates and the scan in the reference's order
    orp, ocand = po.features_in_area(k, Fb.bounds, qx, qy, qr, lo, hi)
    rp, cand = [0], []
    for q in range(nq):
        for c in ocand[orp[q]:orp[q + 1]]:
            if skip[c]:
                continue
            if ur[c] > 0 and abs(np.float32(qxr[q] - ur[c])) > qr[q]:
                continue
            cand.append(c)
        rp.append(len(cand))
    assert np.array_equal(w["row_ptr"], np.array(rp, np.int32)) and np.array_equal(w["cand"], np.array(cand, np.int32))
    bi, bd, si, sd, do = po.nn_csr(Fa.mDescriptors[:nq], d, w["row_ptr"], w["cand"])
    assert np.array_equal(w["dist"], do)
    assert np.array_equal(w["best_idx"], bi) and np.array_equal(w["best_dist"], bd)
    assert np.array_equal(w["second_idx"], si) and np.array_equal(w["second_dist"], sd)
    assert len(cand) > 2000


@pytest.mark.parametrize("th,direction,stereo,ori", [(15.0, 0, False, True), (7.0, 0, True, True), (15.0, 1, True, True), (15.0, 2, True, False),
                                                     (30.0, 0, False, True)])
def test_search_by_projection_last_frame_equals_oracle(frames, th, direction, stereo, ori):
    """Tracking::TrackWithMotionModel's call (src/Tracking.cc:2889,2897): last frame's points projected into the current one."""
    gpu, fr = frames
    m = ORBmatcher(gpu, 0.9, ori)
    rng = np.random.default_rng(int(th) * 8 + direction * 2 + stereo)
    sf = gpu.GetScaleFactors()
    total = 0
    for Fl, Fc in ((fr[0], fr[1]), (fr[1], fr[2])):
        kl = Fl.mvKeysUn
        nl, nc = len(kl), len(Fc.mvKeysUn)
        lp = dict(valid=(rng.random(nl) < 0.85).astype(np.uint8), u=(kl["x"] + 1.5 + rng.normal(0, 2.0, nl)).astype(np.float32),
                  v=(kl["y"] + 0.5 + rng.normal(0, 2.0, nl)).astype(np.float32), invz=rng.uniform(0.02, 1.0, nl).astype(np.float32),
                  octave=kl["octave"].astype(np.int32), angle=(kl["angle"] + rng.choice([0.0, 0.0, 0.0, 45.0, 200.0], nl)).astype(np.float32) % np.float32(360),
                  desc=Fl.mDescriptors, obs=rng.choice([0, 1, 4], nl).astype(np.int32))
        kp_obs = rng.choice([-1, -1, -1, 0, 2], nc).astype(np.int32)
        u_right = np.where(rng.random(nc) < 0.6, Fc.mvKeysUn["x"] - rng.uniform(0.5, 40, nc), -1.0).astype(np.float32) if stereo else None
        Fc.mvScaleFactors, Fc.mvuRight, Fc.kp_obs = sf, u_right, kp_obs.copy()
        n, match = m.SearchByProjectionLastFrame(Fc, lp, th, direction, mbf=38.5)
        on, omatch, oobs = po.search_by_projection_last(Fc.mvKeysUn, Fc.mDescriptors, Fc.bounds, sf, kp_obs, lp, th, direction, ori, u_right, 38.5)
        assert n == on and np.array_equal(match, omatch) and np.array_equal(Fc.kp_obs, oobs)
        total += n
        if ori:
            assert (match == -2).sum() > 0      # the rotation filter removed something
    assert total > 200


def test_frame_grid_with_twenty_thousand_keypoints():
    """The fork's Examples/Monocular/mi.yaml extracts 20 000 features on one level: the frame grid sorts them in LDS (128 KiB)."""
    rng = np.random.default_rng(21)
    img = rng.integers(0, 256, (800, 600)).astype(np.uint8)
    ex = ORBextractor(20000, 1.2, 1, 20, 7)
    _, k, d = ex(img, None, (0, 0))
    assert 16384 < len(k) <= 32768
    m = ORBmatcher(ex)
    nq = 500
    qx = rng.uniform(0, 600, nq).astype(np.float32); qy = rng.uniform(0, 800, nq).astype(np.float32)
    qr = rng.choice([5.0, 15.0, 40.0], nq).astype(np.float32)
    lo = np.full(nq, -1, np.int32); hi = np.full(nq, -1, np.int32)
    bounds = (0.0, 0.0, 600.0, 800.0)
    rp, cand = m.GetFeaturesInArea(k, bounds, qx, qy, qr, lo, hi)
    orp, ocand = po.features_in_area(k, bounds, qx, qy, qr, lo, hi)
    assert np.array_equal(rp, orp) and np.array_equal(cand, ocand) and len(cand) > 20000


@pytest.mark.parametrize("ratio,ori,levelsup", [(0.7, True, 2), (0.75, False, 2), (0.9, True, 3), (0.7, True, 0)])
def test_search_by_bow_equals_oracle(frames, tmp_path, ratio, ori, levelsup):
    """TrackReferenceKeyFrame / Relocalization's matcher (src/ORBmatcher.cc:223-425): features of the same vocabulary node."""
    from orb_slam3_modified_amd import ORBVocabulary
    from tests.vocab_util import make_vocabulary
    gpu, fr = frames
    rng = np.random.default_rng(int(ratio * 100) + levelsup)
    vp = str(tmp_path / "voc.txt")
    make_vocabulary(vp, np.concatenate([f.mDescriptors for f in fr]), 10, 4, seed=5)
    voc = ORBVocabulary(gpu)
    assert voc.loadFromTextFile(vp)
    m = ORBmatcher(gpu, ratio, ori)
    total = 0
    for KF, F_ in ((fr[0], fr[1]), (fr[2], fr[1])):
        kfv = voc.transform(KF.mDescriptors, levelsup)[1]
        ffv = voc.transform(F_.mDescriptors, levelsup)[1]
        valid = (rng.random(len(KF.mvKeysUn)) < 0.7).astype(np.uint8)         # map point present and not bad
        n, match = m.SearchByBoW(KF.mDescriptors, KF.mvKeysUn["angle"], valid, kfv, F_.mDescriptors, F_.mvKeysUn["angle"], ffv)
        on, omatch = po.search_by_bow(KF.mDescriptors, KF.mvKeysUn["angle"], valid, kfv, F_.mDescriptors, F_.mvKeysUn["angle"], ffv, ratio, ori)
        assert n == on and np.array_equal(match, omatch) and n == (match >= 0).sum()
        assert valid[match[match >= 0]].all()
        total += n
    assert total > 100
    # nothing valid / empty inputs
    n, match = m.SearchByBoW(KF.mDescriptors, KF.mvKeysUn["angle"], np.zeros(len(KF.mvKeysUn), np.uint8), kfv, F_.mDescriptors,
                             F_.mvKeysUn["angle"], ffv)
    assert n == 0 and (match == -1).all()
    n, match = m.SearchByBoW(KF.mDescriptors, KF.mvKeysUn["angle"], valid, {}, F_.mDescriptors, F_.mvKeysUn["angle"], ffv)
    assert n == 0
    with pytest.raises(Exception):     # feature indices beyond the descriptor matrix are refused, not read
        m.SearchByBoW(KF.mDescriptors[:10], KF.mvKeysUn["angle"][:10], valid[:10], kfv, F_.mDescriptors, F_.mvKeysUn["angle"], ffv)


def _kf_grid(kps, fbounds, truncate):
    """A KeyFrame's grid: assigned with the Frame's float bounds (Frame::AssignFeaturesToGrid), queried with the bounds the
    KeyFrame keeps — truncated to int (include/KeyFrame.h:403-406) — and the Frame's cell sizes."""
    mnx, mny, mxx, mxy = (np.float32(b) for b in fbounds)
    inv_w = np.float32(64) / np.float32(mxx - mnx); inv_h = np.float32(48) / np.float32(mxy - mny)
    cs, ci = po.assign_grid(kps, mnx, mny, inv_w, inv_h)
    qmx, qmy = (float(int(mnx)), float(int(mny))) if truncate else (float(mnx), float(mny))
    return dict(min_x=qmx, min_y=qmy, inv_w=float(inv_w), inv_h=float(inv_h), cell_start=cs, cell_idx=ci)


@pytest.mark.parametrize("fbounds,truncate,held", [((0.0, 0.0, 640.0, 480.0), False, True), ((-12.7, -9.4, 653.2, 488.9), True, True),
                                                   ((-12.7, -9.4, 653.2, 488.9), False, False)])
def test_window_search_grid_equals_oracle(frames, fbounds, truncate, held):
    _window_search_grid_case(frames, fbounds, truncate, held, 1)


def test_window_search_grid_with_copies_and_synchronisation(frames):
    """The same pass with `window_direct` off: input copy, result copy and a stream synchronisation instead of mapped pinned memory + polling."""
    _window_search_grid_case(frames, (0.0, 0.0, 640.0, 480.0), False, True, 0)


def _window_search_grid_case(frames, fbounds, truncate, held, direct):
    """orbx_window_search_grid (the device pass of the KeyFrame-side routines and the two-camera blocks): candidate lists in
    the reference's order, every distance, best / second — over a grid the caller holds, and over one assigned on the device."""
    gpu, fr = frames
    m = ORBmatcher(gpu)
    m.set_option("window_direct", direct)
    k2, d2, d1 = fr[1].mvKeysUn, fr[1].mDescriptors, fr[0].mDescriptors
    rng = np.random.default_rng(11)
    nq = min(len(d1), 900)
    src = rng.integers(0, len(k2), nq)
    qx = (k2["x"][src] + rng.normal(0, 4, nq)).astype(np.float32); qy = (k2["y"][src] + rng.normal(0, 4, nq)).astype(np.float32)
    qr = rng.choice([0.0, 2.5, 7.0, 15.0, 36.0, 120.0], nq).astype(np.float32)
    lo = rng.choice([-1, 0, 1, 3], nq).astype(np.int32); hi = rng.choice([-1, 0, 2, 7], nq).astype(np.int32)
    grid = _kf_grid(k2, fbounds, truncate)
    if not held:
        grid = dict(grid, cell_start=None, cell_idx=None)
    skip = (rng.random(len(k2)) < 0.2).astype(np.uint8)
    ur = np.where(rng.random(len(k2)) < 0.3, -1.0, k2["x"] - rng.uniform(1, 40, len(k2))).astype(np.float32)
    xr = (qx - rng.uniform(1, 40, nq)).astype(np.float32)
    for kw in (dict(), dict(kp_skip=skip), dict(kp_skip=skip, kp_uright=ur, q_xr=xr)):
        got = m.WindowSearchGrid(k2, d2, grid, qx, qy, qr, lo, hi, d1[:nq], **kw)
        want = po.window_search_grid(k2, d2, grid, qx, qy, qr, lo, hi, d1[:nq], **kw)
        for key in ("row_ptr", "cand", "dist", "best_idx", "best_dist", "second_idx", "second_dist"):
            assert np.array_equal(got[key], want[key]), (key, kw.keys())
        assert got["row_ptr"][-1] > 5000
        best_only = m.WindowSearchGrid(k2, d2, grid, qx, qy, qr, lo, hi, d1[:nq], want_lists=False, **kw)
        assert np.array_equal(best_only["best_idx"], want["best_idx"]) and np.array_equal(best_only["second_dist"], want["second_dist"])
        assert np.array_equal(best_only["row_ptr"], want["row_ptr"])
    # empty sides
    e = m.WindowSearchGrid(k2[:0], d2[:0], dict(grid, cell_start=None, cell_idx=None), qx[:3], qy[:3], qr[:3], lo[:3], hi[:3], d1[:3])
    assert e["row_ptr"].tolist() == [0, 0, 0, 0] and e["best_idx"].tolist() == [-1, -1, -1] and e["best_dist"].tolist() == [256] * 3
    e = m.WindowSearchGrid(k2, d2, grid, qx[:0], qy[:0], qr[:0], lo[:0], hi[:0], d1[:0])
    assert e["row_ptr"].tolist() == [0] and len(e["cand"]) == 0
    m.set_option("window_direct", 1)


@pytest.mark.parametrize("chi2", [False, True])
def test_window_nearest_equals_oracle(frames, chi2):
    """orbx_window_nearest: the arg-min Fuse x2 / SearchBySim3 take from the device, with Fuse's reprojection gate
    (src/ORBmatcher.cc:1269-1296: stereo 7.8 / monocular 5.99, float products compared in double)."""
    gpu, fr = frames
    m = ORBmatcher(gpu)
    k2, d2, d1 = fr[1].mvKeysUn, fr[1].mDescriptors, fr[0].mDescriptors
    rng = np.random.default_rng(12)
    nq = min(len(d1), 2000)
    src = rng.integers(0, len(k2), nq)
    qx = (k2["x"][src] + rng.normal(0, 2.0, nq)).astype(np.float32); qy = (k2["y"][src] + rng.normal(0, 2.0, nq)).astype(np.float32)
    lvl = k2["octave"][src].astype(np.int32)
    sf = np.float32(1.2) ** np.arange(8, dtype=np.float32)
    qr = (np.float32(3.0) * sf[lvl]).astype(np.float32)
    inv_sigma2 = (1.0 / (sf * sf)).astype(np.float32)
    ur = np.where(rng.random(len(k2)) < 0.4, -1.0, k2["x"] - rng.uniform(1, 40, len(k2))).astype(np.float32)
    q_ur = np.where(ur[src] >= 0, ur[src] + rng.normal(0, 1.5, nq), qx - 5).astype(np.float32)
    grid = _kf_grid(k2, (-12.7, -9.4, 653.2, 488.9), True)
    kw = dict(kp_uright=ur, inv_level_sigma2=inv_sigma2, q_ur=q_ur) if chi2 else {}
    bi, bd = m.WindowNearest(k2, d2, grid, qx, qy, qr, lvl - 1, lvl, d1[:nq], **kw)
    obi, obd = po.window_nearest(k2, d2, grid, qx, qy, qr, lvl - 1, lvl, d1[:nq], **kw)
    assert np.array_equal(bi, obi) and np.array_equal(bd, obd)
    assert (bi >= 0).sum() > nq // 3
    if chi2:   # the gate must bite: without it more queries find a candidate
        bi0, _ = po.window_nearest(k2, d2, grid, qx, qy, qr, lvl - 1, lvl, d1[:nq])
        assert (bi0 >= 0).sum() > (obi >= 0).sum()


@pytest.mark.parametrize("held", [True, False])
@pytest.mark.parametrize("direct", [1, 0])
def test_resident_target_searches_equal_oracle(frames, held, direct):
    """orbx_target_*: the frame uploaded once, searched repeatedly with different queries / skip flags / gates — every search equals the
    oracle (and therefore the host-buffer entry points), in the single-kernel direct mode and with copies + synchronisation."""
    gpu, fr = frames
    m = ORBmatcher(gpu)
    m.set_option("window_direct", direct)
    k2, d2, d1 = fr[1].mvKeysUn, fr[1].mDescriptors, fr[0].mDescriptors
    rng = np.random.default_rng(21)
    grid = _kf_grid(k2, (-12.7, -9.4, 653.2, 488.9), True)
    if not held:
        grid = dict(grid, cell_start=None, cell_idx=None)
    sf = np.float32(1.2) ** np.arange(8, dtype=np.float32)
    inv_sigma2 = (1.0 / (sf * sf)).astype(np.float32)
    ur = np.where(rng.random(len(k2)) < 0.3, -1.0, k2["x"] - rng.uniform(1, 40, len(k2))).astype(np.float32)
    tgt = m.Target(k2, d2, grid, kp_uright=ur, inv_level_sigma2=inv_sigma2)
    plain = m.Target(k2, d2, grid)
    assert len(tgt) == len(k2) == len(plain)
    for rep in range(4):
        nq = int(rng.integers(300, min(len(d1), 900)))
        src = rng.integers(0, len(k2), nq)
        qx = (k2["x"][src] + rng.normal(0, 4, nq)).astype(np.float32); qy = (k2["y"][src] + rng.normal(0, 4, nq)).astype(np.float32)
        qr = rng.choice([0.0, 2.5, 7.0, 15.0, 36.0, 120.0], nq).astype(np.float32)
        lo = rng.choice([-1, 0, 1, 3], nq).astype(np.int32); hi = rng.choice([-1, 0, 2, 7], nq).astype(np.int32)
        skip = (rng.random(len(k2)) < 0.2).astype(np.uint8)
        xr = (qx - rng.uniform(1, 40, nq)).astype(np.float32)
        qd = d1[rng.integers(0, len(d1), nq)]
        for T, kw, okw in ((plain, dict(), dict()), (tgt, dict(kp_skip=skip), dict(kp_skip=skip)),
                           (tgt, dict(kp_skip=skip, q_xr=xr), dict(kp_skip=skip, kp_uright=ur, q_xr=xr))):
            got = T.search(qx, qy, qr, lo, hi, qd, **kw)
            want = po.window_search_grid(k2, d2, grid, qx, qy, qr, lo, hi, qd, **okw)
            for key in ("row_ptr", "cand", "dist", "best_idx", "best_dist", "second_idx", "second_dist"):
                assert np.array_equal(got[key], want[key]), (rep, key, kw.keys())
            best_only = T.search(qx, qy, qr, lo, hi, qd, want_lists=False, **kw)
            assert np.array_equal(best_only["best_idx"], want["best_idx"]) and np.array_equal(best_only["row_ptr"], want["row_ptr"])
        lvl = k2["octave"][src].astype(np.int32)
        q_ur = np.where(ur[src] >= 0, ur[src] + rng.normal(0, 1.5, nq), qx - 5).astype(np.float32)
        r3 = (np.float32(3.0) * sf[lvl]).astype(np.float32)
        bi, bd = tgt.nearest(qx, qy, r3, lvl - 1, lvl, qd, q_ur=q_ur)
        obi, obd = po.window_nearest(k2, d2, grid, qx, qy, r3, lvl - 1, lvl, qd, kp_uright=ur, inv_level_sigma2=inv_sigma2, q_ur=q_ur)
        assert np.array_equal(bi, obi) and np.array_equal(bd, obd)
        bi, bd = plain.nearest(qx, qy, r3, lvl - 1, lvl, qd)
        obi, obd = po.window_nearest(k2, d2, grid, qx, qy, r3, lvl - 1, lvl, qd)
        assert np.array_equal(bi, obi) and np.array_equal(bd, obd)
    # a target without kp_uright refuses the gates that need it; empty queries and empty targets behave like the host-buffer calls
    with pytest.raises(OrbxError):
        plain.search(qx, qy, qr, lo, hi, qd, q_xr=xr)
    with pytest.raises(OrbxError):
        plain.nearest(qx, qy, qr, lo, hi, qd, q_ur=xr)
    e = tgt.search(qx[:0], qy[:0], qr[:0], lo[:0], hi[:0], qd[:0])
    assert e["row_ptr"].tolist() == [0] and len(e["cand"]) == 0
    empty = m.Target(k2[:0], d2[:0], dict(grid, cell_start=None, cell_idx=None))
    e = empty.search(qx[:3], qy[:3], qr[:3], lo[:3], hi[:3], qd[:3])
    assert e["row_ptr"].tolist() == [0, 0, 0, 0] and e["best_idx"].tolist() == [-1, -1, -1]
    for T in (tgt, plain, empty):
        T.close()
    m.set_option("window_direct", 1)


def test_list_view_spans_carry_the_two_smallest_of_every_list(frames):
    """orbx_target_search_view: the lists in place (pool segments in any order) equal the copied-out lists, and every span's best / second
    are the two smallest (distance, list position) of its segment — the contract the adapters' order-dependent replays rely on to skip
    the walk over a list (include/orbx.h: orbx_list_span)."""
    gpu, fr = frames
    m = ORBmatcher(gpu)
    k2, d2, d1 = fr[1].mvKeysUn, fr[1].mDescriptors, fr[0].mDescriptors
    rng = np.random.default_rng(33)
    grid = dict(_kf_grid(k2, (0.0, 0.0, 640.0, 480.0), True), cell_start=None, cell_idx=None)
    ur = np.where(rng.random(len(k2)) < 0.3, -1.0, k2["x"] - rng.uniform(1, 40, len(k2))).astype(np.float32)
    tgt = m.Target(k2, d2, grid, kp_uright=ur, inv_level_sigma2=(1.0 / (np.float32(1.2) ** np.arange(8, dtype=np.float32)) ** 2).astype(np.float32))
    for rep in range(3):
        nq = int(rng.integers(200, 900))
        src = rng.integers(0, len(k2), nq)
        qx = (k2["x"][src] + rng.normal(0, 4, nq)).astype(np.float32); qy = (k2["y"][src] + rng.normal(0, 4, nq)).astype(np.float32)
        qr = rng.choice([0.0, 2.5, 7.0, 15.0, 36.0, 120.0], nq).astype(np.float32)
        lo = rng.choice([-1, 0, 1, 3], nq).astype(np.int32); hi = rng.choice([-1, 0, 2, 7], nq).astype(np.int32)
        skip = (rng.random(len(k2)) < 0.2).astype(np.uint8)
        xr = (qx - rng.uniform(1, 40, nq)).astype(np.float32) if rep else None
        qd = d1[rng.integers(0, len(d1), nq)]
        qd[::7] = d2[src[::7]]     # exact copies: distance 0 and ties among the candidates
        want = po.window_search_grid(k2, d2, grid, qx, qy, qr, lo, hi, qd, kp_skip=skip, **({} if xr is None else dict(kp_uright=ur, q_xr=xr)))
        spans, pool = tgt.search_view(qx, qy, qr, lo, hi, qd, kp_skip=skip, q_xr=xr)
        assert len(spans) == nq and int(spans["count"].sum()) == int(want["row_ptr"][-1])
        for q in range(nq):
            a, b = int(want["row_ptr"][q]), int(want["row_ptr"][q + 1])
            seg = pool[int(spans["start"][q]):int(spans["start"][q]) + int(spans["count"][q])]
            assert np.array_equal(seg["idx"], want["cand"][a:b]) and np.array_equal(seg["dist"], want["dist"][a:b]), q
            order = np.lexsort((np.arange(b - a), seg["dist"]))      # by distance, then list position
            exp = [(int(seg["idx"][order[0]]), int(seg["dist"][order[0]])) if b - a > 0 else (-1, 256),
                   (int(seg["idx"][order[1]]), int(seg["dist"][order[1]])) if b - a > 1 else (-1, 256)]
            got = [(int(spans["best_idx"][q]), int(spans["best_dist"][q])), (int(spans["second_idx"][q]), int(spans["second_dist"][q]))]
            assert got == exp, (rep, q, got, exp)
        assert np.array_equal(spans["best_idx"], want["best_idx"]) and np.array_equal(spans["second_dist"], want["second_dist"])
    tgt.close()


def test_a_view_survives_the_capacity_retry_of_the_next_view_call(frames):
    """include/orbx.h: a view stays valid until the context's SECOND next view call (a rig replays the left and the right lists together).
    The right-hand call of a rig may overflow the learnt candidate-pool capacity and repeat itself with a larger pool: the repeat must reuse
    (and may re-allocate) ITS OWN blob, not the one that still backs the left view (round-4 advisor finding: the parity used to flip per
    attempt, so the retry landed on — and could free — the previous view's blob)."""
    gpu, fr = frames
    m = ORBmatcher(gpu)
    k2, d2, d1 = fr[1].mvKeysUn, fr[1].mDescriptors, fr[0].mDescriptors
    grid = dict(_kf_grid(k2, (0.0, 0.0, 640.0, 480.0), True), cell_start=None, cell_idx=None)
    tgt = m.Target(k2, d2, grid)
    rng = np.random.default_rng(5)
    for rep in range(3):
        m.set_option("view_pool_cap", 4096)
        # "left": a small call that fits the pool
        nq = 120
        src = rng.integers(0, len(k2), nq)
        qx, qy = k2["x"][src].copy(), k2["y"][src].copy()
        qr = np.full(nq, 7.0, np.float32)
        lo = np.full(nq, -1, np.int32); hi = np.full(nq, -1, np.int32)
        qd = d1[rng.integers(0, len(d1), nq)]
        spans_l, pool_l = tgt.search_view(qx, qy, qr, lo, hi, qd, copy=False)      # arrays over the blob itself
        keep_s, keep_p = spans_l.copy(), pool_l.copy()
        assert 0 < int(spans_l["count"].sum()) <= 4096
        # "right": far more candidates than the pool holds -> ORBX_E_CAPACITY inside, one repeat with a larger pool (a larger blob)
        nq2 = 900 + 200 * rep
        src2 = rng.integers(0, len(k2), nq2)
        qx2, qy2 = k2["x"][src2].copy(), k2["y"][src2].copy()
        qr2 = np.full(nq2, 200.0, np.float32)   # ~ half the target per query: the repeat needs a blob of several MB (re-allocation)
        lo2 = np.full(nq2, -1, np.int32); hi2 = np.full(nq2, -1, np.int32)
        qd2 = d1[rng.integers(0, len(d1), nq2)]
        want = po.window_search_grid(k2, d2, grid, qx2, qy2, qr2, lo2, hi2, qd2)
        assert int(want["row_ptr"][-1]) > 4 * 4096
        spans_r, pool_r = tgt.search_view(qx2, qy2, qr2, lo2, hi2, qd2, copy=False)
        assert int(spans_r["count"].sum()) == int(want["row_ptr"][-1])
        # the left view is untouched and still readable
        assert np.array_equal(spans_l, keep_s) and np.array_equal(pool_l, keep_p), rep
        for q in range(0, nq2, 37):
            a, b = int(want["row_ptr"][q]), int(want["row_ptr"][q + 1])
            seg = pool_r[int(spans_r["start"][q]):int(spans_r["start"][q]) + int(spans_r["count"][q])]
            assert np.array_equal(seg["idx"], want["cand"][a:b]) and np.array_equal(seg["dist"], want["dist"][a:b]), (rep, q)
    tgt.close()


def test_view_calls_issued_and_waited_for_separately(frames):
    """orbx_target_search_view_begin / _end (the split the drop-in's per-frame routines use to overlap their host passes with the device): two
    calls in flight on the two view blobs, another call on the context in between, ends in either order; a pool that turns out too small makes
    _end fall back to the synchronous call with the same results; a third _begin while two are pending and an _end without a _begin are errors."""
    gpu, fr = frames
    m = ORBmatcher(gpu)
    k2, d2, d1 = fr[1].mvKeysUn, fr[1].mDescriptors, fr[0].mDescriptors
    grid = dict(_kf_grid(k2, (0.0, 0.0, 640.0, 480.0), True), cell_start=None, cell_idx=None)
    tgt = m.Target(k2, d2, grid)
    rng = np.random.default_rng(17)

    def queries(nq, radius):
        src = rng.integers(0, len(k2), nq)
        return (k2["x"][src].copy(), k2["y"][src].copy(), np.full(nq, radius, np.float32), np.full(nq, -1, np.int32), np.full(nq, -1, np.int32),
                d1[rng.integers(0, len(d1), nq)].copy())

    def check_view(view, q):
        spans, pool = view
        want = po.window_search_grid(k2, d2, grid, *q)
        assert int(spans["count"].sum()) == int(want["row_ptr"][-1])
        for i in range(len(q[0])):
            a, b = int(want["row_ptr"][i]), int(want["row_ptr"][i + 1])
            seg = pool[int(spans["start"][i]):int(spans["start"][i]) + int(spans["count"][i])]
            assert np.array_equal(seg["idx"], want["cand"][a:b]) and np.array_equal(seg["dist"], want["dist"][a:b]), i
        assert np.array_equal(spans["best_idx"], want["best_idx"])

    for order in ("ab", "ba"):
        qa, qb, qc = queries(400, 9.0), queries(700, 14.0), queries(50, 5.0)
        ta = tgt.search_view_begin(*qa)
        tb = tgt.search_view_begin(*qb)
        assert {ta[0], tb[0]} == {0, 1}
        with pytest.raises(OrbxError):
            tgt.search_view_begin(*qc)                    # both blobs hold a pending call
        other = tgt.search(*qc)                            # an ordinary call in between
        wc = po.window_search_grid(k2, d2, grid, *qc)
        assert np.array_equal(other["cand"], wc["cand"])
        if order == "ab":
            va = tgt.search_view_end(ta, copy=False); vb = tgt.search_view_end(tb, copy=False)
        else:
            vb = tgt.search_view_end(tb, copy=False); va = tgt.search_view_end(ta, copy=False)
        check_view(va, qa); check_view(vb, qb)            # both views readable at the same time
        with pytest.raises(OrbxError):
            tgt.search_view_end(ta)                        # nothing pending in that slot any more
    # overflow of the learnt capacity inside a pending call: _end repeats it synchronously
    m.set_option("view_pool_cap", 256)
    qa, qb = queries(300, 10.0), queries(600, 60.0)
    ta = tgt.search_view_begin(*qa); tb = tgt.search_view_begin(*qb)
    va = tgt.search_view_end(ta, copy=False)
    keep = (va[0].copy(), va[1].copy())
    vb = tgt.search_view_end(tb, copy=False)
    check_view(vb, qb); check_view(va, qa)
    assert np.array_equal(va[0], keep[0]) and np.array_equal(va[1], keep[1])     # B's repeat did not touch A's blob
    # empty query set and empty target
    te = tgt.search_view_begin(*queries(0, 1.0))
    assert len(tgt.search_view_end(te)[0]) == 0
    # the SYNCHRONOUS view call between a _begin and its _end must not take the pending call's blob; with both slots pending it
    # refuses; a ticket nobody will end is cancelled and the slot is free again
    qa, qb, qc = queries(500, 12.0), queries(300, 8.0), queries(200, 10.0)
    ta = tgt.search_view_begin(*qa)
    vc1 = tgt.search_view(*qc, copy=False)                 # one pending: takes the other blob
    vc2 = tgt.search_view(*qa, copy=False)                 # ... and again (the same non-pending blob: vc1 is dead now, by the lifetime rule)
    check_view(vc2, qa)
    va = tgt.search_view_end(ta, copy=False)
    check_view(va, qa)
    del vc1
    ta = tgt.search_view_begin(*qa); tb = tgt.search_view_begin(*qb)
    with pytest.raises(OrbxError):
        tgt.search_view(*qc)                               # both pending: refused, nothing overwritten
    tgt.search_view_cancel(tb)                             # the ticket of a caller that unwound between the halves
    tgt.search_view_cancel(tb)                             # idempotent
    with pytest.raises(OrbxError):
        tgt.search_view_end(tb)
    check_view(tgt.search_view(*qc, copy=False), qc)       # the cancelled slot serves again
    check_view(tgt.search_view_end(ta, copy=False), qa)    # and the other half is still intact
    t1 = tgt.search_view_begin(*qb); t2 = tgt.search_view_begin(*qc)
    check_view(tgt.search_view_end(t2, copy=False), qc); check_view(tgt.search_view_end(t1, copy=False), qb)
    tgt.close()


def test_failed_target_assign_leaves_an_invalid_target_not_an_empty_one(frames):
    """orbx_target_assign that fails (here: a grid whose indices point past the keypoints) must not leave a target that answers
    searches with 0 candidates and ORBX_OK: it is invalid until a later assign succeeds."""
    gpu, fr = frames
    m = ORBmatcher(gpu)
    k2, d2, d1 = fr[1].mvKeysUn, fr[1].mDescriptors, fr[0].mDescriptors
    grid = _kf_grid(k2, (0.0, 0.0, 640.0, 480.0), False)
    T = m.Target(k2, d2, grid)
    nq = 200
    qx, qy = k2["x"][:nq].copy(), k2["y"][:nq].copy()
    qr = np.full(nq, 15.0, np.float32)
    lo = np.full(nq, -1, np.int32); hi = np.full(nq, -1, np.int32)
    want = po.window_search_grid(k2, d2, grid, qx, qy, qr, lo, hi, d1[:nq])
    assert np.array_equal(T.search(qx, qy, qr, lo, hi, d1[:nq])["cand"], want["cand"]) and len(want["cand"]) > nq
    bad = dict(grid, cell_idx=np.where(np.arange(len(grid["cell_idx"])) == 5, len(k2) + 3, grid["cell_idx"]).astype(np.int32))
    with pytest.raises(OrbxError):
        T.assign(k2, d2, bad)
    with pytest.raises(OrbxError):
        T.search(qx, qy, qr, lo, hi, d1[:nq])
    with pytest.raises(OrbxError):
        T.nearest(qx, qy, qr, lo, hi, d1[:nq])
    with pytest.raises(OrbxError):
        len(T)
    T.assign(k2[:500], d2[:500], _kf_grid(k2[:500], (0.0, 0.0, 640.0, 480.0), False))
    want = po.window_search_grid(k2[:500], d2[:500], _kf_grid(k2[:500], (0.0, 0.0, 640.0, 480.0), False), qx, qy, qr, lo, hi, d1[:nq])
    got = T.search(qx, qy, qr, lo, hi, d1[:nq])
    assert len(T) == 500 and np.array_equal(got["cand"], want["cand"]) and np.array_equal(got["dist"], want["dist"])
    T.close()


def test_search_targets_always_take_the_host_rows_also_after_publish_descriptors():
    """orbx_publish_descriptors names the buffer for the BoW lookup only (include/orbx.h).  The device-to-device hand-over of those rows to
    the first search target (rounds 3-4) was removed in round 5 — it measured slower than the host rows — so a target made from a published
    buffer must hold what the HOST buffer holds at that moment: the same results as before, and a caller who edits the buffer after the
    extraction gets the edited rows (the hand-over would have served the extraction's)."""
    from orb_slam3_modified_amd import synth
    fr = synth.make_stream(2, 480, 640, 31)
    ex = ORBextractor(1000, 1.2, 8, 20, 7)
    m = ORBmatcher(ex)
    mono, k1, d1 = ex(fr[0], None, (0, 1000))
    d1 = np.ascontiguousarray(d1)
    ex.publish_descriptors(d1)
    grid = dict(min_x=0.0, min_y=0.0, inv_w=64 / 640.0, inv_h=48 / 480.0, cell_start=None, cell_idx=None)
    nq = 600
    qx, qy = k1["x"][:nq].copy(), k1["y"][:nq].copy()
    qr = np.full(nq, 12.0, np.float32)
    lo = np.full(nq, -1, np.int32); hi = np.full(nq, -1, np.int32)
    qd = d1[::-1][:nq].copy()
    want = po.window_search_grid(k1, d1, grid, qx, qy, qr, lo, hi, qd)
    T = m.Target(k1, d1, grid)
    got = T.search(qx, qy, qr, lo, hi, qd)
    for key in ("row_ptr", "cand", "dist", "best_idx", "best_dist"):
        assert np.array_equal(got[key], want[key]), key
    assert len(want["cand"]) > nq
    ex.publish_descriptors(d1)
    d1e = d1.copy(); d1[:, 5] ^= 0xff                # the caller edits the published buffer in place
    wante = po.window_search_grid(k1, d1, grid, qx, qy, qr, lo, hi, qd)
    Te = m.Target(k1, d1, grid)
    gote = Te.search(qx, qy, qr, lo, hi, qd)
    assert not np.array_equal(wante["dist"], want["dist"])
    for key in ("row_ptr", "cand", "dist", "best_idx", "best_dist"):
        assert np.array_equal(gote[key], wante[key]), key
    Te.close()
    d1[:] = d1e
    mono2, k2, d2 = ex(fr[1], None, (0, 1000))       # the context moves on: its HBM rows are frame 1's now
    T2 = m.Target(k1, d1, grid)
    got2 = T2.search(qx, qy, qr, lo, hi, qd)
    for key in ("row_ptr", "cand", "dist", "best_idx", "best_dist"):
        assert np.array_equal(got2[key], want[key]), key
    ex.publish_descriptors(d1[:10])                  # a count that does not match the extraction: ignored
    T3 = m.Target(k1[:10], d1[:10], grid)
    assert len(T3) == 10
    for t in (T, T2, T3):
        t.close()
