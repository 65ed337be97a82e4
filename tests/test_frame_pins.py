"""Pins the restatements of src/Frame.cc — the oracle's (oracle/match_oracle.cpp: ComputeStereoMatches :811-981, AssignFeaturesToGrid /
PosInGrid :385-416,:725-735, GetFeaturesInArea :657-723, UndistortKeyPoints :747-780) and the device forms the library ships beside the
drop-in headers (orbx_stereo_matches on the device pyramids, the window grid, the undistortion) — to the REFERENCE'S OWN src/Frame.cc:
tests/golden/frame_world_ref.txt.gz is what that file, compiled unmodified (oracle/_ref/ref_frame_world), leaves in a Frame.  The images
are the ones the scenario driver generates (tests/support/frame_world.cpp, written out with its image-dump argument by one of the built
executables); the expected values are read from the golden text.

  CPU   oracle functions on the oracle extractor's output  == the reference's Frame fields (bit patterns / digests)
  GPU   library calls on the GPU extractor's output         == the same fields
"""
import gzip
import os
import struct
import subprocess

import numpy as np
import pytest

from oracle import pyoracle as po

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REFDIR = os.path.join(ROOT, "oracle", "_ref")
VOC = os.path.join(ROOT, "tests", "golden", "voc_k5_L3.txt")
GOLD = os.path.join(ROOT, "tests", "golden", "frame_world_ref.txt.gz")
f32 = np.float32

# the calibration of tests/support/frame_world.cpp
FX, FY, CX, CY, BF = f32(458.654), f32(457.296), f32(367.215), f32(248.375), f32(47.90639)
D5 = np.array([-0.28340811, 0.07395907, 0.00019359, 1.76187114e-05, 0.011], np.float32)


def _fnv(data: bytes, h: int = 1469598103934665603) -> int:
    for b in data:
        h = ((h ^ b) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return h


class Lcg:   # tests/support/frame_world.cpp
    def __init__(self, seed):
        self.s = (seed * 2862933555777941757 + 3037000493) & 0xFFFFFFFFFFFFFFFF

    def below(self, n):
        self.s = (self.s * 6364136223846793005 + 1442695040888963407) & 0xFFFFFFFFFFFFFFFF
        return (self.s >> 33) % n


def _golden_frames():
    frames, cur = {}, None
    for line in gzip.open(GOLD).read().decode().splitlines():
        if not line.startswith(" "):
            cur = line.split()[0]
            frames[cur] = {"head": line}
        else:
            key = line.split()[0]
            frames[cur].setdefault(key, line)
    return frames


def _hex_floats(line):
    toks = line.split()
    n = int(toks[1].split("=")[1])
    vals = np.array([int(t, 16) for t in toks[3:3 + n]], np.uint32).view(np.float32)
    assert len(vals) == n
    return vals


def _digest(line):
    return int(line.split("digest=")[1].split()[0], 16)


@pytest.fixture(scope="module")
def images(tmp_path_factory):
    exe = next((os.path.join(REFDIR, n) for n in ("dropin_frame_world_cpu", "ref_frame_world") if os.path.exists(os.path.join(REFDIR, n))), None)
    if exe is None:
        pytest.skip("no frame_world executable in oracle/_ref (oracle/ref_fragments.mk builds them from /root/reference)")
    d = tmp_path_factory.mktemp("frame_images")
    subprocess.run([exe, VOC, str(d / "out.txt"), "1", str(d)], check=True, stdout=subprocess.DEVNULL, timeout=600)
    out = {}
    for fn in os.listdir(d):
        if fn.endswith(".raw"):
            name, shape = fn[:-4].rsplit("_", 1)
            r, c = (int(v) for v in shape.split("x"))
            out[name] = np.fromfile(d / fn, np.float32 if name == "rgbd_depth" else np.uint8).reshape(r, c)
    assert set(out) == {"stereo_left", "stereo_right", "rgbd_gray", "rgbd_depth", "mono_distorted"}
    return out


def _grid_digest(cell_start, cell_idx):
    parts = []
    for c in range(64 * 48):   # mGrid[i][j], i = column outer, j = row inner: cell = i * 48 + j
        a, b = int(cell_start[c]), int(cell_start[c + 1])
        parts.append(struct.pack("<I", b - a))
        parts.append(np.asarray(cell_idx[a:b], np.uint32).tobytes())
    return _fnv(b"".join(parts))


def _area_queries(right=False):
    rng = Lcg(77 + (1 if right else 0))
    qx, qy, qr, lo, hi = [], [], [], [], []
    for q in range(200):
        x = f32(rng.below(7000)) * f32(0.1) - f32(20.0)
        y = f32(rng.below(5200)) * f32(0.1) - f32(20.0)
        r = f32(3.0) + f32(rng.below(400)) * f32(0.1)
        a, b = -1, -1
        if q & 1:
            a = rng.below(4)
            b = a + rng.below(4) if q & 2 else -1
        qx.append(x); qy.append(y); qr.append(r); lo.append(a); hi.append(b)
    return np.array(qx, np.float32), np.array(qy, np.float32), np.array(qr, np.float32), np.array(lo, np.int32), np.array(hi, np.int32)


def _area_digest(rp, cand):
    parts = []
    for q in range(len(rp) - 1):
        a, b = int(rp[q]), int(rp[q + 1])
        parts.append(struct.pack("<I", b - a))
        parts.append(np.asarray(cand[a:b], np.uint32).tobytes())
    return _fnv(b"".join(parts)), int(rp[-1])


def _check_stereo(gold, kL, ur, dp):
    gur, gdp = _hex_floats(gold["mvuRight"]), _hex_floats(gold["mvDepth"])
    assert len(kL) == len(gur)
    assert ur.tobytes() == gur.tobytes(), f"mvuRight differs at {np.flatnonzero(ur.view(np.uint32) != gur.view(np.uint32))[:5]}"
    assert dp.tobytes() == gdp.tobytes()
    assert int((gdp > 0).sum()) > 500


def _check_frame_tail(gold, kps, kps_un, bounds, grid, area):
    assert _fnv(np.ascontiguousarray(kps).tobytes()) == _digest(gold["mvKeys"])
    assert _fnv(np.ascontiguousarray(kps_un).tobytes()) == _digest(gold["mvKeysUn"])
    st = gold["statics"]
    gb = np.array([int(t, 16) for t in st.split("bounds=")[1].split()[:4]], np.uint32).view(np.float32)   # minX maxX minY maxY
    assert np.array([bounds[0], bounds[2], bounds[1], bounds[3]], np.float32).tobytes() == gb.tobytes()
    assert _grid_digest(*grid) == _digest(gold["mGrid"])
    dig, total = _area_digest(*area)
    assert f"{total} indices" in gold["GetFeaturesInArea(right=0)"] and dig == _digest(gold["GetFeaturesInArea(right=0)"])


def _undistorted_bounds(rows, cols):
    # Frame::ComputeImageBounds (:782-809): the four corners through cv::undistortPoints
    corners = np.zeros(4, po.KP_DTYPE)
    corners["x"] = [0, cols, 0, cols]
    corners["y"] = [0, 0, rows, rows]
    u = po.undistort_keypoints(corners, FX, FY, CX, CY, D5)
    return (min(u["x"][0], u["x"][2]), min(u["y"][0], u["y"][1]), max(u["x"][1], u["x"][3]), max(u["y"][2], u["y"][3]))


def test_oracle_stereo_matches_equal_the_reference_frame_cc(images):
    gold = _golden_frames()["stereo_752x480"]
    exL, exR = po.OracleExtractor(1200, 1.2, 8, 20, 7), po.OracleExtractor(1200, 1.2, 8, 20, 7)
    kL, dL, _ = exL.extract(images["stereo_left"], (0, 0))
    kR, dR, _ = exR.extract(images["stereo_right"], (0, 0))
    assert _fnv(kL.tobytes()) == _digest(gold["mvKeys"]) and _fnv(kR.tobytes()) == _digest(gold["mvKeysRight"])
    scale, inv_scale = exL.tables()["scale"], exL.tables()["inv_scale"]
    pyrL = [images["stereo_left"]] + [exL.level(l) for l in range(1, 8)]
    pyrR = [images["stereo_right"]] + [exR.level(l) for l in range(1, 8)]
    ur, dp, kept = po.stereo_matches(kL, dL, kR, dR, pyrL, pyrR, scale, inv_scale, BF / FX, BF)
    _check_stereo(gold, kL, ur, dp)
    assert kept == int((dp > 0).sum())


def test_oracle_frame_tail_equals_the_reference_frame_cc(images):
    gold = _golden_frames()["rgbd_640x480_distorted"]
    ex = po.OracleExtractor(1000, 1.2, 8, 20, 7)
    k, d, _ = ex.extract(images["rgbd_gray"], (0, 0))
    ku = po.undistort_keypoints(k, FX, FY, CX, CY, D5)
    bounds = _undistorted_bounds(480, 640)
    inv_w, inv_h = f32(64) / (f32(bounds[2]) - f32(bounds[0])), f32(48) / (f32(bounds[3]) - f32(bounds[1]))
    grid = po.assign_grid(ku, bounds[0], bounds[1], inv_w, inv_h)
    area = po.features_in_area(ku, bounds, *_area_queries())
    _check_frame_tail(gold, k, ku, bounds, grid, area)
    # Frame::ComputeStereoFromRGBD (:984-1005): depth read at the DISTORTED keypoint, uRight from the undistorted one
    dimg = images["rgbd_depth"]
    dd = dimg[k["y"].astype(np.int32), k["x"].astype(np.int32)]
    depth = np.where(dd > 0, dd, f32(-1)).astype(np.float32)
    with np.errstate(divide="ignore"):
        uright = np.where(dd > 0, ku["x"] - BF / dd, f32(-1)).astype(np.float32)
    assert depth.tobytes() == _hex_floats(gold["mvDepth"]).tobytes() and uright.tobytes() == _hex_floats(gold["mvuRight"]).tobytes()


@pytest.mark.gpu
def test_device_stereo_matches_equal_the_reference_frame_cc(images):
    from orb_slam3_modified_amd import ORBextractor, ORBmatcher
    gold = _golden_frames()["stereo_752x480"]
    exL, exR = ORBextractor(1200, 1.2, 8, 20, 7), ORBextractor(1200, 1.2, 8, 20, 7)
    _, kL, dL = exL(images["stereo_left"], None, (0, 0))
    _, kR, dR = exR(images["stereo_right"], None, (0, 0))
    assert _fnv(kL.tobytes()) == _digest(gold["mvKeys"]) and _fnv(kR.tobytes()) == _digest(gold["mvKeysRight"])
    assert _fnv(np.ascontiguousarray(dL).tobytes()) == _digest(gold["mDescriptors"])
    ur, dp, kept = ORBmatcher.ComputeStereoMatches(exL, exR, kL, dL, kR, dR, float(BF / FX), float(BF))
    _check_stereo(gold, kL, ur, dp)
    assert kept == int((dp > 0).sum())


@pytest.mark.gpu
def test_device_frame_tail_equals_the_reference_frame_cc(images):
    from orb_slam3_modified_amd import ORBextractor, ORBmatcher
    gold = _golden_frames()["rgbd_640x480_distorted"]
    ex = ORBextractor(1000, 1.2, 8, 20, 7)
    _, k, d = ex(images["rgbd_gray"], None, (0, 0))
    ku = ex.UndistortKeyPoints(k, float(FX), float(FY), float(CX), float(CY), D5)
    bounds = _undistorted_bounds(480, 640)
    m = ORBmatcher(ex)
    qx, qy, qr, lo, hi = _area_queries()
    rp, cand = m.GetFeaturesInArea(ku, bounds, qx, qy, qr, lo, hi)
    inv_w, inv_h = f32(64) / (f32(bounds[2]) - f32(bounds[0])), f32(48) / (f32(bounds[3]) - f32(bounds[1]))
    grid = po.assign_grid(ku, bounds[0], bounds[1], inv_w, inv_h)     # the grid itself stays inside the library; its lists are what is observable
    _check_frame_tail(gold, k, ku, bounds, grid, (rp, cand))
