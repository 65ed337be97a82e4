"""The reference's OWN phrasing of every call into the drop-in boundary that lives in files this container cannot compile whole
(src/Tracking.cc, src/LocalMapping.cc, src/LoopClosing.cc, src/KeyFrame.cc, src/System.cc: they need Eigen, g2o, Pangolin): 51 statements —
every ORBmatcher construction and Search* / Fuse call, the ORBextractor constructions, the KeyFrameDatabase and vocabulary calls — lifted
verbatim from the reference's files where they lie into one translation unit (tools/gen_callsites.py; nothing of the reference is stored here) and
compiled over include/ORBmatcher.h, ORBextractor.h, KeyFrameDatabase.h, ORBVocabulary.h + tests/support/ref_world.  Argument types, temporaries,
defaulted parameters and overload resolution of the reference's call sites against the replacement's declarations."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
FLAGS = ["-fsyntax-only", "-std=c++17", "-Wall", "-Werror=return-type", "-Wno-unused-variable", "-Wno-unused-but-set-variable", "-DORBX_NO_CV_CALIBRATION",
         "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "tests", "support", "ref_world"), "-I", os.path.join(ROOT, "oracle", "ref_shims")]


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "src")), reason="the reference checkout is not on this machine (GPU box)")
def test_the_reference_call_sites_compile_over_the_drop_in_headers(tmp_path):
    tu = str(tmp_path / "callsites_tu.cpp")
    out = subprocess.check_output([sys.executable, os.path.join(ROOT, "tools", "gen_callsites.py"), REF, tu]).decode()
    assert out.startswith("51 statements of the reference in 17 contexts"), out
    src = open(tu).read()
    for needle in ("matcher.Fuse(pKFi,vpMapPointMatches,true)", "mSensor==System::MONOCULAR || mSensor==System::IMU_MONOCULAR", "new ORBextractor(5*nFeatures",
                   "DetectNBestCandidates(mpCurrentKF, vpLoopBowCand, vpMergeBowCand,3)", "mpORBvocabulary->transform(vCurrentDesc,mBowVec,mFeatVec,4)"):
        assert needle in src, needle          # the statements are the reference's own text
    r = subprocess.run(["g++"] + FLAGS + [tu], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    # the check has teeth: a call the boundary does not offer is refused by the same command
    bad = str(tmp_path / "bad.cpp")
    open(bad, "w").write(src + "\nvoid negative(World& W) { ORBmatcher matcher(0.9, true); vector<MapPoint*> v; matcher.SearchByBoW(W.kf, v); }\n")
    r2 = subprocess.run(["g++"] + FLAGS + [bad], capture_output=True, text=True)
    assert r2.returncode != 0 and "SearchByBoW" in r2.stderr


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "src")), reason="the reference checkout is not on this machine (GPU box)")
def test_tracking_constructs_frames_from_the_drop_in_extractors(tmp_path):
    """The ten `mCurrentFrame = Frame(...)` statements of src/Tracking.cc (GrabImageStereo / RGBD / Monocular, with and without IMU and a second
    camera) hand Tracking's ORBextractor* / ORBVocabulary* members to the constructors of the reference's UNMODIFIED include/Frame.h: lifted
    verbatim and compiled the way oracle/ref_fragments.mk compiles dropin_frame_world — the reference's Frame.h over include/ORBextractor.h and
    include/ORBVocabulary.h."""
    tu = str(tmp_path / "callsites_frame_tu.cpp")
    out = subprocess.check_output([sys.executable, os.path.join(ROOT, "tools", "gen_callsites.py"), "--frames", REF, tu]).decode()
    assert out.startswith("10 Frame constructions"), out
    fw = os.path.join(ROOT, "tests", "support", "frame_world")
    cmd = ["g++", "-fsyntax-only", "-std=c++17", "-ffp-contract=off", "-w", "-pthread", "-include", os.path.join(fw, "prelude.h"), "-I", fw,
           "-I", os.path.join(ROOT, "oracle", "ref_shims"), "-DFRAME_WORLD_DROPIN", "-I", os.path.join(ROOT, "include"), "-I", REF, "-I", os.path.join(REF, "include"),
           "-I", os.path.join(REF, "include", "CameraModels"), tu]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    # teeth: an extractor of another type is refused by the same command
    bad = str(tmp_path / "bad.cpp")
    open(bad, "w").write(open(tu).read() + "\nvoid negative(cv::Mat& im, ORBVocabulary* v, GeometricCamera* c, cv::Mat& d) { int* e = nullptr; Frame f(im, 0.0, e, v, c, d, 1.f, 1.f); }\n")
    r2 = subprocess.run(cmd[:-1] + [bad], capture_output=True, text=True)
    assert r2.returncode != 0


def test_generator_notices_a_moved_reference(tmp_path):
    """Line numbers are pinned together with a token of the statement: against a file whose lines moved the generator stops instead of lifting
    the wrong statement."""
    fake = tmp_path / "ref" / "src"
    fake.mkdir(parents=True)
    for f in ("Tracking.cc", "LocalMapping.cc", "LoopClosing.cc", "KeyFrame.cc", "System.cc"):
        (fake / f).write_text("\n" * 5000)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_callsites.py"), str(tmp_path / "ref"), str(tmp_path / "o.cpp")], capture_output=True, text=True)
    assert r.returncode != 0 and "does not hold" in (r.stdout + r.stderr)
