"""The C++ adapters (include/ORBextractor.h, ORBVocabulary.h, the drop-in ORBmatcher): they compile without OpenCV (against
include/orbx_cv_compat.h on their own, against the container shim together with the matcher's object model), link against the
in-tree liborbx.so, fail loudly without a GPU, and on a GPU return exactly what the oracle returns, through the reference's own
calling convention.  The matcher routines are compared with the reference's src/ORBmatcher.cc in tests/test_matcher_world.py."""
import os
import struct
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SUP = os.path.join(ROOT, "tests", "support")
EXE = os.path.join(SUP, "adapter_demo.bin")
PKG = os.path.join(ROOT, "orb_slam3_modified_amd")


def _build():
    from orb_slam3_modified_amd import build
    build.build()
    from tests import world_util as wu
    src = os.path.join(SUP, "adapter_demo.cpp")
    deps = [src, wu.ADAPTER_SRC, os.path.join(PKG, "liborbx.so")] + [os.path.join(ROOT, "include", f) for f in os.listdir(os.path.join(ROOT, "include"))]
    if not os.path.exists(EXE) or any(os.path.getmtime(d) > os.path.getmtime(EXE) for d in deps):
        # the OpenCV this executable is built with is the shim of oracle/ref_shims, whose cv::GaussianBlur / cv::fastAtan2 forward to the
        # oracle's switchable primitives: include/ORBextractor.h calibrates itself against THEM (include/orbx_cv_calibrate.h), which is what
        # test_adapter_follows_the_opencv_it_is_built_with drives through ORBO_VARIANT
        from oracle import pyoracle
        pyoracle.build()
        odir = os.path.join(ROOT, "oracle")
        subprocess.check_call(["g++"] + wu.CXXFLAGS + wu.INCLUDES + [src, wu.ADAPTER_SRC, "-o", EXE, "-L", PKG, "-lorbx", "-Wl,-rpath," + PKG,
                                                                     "-Wl,--allow-shlib-undefined", "-L", odir, "-lorb_oracle", "-Wl,-rpath," + odir])
    return EXE


def test_extractor_and_vocabulary_adapters_compile_against_their_own_cv_compat(tmp_path):
    """include/orbx_cv_compat.h: the two header-only adapters build where no OpenCV header of any kind is on the path."""
    tu = tmp_path / "compat.cpp"
    tu.write_text('#include "ORBextractor.h"\n#include "ORBVocabulary.h"\nint main() { ORB_SLAM3::ORBVocabulary v; return v.empty() ? 0 : 1; }\n')
    subprocess.check_call(["g++", "-std=c++14", "-Wall", "-DORBX_FORCE_CV_COMPAT", "-I", os.path.join(ROOT, "include"), str(tu), "-o",
                           str(tmp_path / "compat.bin"), "-L", PKG, "-lorbx", "-Wl,-rpath," + PKG, "-Wl,--allow-shlib-undefined"])


def test_adapters_parse_against_real_opencv_signatures(tmp_path):
    """tests/support/opencv_signatures: declarations with OpenCV 4.x's own signatures (_InputArray::getMat(int = -1), _OutputArray::create
    with its defaulted arguments, Mat::step as a MatStep object, Mat(rows, cols, type, void*, size_t = AUTO_STEP), templated ptr<>).  The
    header-only adapters and the calibration must compile against those shapes too, not only against the simplified look-alikes —
    -fsyntax-only: overload resolution, implicit conversions and const-correctness; nothing is linked."""
    tu = tmp_path / "sig.cpp"
    tu.write_text('#include "ORBextractor.h"\n#include "ORBVocabulary.h"\n'
                  '#ifndef ORBX_CV_CALIBRATION\n#error "calibration inactive"\n#endif\n'
                  '#ifndef ORBX_HAVE_OPENCV\n#error "the compat look-alikes were used"\n#endif\n'
                  'int use(ORB_SLAM3::ORBextractor& e, const cv::Mat& im, std::vector<cv::KeyPoint>& k, cv::Mat& d, std::vector<int>& lap) {\n'
                  '  return e(im, cv::Mat(), k, d, lap); }\n'
                  'void bow(const ORB_SLAM3::ORBVocabulary& v, const std::vector<cv::Mat>& f, DBoW2::BowVector& b, DBoW2::FeatureVector& fv) { v.transform(f, b, fv, 4); }\n')
    subprocess.check_call(["g++", "-std=c++14", "-fsyntax-only", "-Wall", "-Wextra", "-I", os.path.join(ROOT, "include"), "-I",
                           os.path.join(SUP, "opencv_signatures"), str(tu)])


def test_adapters_compile_link_and_fail_loudly_without_gpu():
    exe = _build()
    r = subprocess.run([exe, "probe"], capture_output=True, text=True)
    from tests.conftest import HAS_GPU
    if HAS_GPU:
        assert r.returncode == 0 and "DEVICE_OK" in r.stdout, r.stdout + r.stderr
    else:
        assert r.returncode == 3 and "NO_DEVICE" in r.stdout, r.stdout + r.stderr


def _serialize_check(tmp_path, name, extra):
    exe = str(tmp_path / name)
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-Wno-unused-function", "-I", os.path.join(ROOT, "include"), "-I",
                           os.path.join(ROOT, "oracle", "ref_shims")] + extra[0] + [os.path.join(SUP, "serialize_check.cpp"), "-o", exe] + extra[1])
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 0 and "roundtrip=1" in r.stdout, r.stdout + r.stderr
    return r.stdout.split(None, 1)


def test_bow_and_feature_vectors_archive_like_keyframe_h_does(tmp_path):
    """`ar & mBowVec; ar & mFeatVec;` (reference include/KeyFrame.h:130-131, instantiated by src/System.cc:1464-1468) compiles and
    round-trips through boost::serialization::access with the DBoW2 classes include/ORBVocabulary.h provides on its own; the other
    members of the reference's classes (addIfNotExist, normalize, operator<<) are there as well."""
    which, rest = _serialize_check(tmp_path, "ser_own.bin", (["-DORBX_OWN_DBOW2_TYPES"], []))
    assert which == "own" and "words=301 nodes=37" in rest and "<3, 0.5>, <7, 0.25> | <2: [5, 6]>, <9: [1]>" in rest
    if not os.path.isdir("/root/reference/Thirdparty/DBoW2/DBoW2"):
        return
    # inside the reference's tree the header defers to the reference's own BowVector.h / FeatureVector.h (one definition in every
    # translation unit); their out-of-line members come from the DBoW2 library (here: oracle/_ref/libref_dbow2.so)
    refdir = os.path.join(ROOT, "oracle", "_ref")
    if not os.path.exists(os.path.join(refdir, "libref_dbow2.so")):
        pytest.skip("oracle/_ref/libref_dbow2.so not built")
    which2, rest2 = _serialize_check(tmp_path, "ser_tree.bin", (["-w", "-I", "/root/reference"], ["-L", refdir, "-lref_dbow2", "-Wl,-rpath," + refdir]))
    assert which2 == "tree" and rest2 == rest   # same bytes, same values, same printed form as the reference's classes


def _read(path):
    b = open(path, "rb").read()
    off = 0

    def take(fmt):
        nonlocal off
        v = struct.unpack_from(fmt, b, off)
        off += struct.calcsize(fmt)
        return v if len(v) > 1 else v[0]

    from orb_slam3_modified_amd import KP_DTYPE
    n, mono = take("<ii")
    kps = np.frombuffer(b, KP_DTYPE, n, off); off += n * 28
    desc = np.frombuffer(b, np.uint8, n * 32, off).reshape(n, 32); off += n * 32
    pyr = []
    for _ in range(take("<i")):
        h, w = take("<ii")
        pyr.append(np.frombuffer(b, np.uint8, h * w, off).reshape(h, w)); off += h * w
    d01 = take("<i")
    nb = take("<i")
    bow = [take("<Id") for _ in range(nb)]
    fv, self_score = [], None
    if nb:
        fv = [take("<II") for _ in range(take("<i"))]
        self_score = take("<d")
    return mono, kps, desc, pyr, d01, bow, fv, self_score



@pytest.mark.gpu
@pytest.mark.parametrize("rows,cols,lap", [(480, 640, (0, 1000)), (480, 752, (0, 0))])
def test_adapter_results_equal_oracle(tmp_path, rows, cols, lap):
    from oracle import pyoracle as po
    from orb_slam3_modified_amd import synth
    from tests.vocab_util import make_vocabulary
    exe = _build()
    img = synth.make_stream(1, rows, cols)[0]
    raw, out, vocp = str(tmp_path / "im.raw"), str(tmp_path / "out.bin"), str(tmp_path / "voc.txt")
    img.tofile(raw)
    ora = po.OracleExtractor(1000, 1.2, 8, 20, 7)
    okps, odesc, omono = ora.extract(img, lap)
    make_vocabulary(vocp, odesc, 6, 3, seed=3)
    r = subprocess.run([exe, "run", raw, str(rows), str(cols), "1000", str(lap[0]), str(lap[1]), out, vocp], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    mono, kps, desc, pyr, d01, bow, fv, self_score = _read(out)
    assert mono == omono and kps.tobytes() == okps.tobytes() and np.array_equal(desc, odesc)
    for l in range(8):
        assert np.array_equal(pyr[l], ora.level(l))
    assert d01 == po.hamming(odesc[0], odesc[1])
    (oi, ov), ofv = po.OracleVocabulary(vocp).transform(odesc, 2)
    assert [b[0] for b in bow] == list(oi) and np.array([b[1] for b in bow]).tobytes() == ov.tobytes()
    flat = [(int(k), int(f)) for k, v in ofv.items() for f in v]
    assert fv == flat
    assert abs(self_score - 1.0) < 1e-12


@pytest.mark.gpu
@pytest.mark.parametrize("variant", [(1, 0, 0, 0, 0), (1, 2, 16, 1, 0), (1, 1, 4, 0, 0), (0, 2, 8, 1, 0)])
def test_adapter_follows_the_opencv_it_is_built_with(tmp_path, variant):
    """The drop-in chain end to end: an executable built against "another OpenCV" (the shim's cv::GaussianBlur / cv::fastAtan2 switched to
    that release's arithmetic through ORBO_VARIANT) constructs ORB_SLAM3::ORBextractor; the adapter's calibration recognises the variant
    from the cv:: functions alone and the GPU then returns what the CPU path over THAT OpenCV returns — every keypoint bit and
    descriptor byte.  With ORBX_CV_CALIBRATE=0 the same executable keeps the default arithmetic, and the results differ."""
    from oracle import pyoracle as po
    from orb_slam3_modified_amd import synth
    exe = _build()
    img = synth.make_stream(1, 480, 640)[0]
    raw, out = str(tmp_path / "im.raw"), str(tmp_path / "out.bin")
    img.tofile(raw)
    with po.opencv_variant(*variant):
        okps, odesc, omono = po.OracleExtractor(1000, 1.2, 8, 20, 7).extract(img, (0, 1000))
    env = dict(os.environ, ORBO_VARIANT=",".join(str(v) for v in variant), ORBX_CV_VERBOSE="1")
    r = subprocess.run([exe, "run", raw, "480", "640", "1000", "0", "1000", out], capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stdout + r.stderr
    assert f"gauss_kernel={variant[0]} gauss_round={variant[1]} gauss_tail={variant[2]}" in r.stderr and f"atan_fma={variant[3]}" in r.stderr, r.stderr
    mono, kps, desc = _read(out)[:3]
    assert mono == omono and kps.tobytes() == okps.tobytes() and np.array_equal(desc, odesc)
    r = subprocess.run([exe, "run", raw, "480", "640", "1000", "0", "1000", out], capture_output=True, text=True, env=dict(env, ORBX_CV_CALIBRATE="0"))
    assert r.returncode == 0, r.stdout + r.stderr
    mono, kps, desc = _read(out)[:3]
    assert kps.tobytes() != okps.tobytes() or not np.array_equal(desc, odesc)


@pytest.mark.gpu
def test_matchers_and_vocabulary_from_three_threads(tmp_path):
    """Tracking, LocalMapping and LoopClosing construct matchers and call the shared vocabulary concurrently
    (SURVEY §8(b) threading): per-thread default contexts, serialised vocabulary context — results equal the serial ones."""
    from oracle import pyoracle as po
    from orb_slam3_modified_amd import synth
    from tests.vocab_util import make_vocabulary
    exe = _build()
    img = synth.make_stream(1, 480, 640)[0]
    raw, vocp = str(tmp_path / "im.raw"), str(tmp_path / "voc.txt")
    img.tofile(raw)
    okps, odesc, _ = po.OracleExtractor(1000, 1.2, 8, 20, 7).extract(img, (0, 0))
    make_vocabulary(vocp, odesc, 6, 3, seed=3)
    r = subprocess.run([exe, "mt", raw, "480", "640", vocp], capture_output=True, text=True)
    assert r.returncode == 0 and "MT_OK" in r.stdout, r.stdout + r.stderr
