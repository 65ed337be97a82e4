"""tools/validate_opencv.cpp — what a maintainer with a real OpenCV runs (tools/validate_opencv.cmake) — built here over the container
shim (oracle/_ref/validate_opencv, oracle/ref_fragments.mk; the shim's cv:: functions are the oracle's, so the primitive legs prove the
harness, not OpenCV).  CPU: the tool runs, reports the calibration it finds under several "OpenCV builds", and a deliberately broken
primitive is reported with its first mismatch.  GPU: the operator() leg — the reference's own src/ORBextractor.cc, compiled into the tool,
against liborbx.so on the natural crops and synthetic frames, under the default and under another OpenCV variant."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "oracle", "_ref", "validate_opencv")

pytestmark = pytest.mark.skipif(not os.path.exists(EXE), reason="oracle/_ref/validate_opencv not built (needs /root/reference and liborbx.so: make -C oracle -f ref_fragments.mk)")


def _set(tmp_path, nsyn=1):
    out = str(tmp_path / "validate_set.bin")
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "make_validate_set.py"), out, "--synthetic", str(nsyn)])
    return out


def test_primitive_legs_run_and_report_the_calibration(tmp_path):
    r = subprocess.run([EXE], capture_output=True, text=True)
    assert r.returncode == 0 and "RESULT: ALL MATCH" in r.stdout and "gauss_kernel=0 gauss_round=0 gauss_tail=0 (exact, 1 candidate(s))" in r.stdout, r.stdout + r.stderr
    env = dict(os.environ, ORBO_VARIANT="1,2,16,1,0")
    r = subprocess.run([EXE], capture_output=True, text=True, env=env)
    assert r.returncode == 0 and "gauss_kernel=1 gauss_round=2 gauss_tail=16 (exact" in r.stdout and "atan_fma=1 (exact)" in r.stdout, r.stdout + r.stderr
    assert r.stdout.count("MATCH") >= 7 and "MISMATCH" not in r.stdout


def test_natural_crop_set_is_read(tmp_path):
    r = subprocess.run([EXE, "--set", _set(tmp_path)], capture_output=True, text=True)
    assert r.returncode == 0 and "12 images" in r.stdout and "RESULT: ALL MATCH" in r.stdout, r.stdout + r.stderr
    r = subprocess.run([EXE, "--set", str(tmp_path / "missing.bin")], capture_output=True, text=True)
    assert r.returncode == 2


def test_the_tool_names_the_profile_of_the_opencv_at_hand(tmp_path):
    """tools/opencv_pin: the digest table (expected_digests.txt) is current, and under every named profile — the shim's cv:: functions then ARE
    that profile's arithmetic — the program names it from the cv:: outputs alone, with the orbx_set_cpu_profile call to make."""
    import ctypes as C
    import numpy as np
    from orb_slam3_modified_amd import _lib
    pin = os.path.join(ROOT, "tools", "opencv_pin")
    assert subprocess.run([sys.executable, os.path.join(pin, "make_expected.py"), "--check"]).returncode == 0, "tools/opencv_pin/expected_digests.txt is stale"
    table = os.path.join(pin, "expected_digests.txt")
    st = _set(tmp_path)
    L = _lib.lib()
    i = 0
    while L.orbx_cpu_profile_name(i):
        nm = L.orbx_cpu_profile_name(i).decode()
        v = np.zeros(5, np.int32)
        for fma in (0, 2):
            if L.orbx_cpu_profile_values(nm.encode(), fma, _lib.ptr(v)) != 0:
                continue                                    # this profile's OpenCV has no FMA copy of fastAtan2
            env = dict(os.environ, ORBO_VARIANT=",".join(str(int(x)) for x in v))
            r = subprocess.run([EXE, "--set", st, "--expect", table], capture_output=True, text=True, env=env)
            assert r.returncode == 0 and "RESULT: ALL MATCH" in r.stdout, r.stdout[-1500:]
            assert f'-> orbx_set_cpu_profile(ctx, "{nm}", {fma})' in r.stdout and "IN NO PROFILE" not in r.stdout, r.stdout[-1500:]
        i += 1
    assert i == 6
    # another image set: the table does not apply, and the tool says so instead of naming a profile
    other = str(tmp_path / "other.bin")
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "make_validate_set.py"), other, "--synthetic", "2"])
    r = subprocess.run([EXE, "--set", other, "--expect", table], capture_output=True, text=True)
    assert r.returncode == 1 and "ANOTHER image set" in r.stdout
    # the stand-alone project and the one-command script are in place (cmake needs a real OpenCV: not run here)
    for f in ("CMakeLists.txt", "run.sh", "no_orbx.cpp", "README.md"):
        assert os.path.exists(os.path.join(pin, f)), f
    assert os.access(os.path.join(pin, "run.sh"), os.X_OK) and "find_package(OpenCV REQUIRED)" in open(os.path.join(pin, "CMakeLists.txt")).read()


def test_the_one_command_recipe_end_to_end_over_the_shim(tmp_path):
    """tools/opencv_pin/run.sh as a maintainer runs it — cmake finds "OpenCV" (here: a two-line OpenCVConfig.cmake that points at the container's
    shim headers, whose cv:: functions forward to the oracle the project builds), builds the oracle and the tool, the reference's own
    src/ORBextractor.cc compiled in, writes the set, runs every leg and names the profile."""
    import shutil
    if not shutil.which("cmake") or not os.path.isdir("/root/reference"):
        pytest.skip("needs cmake and /root/reference")
    cfg = tmp_path / "fake_opencv"
    cfg.mkdir()
    (cfg / "OpenCVConfig.cmake").write_text(f'set(OpenCV_VERSION "0.0-shim")\nset(OpenCV_INCLUDE_DIRS "{ROOT}/oracle/ref_shims")\nset(OpenCV_LIBS "")\n')
    env = dict(os.environ, CMAKE_PREFIX_PATH=str(cfg), OpenCV_DIR=str(cfg), OPENCV_PIN_CMAKE_ARGS="-DWITH_EXTRAS=OFF")   # the shim has no calib3d
    env.pop("ORBO_VARIANT", None)
    r = subprocess.run([os.path.join(ROOT, "tools", "opencv_pin", "run.sh"), "/root/reference"], capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    assert "RESULT: ALL MATCH" in r.stdout and "12 images" in r.stdout and '-> orbx_set_cpu_profile(ctx, "opencv>=4.5.1", ' in r.stdout, r.stdout[-2000:]
    assert r.stdout.count("MATCH") >= 7 and "0.0-shim" not in r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("variant", ["", "1,0,0,0,0", "1,2,8,1,0"])
def test_operator_leg_reference_compiled_vs_liborbx(tmp_path, variant):
    env = dict(os.environ)
    if variant:
        env["ORBO_VARIANT"] = variant
    r = subprocess.run([EXE, "--set", _set(tmp_path), "--orbx"], capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "keypoints (all 7 fields as bit patterns), descriptors, return value" in r.stdout and "RESULT: ALL MATCH" in r.stdout, r.stdout
