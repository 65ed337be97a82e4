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


@pytest.mark.gpu
@pytest.mark.parametrize("variant", ["", "1,0,0,0,0", "1,2,8,1,0"])
def test_operator_leg_reference_compiled_vs_liborbx(tmp_path, variant):
    env = dict(os.environ)
    if variant:
        env["ORBO_VARIANT"] = variant
    r = subprocess.run([EXE, "--set", _set(tmp_path), "--orbx"], capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "keypoints (all 7 fields as bit patterns), descriptors, return value" in r.stdout and "RESULT: ALL MATCH" in r.stdout, r.stdout
