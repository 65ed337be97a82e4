"""The host side of the drop-ins under AddressSanitizer + UndefinedBehaviorSanitizer (CPU builds only: GPU sanitizers are not available on the
pool).  csrc/ref_adapter/ORBmatcher.cc (1 500 lines of pre-passes and replays over the reference's objects), ref_adapter/KeyFrameDatabase.cc,
the header-only extractor / vocabulary adapters and the scenario drivers, linked against the oracle-backed stub of the C ABI: every scenario of
the matcher world and the keyframe-database world, and a 30-frame stream through the ring of 8 Frames, must run clean AND produce what the
unsanitised builds produce (tests/test_matcher_world.py, test_kfdb_world.py and test_streamed_frontend.py hold those to the reference)."""
import os
import shutil
import subprocess

import pytest

from tests import world_util as wu

SAN = ["-O1", "-g", "-std=c++17", "-ffp-contract=off", "-w", "-pthread", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer"]
ENV = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0:exitcode=86", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")


def _build(out, srcs, defines=()):
    from oracle import pyoracle
    pyoracle.build()
    odir = os.path.join(wu.ROOT, "oracle")
    stub = os.path.join(wu.SUP, "orbx_oracle_stub.cpp")
    subprocess.check_call(["g++"] + SAN + list(defines) + wu.INCLUDES + srcs + [stub, "-o", out, "-L", odir, "-lorb_oracle", "-Wl,-rpath," + odir])
    return out


def _clean(r):
    assert r.returncode == 0 and "ERROR: AddressSanitizer" not in r.stderr and "runtime error:" not in r.stderr and "LeakSanitizer" not in r.stderr, \
        (r.returncode, r.stderr[-3000:])


@pytest.mark.skipif(shutil.which("g++") is None, reason="g++ not installed")
def test_matcher_and_database_adapters_are_clean_under_asan_and_ubsan(tmp_path):
    world = str(tmp_path / "world.bin")
    wu.write_world(world)
    exe = _build(str(tmp_path / "matcher_world_san"), [os.path.join(wu.SUP, "matcher_world.cpp"), wu.ADAPTER_SRC])
    r = subprocess.run([exe, world, str(tmp_path / "san.txt")], capture_output=True, text=True, env=ENV, timeout=900)
    _clean(r)
    plain = wu.run_world(wu.build_adapter_world("oracle"), world, str(tmp_path / "plain.txt"))
    assert open(tmp_path / "san.txt").read() == plain
    kexe = _build(str(tmp_path / "kfdb_world_san"), [wu.KFDB_SRC, wu.KFDB_ADAPTER_SRC])
    r = subprocess.run([kexe, world, world + ".voc.txt", str(tmp_path / "ksan.txt")], capture_output=True, text=True, env=ENV, timeout=900)
    _clean(r)
    assert open(tmp_path / "ksan.txt").read() == wu.run_kfdb_world(wu.build_kfdb_world("oracle"), world, str(tmp_path / "kplain.txt"))


@pytest.mark.skipif(shutil.which("g++") is None, reason="g++ not installed")
def test_the_streamed_frontend_adapters_are_clean_under_asan_and_ubsan(tmp_path):
    import json
    raw, voc = wu.frontend_inputs(str(tmp_path), 8, 480, 640, 1000)
    exe = _build(str(tmp_path / "frontend_san"), [wu.FRONTEND_SRC, wu.ADAPTER_SRC], defines=["-DORBX_STUB_BACKEND"])
    r = subprocess.run([exe, raw, "480", "640", "8", "1000", voc, "1", "--frames", "30"], capture_output=True, text=True, env=ENV, timeout=900)
    _clean(r)
    got = json.loads(r.stdout.strip().splitlines()[-1])
    plain = wu.run_frontend(wu.build_frontend("oracle"), raw, 480, 640, 8, 1000, voc, 1, frames=30)
    assert got["results_digest"] == plain["results_digest"] and got["frames_timed"] == 22
