#!/usr/bin/env python3
"""Generates tools/opencv_pin/expected_digests.txt: what every NAMED CPU-path profile (orbx_set_cpu_profile) computes,
primitive by primitive, on the validation image set — as 64-bit digests, so that tools/validate_opencv.cpp can tell a maintainer WHICH profile
the OpenCV at hand is, from OpenCV's own outputs alone.  The digests are made by running that same program over the container shim, whose cv::
functions are the oracle's under ORBO_VARIANT (oracle/_ref/validate_opencv, built by oracle/ref_fragments.mk where /root/reference exists).
They pin the ORACLE's arithmetic per profile, not OpenCV's: no OpenCV exists in the build container.

    python tools/opencv_pin/make_expected.py [--check]      (--check: compare with the committed file instead of writing it)"""
import os
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
EXE = os.path.join(ROOT, "oracle", "_ref", "validate_opencv")
OUT = os.path.join(HERE, "expected_digests.txt")
sys.path.insert(0, ROOT)


def digests(set_path, variant):
    env = dict(os.environ, ORBO_VARIANT=",".join(str(v) for v in variant))
    r = subprocess.run([EXE, "--set", set_path, "--print-digests"], capture_output=True, text=True, env=env)
    if r.returncode != 0:
        raise SystemExit(f"validate_opencv failed under {variant}:\n{r.stdout[-2000:]}{r.stderr[-500:]}")
    return dict(l.split()[1:3] for l in r.stdout.splitlines() if l.startswith("DIGEST "))


def table():
    from orb_slam3_modified_amd import _lib
    import ctypes as C
    import numpy as np
    L = _lib.lib()
    td = tempfile.mkdtemp()
    set_path = os.path.join(td, "validate_set.bin")
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "make_validate_set.py"), set_path, "--synthetic", "1"], stdout=subprocess.DEVNULL)
    lines = ["# expected digests per named CPU-path profile, made by tools/opencv_pin/make_expected.py over the oracle (recalled arithmetic: DESIGN.md section 2)",
             "# <profile | *> <primitive> <fnv-1a 64 of the primitive's outputs on the set of tools/make_validate_set.py --synthetic 1>"]
    base = digests(set_path, (0, 0, 0, 0, 0))
    lines.append(f"* set {base['set']}")
    for k in ("resize", "fast20", "fast7"):
        lines.append(f"* {k} {base[k]}")
    i = 0
    while True:
        nm = L.orbx_cpu_profile_name(i)
        if not nm:
            break
        v = np.zeros(5, np.int32)
        assert L.orbx_cpu_profile_values(nm, 0, _lib.ptr(v)) == 0
        d = digests(set_path, tuple(int(x) for x in v))
        assert all(d[k] == base[k] for k in ("set", "resize", "fast20", "fast7")), "a profile changed a profile-independent primitive"
        lines.append(f"{nm.decode()} blur {d['blur']}")
        i += 1
    for fma in (0, 1):
        lines.append(f"atan_fma={fma} atan {digests(set_path, (0, 0, 0, fma, 0))['atan']}")
    return "\n".join(lines) + "\n"


if __name__ == "__main__":
    if not os.path.exists(EXE):
        raise SystemExit(f"{EXE} not built (oracle/ref_fragments.mk, needs /root/reference)")
    txt = table()
    if "--check" in sys.argv:
        sys.exit(0 if open(OUT).read() == txt else 1)
    open(OUT, "w").write(txt)
    print(txt, end="")
