"""The two exactness shims the HIP path depends on, checked on the CPU against the host toolchain:
glibc cosf/sinf (csrc/glibc_sincosf.h) and libstdc++ std::sort (csrc/gnu_sort.h); plus the array formulation of
the quadtree that the kernel implements (tests/support/quadtree_model.cpp) against the std::list oracle."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
SUP = os.path.join(HERE, "support")


def _build(src, out, extra=()):
    out = os.path.join(SUP, out)
    srcp = os.path.join(SUP, src)
    deps = [srcp, os.path.join(HERE, "..", "orb_slam3_modified_amd", "csrc", "gnu_sort.h"),
            os.path.join(HERE, "..", "orb_slam3_modified_amd", "csrc", "glibc_sincosf.h")]
    if not os.path.exists(out) or any(os.path.getmtime(d) > os.path.getmtime(out) for d in deps):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", *extra, srcp, "-o", out, "-lpthread"])
    return out


def test_glibc_sincosf_restatement_exhaustive():
    exe = _build("check_sincosf.cpp", "check_sincosf.bin", ["-mfma"])
    # stride 1: EVERY float in [0, 2*pi * 1.0001] (1.09e9 arguments, a few seconds on 8 cores)
    r = subprocess.run([exe, "1"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "0 mismatches" in r.stdout


def test_gnu_sort_equals_std_sort():
    exe = _build("check_gnu_sort.cpp", "check_gnu_sort.bin")
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "heapsort fallbacks exercised" in r.stdout


def _model():
    lib = _build("quadtree_model.cpp", "libquadtree_model.bin", ["-shared", "-fPIC"])
    L = C.CDLL(lib)
    L.qtm_distribute.restype = C.c_int
    return L


def _run_model(L, xs, ys, rs, box, N):
    n = len(xs); cap = n + 8
    xs, ys, rs = (np.ascontiguousarray(v, np.int32) for v in (xs, ys, rs))
    ox, oy, orr = (np.zeros(cap, np.int32) for _ in range(3))
    vp = lambda a: a.ctypes.data_as(C.c_void_p)
    k = L.qtm_distribute(vp(xs), vp(ys), vp(rs), n, box[0], box[1], box[2], box[3], N, vp(ox), vp(oy), vp(orr), cap)
    return ox[:k], oy[:k], orr[:k]


def _run_oracle(po, xs, ys, rs, box, N):
    c = np.zeros(len(xs), po.KP_DTYPE)
    c["x"], c["y"], c["response"] = xs, ys, rs
    r = po.distribute(c, box[0], box[1], box[2], box[3], N)
    return r["x"].astype(np.int32), r["y"].astype(np.int32), r["response"].astype(np.int32)


def test_quadtree_array_model_matches_list_oracle(oracle_lib):
    po = oracle_lib
    L = _model()
    rng = np.random.default_rng(11)
    cases = 0
    for rep in range(600):
        W, H = int(rng.integers(40, 900)), int(rng.integers(40, 700))
        if rng.random() < 0.35:
            W = int(H * rng.uniform(1.5, 3.4))   # 2 and 3 root nodes
        if round(np.float32(W) / np.float32(H)) < 1:
            continue
        n, N = int(rng.integers(0, 1500)), int(rng.integers(1, 400))
        idx = rng.choice(W * H, size=min(n, W * H), replace=False)
        ys, xs = idx // W, idx % W
        rs = rng.integers(7, 7 + int(rng.integers(1, 60)), len(xs))
        box = (16, 16 + W, 16, 16 + H)
        a, b = _run_model(L, xs, ys, rs, box, N), _run_oracle(po, xs, ys, rs, box, N)
        assert len(a[0]) == len(b[0]) and all(np.array_equal(p, q) for p, q in zip(a, b)), (W, H, n, N)
        cases += 1
    assert cases > 400
