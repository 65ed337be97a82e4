"""Inputs and builds of tests/support/matcher_world.cpp (see its header): the world file (four views of one synthetic
stream: keypoints, descriptors, feature vectors) and the two executables — the reference's src/ORBmatcher.cc
(oracle/_ref/ref_matcher_world, built by oracle/ref_fragments.mk where /root/reference exists; prebuilt elsewhere) and
this repository's drop-in (tests/support/matcher_world*.bin)."""
from __future__ import annotations

import os
import struct
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SUP = os.path.join(ROOT, "tests", "support")
PKG = os.path.join(ROOT, "orb_slam3_modified_amd")
REF_EXE = os.path.join(ROOT, "oracle", "_ref", "ref_matcher_world")
ADAPTER_SRC = os.path.join(PKG, "csrc", "ref_adapter", "ORBmatcher.cc")
INCLUDES = ["-I", os.path.join(ROOT, "include"), "-I", os.path.join(SUP, "ref_world"), "-I", os.path.join(ROOT, "oracle", "ref_shims")]
CXXFLAGS = ["-O2", "-std=c++17", "-ffp-contract=off", "-Wall", "-Wno-unused-function", "-pthread"]


def write_world(path: str, rows: int = 480, cols: int = 752, nfeatures: int = 1000, steps=(0, 2, 4, 6), seed: int = 20260925,
                levelsup: int = 2) -> dict:
    """Views = frames `steps` of one synthetic stream, extracted by the oracle; vocabulary k=8, L=3 over their descriptors."""
    from oracle import pyoracle as po
    from orb_slam3_modified_amd import synth
    from tests.vocab_util import make_vocabulary
    frames = synth.make_stream(max(steps) + 1, rows, cols, seed)
    ora = po.OracleExtractor(nfeatures, 1.2, 8, 20, 7)
    views = []
    for t in steps:
        kps, desc, _ = ora.extract(frames[t], (0, 0))
        views.append((kps, desc))
    vocp = path + ".voc.txt"
    make_vocabulary(vocp, np.concatenate([d for _, d in views]), 8, 3, seed=5)
    voc = po.OracleVocabulary(vocp)
    tb = ora.tables()
    with open(path, "wb") as f:
        f.write(struct.pack("<iiii", 0x0b5e55ed, rows, cols, 8))
        f.write(tb["scale"].astype("<f4").tobytes())
        f.write(tb["sigma2"].astype("<f4").tobytes())
        f.write(tb["inv_sigma2"].astype("<f4").tobytes())
        f.write(struct.pack("<ff", np.float32(1.2), np.float32(np.log(np.float32(1.2)))))
        f.write(struct.pack("<i", len(views)))
        for kps, desc in views:
            _, fv = voc.transform(desc, levelsup)
            f.write(struct.pack("<i", len(kps)))
            f.write(kps.tobytes())
            f.write(np.ascontiguousarray(desc).tobytes())
            pairs = [(k, i) for k in sorted(fv) for i in fv[k]]
            f.write(struct.pack("<i", len(pairs)))
            f.write(np.array(pairs, "<u4").reshape(-1, 2).tobytes())
    return dict(n=[len(k) for k, _ in views])


def _stale(out: str, deps) -> bool:
    return not os.path.exists(out) or any(os.path.getmtime(d) > os.path.getmtime(out) for d in deps if os.path.exists(d))


def _deps():
    d = [os.path.join(SUP, "matcher_world.cpp"), os.path.join(SUP, "world_scene.h"), ADAPTER_SRC]
    for base in (os.path.join(ROOT, "include"), os.path.join(SUP, "ref_world"), os.path.join(ROOT, "oracle", "ref_shims", "opencv2", "core")):
        for dp, _, fs in os.walk(base):
            d += [os.path.join(dp, f) for f in fs]
    return d


def build_adapter_world(backend: str) -> str:
    """backend 'orbx': link the drop-in against liborbx.so (needs a GPU to run); 'oracle': against the oracle-backed stub
    of the C-ABI (tests/support/orbx_oracle_stub.cpp) — the drop-in's host logic on a CPU."""
    out = os.path.join(SUP, f"matcher_world_{backend}.bin")
    srcs = [os.path.join(SUP, "matcher_world.cpp"), ADAPTER_SRC]
    if backend == "orbx":
        from orb_slam3_modified_amd import build
        build.build()
        # no OpenCV calibration in the product-path builds: the shim's cv::GaussianBlur would pull the oracle in as "the OpenCV at hand"
        # (tests/test_adapters.py has the one GPU build that does exactly that, on purpose)
        link = ["-DORBX_NO_CV_CALIBRATION", "-L", PKG, "-lorbx", "-Wl,-rpath," + PKG, "-Wl,--allow-shlib-undefined"]
        deps = _deps() + [os.path.join(PKG, "liborbx.so")]
    else:
        from oracle import pyoracle
        pyoracle.build()
        srcs.append(os.path.join(SUP, "orbx_oracle_stub.cpp"))
        odir = os.path.join(ROOT, "oracle")
        link = ["-L", odir, "-lorb_oracle", "-Wl,-rpath," + odir]
        deps = _deps() + [srcs[-1], os.path.join(odir, "liborb_oracle.so")]
    if _stale(out, deps):
        subprocess.check_call(["g++"] + CXXFLAGS + INCLUDES + srcs + ["-o", out] + link)
    return out


def run_world(exe: str, world: str, out: str, only: str = "", time_json: str = "") -> str:
    r = subprocess.run([exe, world, out] + ([only] if only else []) + (["--time", time_json] if time_json else []), capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"{exe} failed ({r.returncode}): {r.stdout}{r.stderr}")
    return open(out).read()


def first_difference(a: str, b: str) -> str:
    la, lb = a.splitlines(), b.splitlines()
    scen = ""
    for i, (x, y) in enumerate(zip(la, lb)):
        if not x.startswith("  "):
            scen = x
        if x != y:
            xs, ys = x.split(), y.split()
            k = next((j for j, (p, q) in enumerate(zip(xs, ys)) if p != q), min(len(xs), len(ys)))
            return f"scenario '{scen}', line {i}: token {k}: {xs[k:k + 6]} != {ys[k:k + 6]} ({x[:60]}...)"
    return f"lengths differ: {len(la)} vs {len(lb)} lines" if len(la) != len(lb) else ""


# ---- tools/streamed_frontend.cpp: the per-frame Tracking sequence (extract -> BoW -> SearchByProjection x2) as one loop ----
FRONTEND_SRC = os.path.join(ROOT, "tools", "streamed_frontend.cpp")
REF_FRONTEND_EXE = os.path.join(ROOT, "oracle", "_ref", "ref_streamed_frontend")


def build_frontend(backend: str, outdir: str = SUP) -> str:
    """The drop-in build of tools/streamed_frontend.cpp: backend 'orbx' links liborbx.so, 'oracle' the oracle-backed C-ABI stub."""
    out = os.path.join(outdir, f"streamed_frontend_{backend}.bin")
    srcs = [FRONTEND_SRC, ADAPTER_SRC]
    if backend == "orbx":
        from orb_slam3_modified_amd import build
        build.build()
        link = ["-DORBX_NO_CV_CALIBRATION", "-L", PKG, "-lorbx", "-Wl,-rpath," + PKG, "-Wl,--allow-shlib-undefined"]
        deps = _deps() + [FRONTEND_SRC, os.path.join(PKG, "liborbx.so")]
    else:
        from oracle import pyoracle
        pyoracle.build()
        srcs.append(os.path.join(SUP, "orbx_oracle_stub.cpp"))
        odir = os.path.join(ROOT, "oracle")
        link = ["-DORBX_STUB_BACKEND", "-L", odir, "-lorb_oracle", "-Wl,-rpath," + odir]
        deps = _deps() + [FRONTEND_SRC, srcs[-1], os.path.join(odir, "liborb_oracle.so")]
    if _stale(out, deps):
        subprocess.check_call(["g++"] + CXXFLAGS + INCLUDES + srcs + ["-o", out] + link)
    return out


def frontend_inputs(tmpdir: str, nframes: int = 12, rows: int = 480, cols: int = 640, nfeatures: int = 1000, seed: int = 20260925, k: int = 10,
                    L: int = 4):
    """frames.raw of one synthetic stream + a k-ary vocabulary over the descriptors of a few of its frames."""
    from oracle import pyoracle as po
    from orb_slam3_modified_amd import synth
    from tests.vocab_util import make_vocabulary
    frames = synth.make_stream(nframes, rows, cols, seed)
    raw, vocp = os.path.join(tmpdir, "frames.raw"), os.path.join(tmpdir, "voc.txt")
    np.ascontiguousarray(frames).tofile(raw)
    ora = po.OracleExtractor(nfeatures, 1.2, 8, 20, 7)
    descs = [ora.extract(frames[t], (0, 1000))[1] for t in range(0, nframes, max(1, nframes // 4))]
    make_vocabulary(vocp, np.concatenate(descs), k, L, seed=9)
    return raw, vocp


MH01_STAMPS_GZ = os.path.join(ROOT, "tests", "golden", "mh01_stamps.txt.gz")


def mh01_stamps(tmpdir: str) -> str:
    """The 3 682 image time stamps of EuRoC MH_01 (fixture made by tools/make_mh01_stamps.py), unpacked for --timestamps."""
    import gzip
    out = os.path.join(tmpdir, "MH01.txt")
    with gzip.open(MH01_STAMPS_GZ, "rb") as f, open(out, "wb") as g:
        g.write(f.read())
    return out


def run_frontend(exe: str, raw: str, rows: int, cols: int, nframes: int, nfeatures: int, vocp: str, passes: int = 1, timeout: int = 600,
                 frames: int = 0, stamps: str = "", pace: int = 0) -> dict:
    """frames > 0: the long form (`--frames`: that many frames per pass, forth and back through the `nframes` images, ring of 8)."""
    import json
    extra = (["--frames", str(frames)] if frames else []) + (["--timestamps", stamps] if stamps else []) + (["--pace", str(pace)] if pace else [])
    r = subprocess.run([exe, raw, str(rows), str(cols), str(nframes), str(nfeatures), vocp, str(passes)] + extra, capture_output=True, text=True,
                       timeout=timeout)
    if r.returncode != 0:
        raise RuntimeError(f"{exe} failed ({r.returncode}): {r.stdout}{r.stderr}")
    return json.loads(r.stdout.strip().splitlines()[-1])


# ---- tests/support/kfdb_world.cpp: the KeyFrameDatabase scenarios ----
KFDB_SRC = os.path.join(SUP, "kfdb_world.cpp")
KFDB_ADAPTER_SRC = os.path.join(PKG, "csrc", "ref_adapter", "KeyFrameDatabase.cc")
REF_KFDB_EXE = os.path.join(ROOT, "oracle", "_ref", "ref_kfdb_world")


def build_kfdb_world(backend: str) -> str:
    """The drop-in build of tests/support/kfdb_world.cpp (include/KeyFrameDatabase.h + csrc/ref_adapter/KeyFrameDatabase.cc)."""
    out = os.path.join(SUP, f"kfdb_world_{backend}.bin")
    srcs = [KFDB_SRC, KFDB_ADAPTER_SRC]
    if backend == "orbx":
        from orb_slam3_modified_amd import build
        build.build()
        link = ["-L", PKG, "-lorbx", "-Wl,-rpath," + PKG, "-Wl,--allow-shlib-undefined"]
        deps = _deps() + srcs + [os.path.join(PKG, "liborbx.so")]
    else:
        from oracle import pyoracle
        pyoracle.build()
        srcs.append(os.path.join(SUP, "orbx_oracle_stub.cpp"))
        odir = os.path.join(ROOT, "oracle")
        link = ["-L", odir, "-lorb_oracle", "-Wl,-rpath," + odir]
        deps = _deps() + srcs + [os.path.join(odir, "liborb_oracle.so")]
    if _stale(out, deps):
        subprocess.check_call(["g++"] + CXXFLAGS + INCLUDES + srcs + ["-o", out] + link)
    return out


def run_kfdb_world(exe: str, world: str, out: str, time_json: str = "") -> str:
    r = subprocess.run([exe, world, world + ".voc.txt", out] + (["--time", time_json] if time_json else []), capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"{exe} failed ({r.returncode}): {r.stdout}{r.stderr}")
    return open(out).read()
