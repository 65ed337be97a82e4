#!/usr/bin/env python3
"""The fork's own configuration (Examples/Monocular/mi.yaml: 600 x 800 portrait, 20 000 features on ONE level, iniTh 20 / minTh 7) by corner
density: operator() per frame and the per-kernel split of a one-frame batch, every result checked against the oracle.
    python tools/mi_yaml_times.py"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import pyoracle as po
from orb_slam3_modified_amd import ORBextractor, synth

H, W = 800, 600
nat = np.load(os.path.join(ROOT, "tests", "golden", "natural_crops.npz"))
crop = np.ascontiguousarray(nat["result_640x480_img"])
rng = np.random.default_rng(5)
images = {
    "synthetic": synth.make_stream(1, H, W, synth.DEFAULT_SEED + 77)[0],
    "natural (crop tiled)": np.ascontiguousarray(np.tile(crop, (2, 2))[:H, :W]),
    "smoothed noise": None,
    "white noise": rng.integers(0, 256, (H, W), dtype=np.uint8),
}
n = rng.normal(0, 1, (H, W)).astype(np.float32)
k = np.array([1, 4, 6, 4, 1], np.float32) / 16
for ax in (0, 1): n = sum(np.roll(n, s - 2, ax) * k[s] for s in range(5))
images["smoothed noise"] = np.clip(128 + 160 * n, 0, 255).astype(np.uint8)
for name, img in images.items():
    ex = ORBextractor(20000, 1.2, 1, 20, 7)
    for _ in range(3): r = ex(img, None, (0, 0))
    t0 = time.perf_counter()
    for _ in range(20): r = ex(img, None, (0, 0))
    ms = (time.perf_counter() - t0) / 20 * 1e3
    frames = img[None]
    ex.extract_batch(frames, (0, 0))
    ex.profile_enable(True)
    for _ in range(5): ex.extract_batch(frames, (0, 0))
    pr = ex.profile_read(); ex.profile_enable(False)
    split = "  ".join(f"{k_.split('(')[0]} {1000.0 * ms_ / max(n_, 1):.0f}" for k_, (ms_, n_) in pr.items())
    ref_k, ref_d, _ = po.OracleExtractor(20000, 1.2, 1, 20, 7).extract(img, (0, 0))
    ok = "equal to the oracle" if (len(ref_k) == len(r[1]) and np.array_equal(ref_k, r[1]) and np.array_equal(ref_d, r[2])) else "MISMATCH"
    print(f"{name:22s} features {len(r[1]):6d}  operator() {ms:7.3f} ms/frame   one-frame batch, us: {split}   [{ok}]", flush=True)
    ex.close()
