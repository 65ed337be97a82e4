#!/usr/bin/env python3
"""Randomised pin of the oracle (CPU only, no GPU): the REFERENCE's own src/ORBextractor.cc compiled where it lies
(oracle/_ref/libref_orbextractor.so, and its -O3 -mfma build) against oracle/orb_oracle.cpp's restatement, on the configurations of
tools/fuzz_extractor.py (shapes, scale factors, level counts, feature counts, thresholds, lapping areas, synthetic / noise / smooth /
natural windows), optionally under random OpenCV / build variants — both sit over the same restated OpenCV primitives, so what this
sweep pins is everything the reference itself wrote: tables, cell loop, quadtree with its std::sort tie order, IC_Angle, rBRIEF, output order.
    python tools/fuzz_oracle_vs_reference.py [first] [count] [--variants]"""
import os
import sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import pyoracle as po
from orb_slam3_modified_amd import synth

variants = "--variants" in sys.argv
argv = [a for a in sys.argv[1:] if not a.startswith("--")]
first = int(argv[0]) if len(argv) > 0 else 100
count = int(argv[1]) if len(argv) > 1 else 60
if not po.ref_extractor_available(False):
    print("oracle/_ref/libref_orbextractor.so is missing: make -C oracle -f ref_fragments.mk (needs /root/reference)")
    sys.exit(2)
have_fma = po.ref_extractor_available(True)
bad = skipped = 0
for seed in range(first, first + count):
    rng = np.random.default_rng(1000 + seed)
    sf = float(np.float32(rng.choice([1.1, 1.15, 1.2, 1.25, 1.33, 1.5, 1.7, 1.9])))
    nlev = int(rng.integers(1, 9))
    lo = max(int(np.ceil(70 * sf ** (nlev - 1))) + 2, 90)
    rows = int(rng.integers(lo, max(700, lo + 200))); cols = int(rng.integers(lo, max(900, lo + 300)))
    if cols > 8 * rows or rows > 2 * cols:
        rows = cols = max(rows, cols) // 2 + lo
    nf = int(rng.choice([30, 150, 700, 1000, 2500, 6000]))
    ini = int(rng.choice([12, 20, 35])); mn = int(rng.choice([3, 7, ini]))
    lap = tuple(sorted(rng.integers(0, cols + 50, 2).tolist()))
    kind = rng.choice(["synth", "noise", "smooth", "natural"])
    if kind == "natural":
        g = np.load(os.path.join(ROOT, "tests", "golden", "natural_crops.npz"))
        big = g[str(rng.choice(["pineapple_1024x1024_img", "result_752x480_img", "teaser_752x480_img", "result_640x480_img"]))]
        rows, cols = min(rows, big.shape[0]), min(cols, big.shape[1])
        if rows < lo or cols < lo:
            big = g["pineapple_1024x1024_img"]; rows, cols = min(max(rows, lo), 1024), min(max(cols, lo), 1024)
        y0 = int(rng.integers(0, big.shape[0] - rows + 1)); x0 = int(rng.integers(0, big.shape[1] - cols + 1))
        img = np.ascontiguousarray(big[y0:y0 + rows, x0:x0 + cols])
        lap = tuple(sorted(rng.integers(0, cols + 50, 2).tolist()))
    elif kind == "synth":
        img = synth.make_stream(1, rows, cols, 4242 + seed)[0]
    elif kind == "noise":
        img = rng.integers(0, 256, (rows, cols)).astype(np.uint8)
    else:
        img = (synth.make_stream(1, rows, cols, 7 + seed)[0].astype(np.float32) * 0.25 + 90).astype(np.uint8)
    var = (0, 0, 0, 0, 0)
    if variants:
        rnd = int(rng.integers(0, 3))
        var = (int(rng.integers(0, 2)), rnd, int(rng.choice([0, 4, 8, 16, 32, 64])) if rnd else 0, int(rng.integers(0, 2)), int(rng.integers(0, 2)) if have_fma else 0)
    tag = f"seed {seed}: {cols}x{rows} sf{sf:.2f} L{nlev} nf{nf} th{ini}/{mn} lap{lap} {kind}" + (f" variant{var}" if variants else "")
    # the reference divides by zero where a level's nIni = round(w / h) is 0 and has no level-size floor: the configurations liborbx refuses
    undefined = False
    for l in range(nlev):
        wl, hl = round(cols / sf ** l), round(rows / sf ** l)
        # src/ORBextractor.cc:789-803, :559: borders at 16 px, cells of 35 px, nIni = round(width / height) root nodes (0 roots: division by zero)
        if min(wl, hl) < 67 or not (0.5 <= (wl - 32) / max(hl - 32, 1) < 8.5):
            undefined = True
    if undefined:
        skipped += 1
        continue
    with po.opencv_variant(*var):
        okps, odesc, omono = po.OracleExtractor(nf, sf, nlev, ini, mn).extract(img, lap)
        rkps, rdesc, rmono = po.RefExtractor(nf, sf, nlev, ini, mn, fma=bool(var[4])).extract(img, lap)
    ok = omono == rmono and okps.tobytes() == rkps.tobytes() and np.array_equal(odesc, rdesc)
    if not ok:
        bad += 1
        print("MISMATCH", tag, len(okps), len(rkps), omono, rmono)
print(f"{count} configurations: {bad} mismatches, {skipped} skipped (the reference's own undefined cases)")
sys.exit(1 if bad else 0)
