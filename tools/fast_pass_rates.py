#!/usr/bin/env python3
"""where do the FAST cells of natural images spend what the synthetic stream does not?  Pass rates per stage of k_fast_cells,
computed on the CPU (numpy + the oracle's FAST), per 35-px cell of every pyramid level:
  pre@7 / pre@20   pixels that pass the 4-pair necessary test (stage B) at minTh / iniTh          -> entries of the candidate list
  corner@7 / @20   pixels whose exact score reaches the threshold (stage C decides)
  cells with an iniTh survivor (the reference keeps iniTh's keypoints there, src/ORBextractor.cc:826-850) and exact-score TRIPS per cell
  (128-thread workgroup: ceil(list / 128)) with one list (product until round 5) and with the list split by class (round 6).
   python tools/fast_pass_rates.py            (no GPU)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import pyoracle as po  # noqa: E402
from orb_slam3_modified_amd import synth  # noqa: E402


def pretest(img, t):
    """The necessary test of stage B on the whole image: for the four opposite pairs of the 16-circle, every pair holds a pixel darker than
    v - t, or every pair holds a pixel brighter than v + t."""
    I = img.astype(np.int32)
    H, W = I.shape
    c = I[3:H - 3, 3:W - 3]
    def at(dx, dy): return I[3 + dy:H - 3 + dy, 3 + dx:W - 3 + dx]
    pairs = [(at(0, 3), at(0, -3)), (at(3, 0), at(-3, 0)), (at(2, 2), at(-2, -2)), (at(2, -2), at(-2, 2))]
    mins = np.stack([np.minimum(a, b) for a, b in pairs]); maxs = np.stack([np.maximum(a, b) for a, b in pairs])
    ok = (mins.max(0) < c - t) | (maxs.min(0) > c + t)
    out = np.zeros((H, W), bool)
    out[3:H - 3, 3:W - 3] = ok
    return out


def stats(frames, name):
    ora = po.OracleExtractor(1000, 1.2, 8, 20, 7)
    tot = dict(cells=0, px=0, pre7=0, pre20=0, c7=0, c20=0, ini_cells=0, trips_one=0, trips_a=0, trips_both=0, empty20=0)
    for f in frames:
        ora.extract(f, (0, 1000))
        for l in range(8):
            img = ora.level(l)
            H, W = img.shape
            p7, p20 = pretest(img, 7), pretest(img, 20)
            sc = np.zeros((H, W), np.int32)          # 7 where the pixel is a corner at minTh, 20 where it is one at iniTh (cv::FAST without NMS)
            for th in (7, 20):
                k = po.fast(img, th, nms=False)
                sc[k["y"].astype(int), k["x"].astype(int)] = th
            x0, y0, bw, bh = 16 - 3, 16 - 3, W - 32 + 6, H - 32 + 6          # the border box (src/ORBextractor.cc:789-803)
            nc, nr = max((W - 32) // 35, 1), max((H - 32) // 35, 1)
            cw, ch = int(np.ceil((W - 32) / nc)), int(np.ceil((H - 32) / nr))
            for r in range(nr):
                for c in range(nc):
                    ys, xs = 16 + r * ch, 16 + c * cw
                    ye, xe = min(ys + ch, H - 16), min(xs + cw, W - 16)
                    a7, a20 = int(p7[ys:ye, xs:xe].sum()), int(p20[ys:ye, xs:xe].sum())
                    s = sc[ys:ye, xs:xe]
                    tot["cells"] += 1; tot["px"] += s.size; tot["pre7"] += a7; tot["pre20"] += a20
                    tot["c7"] += int((s >= 7).sum()); tot["c20"] += int((s >= 20).sum())
                    ini = bool((s >= 20).any())       # a corner at iniTh exists <=> an NMS survivor at iniTh exists (the cell's maximum survives)
                    tot["ini_cells"] += ini
                    tot["empty20"] += a20 == 0
                    tot["trips_one"] += -(-a7 // 128)
                    tot["trips_a"] += -(-a20 // 128)
                    tot["trips_both"] += -(-a20 // 128) + (0 if ini else -(-(a7 - a20) // 128))
    n = tot["cells"]
    print(f"| {name} | {n / len(frames):.0f} | {100 * tot['pre7'] / tot['px']:.1f} % | {100 * tot['pre20'] / tot['px']:.1f} % | {tot['pre7'] / n:.0f} | {tot['pre20'] / n:.0f} | "
          f"{tot['c7'] / n:.0f} | {tot['c20'] / n:.0f} | {100 * tot['ini_cells'] / n:.0f} % | {100 * tot['empty20'] / n:.0f} % | {tot['trips_one'] / n:.2f} | {tot['trips_both'] / n:.2f} |")


if __name__ == "__main__":
    nat = np.load(os.path.join(ROOT, "tests", "golden", "natural_crops.npz"))
    print("| stream | cells / frame | pre-test pass @7 | @20 | list entries / cell @7 | @20 | corners / cell @7 | @20 | cells with an iniTh corner | cells without a @20 entry | exact-score trips / cell, one list | split by class |")
    print("|---|---:|---:|---:|---:|---:|---:|---:|---:|---:|---:|---:|")
    stats(list(synth.make_stream(6)), "S-EuRoC-640 (6 frames)")
    stats([np.ascontiguousarray(nat[k]) for k in ("result_640x480_img", "pineapple_640x480_img")], "natural crops 640x480 (2)")
