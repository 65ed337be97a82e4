#!/usr/bin/env python3
"""Steady-state latency of a search on a resident target (orbx_target_search / _nearest): the frame is uploaded once, then searched repeatedly.
ORBX_TRACE_WINDOW=1 prints the phases of every call."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from orb_slam3_modified_amd import ORBextractor, ORBmatcher, synth

gpu = ORBextractor(1000, 1.2, 8, 20, 7)
fr = synth.make_stream(2)
out = gpu.extract_batch(fr, (0, 0))
k2, d2, d1 = out[1][1], out[1][2], out[0][2]
m = ORBmatcher(gpu)
rng = np.random.default_rng(3)
nq = min(len(d1), 800)
src = rng.integers(0, len(k2), nq)
qx = (k2["x"][src] + rng.normal(0, 2, nq)).astype(np.float32); qy = (k2["y"][src] + rng.normal(0, 2, nq)).astype(np.float32)
lvl = k2["octave"][src].astype(np.int32)
grid = dict(min_x=0.0, min_y=0.0, inv_w=64 / 640.0, inv_h=48 / 480.0, cell_start=None, cell_idx=None)
T = m.Target(k2, d2, grid)
for name, r in (("radius 3 px x scale (SearchByProjection, th = 3)", 3.0), ("radius 15 px x scale", 15.0)):
    qr = (np.float32(r) * np.float32(1.2) ** lvl).astype(np.float32)
    for want_lists in (True, False):
        for _ in range(5): T.search(qx, qy, qr, lvl - 1, lvl, d1[:nq], want_lists=want_lists)
        t0 = time.perf_counter()
        for _ in range(200): res = T.search(qx, qy, qr, lvl - 1, lvl, d1[:nq], want_lists=want_lists)
        dt = (time.perf_counter() - t0) / 200
        print(f"{name}: {nq} queries, lists={want_lists}, {int(res['row_ptr'][-1])} candidates: {dt * 1e6:.1f} us per call (python caller)")

# ---- first search of a NEW frame (target created from host arrays, then searched) against a search on a resident target: what the
# front-end tail is measured by.  Two ways of making the target: host arrays with a caller-held grid (round 2) and host arrays
# with the grid built on the device (what the adapters do).  (Rounds 3-4 had a third: descriptor rows handed over from the extractor's staging
# block — 58.0 us against 51.2, profiles/target_latency_r4.txt — removed in round 5.)
import ctypes as C
print()
qr = (np.float32(3.0) * np.float32(1.2) ** lvl).astype(np.float32)
from oracle import pyoracle as po
cs, ci = po.assign_grid(k2, np.float32(0), np.float32(0), np.float32(64 / 640.0), np.float32(48 / 480.0))
held = dict(grid, cell_start=cs, cell_idx=ci)
img = fr[1]
R = m.Target(k2, d2, held)     # the recycled target: the adapters' LRU refills a device block (orbx_target_assign), it does not allocate
def first_search(make):
    ts = []
    for rep in range(60):
        mono, kk, dd = gpu(img, None, (0, 0))          # a fresh extraction: the context's staging block holds these rows
        dd = np.ascontiguousarray(dd)
        t0 = time.perf_counter()
        make(kk, dd)
        R.search(qx, qy, qr, lvl - 1, lvl, d1[:nq], want_lists=False)
        ts.append(time.perf_counter() - t0)
    return 1e6 * float(np.median(ts[10:]))
def resident():
    for _ in range(5): T.search(qx, qy, qr, lvl - 1, lvl, d1[:nq], want_lists=False)
    t0 = time.perf_counter()
    for _ in range(200): T.search(qx, qy, qr, lvl - 1, lvl, d1[:nq], want_lists=False)
    return 1e6 * (time.perf_counter() - t0) / 200
r = resident()
a = first_search(lambda kk, dd: R.assign(kk, dd, held))
b = first_search(lambda kk, dd: R.assign(kk, dd, grid))
print(f"search on a resident target: {r:.1f} us")
print(f"first search of a new frame = refill a recycled target (orbx_target_assign) + search (python caller, median of 50):")
print(f"  host arrays + caller-held grid (round 2's way): {a:.1f} us  (+{a - r:.1f})")
print(f"  host arrays, grid built on the device:           {b:.1f} us  (+{b - r:.1f})")
