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
