#!/usr/bin/env python3
"""Isolated per-kernel times (HIP events, orbx_profile_*) of one 256-frame 640x480 batch; median of several passes."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from orb_slam3_modified_amd import ORBextractor, synth

nf = int(sys.argv[1]) if len(sys.argv) > 1 else 256
frames = synth.make_stream(nf)
ex = ORBextractor(1000, 1.2, 8, 20, 7)
ex.extract_batch(frames, (0, 1000))
rows = []
for rep in range(7):
    ex.profile_enable(True)
    for _ in range(5): ex.extract_batch(frames, (0, 1000))
    pr = ex.profile_read()
    ex.profile_enable(False)
    rows.append({k: 1000.0 * ms / max(n, 1) for k, (ms, n) in pr.items()})
keys = list(rows[0].keys())
med = {k: float(np.median([r[k] for r in rows])) for k in keys}
print("  ".join(f"{k.split('(')[0]} {v:7.1f}" for k, v in med.items()), f" sum {sum(med.values()):7.1f} us")
