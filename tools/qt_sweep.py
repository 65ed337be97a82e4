#!/usr/bin/env python3
"""k_quadtree alone (HIP events, orbx_profile_*) for several workgroup sizes: 256-frame batch and a single frame."""
import os, sys, hashlib
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from orb_slam3_modified_amd import ORBextractor, synth

frames = synth.make_stream(256)
for nf in (256, 1):
    ref = None
    for qt in (0, 64, 128, 256, 512):
        ex = ORBextractor(1000, 1.2, 8, 20, 7)
        ex.set_option("qt_threads", qt)
        fr = frames[:nf]
        out = ex.extract_batch(fr, (0, 1000))
        dig = hashlib.sha1(b"".join(np.ascontiguousarray(a).tobytes() for o in out for a in o[1:3])).hexdigest()[:12]
        if ref is None: ref = dig
        ex.profile_enable(True)
        for _ in range(20): ex.extract_batch(fr, (0, 1000))
        pr = ex.profile_read()
        ex.profile_enable(False)
        ms, n = pr["k_quadtree"]
        print(f"frames {nf:3d} qt_threads {qt:3d}: k_quadtree {1000 * ms / max(n, 1):8.1f} us per pass   same results: {dig == ref}", flush=True)
