#!/usr/bin/env python3
"""k_quadtree on the natural crops (1 650 - 2 300 corners on level 0: more than the default 2 048 LDS-resident points per level) against the
synthetic stream, by `qt_points` (LDS point capacity per big level): HIP-event kernel times of 256-frame batches.
    python tools/natural_qt_points.py"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from orb_slam3_modified_amd import ORBextractor, synth
from orb_slam3_modified_amd.replay import BlockLayout

dev = torch.device("cuda", 0)
nat = np.load(os.path.join(ROOT, "tests", "golden", "natural_crops.npz"))
crops = [np.ascontiguousarray(nat[k]) for k in ("result_640x480_img", "pineapple_640x480_img")]
B = 256
sets = {"natural": np.stack([np.roll(crops[i % 2], (7 * (i // 2) % 480, 13 * (i // 2) % 640), (0, 1)) for i in range(B)]), "synthetic": synth.make_stream(B)}
for name, host in sets.items():
    frames = torch.from_numpy(host).to(dev)
    for qp in (2048, 2560, 3072, 4096):
        ex = ORBextractor(1000, 1.2, 8, 20, 7)
        ex.set_option("qt_points", qp)
        lo = BlockLayout(B, ex.capacity)
        blk = torch.zeros(lo.nbytes, dtype=torch.uint8, device=dev)
        st = torch.cuda.Stream(device=dev)
        def run():
            ex.extract_batch_device(frames.data_ptr(), B, 480, 640, frames.stride(1), frames.stride(0), blk.data_ptr(), blk.data_ptr() + lo.desc_off,
                                    blk.data_ptr() + lo.counts_off, (0, 1000), st.cuda_stream)
        run(); st.synchronize()
        rows = []
        for rep in range(5):
            ex.profile_enable(True)
            for _ in range(4): run()
            st.synchronize()
            pr = ex.profile_read(); ex.profile_enable(False)
            rows.append({k: 1000.0 * ms / max(n, 1) for k, (ms, n) in pr.items()})
        med = {k: float(np.median([r[k] for r in rows])) for k in rows[0]}
        print(f"{name:9s} qt_points {qp}: k_quadtree {med['k_quadtree']:.1f} us, sum of kernels {sum(med.values()):.1f} us", flush=True)
