"""Single-frame operator() loop for `rocprofv3 --kernel-trace` (where do the ~0.17 ms of one live frame go?)."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from orb_slam3_modified_amd.extractor import ORBextractor
from orb_slam3_modified_amd import synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
img = synth.make_stream(1, 480, 640, synth.DEFAULT_SEED)[0]
ex = ORBextractor(1000, 1.2, 8, 20, 7)
for _ in range(5):
    ex(img)
t0 = time.perf_counter()
for _ in range(n):
    mono, kps, desc = ex(img)
dt = (time.perf_counter() - t0) / n
print(f"frames {n}  ms/frame {dt*1e3:.4f}  features {len(kps)}")
