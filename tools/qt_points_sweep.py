import os, sys, hashlib, subprocess, json
import numpy as np
sys.path.insert(0, ".")
from orb_slam3_modified_amd import ORBextractor, synth
frames = synth.make_stream(256)
ref = None
for pts in (2048, 1536, 1280, 1024, 768):
    ex = ORBextractor(1000, 1.2, 8, 20, 7)
    ex.set_option("qt_points", pts)
    out = ex.extract_batch(frames, (0, 1000))
    dig = hashlib.sha1(b"".join(np.ascontiguousarray(a).tobytes() for o in out for a in o[1:3])).hexdigest()[:12]
    ref = ref or dig
    ex.profile_enable(True)
    for _ in range(10): ex.extract_batch(frames, (0, 1000))
    pr = ex.profile_read(); ex.profile_enable(False)
    ms, n = pr["k_quadtree"]
    print(f"qt_points {pts}: k_quadtree {1000*ms/max(n,1):.1f} us per 256 frames, same={dig==ref}", flush=True)
for pts in (2048, 1536, 1024, 768):
    r = subprocess.run([sys.executable, "bench.py", "--steps", "30", "--warmup", "5", "--no-cpu-baseline", "--no-frontend", "--no-verify"],
                       capture_output=True, text=True, env=dict(os.environ, ORBX_QT_POINTS=str(pts)))
    j = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    nat = j["secondary_natural"]
    print(f"bench ORBX_QT_POINTS={pts}: ms_per_step {j['ms_per_step']} value {j['value']} k_quadtree {j['roofline']['kernels_ms_per_launch']['k_quadtree']}"
          f" | natural {nat['value']} k_quadtree {nat['kernels_ms_per_launch']['k_quadtree']} | config 4 {j['secondary']['value']}", flush=True)
