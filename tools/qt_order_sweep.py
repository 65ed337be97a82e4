import os, sys, hashlib, subprocess, json
import numpy as np
sys.path.insert(0, ".")
from orb_slam3_modified_amd import ORBextractor, synth
frames = synth.make_stream(256)
ref = None
for lm, one in ((0, 0), (1, 0), (0, 1), (1, 1)):
    ex = ORBextractor(1000, 1.2, 8, 20, 7)
    ex.set_option("qt_level_major", lm); ex.set_option("qt_one_launch", one)
    out = ex.extract_batch(frames, (0, 1000))
    dig = hashlib.sha1(b"".join(np.ascontiguousarray(a).tobytes() for o in out for a in o[1:3])).hexdigest()[:12]
    ref = ref or dig
    ex.profile_enable(True)
    for _ in range(10): ex.extract_batch(frames, (0, 1000))
    pr = ex.profile_read(); ex.profile_enable(False)
    ms, n = pr["k_quadtree"]
    print(f"level_major {lm} one_launch {one}: k_quadtree {1000*ms/max(n,1):.1f} us per 256 frames, same={dig==ref}", flush=True)
for lm, one in ((0, 0), (1, 0), (1, 1)):
    r = subprocess.run([sys.executable, "bench.py", "--steps", "40", "--warmup", "5", "--no-secondary", "--no-cpu-baseline", "--no-frontend", "--no-verify"],
                       capture_output=True, text=True, env=dict(os.environ, ORBX_QT_LEVEL_MAJOR=str(lm), ORBX_QT_ONE_LAUNCH=str(one)))
    j = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    print(f"bench level_major={lm} one_launch={one}: ms_per_step {j['ms_per_step']} value {j['value']}", flush=True)
