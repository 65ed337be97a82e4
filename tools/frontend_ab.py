#!/usr/bin/env python3
"""The streamed front-end leg of bench.py on its own, optionally under rocprofv3: python tools/frontend_ab.py [--print-cmd]
Environment knobs under test are simply set by the caller (e.g. ORBX_BOW_IN_GRAPH=0)."""
import json, os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from orb_slam3_modified_amd import ORBextractor, synth
from tests import world_util as wu
from tests.vocab_util import make_vocabulary

host = synth.make_stream(48, 480, 640, synth.DEFAULT_SEED)
tmp = tempfile.mkdtemp(prefix="orbx_front_")
raw, vocp = os.path.join(tmp, "frames.raw"), os.path.join(tmp, "voc.txt")
host.tofile(raw)
fex = ORBextractor(1000, 1.2, 8, 20, 7)
vdesc = [fex(host[t], None, (0, 1000))[2] for t in range(0, 48, 8)]
del fex
make_vocabulary(vocp, np.concatenate(vdesc), 10, 5, seed=9)
exe = wu.build_frontend("orbx", tmp)
cmd = [exe, raw, "480", "640", "48", "1000", vocp, "3"]
if "--print-cmd" in sys.argv:
    print(" ".join(cmd))
else:
    out = wu.run_frontend(exe, raw, 480, 640, 48, 1000, vocp, passes=3, timeout=300)
    print(json.dumps({k: out[k] for k in out if k.endswith("_ms") or k == "ms_per_frame"}))
