#!/usr/bin/env python3
"""Phase timing of k_quadtree (needs a build with ORBX_EXTRA_FLAGS=-DORBX_QT_PROFILE).
Rounds 4 - 5 tool (profiles/qt_profile_r4*.txt).  Since the diagnostic ABI moved into liborbx_debug.so (round 6) its read-out hook sits in a library whose
copy of the counters no kernel writes: to use it again, build the hook (orbx_debug_qt_profile, csrc/orbx_extractor.hip) into the profiling library itself."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from orb_slam3_modified_amd import ORBextractor, synth, _lib
L = _lib.lib()
ex = ORBextractor(1000, 1.2, 8, 20, 7)
frames = synth.make_stream(8)
ex.extract_batch(frames, (0, 1000))
buf = (C.c_longlong * (16 * 8))()
L.orbx_debug_qt_profile(ex._ctx, buf, 1)
ex.extract_batch(frames, (0, 1000))
L.orbx_debug_qt_profile(ex._ctx, buf, 1)
a = np.array(buf[:]).reshape(16, 8)[:8, :8]
print("cycles (100 MHz wall clock -> x10 ns): gather, full passes, sort, sorted rest, final | sort: rounds, final rank, number of sorts")
print(a * 10 / 1000.0, "us")
one = frames[:1]
ex.extract_batch(one, (0, 1000))
L.orbx_debug_qt_profile(ex._ctx, buf, 1)
ex.extract_batch(one, (0, 1000))
L.orbx_debug_qt_profile(ex._ctx, buf, 1)
b = np.array(buf[:]).reshape(16, 8)[:8, :8]
print("single frame:")
print(b * 10 / 1000.0, "us")
for l in range(8):
    print(l, len(ex.debug_level_points(l, 0)[0]), len(ex.debug_level_points(l, 1)[0]))
