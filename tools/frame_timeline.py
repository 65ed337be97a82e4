#!/usr/bin/env python3
"""Timeline of the LAST `n` kernel dispatches of a rocprofv3 --kernel-trace run (rocpd database): start offset, duration and the gap
to the previous kernel's end — where the microseconds of one single-frame operator() call go.
    python tools/frame_timeline.py prof_results.db [n]"""
import sqlite3
import sys

db, n = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 14
c = sqlite3.connect(db)
rows = c.execute("select name, start, end, grid_x / workgroup_x from kernels order by start").fetchall()
rows = rows[-n:]
t0 = rows[0][1]
prev_end = t0
print(f"{'kernel':40s} {'start us':>9s} {'dur us':>8s} {'gap us':>7s} {'wgs':>6s}")
for name, s, e, wgs in rows:
    print(f"{name[:40]:40s} {(s - t0) / 1e3:9.2f} {(e - s) / 1e3:8.2f} {(s - prev_end) / 1e3:7.2f} {wgs:6d}")
    prev_end = e
print(f"span {(rows[-1][2] - t0) / 1e3:.2f} us, kernels {sum(e - s for _, s, e, _ in rows) / 1e3:.2f} us")
