#!/usr/bin/env python3
"""Timeline of the LAST `n` device operations (kernel dispatches and, when the run traced them, memory copies) of a rocprofv3 run
(rocpd database): start offset, duration and the gap to the previous operation's end — where the microseconds of one single-frame
operator() call go.
    python tools/frame_timeline.py prof_results.db [n]"""
import sqlite3
import sys

db, n = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 14
c = sqlite3.connect(db)
rows = [(r[0], r[1], r[2], r[3]) for r in c.execute("select name, start, end, grid_x / workgroup_x from kernels")]
try:
    rows += [(f"copy {r[0]} ({r[3]} B)", r[1], r[2], 0) for r in c.execute("select name, start, end, size from memory_copies")]
except sqlite3.Error as e:
    print("(no memory copies in this database:", e, ")")
rows.sort(key=lambda r: r[1])
rows = rows[-n:]
t0 = rows[0][1]
prev_end = t0
print(f"{'operation':44s} {'start us':>9s} {'dur us':>8s} {'gap us':>7s} {'wgs':>6s}")
for name, s, e, wgs in rows:
    print(f"{name[:44]:44s} {(s - t0) / 1e3:9.2f} {(e - s) / 1e3:8.2f} {(s - prev_end) / 1e3:7.2f} {wgs:6d}")
    prev_end = e
print(f"span {(rows[-1][2] - t0) / 1e3:.2f} us, busy {sum(e - s for _, s, e, _ in rows) / 1e3:.2f} us")
