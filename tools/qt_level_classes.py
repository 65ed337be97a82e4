#!/usr/bin/env python3
"""k_quadtree per LEVEL CLASS.  The batch quadtree runs as two launches — the big levels (0 .. qt_big_levels-1) and the
small ones — each with its own workgroup size and LDS point capacity.  This tool measures, per configuration, every launch of k_quadtree
on its own (rocprofv3 --kernel-trace, launches told apart by grid and workgroup size), the kernel's total per 256-frame pass, the residency the
launch shape allows (workgroups per CU by LDS, waves per SIMD) and the product-shape step (2 replay lanes x 128 frames) — including the
"one tree per wave" shape the review asked for (64-thread workgroups for the small levels / for all levels).

   python tools/qt_level_classes.py            (driver: runs itself under rocprofv3, prints the table)
   python tools/qt_level_classes.py --workload (what is profiled)"""
import json
import os
import sqlite3
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

# (name, qt_big_levels, threads of the big launch (0 = default 256), threads of the small launch, qt_points)
CONFIGS = [
    ("product: levels 0-1 @256 thr, 2-7 @128 thr", 2, 0, 128, 2048),
    ("small levels one wave per tree (64 thr)", 2, 0, 64, 2048),
    ("small levels @256 thr", 2, 0, 256, 2048),
    ("levels 0-1 @128 thr, 2-7 one wave per tree", 2, 128, 64, 2048),
    ("levels 0-1 @512 thr, 2-7 @128", 2, 512, 128, 2048),
    ("level 0 big, 1-7 small @128", 1, 0, 128, 2048),
    ("level 0 big, 1-7 one wave per tree", 1, 0, 64, 2048),
    ("levels 0-2 big, 3-7 one wave per tree", 3, 0, 64, 2048),
    ("levels 0-2 big, 3-7 @128", 3, 0, 128, 2048),
    ("product shape, 1024 LDS points (512 small)", 2, 0, 128, 1024),
    ("one wave per tree for the small levels, 1024 LDS points (512 small)", 2, 0, 64, 1024),
]
NATURAL = "--natural" in sys.argv


def frames_for(idx):
    import numpy as np
    from orb_slam3_modified_amd import synth
    n = 256 - idx            # a different frame count per configuration: its launches have their own grid in the trace
    if NATURAL:   # bench.py's natural leg: the two 640x480 crops, every copy shifted cyclically by its own (dx, dy)
        nat = np.load(os.path.join(ROOT, "tests", "golden", "natural_crops.npz"))
        crops = [np.ascontiguousarray(nat[k]) for k in ("result_640x480_img", "pineapple_640x480_img")]
        return np.stack([np.roll(crops[i % 2], (7 * (i // 2) % 480, 13 * (i // 2) % 640), (0, 1)) for i in range(n)])
    return synth.make_stream(256)[:n]


def workload():
    import torch
    from orb_slam3_modified_amd import ORBextractor
    from orb_slam3_modified_amd.replay import ReplayEngine
    import time
    res = []
    for idx, (name, big, tbig, tsmall, pts) in enumerate(CONFIGS):
        os.environ["ORBX_QT_BIG_LEVELS"] = str(big)
        os.environ["ORBX_QT_THREADS_SMALL"] = str(tsmall)
        os.environ["ORBX_QT_POINTS"] = str(pts)
        if tbig:
            os.environ["ORBX_QT_THREADS"] = str(tbig)
        else:
            os.environ.pop("ORBX_QT_THREADS", None)
        fr = frames_for(idx)
        ex = ORBextractor(1000, 1.2, 8, 20, 7)
        if tbig:   # "qt_threads" overrides both launches: the small one keeps its own
            ex.set_option("qt_threads", 0)
            ex.set_option("qt_threads_small", tsmall)
        dev = torch.device("cuda", 0)
        d = torch.from_numpy(fr).to(dev)
        # (a) the kernels one after the other, one context: what the trace's per-launch rows are taken from
        kp = torch.empty((len(fr), ex.capacity, 28), dtype=torch.uint8, device=dev)
        ds = torch.empty((len(fr), ex.capacity, 32), dtype=torch.uint8, device=dev)
        ct = torch.empty((len(fr), 2), dtype=torch.int32, device=dev)
        if tbig:
            os.environ["ORBX_QT_THREADS"] = str(tbig)
        for _ in range(12):
            ex.extract_batch_device(d.data_ptr(), len(fr), 480, 640, 640, 480 * 640, kp.data_ptr(), ds.data_ptr(), ct.data_ptr(), (0, 1000))
        torch.cuda.synchronize()
        # (b) the product shape: two lanes on free-running streams
        eng = ReplayEngine(ex, d, lapping=(0, 1000), gather=False, lanes=2)
        for _ in range(5):
            eng.step()
        eng.drain()
        ts = []
        for _ in range(5):
            t0 = time.perf_counter()
            for _ in range(20):
                eng.step()
            eng.drain()
            ts.append((time.perf_counter() - t0) / 20 * 1e3 * 256 / len(fr))
        feats = int(eng.counts(0)[:, 0].sum())
        eng.close()
        res.append(dict(idx=idx, name=name, frames=len(fr), step_ms_per_256=sorted(ts)[2], step_min=min(ts), step_max=max(ts), features_per_frame=feats / len(fr)))
        del ex, eng, d
    print("RESULT " + json.dumps(res))


def main():
    if "--workload" in sys.argv:
        return workload()
    out = os.path.join(ROOT, "gpurun_out", "qt_classes" + ("_nat" if NATURAL else ""))
    os.makedirs(out, exist_ok=True)
    cmd = ["rocprofv3", "--kernel-trace", "-d", out, "-o", "qt", "--", sys.executable, os.path.abspath(__file__), "--workload"] + (["--natural"] if NATURAL else [])
    r = subprocess.run(cmd, env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")]
    if r.returncode != 0 or not line:
        print(r.stdout[-3000:])
        return 1
    res = json.loads(line[-1][7:])
    db = None
    for dp, _, fs in os.walk(out):
        for f in fs:
            if f.endswith("_results.db"):
                db = os.path.join(dp, f)
    c = sqlite3.connect(db)
    rows = c.execute("select grid_x / workgroup_x, grid_y, workgroup_x, count(*), avg(end - start), min(end - start) from kernels "
                     "where name like '%k_quadtree%' group by grid_x / workgroup_x, grid_y, workgroup_x").fetchall()
    from orb_slam3_modified_amd.build import stamp
    print(f"# k_quadtree per level class ({'natural crops' if NATURAL else 'S-EuRoC-640'}, 640x480, 1000 features); {stamp()}")
    print("# per launch: average over the single-context passes (kernels back to back); the step: 2 replay lanes, median of 5 x 20 steps, scaled to 256 frames")
    print("| configuration | big launch (levels x wg threads): us | small launch: us | sum us / 256 frames | step ms / 256 frames (min .. max) |")
    print("|---|---:|---:|---:|---:|")
    for e in res:
        mine = [r_ for r_ in rows if r_[0] == e["frames"] and r_[3] >= 10]
        big = CONFIGS[e["idx"]][1]
        cells = {}
        for gx, gy, wg, n, avg, mn in mine:
            cells["big" if gy == big else "small"] = f"{gy} x {wg}: {avg / 1e3 * 256 / e['frames']:.1f}"
        tot = sum(r_[4] for r_ in mine) / 1e3 * 256 / e["frames"]
        print(f"| {e['name']} | {cells.get('big', '-')} | {cells.get('small', '-')} | {tot:.1f} | {e['step_ms_per_256']:.4f} ({e['step_min']:.4f} .. {e['step_max']:.4f}) |")
    print()
    print(json.dumps(res))
    subprocess.run(["rm", "-rf", out])
    return 0


if __name__ == "__main__":
    sys.exit(main())
