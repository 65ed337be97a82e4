#!/usr/bin/env python3
"""HBM traffic of the extractor kernels from rocprofv3 PMC counters, collected and corrected the way
/opt/skills/guides/MI355X_MICROARCH.md ("HBM", "rocprofv3 PMC slots") prescribes:

  * FETCH_SIZE and WRITE_SIZE are collected in SEPARATE passes (they do not fit one TCC pass), each pass with
    `--kernel-trace` only (never with sys/hip/hsa tracing);
  * the counters are in KiB and, on gfx950, FETCH_SIZE under-reports wide streaming reads by 2x while other access
    widths are uncalibrated -> every pass also runs `orbx_debug_calib_copy` (a copy kernel with exactly known
    traffic) at 1, 4 and 16 bytes per lane over 512 MiB, and the per-width correction factor
    known_bytes / (counter * 1024) is reported and applied (the extractor kernels use 4-byte accesses for their
    streaming reads/writes, so the 4-byte factor is the one applied).

Run on the GPU box:   python tools/pmc_traffic.py            (driver: two rocprofv3 passes -> gpurun_out/pmc_traffic.json)
The profiled process: python tools/pmc_traffic.py --workload
"""
import argparse
import csv
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CAL_BYTES = 512 << 20
B, H, W = 256, 480, 640
STEPS = 3


def workload():
    import numpy as np
    import torch
    from orb_slam3_modified_amd import ORBextractor, synth
    from orb_slam3_modified_amd.replay import ReplayEngine
    dev = torch.device("cuda", 0)
    ex = ORBextractor(1000, 1.2, 8, 20, 7, device_id=0)
    src = torch.randint(0, 255, (CAL_BYTES,), dtype=torch.uint8, device=dev)
    dst = torch.empty_like(src)
    torch.cuda.synchronize()
    st = torch.cuda.current_stream().cuda_stream
    for width in (1, 4, 16):
        ex.debug_calib_copy(src.data_ptr(), dst.data_ptr(), CAL_BYTES, width, st)
        torch.cuda.synchronize()
    del src, dst
    host = synth.make_stream(B, H, W)            # 256 distinct frames (bench.py's first batch)
    frames = torch.from_numpy(host).to(dev)
    eng = ReplayEngine(ex, frames, lapping=(0, 1000), gather=False)
    for _ in range(STEPS):
        eng.step()
    torch.cuda.synchronize()


def run_pass(counter, outdir):
    os.makedirs(outdir, exist_ok=True)
    env = dict(os.environ, TMPDIR="/tmp")
    cmd = ["rocprofv3", "--kernel-trace", "--pmc", counter, "-f", "csv", "-d", outdir, "-o", counter.lower(), "--",
           sys.executable, os.path.abspath(__file__), "--workload"]
    subprocess.check_call(cmd, env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    files = glob.glob(os.path.join(outdir, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        raise SystemExit(f"no counter_collection.csv under {outdir}")
    per = {}
    for f in files:
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            name = r["Kernel_Name"].split("(")[0]
            per.setdefault(name, []).append(float(r["Counter_Value"]))
    return per


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", action="store_true")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "pmc_traffic.json"))
    a = ap.parse_args()
    if a.workload:
        return workload()
    base = os.path.join(ROOT, "gpurun_out", "pmc")
    fetch = run_pass("FETCH_SIZE", os.path.join(base, "fetch"))
    write = run_pass("WRITE_SIZE", os.path.join(base, "write"))

    def calib(per, suffix):
        # dispatch order of the three calibration copies: width 1, 4, 16 (distinct template instantiations)
        out = {}
        for name, vals in per.items():
            if "k_calib_copy" in name:
                t = name.split("<")[-1].split(">")[0].strip()
                width = {"unsigned char": 1, "unsigned int": 4}.get(t, 16)
                out[width] = CAL_BYTES / (vals[0] * 1024.0) if vals[0] > 0 else None
        return out

    cf, cw = calib(fetch, "f"), calib(write, "w")
    f4, w4 = cf.get(4) or 1.0, cw.get(4) or 1.0
    kernels = {}
    names = sorted(set(list(fetch) + list(write)))
    for n in names:
        if "orbx::k_" not in n or "calib" in n:
            continue
        short = n.split("::")[-1].split("<")[0]
        fv, wv = fetch.get(n, []), write.get(n, [])
        # "launch" = one pass of the kernel over the batch: the 7 level launches of the pyramid chain, the two residency groups of
        # k_fast_cells, the two level groups of k_quadtree are summed (STEPS passes were run)
        fpl = sum(fv) / STEPS * 1024.0
        wpl = sum(wv) / STEPS * 1024.0
        ent = {"dispatches_per_pass": len(fv) / STEPS, "fetch_raw_bytes_per_launch": fpl, "write_raw_bytes_per_launch": wpl,
               "fetch_bytes_per_launch": fpl * f4, "write_bytes_per_launch": wpl * w4,
               "hbm_bytes_per_launch": fpl * f4 + wpl * w4}
        kernels[short] = ent
    from orb_slam3_modified_amd.build import stamp
    res = {"stamp": stamp(), "batch": B, "rows": H, "cols": W, "steps": STEPS,
           "method": "rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE in separate passes; KiB -> bytes; corrected by the "
                     "4-byte-per-lane factor of the known-traffic calibration copy (512 MiB) run in the same pass",
           "calibration_factor_fetch": {str(k): v for k, v in sorted(cf.items())},
           "calibration_factor_write": {str(k): v for k, v in sorted(cw.items())},
           "kernels": kernels}
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    json.dump(res, open(a.out, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
