#!/usr/bin/env python3
"""SQ/TA counter passes over the extractor workload (diagnostics for kernel tuning):
   python tools/pmc_sq.py "SQ_WAVES SQ_WAVE_CYCLES ..." "second pass counters" ...   -> gpurun_out/pmc_sq.md"""
import csv, glob, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
TAG = os.environ.get("PMC_SQ_TAG", "")     # suffix of the output files (A/B runs under different ORBX_* settings)
out = os.path.join(ROOT, "gpurun_out", "pmc_sq" + TAG)
rows = {}
order = []
for i, cs in enumerate(sys.argv[1:]):
    d = os.path.join(out, f"pass{i}")
    os.makedirs(d, exist_ok=True)
    cmd = ["rocprofv3", "--kernel-trace", "--pmc"] + cs.split() + ["-f", "csv", "-d", d, "-o", "p", "--", sys.executable,
           os.path.join(ROOT, "tools", "pmc_traffic.py"), "--workload"]
    r = subprocess.run(cmd, env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        print(f"pass {i} failed: {cs}\n{r.stdout[-2000:]}")
        continue
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"].split("(")[0].split("::")[-1]
            if "calib" in k or not k.startswith("k_"):
                continue
            c = row["Counter_Name"]
            if c not in order:
                order.append(c)
            a = rows.setdefault(k, {}).setdefault(c, [0.0, 0])
            a[0] += float(row["Counter_Value"]); a[1] += 1
lines = ["| kernel | " + " | ".join(order) + " |", "|---|" + "---:|" * len(order)]
for k, cs in rows.items():
    lines.append(f"| {k} | " + " | ".join(f"{cs[c][0] / cs[c][1]:.4g}" if c in cs else "-" for c in order) + " |")
import json
derived = {}
# "launch" = one pass of a kernel over the batch (7 level launches of the pyramid chain, the residency groups of k_fast_cells, the level
# groups of k_quadtree summed); k_assemble runs once per pass and so counts the passes
npass = max((cs[c][1] for k, cs in rows.items() if k.startswith("k_assemble") for c in cs), default=1)
for k, cs in rows.items():
    avg = {c: v[0] / npass for c, v in cs.items()}
    if "GRBM_GUI_ACTIVE" in avg and "SQ_ACTIVE_INST_VALU" in avg:
        cyc = avg["GRBM_GUI_ACTIVE"] / 8.0                      # the counter is summed over the 8 XCDs
        d = {"dispatches_per_launch": cs["GRBM_GUI_ACTIVE"][1] / npass, "cycles_per_launch": cyc,
             "valu_busy": avg["SQ_ACTIVE_INST_VALU"] * 4.0 / (1024.0 * cyc)}   # quad-cycles, 1024 SIMDs
        if "SQ_ACTIVE_INST_LDS" in avg:
            d["lds_busy"] = avg["SQ_ACTIVE_INST_LDS"] * 4.0 / (256.0 * cyc)
            if "SQ_LDS_BANK_CONFLICT" in avg and avg["SQ_ACTIVE_INST_LDS"] > 0:   # share of the LDS-active cycles that is conflict replay
                d["lds_conflict_frac"] = avg["SQ_LDS_BANK_CONFLICT"] / (4.0 * avg["SQ_ACTIVE_INST_LDS"])
        if "TA_BUSY_avr" in avg:
            d["ta_busy"] = avg["TA_BUSY_avr"] / cyc
        if "SQ_INSTS_VALU" in avg:
            d["valu_insts_per_launch"] = avg["SQ_INSTS_VALU"]
        if "SQ_WAVE_CYCLES" in avg:
            d["waves_per_simd"] = avg["SQ_WAVE_CYCLES"] * 4.0 / (1024.0 * cyc)   # average residency
        derived[k.split("<")[0]] = d
from orb_slam3_modified_amd.build import stamp
json.dump({"stamp": stamp(), "derived": derived, "raw_per_dispatch_avg": {k: {c: v[0] / v[1] for c, v in cs.items()} for k, cs in rows.items()}},
          open(os.path.join(ROOT, "gpurun_out", f"pmc_sq{TAG}.json"), "w"), indent=1)
lines += ["", "derived: " + json.dumps(derived)]
open(os.path.join(ROOT, "gpurun_out", f"pmc_sq{TAG}.md"), "w").write("per-dispatch averages\n\n" + "\n".join(lines) + "\n")
print("\n".join(lines))
