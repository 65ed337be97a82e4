#!/usr/bin/env python3
"""Per-call wall time of the 12 ORBmatcher routines on the object graphs of tests/support/matcher_world.cpp: the reference's own
src/ORBmatcher.cc on this machine's CPU (oracle/_ref/ref_matcher_world) beside the drop-in on the GPU (host buffers in and out:
PCIe-inclusive).  python tools/matcher_times.py -> gpurun_out/matcher_times.json + a markdown table on stdout."""
import gzip, json, os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests import world_util as wu

td = tempfile.mkdtemp()
world = os.path.join(td, "world.bin")
open(world, "wb").write(gzip.open(os.path.join(ROOT, "tests", "golden", "matcher_world.bin.gz")).read())
res = {}
if os.path.exists(wu.REF_EXE):
    wu.run_world(wu.REF_EXE, world, os.path.join(td, "ref.txt"), time_json=os.path.join(td, "ref.json"))
    res["reference_cpu"] = json.load(open(os.path.join(td, "ref.json")))
exe = wu.build_adapter_world("orbx")
out = wu.run_world(exe, world, os.path.join(td, "gpu.txt"), time_json=os.path.join(td, "gpu.json"))
res["dropin_gpu"] = json.load(open(os.path.join(td, "gpu.json")))
gold = gzip.open(os.path.join(ROOT, "tests", "golden", "matcher_world_ref.txt.gz")).read().decode()
res["results_identical_to_reference"] = out == gold
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(res, open(os.path.join(ROOT, "gpurun_out", "matcher_times.json"), "w"), indent=1)
print("| scenario | calls | reference src/ORBmatcher.cc, 1 CPU core (ms/call) | drop-in on the GPU, host buffers (ms/call) |\n|---|---:|---:|---:|")
for k, v in res["dropin_gpu"].items():
    r = res.get("reference_cpu", {}).get(k, {}).get("ms_per_call")
    print(f"| {k} | {v['calls_per_run']} | {r if r is None else round(r, 3)} | {round(v['ms_per_call'], 3)} |")
print("results identical to the reference:", res["results_identical_to_reference"])
