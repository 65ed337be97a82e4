#!/usr/bin/env python3
"""Fuzz of the Frame.cc drop-in evidence: for every variant v (other scenes, image sizes, feature counts: FRAME_WORLD_VARIANT in
tests/support/frame_world.cpp) run the reference build (oracle/_ref/ref_frame_world: the reference's src/Frame.cc over its own extractor and DBoW2)
and the drop-in build (oracle/_ref/dropin_frame_world: the same src/Frame.cc over include/ORBextractor.h + ORBVocabulary.h + liborbx.so) and
compare every printed Frame field.  usage: tools/fuzz_frame_world.py [first_variant] [count] [--cpu]   (the executables are prebuilt;
--cpu: the drop-in over the oracle-backed stub, no GPU)"""
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CPU_ONLY = "--cpu" in sys.argv   # the drop-in build over the oracle-backed C-ABI stub (oracle/_ref/dropin_frame_world_cpu): no GPU needed
sys.argv = [a for a in sys.argv if a != "--cpu"]
REF, GPU = (os.path.join(ROOT, "oracle", "_ref", n) for n in ("ref_frame_world", "dropin_frame_world_cpu" if CPU_ONLY else "dropin_frame_world"))
VOC = os.path.join(ROOT, "tests", "golden", "voc_k5_L3.txt")
first, count = (int(sys.argv[1]) if len(sys.argv) > 1 else 1), (int(sys.argv[2]) if len(sys.argv) > 2 else 20)
bad = 0
with tempfile.TemporaryDirectory() as td:
    for v in range(first, first + count):
        env = dict(os.environ, FRAME_WORLD_VARIANT=str(v))
        outs = []
        for exe in (REF, GPU):
            out = os.path.join(td, os.path.basename(exe) + ".txt")
            subprocess.run([exe, VOC, out], check=True, stdout=subprocess.DEVNULL, env=env, timeout=900)
            outs.append(open(out).read())
        heads = [l for l in outs[0].splitlines() if not l.startswith(" ")]
        shapes = " ".join(f"{h.split()[0]}:N={h.split('N=')[1].split()[0]}" for h in heads)
        if outs[0] == outs[1]:
            print(f"variant {v}: identical ({len(outs[0])} bytes of fields) {shapes}", flush=True)
        else:
            bad += 1
            la, lb = outs[0].splitlines(), outs[1].splitlines()
            i = next((k for k in range(min(len(la), len(lb))) if la[k] != lb[k]), min(len(la), len(lb)))
            print(f"variant {v}: MISMATCH at line {i + 1}\n  ref    {la[i][:200] if i < len(la) else '<eof>'}\n  dropin {lb[i][:200] if i < len(lb) else '<eof>'}", flush=True)
print(f"{count} variants: {bad} mismatches")
sys.exit(1 if bad else 0)
