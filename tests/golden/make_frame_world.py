#!/usr/bin/env python3
"""Generates tests/golden/frame_world_ref.txt.gz: what the REFERENCE'S OWN src/Frame.cc (all four constructors, ComputeBoW, GetFeaturesInArea,
isInFrustum, ProjectPointDistort, the copy constructor) leaves in a Frame when it runs over the reference's own src/ORBextractor.cc and DBoW2 —
the output of oracle/_ref/ref_frame_world, which oracle/ref_fragments.mk compiles from /root/reference (scenario driver:
tests/support/frame_world.cpp; the images are generated inside the driver).  Run in the build container (needs /root/reference)."""
import gzip
import os
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "-f", "ref_fragments.mk", "_ref/ref_frame_world"])
with tempfile.TemporaryDirectory() as td:
    out = os.path.join(td, "ref.txt")
    subprocess.run([os.path.join(ROOT, "oracle", "_ref", "ref_frame_world"), os.path.join(HERE, "voc_k5_L3.txt"), out], check=True, stdout=subprocess.DEVNULL)
    txt = open(out).read()
with gzip.GzipFile(os.path.join(HERE, "frame_world_ref.txt.gz"), "wb", 9, mtime=0) as f:
    f.write(txt.encode())
print("frame world golden written:", sum(1 for l in txt.splitlines() if not l.startswith(" ")), "frames,", len(txt), "bytes of text")
sys.exit(0)
