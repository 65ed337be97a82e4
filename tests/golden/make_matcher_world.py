#!/usr/bin/env python3
"""Generates tests/golden/matcher_world.bin.gz (the object-graph inputs of tests/support/matcher_world.cpp) and
tests/golden/matcher_world_ref.txt.gz — the results of the REFERENCE'S OWN src/ORBmatcher.cc on them (all 12 public
routines incl. the two-camera blocks), produced by oracle/_ref/ref_matcher_world, which oracle/ref_fragments.mk compiles
from /root/reference.  Run in the build container (needs /root/reference)."""
import gzip
import os
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from tests import world_util as wu  # noqa: E402

subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "-f", "ref_fragments.mk"])
with tempfile.TemporaryDirectory() as td:
    world, out = os.path.join(td, "world.bin"), os.path.join(td, "ref.txt")
    print(wu.write_world(world))
    txt = wu.run_world(wu.REF_EXE, world, out)
    with gzip.GzipFile(os.path.join(HERE, "matcher_world.bin.gz"), "wb", 9, mtime=0) as f:
        f.write(open(world, "rb").read())
    with gzip.GzipFile(os.path.join(HERE, "matcher_world_ref.txt.gz"), "wb", 9, mtime=0) as f:
        f.write(txt.encode())
    # the same world through the REFERENCE'S OWN src/KeyFrameDatabase.cc (oracle/_ref/ref_kfdb_world); its vocabulary is the
    # world's (world.bin.voc.txt, regenerated deterministically by tests/world_util.write_world)
    ktxt = wu.run_kfdb_world(wu.REF_KFDB_EXE, world, os.path.join(td, "kfdb.txt"))
    with gzip.GzipFile(os.path.join(HERE, "kfdb_world_ref.txt.gz"), "wb", 9, mtime=0) as f:
        f.write(ktxt.encode())
    print("kfdb world golden written:", sum(1 for l in ktxt.splitlines() if not l.startswith("  ")), "result records")
print("matcher world golden written:", sum(1 for l in txt.splitlines() if not l.startswith("  ")), "result records")
