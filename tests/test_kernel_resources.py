"""Resource guards on the compiled gfx950 kernels (no GPU needed: hipcc cross-compiles to assembly).

A kernel with a private segment (scratch: register spills, or a local object the compiler could not keep in registers) pays a scratch
set-up on every queue that first runs it and makes every dispatch depend on the scratch allocation — measured in round 2 as a 1.5 ms
first launch of k_quadtree on a new stream and an 11x slower kernel under counter collection.  Every kernel of the library must
therefore report `.amdhsa_private_segment_fixed_size 0`."""
import os
import re
import shutil
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "orb_slam3_modified_amd", "csrc")


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="hipcc not installed")
def test_no_kernel_uses_scratch_memory():
    from orb_slam3_modified_amd.build import FLAGS, SOURCES
    flags = [f for f in FLAGS if f not in ("-shared", "-fPIC")]
    tmp = tempfile.mkdtemp(prefix="orbx_asm_")
    procs = []
    for src in SOURCES:
        out = os.path.join(tmp, src + ".s")
        procs.append((src, out, subprocess.Popen(["hipcc"] + flags + ["-S", "--cuda-device-only", "-o", out, os.path.join(CSRC, src)],
                                                 stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    kernels, bad = 0, []
    dma = {}   # kernel symbol -> number of LDS-DMA loads in its body
    for src, out, p in procs:
        log, _ = p.communicate()
        assert p.returncode == 0, log[-2000:]
        name = None
        body = None
        for line in open(out):
            m = re.match(r"(_Z\w+):\s", line)
            if m:
                body = m.group(1)
            if body and "global_load_lds_dword" in line:
                dma[body] = dma.get(body, 0) + 1
            m = re.match(r"\s*\.amdhsa_kernel\s+(\S+)", line)
            if m:
                name = m.group(1)
                kernels += 1
            m = re.match(r"\s*\.amdhsa_private_segment_fixed_size\s+(\d+)", line)
            if m and int(m.group(1)) != 0:
                bad.append((src, name, int(m.group(1))))
    shutil.rmtree(tmp, ignore_errors=True)
    assert kernels >= 30
    assert not bad, bad
    # the kernels that stage a tile / rectangle / window fill it by LDS-DMA: the builtin must have survived the compiler
    for frag in ("k_fast_cellsILi128ELi64ELb1", "k_fast_blurILi64ELb1", "k_resize", "k_blur7", "k_describeILi4ELb1ELb0"):
        hit = [k for k in dma if frag in k]
        assert hit, (frag, sorted(dma))
