"""In-tree build of liborbx.so (HIP, gfx950 only).  `python -m orb_slam3_modified_amd.build [--force]`."""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "liborbx.so")
SOURCES = ["orbx_extractor.hip", "orbx_matcher.hip", "orbx_search.hip", "orbx_window.hip", "orbx_kfdb.hip"]
# -ffp-contract=off: the float paths (fastAtan2 polynomial, BRIEF rotation) must not be fused into FMAs,
# the CPU reference evaluates them as separate IEEE operations (DESIGN.md "bit-exactness").
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", "-Wall",
         "-Wno-unused-function"]


def kernels_hash() -> str:
    """sha256 over the kernel sources (csrc/*.hip, *.h, *.inc, sorted by name), first 16 hex digits: what the committed counter files
    (profiles/pmc_*.json) are stamped with, so that bench.py can tell whether they were measured on THIS code."""
    import hashlib
    h = hashlib.sha256()
    for f in sorted(os.listdir(CSRC)):
        if f.endswith((".hip", ".h", ".inc")):
            h.update(f.encode() + b"\0" + open(os.path.join(CSRC, f), "rb").read())
    return h.hexdigest()[:16]


def stamp() -> dict:
    """{kernels_hash, commit, date} for evidence files: the commit comes from ORBX_COMMIT (the GPU box has no .git) or `git rev-parse`."""
    import datetime
    commit = os.environ.get("ORBX_COMMIT")
    if not commit:
        try:
            commit = subprocess.check_output(["git", "-C", HERE, "rev-parse", "--short=12", "HEAD"], stderr=subprocess.DEVNULL).decode().strip()
        except Exception:   # noqa: BLE001
            commit = "unknown"
    return {"kernels_hash": kernels_hash(), "commit": commit, "date": datetime.datetime.utcnow().strftime("%Y-%m-%dT%H:%MZ")}


def _stale() -> bool:
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, "..", "include", "orbx.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = True) -> str:
    if not (force or _stale()):
        return OUT
    hipcc = os.environ.get("HIPCC", "hipcc")
    cmd = [hipcc] + FLAGS + os.environ.get("ORBX_EXTRA_FLAGS", "").split() + ["-o", OUT] + [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd, cwd=CSRC)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
