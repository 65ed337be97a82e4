"""In-tree build of liborbx.so — the product — and of liborbx_debug.so — the diagnostic ABI (include/orbx_debug.h: stage dumps and numeric test
hooks for the parity tests and the profiling tools; the product library has none of them).  HIP, gfx950 only.
`python -m orb_slam3_modified_amd.build [--force]`."""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.environ.get("ORBX_BUILD_OUT") or os.path.join(HERE, "liborbx.so")   # ORBX_BUILD_OUT (+ ORBX_EXTRA_FLAGS): experiment builds beside the product
SOURCES = ["orbx_extractor.hip", "orbx_matcher.hip", "orbx_search.hip", "orbx_window.hip", "orbx_kfdb.hip", "orbx_replay.hip"]
DEBUG_OUT = os.path.join(os.path.dirname(OUT), "liborbx_debug.so")
DEBUG_SOURCE = "orbx_debug.hip"     # = the extractor's translation unit with ORBX_DEBUG_ABI; -fvisibility=hidden: exports orbx_debug_* alone
# -ffp-contract=off: the float paths (fastAtan2 polynomial, BRIEF rotation) must not be fused into FMAs,
# the CPU reference evaluates them as separate IEEE operations (DESIGN.md "bit-exactness").
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", "-Wall",
         "-Wno-unused-function", "-ldl"]


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
    if not os.path.exists(OUT) or not os.path.exists(DEBUG_OUT):
        return True
    t = min(os.path.getmtime(OUT), os.path.getmtime(DEBUG_OUT))
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, "..", "include", "orbx.h"), os.path.join(HERE, "..", "include", "orbx_debug.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = True) -> str:
    """One object per source (compiled in parallel, rebuilt only when the source or a header is newer), then one link."""
    if not (force or _stale()):
        return OUT
    from concurrent.futures import ThreadPoolExecutor
    hipcc = os.environ.get("HIPCC", "hipcc")
    extra = os.environ.get("ORBX_EXTRA_FLAGS", "").split()
    objdir = os.path.join(HERE, "build") if OUT.endswith("/liborbx.so") else OUT + ".obj"
    os.makedirs(objdir, exist_ok=True)
    cflags = [f for f in FLAGS if f not in ("-shared", "-ldl")] + extra
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if not f.endswith(".hip") or f == "orbx_kernels.hip"] + \
              [os.path.join(HERE, "..", "include", "orbx.h"), os.path.join(HERE, "..", "include", "orbx_debug.h")]
    hnew = max(os.path.getmtime(h) for h in headers if os.path.isfile(h))
    tag = os.path.join(objdir, ".flags")
    flags_now = " ".join([hipcc] + cflags)
    same_flags = os.path.exists(tag) and open(tag).read() == flags_now

    def compile_one(src):
        obj = os.path.join(objdir, src.replace(".hip", ".o"))
        sp = os.path.join(CSRC, src)
        newest = max(os.path.getmtime(sp), hnew)
        if src == DEBUG_SOURCE:   # it IS the extractor's translation unit
            newest = max(newest, os.path.getmtime(os.path.join(CSRC, "orbx_extractor.hip")))
        if not force and same_flags and os.path.exists(obj) and os.path.getmtime(obj) > newest:
            return obj
        cmd = [hipcc] + cflags + (["-fvisibility=hidden"] if src == DEBUG_SOURCE else []) + ["-c", sp, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd, cwd=CSRC)
        return obj

    with ThreadPoolExecutor(max_workers=len(SOURCES) + 1) as ex:
        objs = list(ex.map(compile_one, SOURCES + [DEBUG_SOURCE]))
    open(tag, "w").write(flags_now)
    # the debug library is the extractor's translation unit alone: what that unit takes from the others (the vocabulary's device view, ...) it takes
    # from liborbx.so at load time ($ORIGIN)
    for out, oo, more in ((OUT, objs[:-1], []), (DEBUG_OUT, objs[-1:], ["-L", os.path.dirname(OUT), "-l:" + os.path.basename(OUT), "-Wl,-rpath,$ORIGIN"])):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + oo + more + ["-ldl"]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd, cwd=CSRC)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
