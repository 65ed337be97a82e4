#!/usr/bin/env python3
"""Randomised worlds through the three scenario drivers: the REFERENCE's own src/ORBmatcher.cc / src/KeyFrameDatabase.cc /
per-frame loop (oracle/_ref/*, prebuilt) against the drop-ins on the GPU, outputs compared as text / digests.
    python tools/fuzz_worlds.py [first_seed] [count] [--backend oracle]
--backend oracle: the drop-ins linked against the oracle-backed C-ABI stub instead of liborbx.so — no GPU needed; what it pins at scale is the
adapters' own logic (pre-pass, replay order, side effects, the keyframe database's bookkeeping) against the reference's files."""
import os, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests import world_util as wu

backend = "oracle" if "--backend" in sys.argv and sys.argv[sys.argv.index("--backend") + 1] == "oracle" else "orbx"
_argv = [a for i, a in enumerate(sys.argv) if not a.startswith("--") and (i == 0 or sys.argv[i - 1] != "--backend")]
first = int(_argv[1]) if len(_argv) > 1 else 100
count = int(_argv[2]) if len(_argv) > 2 else 20
mexe, kexe, fexe = wu.build_adapter_world(backend), wu.build_kfdb_world(backend), wu.build_frontend(backend)
bad = 0
for seed in range(first, first + count):
    rng = np.random.default_rng(seed)
    rows, cols = [(480, 640), (480, 752), (376, 1241), (512, 512), (350, 600), (600, 800)][int(rng.integers(0, 6))]
    nf = int(rng.choice([500, 800, 1000, 1500, 2000]))
    steps = tuple(sorted(rng.choice(np.arange(0, 10), 4, replace=False).tolist()))
    td = tempfile.mkdtemp(prefix="orbx_fw_")
    world = os.path.join(td, "world.bin")
    info = wu.write_world(world, rows=rows, cols=cols, nfeatures=nf, steps=steps, seed=seed)
    tag = f"seed {seed}: {cols}x{rows} nf{nf} steps{steps} n={info['n']}"
    ok = True
    a, b = wu.run_world(wu.REF_EXE, world, os.path.join(td, "r.txt")), wu.run_world(mexe, world, os.path.join(td, "g.txt"))
    if a != b: ok = False; print("MATCHER MISMATCH", tag, wu.first_difference(a, b))
    a, b = wu.run_kfdb_world(wu.REF_KFDB_EXE, world, os.path.join(td, "kr.txt")), wu.run_kfdb_world(kexe, world, os.path.join(td, "kg.txt"))
    if a != b: ok = False; print("KFDB MISMATCH", tag, wu.first_difference(a, b))
    nfr = 8
    raw, voc = wu.frontend_inputs(td, nfr, rows, cols, nf, seed=seed, k=int(rng.choice([6, 8, 10])), L=int(rng.choice([3, 4, 5])))
    r, g = wu.run_frontend(wu.REF_FRONTEND_EXE, raw, rows, cols, nfr, nf, voc, 1), wu.run_frontend(fexe, raw, rows, cols, nfr, nf, voc, 1)
    if r["results_digest"] != g["results_digest"]: ok = False; print("FRONTEND MISMATCH", tag, r["results_digest"], g["results_digest"])
    bad += not ok
    print(("ok       " if ok else "MISMATCH ") + tag, flush=True)
print(f"{count} worlds: {bad} mismatching")
sys.exit(1 if bad else 0)
