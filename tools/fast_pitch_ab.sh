#!/bin/bash
# Round-5 experiment: k_fast_cells on the 64- / 80- / 96-byte LDS tile pitch: isolated kernel times (HIP events, median of 7 x 5
# passes over 256 frames) and the SQ counters of the same workload per pitch -> gpurun_out/fast_pitch_ab.txt, gpurun_out/pmc_sq_pitch*.{json,md}
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
mkdir -p gpurun_out
STAMP=$(python -c "from orb_slam3_modified_amd.build import stamp; s = stamp(); print('commit', s['commit'], 'kernel sources', s['kernels_hash'], s['date'])")
{ echo "$STAMP"; echo
  for rep in 1 2; do for P in 0 80 96; do echo "fast_pitch $P (rep $rep): $(ORBX_FAST_PITCH=$P python tools/kernel_times.py 256)"; done; done
  for P in 0 80 96; do
    ORBX_FAST_PITCH=$P PMC_SQ_TAG=_pitch$P timeout 600 python tools/pmc_sq.py "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_LDS_IDX_ACTIVE" > /dev/null 2>&1
    python - <<PY
import json
j = json.load(open("gpurun_out/pmc_sq_pitch$P.json"))
d = j["derived"].get("k_fast_cells", {})
raw = [v for k, v in j["raw_per_dispatch_avg"].items() if k.startswith("k_fast_cells")]
print("fast_pitch $P counters:", {k: round(v, 4) for k, v in d.items()}, {k: raw[0].get(k) for k in ("SQ_LDS_BANK_CONFLICT", "SQ_ACTIVE_INST_LDS", "SQ_LDS_IDX_ACTIVE", "SQ_INSTS_LDS")} if raw else None)
PY
  done; } 2>&1 | tee gpurun_out/fast_pitch_ab.txt
