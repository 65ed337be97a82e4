#!/bin/bash
# A/B of the batch FAST kernel's two forms on ONE box (alternating): ORBX_FAST_PASSES=1 (one pass at minTh, the threshold chosen afterwards: the
# product until round 5) vs 2 (iniTh first, minTh where the cell stayed empty: round 6) — the step, the kernel alone, the natural-crop leg, config 4,
# and the SQ counters of k_fast_cells.   bash tools/fast_passes_ab.sh > gpurun_out/fast_passes_ab.txt
cd "$(dirname "$0")/.."
python -c "from orb_slam3_modified_amd.build import stamp; print(stamp())"
for rep in 1 2 3; do for P in 1 2; do
  echo "fast_passes $P (rep $rep): $(ORBX_FAST_PASSES=$P python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-frontend --no-fixed-streams 2>/dev/null | tail -1 | python -c "
import sys,json; j=json.loads(sys.stdin.read()); print('step', j['ms_per_step'], 'min/max', j['timing']['ms_per_step_min'], j['timing']['ms_per_step_max'], 'value', j['value'], 'k_fast_cells', j['roofline']['kernels_ms_per_launch']['k_fast_cells'], '| natural', j['secondary_natural']['value'], 'k_fast_cells', j['secondary_natural']['kernels_ms_per_launch']['k_fast_cells'], '| config4', j['secondary']['value'])")"
done; done
for P in 1 2; do
  ORBX_FAST_PASSES=$P PMC_SQ_TAG=_passes$P timeout 900 python tools/pmc_sq.py "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES" "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_LDS" > /dev/null 2>&1
  python - <<PY
import json
j=json.load(open("gpurun_out/pmc_sq_passes$P.json"))
for k, v in j["raw_per_dispatch_avg"].items():
    if "fast_cells" in k: print("fast_passes $P counters (synthetic stream, per dispatch):", {c: round(x) for c, x in v.items()})
print("   derived per launch:", {k: {a: round(b, 4) for a, b in v.items()} for k, v in j["derived"].items() if "fast" in k})
PY
done
