# Scratch script for one gpurun call during development (overwritten freely): `gpurun -- 'bash tools/gpu_iter.sh'`.
# The reproducible end-of-round sequence is tools/final_refresh.sh.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
export ORBX_COMMIT=$(cat .commit_stamp 2>/dev/null)
timeout 1200 python -m pytest tests/test_gpu_extractor.py tests/test_natural_images.py -m gpu -x -q 2>&1 | tail -4
timeout 300 python tools/fuzz_extractor.py 4000 120 2>&1 | tail -2
for P in 0 1; do
  ORBX_FAST_TWOPASS=$P PMC_SQ_TAG=_twopass$P timeout 900 python tools/pmc_sq.py "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES" "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_LDS" > /dev/null 2>&1
  python - <<PY
import json
j=json.load(open("gpurun_out/pmc_sq_twopass$P.json"))
for k, v in j["raw_per_dispatch_avg"].items():
    if "fast_cells" in k: print("twopass $P", k[:60], {c: round(x) for c, x in v.items()})
print("  derived", {k: {a: round(b, 4) for a, b in v.items()} for k, v in j["derived"].items() if "fast" in k})
PY
done
{ python -c "from orb_slam3_modified_amd.build import stamp; print(stamp())"
echo "A/B on one box, alternating: ORBX_FAST_TWOPASS=0 (one pass at minTh, threshold chosen afterwards: the product until round 5) vs 1 (round 6)"
for rep in 1 2 3; do for P in 0 1; do
  echo "twopass $P (rep $rep): $(ORBX_FAST_TWOPASS=$P python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-frontend --no-fixed-streams 2>/dev/null | tail -1 | python -c "
import sys,json; j=json.loads(sys.stdin.read()); print('step', j['ms_per_step'], 'min/max', j['timing']['ms_per_step_min'], j['timing']['ms_per_step_max'], 'value', j['value'], 'k_fast_cells', j['roofline']['kernels_ms_per_launch']['k_fast_cells'], '| natural', j['secondary_natural']['value'], j['secondary_natural']['kernels_ms_per_launch']['k_fast_cells'], '| config4', j['secondary']['value'])")"
done; done; } 2>&1 | tee gpurun_out/fast_twopass_ab.txt
