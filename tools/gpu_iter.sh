# Scratch script for one gpurun call during development (overwritten freely): `gpurun -- 'bash tools/gpu_iter.sh'`.
# The reproducible end-of-round sequence is tools/final_refresh.sh.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
B="python bench.py --no-cpu-baseline --no-frontend --no-secondary --no-gather --steps 20 --warmup 5"
{ for rep in 1 2 3; do
    echo "default      : $($B 2>/dev/null | tail -1 | python -c 'import json,sys; j=json.loads(sys.stdin.read()); print(j["ms_per_step"], j["timing"]["ms_per_step_min"], j["timing"]["ms_per_step_max"], j["roofline"]["kernels_ms_per_launch"])')"
    echo "chain_batch  : $(ORBX_CHAIN_BATCH=1 ORBX_CHAIN_THREADS=256 $B 2>/dev/null | tail -1 | python -c 'import json,sys; j=json.loads(sys.stdin.read()); print(j["ms_per_step"], j["timing"]["ms_per_step_min"], j["timing"]["ms_per_step_max"], j["roofline"]["kernels_ms_per_launch"])')"
  done; } 2>&1 | tee gpurun_out/chain_batch_ab.txt
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 | tee gpurun_out/iter_tests_full.log
