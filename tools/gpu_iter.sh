# Scratch script for one gpurun call during development (overwritten freely): `gpurun -- 'bash tools/gpu_iter.sh'`.
# The reproducible end-of-round sequence is tools/final_refresh.sh.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
export ORBX_COMMIT=$(cat .commit_stamp 2>/dev/null)
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -8
timeout 400 python tools/fuzz_extractor.py 400 150 2>&1 | tail -3
python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>gpurun_out/bench.err | tail -1 > gpurun_out/bench_iter.json; python -c "
import json; j=json.load(open('gpurun_out/bench_iter.json')); print(j['value'], j['ms_per_step'], j['timing']['ms_per_step_all'], j['roofline']['kernels_ms_per_launch'], j['secondary_natural']['value'], j['strong']['value'])"
