# Scratch script for one gpurun call during development (overwritten freely): `gpurun -- 'bash tools/gpu_iter.sh'`.
# The reproducible end-of-round sequence is tools/final_refresh.sh.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
export ORBX_COMMIT=$(cat .commit_stamp 2>/dev/null)
timeout 900 python tools/qt_level_classes.py > gpurun_out/qt_level_classes.txt 2>&1; tail -20 gpurun_out/qt_level_classes.txt | cut -c1-400
timeout 900 python tools/qt_level_classes.py --natural > gpurun_out/qt_level_classes_nat.txt 2>&1; tail -18 gpurun_out/qt_level_classes_nat.txt | cut -c1-400
python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>gpurun_out/bench.err | tail -1 > gpurun_out/bench_iter.json; python -c "
import json; j=json.load(open('gpurun_out/bench_iter.json')); print(j['value'], j['ms_per_step'], json.dumps(j['matcher_roofline'])[:1500])"
timeout 600 python -m pytest tests/test_gpu_search.py tests/test_bench_contract.py -m gpu -x -q 2>&1 | tail -5
