# Scratch script for one gpurun call during development (overwritten freely): `gpurun -- 'bash tools/gpu_iter.sh'`.
# The reproducible end-of-round sequence is tools/final_refresh.sh.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_replay.py tests/test_frame_world.py tests/test_bench_contract.py -m gpu -x -q 2>&1 | tail -15
python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>gpurun_out/bench.err | tail -1 > gpurun_out/bench_iter.json; python -c "
import json; j=json.load(open('gpurun_out/bench_iter.json')); print(j['value'], j['ms_per_step'], j['strong'], j['frame_constructor'])"
