# Scratch script for one gpurun call during development (overwritten freely): `gpurun -- 'bash tools/gpu_iter.sh'`.
# The reproducible end-of-round sequence is tools/final_refresh.sh.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_replay.py tests/test_gpu_matcher_bow.py -x -q 2>&1 | tail -15 | tee gpurun_out/iter_tests_a.log
timeout 900 python -m pytest tests/test_gpu_search.py tests/test_gpu_extractor.py tests/test_opencv_variants.py -x -q -m gpu -k "view or host_rows or staging or named_profile or invalid_target" 2>&1 | tail -8 | tee gpurun_out/iter_tests_b.log
timeout 600 python bench.py --no-cpu-baseline --no-frontend --no-secondary --steps 20 --warmup 5 2> gpurun_out/iter_bench.err | tail -1 > gpurun_out/iter_bench.json; cut -c1-1500 gpurun_out/iter_bench.json; tail -5 gpurun_out/iter_bench.err
timeout 1200 bash tools/fast_pitch_ab.sh 2>&1 | tail -12
