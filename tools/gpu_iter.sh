# Scratch script for one gpurun call during development (overwritten freely): `gpurun -- 'bash tools/gpu_iter.sh'`.
# The reproducible end-of-round sequence is tools/final_refresh.sh.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_replay.py tests/test_gpu_matcher_bow.py -x -q 2>&1 | tail -15 | tee gpurun_out/iter_tests_a.log
timeout 900 python -m pytest tests/test_gpu_search.py tests/test_gpu_extractor.py tests/test_opencv_variants.py -x -q -m gpu -k "view or host_rows or staging or named_profile or invalid_target" 2>&1 | tail -8 | tee gpurun_out/iter_tests_b.log
timeout 600 python bench.py --no-cpu-baseline --no-frontend --no-secondary --steps 20 --warmup 5 2> gpurun_out/iter_bench.err | tail -1 > gpurun_out/iter_bench.json; cut -c1-1500 gpurun_out/iter_bench.json; tail -5 gpurun_out/iter_bench.err
timeout 1200 bash tools/fast_pitch_ab.sh 2>&1 | tail -12
# k_describe: the double-buffered window (experiment build gpurun_exp/liborbx_dbuf.so, -DORBX_DESC_DBUF) against the product
{ for rep in 1 2 3; do echo "default: $(python tools/kernel_times.py 256)"; echo "dbuf:    $(ORBX_LIB=$PWD/gpurun_exp/liborbx_dbuf.so python tools/kernel_times.py 256)"; done
  ORBX_LIB=$PWD/gpurun_exp/liborbx_dbuf.so timeout 600 python -m pytest tests/test_gpu_extractor.py -x -q -k "bit_exact or degenerate or other_parameters" 2>&1 | tail -2; } 2>&1 | tee gpurun_out/desc_dbuf_ab.txt
