cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
STAMP=$(python -c "from orb_slam3_modified_amd.build import stamp; s = stamp(); print('commit', s['commit'], 'kernel sources', s['kernels_hash'], s['date'])")
timeout 900 python -m pytest tests/test_gpu_extractor.py tests/test_natural_images.py tests/test_frame_world.py -x -q -m gpu 2>&1 | tail -3
timeout 400 python tools/fuzz_extractor.py 5000 200 2>&1 | tail -1
{ echo "$STAMP"; echo "quadtree: a sort of <= 257 elements by ONE wave (product) against the level-synchronous rounds of the whole workgroup (-DORBX_QT_NO_WAVE_SORT)"
  for rep in 1 2 3; do echo "one-wave sort: $(python tools/kernel_times.py 256)"; echo "block rounds:  $(ORBX_LIB=$PWD/gpurun_exp/liborbx_nowsort.so python tools/kernel_times.py 256)"; done
  for rep in 1 2 3; do echo "one-wave sort, single frame: $(python tools/one_frame_trace.py 300)"; echo "block rounds,  single frame: $(ORBX_LIB=$PWD/gpurun_exp/liborbx_nowsort.so python tools/one_frame_trace.py 300)"; done; } 2>&1 | tee gpurun_out/qt_wave_sort_ab.txt
for L in product nowsort; do
  if [ $L = product ]; then unset ORBX_LIB; else export ORBX_LIB=$PWD/gpurun_exp/liborbx_$L.so; fi
  rocprofv3 --kernel-trace -d gpurun_out/tl_$L -o tl -- python tools/one_frame_trace.py 100 > gpurun_out/tl_$L.log 2>&1
  TDB=$(ls gpurun_out/tl_$L/*/tl_results.db gpurun_out/tl_$L/tl_results.db 2>/dev/null | head -1)
  echo "== $L"; python tools/frame_timeline.py "$TDB" 6 | tee -a gpurun_out/qt_wave_sort_ab.txt; rm -rf gpurun_out/tl_$L
done
