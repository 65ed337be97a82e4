cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
STAMP=$(python -c "from orb_slam3_modified_amd.build import stamp; s = stamp(); print('commit', s['commit'], 'kernel sources', s['kernels_hash'], s['date'])")
ORBX_LIB=$PWD/gpurun_exp/liborbx_ends.so timeout 900 python -m pytest tests/test_gpu_extractor.py tests/test_natural_images.py -x -q -m gpu 2>&1 | tail -3
ORBX_LIB=$PWD/gpurun_exp/liborbx_ends.so timeout 300 python tools/fuzz_extractor.py 3000 80 2>&1 | tail -1
{ echo "$STAMP"; echo "k_fast_cells: candidate list filled from both ends by the two waves (-DORBX_FAST_ENDS, no LDS atomic per trip) against the product; tools/kernel_times.py, us per 256 frames"
  for rep in 1 2 3; do echo "product: $(python tools/kernel_times.py 256)"; echo "ends:    $(ORBX_LIB=$PWD/gpurun_exp/liborbx_ends.so python tools/kernel_times.py 256)"; done
  B="python bench.py --no-cpu-baseline --no-frontend --no-secondary --no-gather --steps 20 --warmup 5"
  brief='import json,sys; j=json.loads(sys.stdin.read()); print(j["ms_per_step"], j["timing"]["ms_per_step_min"], j["timing"]["ms_per_step_max"], j["value"])'
  for rep in 1 2; do echo "product bench: $($B 2>/dev/null | tail -1 | python -c "$brief")"; echo "ends bench:    $(ORBX_LIB=$PWD/gpurun_exp/liborbx_ends.so $B 2>/dev/null | tail -1 | python -c "$brief")"; done; } 2>&1 | tee gpurun_out/fast_ends_ab.txt
