# Scratch script for one gpurun call during development (overwritten freely): `gpurun -- 'bash tools/gpu_iter.sh'`.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
STAMP=$(python -c "from orb_slam3_modified_amd.build import stamp; s = stamp(); print('commit', s['commit'], 'kernel sources', s['kernels_hash'], s['date'])")
timeout 900 python -m pytest tests/test_opencv_variants.py tests/test_validate_opencv.py tests/test_adapters.py -x -q -m gpu 2>&1 | tail -4
{ echo "$STAMP"; echo "k_blur7 per 256 frames of 640x480 under the named CPU-path profiles (tools/kernel_times.py: us, median of 7 x 5 passes), after the general path's rewrite"
  for V in "0 0 0" "1 2 16" "1 2 8" "1 0 0" "1 1 4"; do set -- $V
    echo "gauss_kernel=$1 gauss_round=$2 gauss_tail=$3: $(ORBX_GAUSS_KERNEL=$1 ORBX_GAUSS_ROUND=$2 ORBX_GAUSS_TAIL=$3 python tools/kernel_times.py 256)"; done; } 2>&1 | tee gpurun_out/blur_variants_after.txt
