# Scratch script for one gpurun call during development (overwritten freely): `gpurun -- 'bash tools/gpu_iter.sh'`.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
export ORBX_COMMIT=$(cat .commit_stamp 2>/dev/null)
STAMP=$(python -c "from orb_slam3_modified_amd.build import stamp; s = stamp(); print('commit', s['commit'], 'kernel sources', s['kernels_hash'], s['date'])")
{ echo "$STAMP"; timeout 1500 python tools/fuzz_extractor.py 5000 900 2>&1 | tail -4; } | tee gpurun_out/fuzz_r6b_900.log
{ echo "$STAMP"; timeout 1200 python tools/fuzz_extractor.py 7000 500 --variants 2>&1 | tail -4; } | tee gpurun_out/fuzz_r6b_variants_500.log
{ echo "$STAMP"; timeout 900 python tools/fuzz_worlds.py 900 20 2>&1 | tail -3; } | tee gpurun_out/fuzz_worlds_r6b_20.log
{ echo "$STAMP"; timeout 900 python tools/fuzz_frame_world.py 301 30 2>&1 | tail -2; } | tee gpurun_out/fuzz_frame_world_r6b_30.log
