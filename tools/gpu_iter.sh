cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
STAMP=$(python -c "from orb_slam3_modified_amd.build import stamp; s = stamp(); print('commit', s['commit'], 'kernel sources', s['kernels_hash'], s['date'])")
{ echo "$STAMP"; timeout 1500 python tools/fuzz_extractor.py 20000 900 2>&1 | tail -25; } > gpurun_out/fuzz_r5c_900.log; tail -1 gpurun_out/fuzz_r5c_900.log
{ echo "$STAMP"; timeout 1200 python tools/fuzz_extractor.py 30000 500 --variants 2>&1 | tail -25; } > gpurun_out/fuzz_r5c_variants_500.log; tail -1 gpurun_out/fuzz_r5c_variants_500.log
{ echo "$STAMP"; timeout 1200 python tools/fuzz_worlds.py 900 20 2>&1 | tail -24; } > gpurun_out/fuzz_worlds_r5c_20.log; tail -1 gpurun_out/fuzz_worlds_r5c_20.log
{ echo "$STAMP"; timeout 1200 python tools/fuzz_frame_world.py 300 30 2>&1 | tail -34; } > gpurun_out/fuzz_frame_world_r5c_30.log; tail -1 gpurun_out/fuzz_frame_world_r5c_30.log
