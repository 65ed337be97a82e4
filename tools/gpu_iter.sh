cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
export ORBX_COMMIT=14f17ed20d20
STAMP=$(python -c "from orb_slam3_modified_amd.build import stamp; s = stamp(); print('commit', s['commit'], 'kernel sources', s['kernels_hash'], s['date'])")
{ echo "$STAMP"; timeout 1500 python tools/fuzz_extractor.py 5000 900 2>&1 | tail -4; } > gpurun_out/fuzz_r6f_900.log
{ echo "$STAMP"; timeout 1200 python tools/fuzz_extractor.py 7000 500 --variants 2>&1 | tail -4; } > gpurun_out/fuzz_r6f_variants_500.log
{ echo "$STAMP"; timeout 900 python tools/fuzz_worlds.py 900 20 2>&1 | tail -3; } > gpurun_out/fuzz_worlds_r6f_20.log
{ echo "$STAMP"; timeout 900 python tools/fuzz_frame_world.py 301 30 2>&1 | tail -2; } > gpurun_out/fuzz_frame_world_r6f_30.log
tail -1 gpurun_out/fuzz_r6f_900.log gpurun_out/fuzz_r6f_variants_500.log gpurun_out/fuzz_worlds_r6f_20.log gpurun_out/fuzz_frame_world_r6f_30.log | cut -c1-200
