cd $GRAFT_REPO_ROOT
STAMP=$(python -c "from orb_slam3_modified_amd.build import stamp; s = stamp(); print('commit', s['commit'], 'kernel sources', s['kernels_hash'], s['date'])")
{ echo "$STAMP"; timeout 120 python tools/target_latency.py 2>&1 | tail -25; } > gpurun_out/target_latency_r4.txt
{ echo "$STAMP"; timeout 120 python tools/stereo_latency.py 2>&1 | tail -15; } > gpurun_out/stereo_latency_r4.txt
tail -12 gpurun_out/target_latency_r4.txt; tail -6 gpurun_out/stereo_latency_r4.txt
