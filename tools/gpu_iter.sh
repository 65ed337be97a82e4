cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_search.py tests/test_matcher_world.py tests/test_streamed_frontend.py tests/test_frame_world.py -x -q -m gpu 2>&1 | tail -12
STAMP=$(python -c "from orb_slam3_modified_amd.build import stamp; s = stamp(); print('commit', s['commit'], 'kernel sources', s['kernels_hash'], s['date'])")
{ echo "$STAMP"; echo "the per-frame Tracking sequence (tools/frontend_ab.py), alternating: pipelined searches (default) / ORBX_SEARCH_PIPELINE=0"
  for i in 1 2 3; do echo "pipelined: $(python tools/frontend_ab.py 2>/dev/null)"; echo "serial:    $(ORBX_SEARCH_PIPELINE=0 python tools/frontend_ab.py 2>/dev/null)"; done; } 2>&1 | tee gpurun_out/search_pipeline_ab.txt
CMD=$(python tools/frontend_ab.py --print-cmd)
ORBX_TRACE_MATCHER=1 $CMD 2>&1 >/dev/null | tail -6
