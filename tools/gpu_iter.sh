cd $GRAFT_REPO_ROOT
CMD=$(python tools/frontend_ab.py --print-cmd 2>/dev/null | tail -1)
ORBX_TRACE_MATCHER=1 $CMD 2> gpurun_out/trace_matcher.txt | tail -1
grep -c . gpurun_out/trace_matcher.txt
tail -40 gpurun_out/trace_matcher.txt | cut -c1-200
