cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
CMD=$(python tools/frontend_ab.py --print-cmd)
ORBX_TRACE_MATCHER=1 ORBX_TRACE_BOW=1 ORBX_TRACE_EXTRACT=1 ORBX_TRACE_WINDOW=1 $CMD > /dev/null 2> gpurun_out/frontend_trace.txt
grep -c . gpurun_out/frontend_trace.txt; tail -60 gpurun_out/frontend_trace.txt
