# Scratch script for one gpurun call during development (overwritten freely): `gpurun -- 'bash tools/gpu_iter.sh'`.
# The reproducible end-of-round sequence is tools/final_refresh.sh.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python tools/config3_full.py > gpurun_out/config3_full.txt 2> gpurun_out/config3_full.err; echo "config3 exit $?"
head -c 6000 gpurun_out/config3_full.txt | grep -v '^{'
timeout 300 python -m pytest tests/test_streamed_frontend.py -m gpu -x -q 2>&1 | tail -3
