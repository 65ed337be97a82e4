cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python tools/fuzz_extractor.py 5000 200 2>&1 | tail -2 | tee gpurun_out/fuzz_r5_5000_200.log
