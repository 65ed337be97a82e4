# Scratch script for one gpurun call during development (overwritten freely): `gpurun -- 'bash tools/gpu_iter.sh'`.
# The reproducible end-of-round sequence is tools/final_refresh.sh.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
export ORBX_COMMIT=$(cat .commit_stamp 2>/dev/null)
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
bash tools/fast_passes_ab.sh 2>&1 | tee gpurun_out/fast_passes_ab.txt
