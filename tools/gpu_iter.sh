# Scratch script for one gpurun call during development (overwritten freely): `gpurun -- 'bash tools/gpu_iter.sh'`.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
export ORBX_COMMIT=$(cat .commit_stamp 2>/dev/null)
{ for KW in 0 2000 500 100; do
  echo "== ORBX_KEEP_WARM_US=$KW"
  python tools/config3_full.py --frames 700 --no-ref --env ORBX_KEEP_WARM_US=$KW 2>&1 | grep -E "^## |four_calls_ms|extract_ms|search_last_ms|^paced|rror"
done; } 2>&1 | tee gpurun_out/keep_warm.txt
