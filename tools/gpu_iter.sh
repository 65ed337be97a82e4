#!/bin/bash
# The short build -> measure loop on the GPU box (one gpurun call, ~2 minutes):
#   gpurun --timeout 900 -- 'bash tools/gpu_iter.sh'
# kernel times of a 256-frame batch (HIP events per kernel), the extractor parity tests, and two short headline runs with the natural-crop
# and config-4 legs.  Everything an experiment needs before it is worth a full tools/final_refresh.sh.
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
mkdir -p gpurun_out
python tools/kernel_times.py 256 2>/dev/null | tail -1
python -m pytest tests/test_gpu_extractor.py -m gpu -x -q 2>&1 | tail -2
for rep in 1 2; do
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-frontend --no-fixed-streams --no-gather 2>/dev/null | tail -1 | python -c "
import sys, json
j = json.loads(sys.stdin.read())
print('bench (rep $rep): step', j['ms_per_step'], 'value', j['value'], 'nat', j.get('secondary_natural', {}).get('value'), 'cfg4', j.get('secondary', {}).get('value'),
      j['roofline'].get('kernels_ms_per_launch'))"
done
