#!/bin/bash
# A/B of experiment builds of the library (ORBX_BUILD_OUT=gpurun_exp/lib_<name>.so ORBX_EXTRA_FLAGS=... python -m orb_slam3_modified_amd.build) against the
# product build on ONE box: bench.py --steps 20 --repeats 5, alternating; usage: bash tools/lib_ab.sh [reps] lib_a.so lib_b.so ...
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
REPS=$1; shift
run() {
  local label="$1"; shift
  env "$@" python bench.py --steps 20 --warmup 5 --repeats 5 --no-cpu-baseline --no-frontend --no-fixed-streams --no-gather 2>/dev/null | tail -1 | python -c "
import sys, json
j = json.loads(sys.stdin.read())
print('%-28s step %.4f (min %.4f max %.4f)  natural %.4f  config4 %.4f  verified %s  %s' % ('$label', j['ms_per_step'], j['timing']['ms_per_step_min'], j['timing']['ms_per_step_max'],
      j.get('secondary_natural', {}).get('ms_per_step', float('nan')), j.get('secondary', {}).get('ms_per_step', float('nan')), j.get('verified_frames'), j['roofline'].get('kernels_ms_per_launch')))"
}
for rep in $(seq 1 $REPS); do
  run "product" X=1
  for lib in "$@"; do run "$(basename $lib .so)" ORBX_LIB=$PWD/$lib; done
done
