#!/bin/bash
# The launch-shape options that were last measured under older kernels / the split lane schedule, re-measured under the current ones
# (alternate lanes, the Gaussian inside the descriptor kernel): every configuration twice, alternating with the default, one box.
#   gpurun --timeout 900 -- 'bash tools/option_sweep.sh > gpurun_out/option_sweep.txt 2>&1'
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
python -c "from orb_slam3_modified_amd.build import stamp; print(stamp())"
run() {  # label, extra bench args, env assignments...
  local label="$1" extra="$2"; shift 2
  env "$@" python bench.py --steps 20 --warmup 5 --repeats 5 --no-cpu-baseline --no-frontend --no-fixed-streams --no-gather --no-verify $extra 2>/dev/null | tail -1 | python -c "
import sys, json
j = json.loads(sys.stdin.read())
print('%-44s step %.4f (min %.4f max %.4f)  natural %.4f  config4 %.4f' % ('$label', j['ms_per_step'], j['timing']['ms_per_step_min'], j['timing']['ms_per_step_max'],
      j.get('secondary_natural', {}).get('ms_per_step', float('nan')), j.get('secondary', {}).get('ms_per_step', float('nan'))))"
}
for rep in 1; do
  run "default" "" X=1
  run "chain_batch=1" "" ORBX_CHAIN_BATCH=1
  run "chain_batch=1 chain_threads=256" "" ORBX_CHAIN_BATCH=1 ORBX_CHAIN_THREADS=256
  run "default" "" X=1
  run "fork_fast0=0" "" ORBX_FORK_FAST0=0
  run "fork_qt=1" "" ORBX_FORK_QT=1
  run "qt_level_major=0" "" ORBX_QT_LEVEL_MAJOR=0
  run "default" "" X=1
  run "fast_threads=256" "" ORBX_FAST_THREADS=256
  run "fast_threads=64" "" ORBX_FAST_THREADS=64
  run "fast_split=0" "" ORBX_FAST_SPLIT=0
  run "lanes 3" "--lanes 3" X=1
done
