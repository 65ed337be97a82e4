# Scratch script for one gpurun call during development (overwritten freely): `gpurun -- 'bash tools/gpu_iter.sh'`.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
python -c "
import ctypes
h=ctypes.CDLL('libamdhip64.so'); lo=ctypes.c_int(); hi=ctypes.c_int(); print('priority range rc', h.hipDeviceGetStreamPriorityRange(ctypes.byref(lo), ctypes.byref(hi)), lo.value, hi.value)"
run() { echo "$1: $(env $1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-frontend --no-fixed-streams --no-secondary 2>/dev/null | tail -1 | python -c "
import sys,json; j=json.loads(sys.stdin.read()); print('step', j['ms_per_step'], 'min/max', j['timing']['ms_per_step_min'], j['timing']['ms_per_step_max'], 'value', j['value'])")"; }
{ for rep in 1 2; do
for E in "ORBX_AUX_PRIO=0,0,0" "ORBX_AUX_PRIO=1,0,0" "ORBX_AUX_PRIO=0,0,-1" "ORBX_AUX_PRIO=1,0,-1" "ORBX_AUX_PRIO=1,-1,-1" "ORBX_AUX_PRIO=-1,0,0" "ORBX_LANE_PRIO=-1" "ORBX_LANE_PRIO=1"; do run "$E"; done; done; } 2>&1 | tee gpurun_out/prio_ab.txt
