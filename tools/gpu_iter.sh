cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
ORBX_REPLAY_GRAPH=1 timeout 900 python -m pytest tests/test_gpu_replay.py -m gpu -x -q 2>&1 | tail -3
{ python -c "from orb_slam3_modified_amd.build import stamp; print(stamp())"
for rep in 1 2 3; do for G in 0 1; do
  echo "replay_graph $G (rep $rep): $(ORBX_REPLAY_GRAPH=$G python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-frontend --no-fixed-streams --no-secondary 2>/dev/null | tail -1 | python -c "
import sys,json; j=json.loads(sys.stdin.read()); print('step', j['ms_per_step'], 'min/max', j['timing']['ms_per_step_min'], j['timing']['ms_per_step_max'], 'value', j['value'], 'verified', j['verified_frames'])")"
done; done; } 2>&1 | tee gpurun_out/replay_graph_ab.txt
