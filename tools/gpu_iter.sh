cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
{ python -c "from orb_slam3_modified_amd.build import stamp; print(stamp())"
for rep in 1 2; do for B in 256 512 1024 2048; do
  echo "batch $B lanes 2 (rep $rep): $(python bench.py --batch $B --batches 2 --steps $((5120 / B)) --warmup 3 --no-cpu-baseline --no-frontend --no-fixed-streams --no-secondary 2>/dev/null | tail -1 | python -c "
import sys,json; j=json.loads(sys.stdin.read()); print('step', j['ms_per_step'], 'ms per 256 frames', round(j['ms_per_step']*256/j['config']['frames_per_step_per_gpu'],4), 'value', j['value'], 'spread', j['timing']['spread_frac'])")"
done; done
for L in 1 3 4; do
  echo "batch 1024 lanes $L: $(python bench.py --batch 1024 --batches 2 --steps 5 --warmup 3 --lanes $L --no-cpu-baseline --no-frontend --no-fixed-streams --no-secondary 2>/dev/null | tail -1 | python -c "
import sys,json; j=json.loads(sys.stdin.read()); print('step', j['ms_per_step'], 'ms per 256 frames', round(j['ms_per_step']*256/j['config']['frames_per_step_per_gpu'],4), 'value', j['value'])")"
done; } 2>&1 | tee gpurun_out/batch_size_curve.txt
