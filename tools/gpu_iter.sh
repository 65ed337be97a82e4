cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
rocprofv3 --kernel-trace -d gpurun_out/t1 -o t -- python tools/kernel_times.py 64 > /dev/null 2>&1
DB=$(ls gpurun_out/t1/*/t_results.db gpurun_out/t1/t_results.db 2>/dev/null | head -1)
python - <<PY
import sqlite3
c=sqlite3.connect("$DB")
print([r[1] for r in c.execute("pragma table_info(kernels)")])
print(c.execute("select * from kernels limit 1").fetchall())
PY
rm -rf gpurun_out/t1
for rep in 1 2; do for L in 2 3 4; do
  echo "alternate NOWAIT lanes $L (rep $rep): $(ORBX_REPLAY_NOWAIT=1 python bench.py --steps 20 --warmup 5 --lanes $L --no-verify --no-cpu-baseline --no-frontend --no-fixed-streams --no-secondary --no-gather 2>/dev/null | tail -1 | python -c "
import sys,json; j=json.loads(sys.stdin.read()); print('step', j['ms_per_step'], 'min/max', j['timing']['ms_per_step_min'], j['timing']['ms_per_step_max'], 'value', j['value'])")"
done; done
