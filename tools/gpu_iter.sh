cd $GRAFT_REPO_ROOT
for l in 2 3 4 2 3; do
timeout 200 python bench.py --lanes $l --steps 30 --warmup 5 --no-cpu-baseline --no-frontend --no-secondary 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('lanes $l', d['value'], d['ms_per_step'])"
done
for e in "ORBX_FORK_BLUR=0" "ORBX_FORK_FAST0=0" "ORBX_FORK_QT=0" "ORBX_CHAIN_BATCH=1" "ORBX_QT_ONE_LAUNCH=1"; do
env $e timeout 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-frontend --no-secondary 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$e', d['value'], d['ms_per_step'])"
done
