cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_gpu_extractor.py -x -q -m gpu -k "odd_row_strides or strided or staging" 2>&1 | tail -3
for c in 1240 1241 1226; do
timeout 200 python bench.py --rows 376 --cols $c --batch 128 --steps 10 --warmup 3 --no-cpu-baseline --no-frontend --no-secondary 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('cols $c', d['value'], d['ms_per_step'])"
done
