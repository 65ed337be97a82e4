cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_extractor.py tests/test_natural_images.py -x -q -m gpu -k "not exhaustive and not billion and not capacities" 2>&1 | tail -2
timeout 300 python tools/fuzz_extractor.py 14000 100 2>&1 | tail -1
for rep in 1 2 3; do for c in 0 1; do
ORBX_DESC_MASK_TABLE=$c timeout 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-frontend --no-secondary 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
k=d['roofline']['kernels_ms_per_launch']
print('mask_table $c', d['value'], d['ms_per_step'], k['k_describe'])"
done; done
