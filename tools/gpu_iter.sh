cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_extractor.py tests/test_natural_images.py -x -q -m gpu -k "not exhaustive and not billion" 2>&1 | tail -3
timeout 200 python tools/fuzz_extractor.py 5000 80 2>&1 | tail -2
for r in 0 1; do
ORBX_QT_RANK=$r python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-frontend 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
e=d['end_to_end_operator']
print('rank $r', d['value'], d['roofline']['kernels_ms_per_launch']['k_quadtree'], 'nat', d['secondary_natural']['value'], d['secondary_natural']['kernels_ms_per_launch']['k_quadtree'], 'c4', d['secondary']['value'], 'e2e', e['ms_per_frame'], e['device_ms_per_frame'], e['ms_per_frame_without_host_pyramid'], e['device_ms_per_frame_without_host_pyramid'])"
done
