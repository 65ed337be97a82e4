cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_extractor.py tests/test_opencv_variants.py -x -q -m gpu -k "stage_parity or random_shapes or blur or other_param" 2>&1 | tail -2
timeout 300 python tools/fuzz_extractor.py 13000 100 2>&1 | tail -1
b() { timeout 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-frontend --no-secondary 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
k=d['roofline']['kernels_ms_per_launch']
print('$1', d['value'], d['ms_per_step'], k['k_blur7'])"; }
b blurdma; b blurdma; b blurdma
python -m orb_slam3_modified_amd.build --force > /dev/null 2>&1
b plain; b plain; b plain
