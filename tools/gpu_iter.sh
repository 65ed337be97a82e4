cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_extractor.py -x -q -m gpu -k "stage_parity or random_shapes or staging or other_param" 2>&1 | tail -2
b() { timeout 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-frontend --no-secondary 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
k=d['roofline']['kernels_ms_per_launch']
print('$1', d['value'], d['ms_per_step'], k['k_fast_cells'])"; }
b noearly; b noearly; b noearly
ORBX_EXTRA_FLAGS=-DORBX_FAST_EARLY_OPTION python -m orb_slam3_modified_amd.build --force > /dev/null 2>&1
b option; b option; b option
