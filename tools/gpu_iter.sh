cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_extractor.py -x -q -m gpu -k "stage_parity or random_shapes or strided or other_param or launch_shapes" 2>&1 | tail -2
timeout 300 python tools/fuzz_extractor.py 12000 120 2>&1 | tail -1
b() { timeout 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-frontend --no-secondary 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
k=d['roofline']['kernels_ms_per_launch']
print('$1', d['value'], d['ms_per_step'], k['k_describe'], k['k_fast_cells'])"; }
b dma; b dma; b dma
ORBX_EXTRA_FLAGS=-DORBX_DESC_NO_DMA python -m orb_slam3_modified_amd.build --force > /dev/null 2>&1
b nodma; b nodma; b nodma
