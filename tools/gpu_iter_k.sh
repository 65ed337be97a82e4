#!/bin/bash
timeout 300 python -m pytest tests/test_gpu_extractor.py tests/test_golden.py -m gpu -x -q 2>&1 | tail -3
for envs in "ORBX_DESC_K=4 ORBX_LANES=1" "ORBX_DESC_K=8 ORBX_LANES=1" "ORBX_DESC_K=2 ORBX_LANES=1" "ORBX_DESC_K=16 ORBX_LANES=1" "ORBX_DESC_K=4 ORBX_LANES=2" "ORBX_DESC_K=8 ORBX_LANES=2"; do
  echo "== $envs"
  env $envs python bench.py --steps 40 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'], j['roofline']['kernels_ms_per_launch'])"
done
