#!/bin/bash
# One GPU iteration: extractor parity tests, then the bench line's per-kernel times (optionally for several env settings).
timeout 300 python -m pytest tests/test_gpu_extractor.py tests/test_golden.py -m gpu -x -q 2>&1 | tail -4
for envs in "$@"; do
  echo "== $envs"
  env $envs python bench.py --steps 10 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'], j['roofline']['kernels_ms_per_launch'])"
done
