#!/bin/bash
# Timing experiment: where does k_fast_cells' time go?  ORBX_FAST_STOP=1 returns after the tile is staged, =2 after the
# necessary test (results are void then: bench.py's verification is skipped by --no-verify).
#   r2 (256 distinct frames per launch): whole kernel 0.535 ms, stops after B 0.299, after A 0.165
for E in "${@:-ORBX_LANES=1}"; do for S in 0 2 1; do
  echo -n "$E stop=$S: "
  env $E ORBX_FAST_STOP=$S python bench.py --no-cpu-baseline --no-secondary --no-verify --lanes 1 --steps 10 2>/dev/null | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['roofline']['kernels_ms_per_launch']['k_fast_cells'], j['ms_per_step'])"
done; done
