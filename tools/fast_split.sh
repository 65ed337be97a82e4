#!/bin/bash
# Timing experiment: where does k_fast_cells' time go?  Build two variants HERE (the up-to-date check of build.py ignores flag
# changes, and the variants must travel to the GPU box prebuilt):
#   cd orb_slam3_modified_amd/csrc && for V in A B; do hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared \
#     -DORBX_FAST_STOP_AFTER_$V -o ../../tools/var_$V.so orbx_extractor.hip orbx_matcher.hip orbx_search.hip orbx_kfdb.hip; done
# then on the GPU:  bash tools/fast_split.sh      (r1: whole kernel 0.470 ms, stops after B 0.292, after A 0.160 per 256 frames)
for V in "" tools/var_B.so tools/var_A.so; do
  echo "== lib: ${V:-default}"
  if [ -n "$V" ]; then export ORBX_LIB=$PWD/$V; else unset ORBX_LIB; fi
  python bench.py --no-cpu-baseline --lanes 1 --steps 10 2>&1 | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['roofline']['kernels_ms_per_launch'])"
done
