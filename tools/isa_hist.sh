#!/bin/bash
# usage: tools/isa_hist.sh <kernel-name-substring> [top-n]  — instruction histogram of one gfx950 kernel
set -e
D=/tmp/isa; mkdir -p $D
ROOT=$(cd "$(dirname "$0")/.." && pwd)
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -S --cuda-device-only ${ORBX_DEFS} -I$ROOT/orb_slam3_modified_amd/csrc $ROOT/orb_slam3_modified_amd/csrc/orbx_extractor.hip -o $D/ext.s 2>/dev/null
awk -v k="$1" '$0 ~ "^_Z.*"k".*:" {on=1} on {print} on && /s_endpgm/ {exit}' $D/ext.s > $D/k.s
echo "lines: $(wc -l < $D/k.s)"
grep -E "^\s+[a-z]" $D/k.s | awk '{print $1}' | sort | uniq -c | sort -rn | head -${2:-40}
grep -E "vgpr_count|sgpr_count|lds_size|scratch" $D/ext.s | head -0
