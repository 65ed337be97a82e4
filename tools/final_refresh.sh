#!/bin/bash
# End-of-round evidence refresh on the GPU box: everything lands in gpurun_out/ (merged back), then copied into profiles/.
# usage (from the repo root on the box): bash tools/final_refresh.sh r6a <commit>
TAG=${1:-r6x}
export ORBX_COMMIT=${2:-unknown}      # the GPU box has no .git: the caller passes `git rev-parse --short=12 HEAD`; every evidence file is stamped with it
mkdir -p gpurun_out
export TMPDIR=/tmp
STAMP=$(python -c "from orb_slam3_modified_amd.build import stamp; s = stamp(); print('commit', s['commit'], 'kernel sources', s['kernels_hash'], s['date'])")
echo "$STAMP"
timeout 1200 python -m pytest tests -m gpu -q > gpurun_out/pytest_gpu_$TAG.log 2>&1; grep -E "passed|failed" gpurun_out/pytest_gpu_$TAG.log | tail -1
# HBM traffic (separate FETCH_SIZE / WRITE_SIZE passes) and issue-side counters, full-batch launches
timeout 600 python tools/pmc_traffic.py > /dev/null 2>&1 && cp gpurun_out/pmc_traffic.json profiles/pmc_traffic.json
timeout 900 python tools/pmc_sq.py "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS" \
   "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT TA_BUSY_avr" > /dev/null 2>&1
[ -f gpurun_out/pmc_sq.json ] && cp gpurun_out/pmc_sq.json profiles/pmc_sq.json
# kernel trace of the bench command FIRST (without the CPU leg): the bench line below cites this summary (roofline.rocprof) when it carries this build's hash
# (--no-secondary: the natural-crop leg launches the same grids as the headline's roofline passes — 256 frames of 640x480 — on slower input and would
# sit in the same by-grid rows)
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$TAG -o $TAG -- python bench.py --no-cpu-baseline --no-secondary > gpurun_out/prof_$TAG.log 2>&1
DB=$(ls gpurun_out/prof_$TAG/*/${TAG}_results.db gpurun_out/prof_$TAG/${TAG}_results.db 2>/dev/null | head -1)
{ python tools/rocpd_summary.py "$DB" --title "rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline --no-secondary ($TAG): all launches";
  echo; python tools/rocpd_summary.py "$DB" --by-grid --title "the same run, one row per launch shape (timed region: 2 lanes x 128 frames, overlapping; roofline passes: 256 frames back to back, 8 timed + one warm-up of 248 frames)"; } > gpurun_out/${TAG}_kernel_stats.md
sed -i "1i $STAMP\n" gpurun_out/${TAG}_kernel_stats.md
cp gpurun_out/${TAG}_kernel_stats.md profiles/${TAG}_kernel_stats.md
head -12 gpurun_out/${TAG}_kernel_stats.md
rm -rf gpurun_out/prof_$TAG/*/*.db gpurun_out/prof_$TAG/*.db   # keep the merge-back small
# the bench line (reads profiles/pmc_*.json and profiles/*_kernel_stats.md refreshed above)
python bench.py 2> gpurun_out/bench_$TAG.err | tail -1 > gpurun_out/bench_$TAG.json; cat gpurun_out/bench_$TAG.json | cut -c1-600
# the device timeline of one single-frame operator()
rocprofv3 --kernel-trace -d gpurun_out/tl_$TAG -o tl -- python tools/one_frame_trace.py 100 > gpurun_out/tl_$TAG.log 2>&1
TDB=$(ls gpurun_out/tl_$TAG/*/tl_results.db gpurun_out/tl_$TAG/tl_results.db 2>/dev/null | head -1)
{ echo "$STAMP"; grep "ms/frame" gpurun_out/tl_$TAG.log; python tools/frame_timeline.py "$TDB" 12; } > gpurun_out/frame_timeline_$TAG.txt 2>&1; tail -16 gpurun_out/frame_timeline_$TAG.txt
rm -rf gpurun_out/tl_$TAG
timeout 300 python tools/bench_aux.py > /dev/null 2>&1; ls -la gpurun_out/bench_aux.json
{ echo "$STAMP"; echo; timeout 300 python tools/matcher_times.py 2>&1 | grep -v "^\["; } > gpurun_out/matcher_times_$TAG.md; tail -2 gpurun_out/matcher_times_$TAG.md
{ echo "$STAMP"; for i in 1 2; do python tools/frontend_ab.py 2>/dev/null; ORBX_BOW_IN_GRAPH=0 python tools/frontend_ab.py 2>/dev/null; done; } > gpurun_out/frontend_ab_$TAG.txt 2>&1; cat gpurun_out/frontend_ab_$TAG.txt
{ echo "$STAMP"; python tools/target_latency.py 2>&1 | tail -8; python tools/stereo_latency.py 2>&1 | tail -3; } > gpurun_out/latency_$TAG.txt; tail -6 gpurun_out/latency_$TAG.txt
timeout 400 python tools/fuzz_extractor.py 400 150 2>&1 | tail -3 | tee gpurun_out/fuzz_$TAG.log
timeout 500 python tools/fuzz_extractor.py 900 200 --variants 2>&1 | tail -3 | tee gpurun_out/fuzz_${TAG}_variants.log
timeout 600 python tools/fuzz_worlds.py 700 8 2>&1 | tail -4 | tee gpurun_out/fuzz_worlds_$TAG.log
timeout 600 python tools/fuzz_frame_world.py 201 10 2>&1 | tail -4 | tee gpurun_out/fuzz_frame_world_$TAG.log
# BASELINE config 3 at length: 3 682 frames back to back, paced at MH_01's 20 Hz (sleeping, and busy-waiting: whose idle state is the paced tail?), and the
# reference-compiled loop; digests over ALL frames
timeout 1500 python tools/config3_full.py --spin > gpurun_out/config3_full_$TAG.txt 2> gpurun_out/config3_full_$TAG.err; echo "config3 exit $?"; grep -E "^digest|^paced" gpurun_out/config3_full_$TAG.txt
