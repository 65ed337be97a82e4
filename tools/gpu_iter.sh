cd $GRAFT_REPO_ROOT
( time timeout 420 python tools/fuzz_extractor.py 20000 900 ) > gpurun_out/fuzz_r4e_900.log 2>&1; grep "configurations" gpurun_out/fuzz_r4e_900.log; grep -c MISMATCH gpurun_out/fuzz_r4e_900.log
( time timeout 300 python tools/fuzz_extractor.py 30000 500 --variants ) > gpurun_out/fuzz_r4e_variants_500.log 2>&1; grep "configurations" gpurun_out/fuzz_r4e_variants_500.log; grep -c MISMATCH gpurun_out/fuzz_r4e_variants_500.log
( time timeout 200 python tools/fuzz_worlds.py 700 10 ) > gpurun_out/fuzz_worlds_r4e.log 2>&1; tail -4 gpurun_out/fuzz_worlds_r4e.log | head -1
( time timeout 200 python tools/fuzz_frame_world.py 60 20 ) > gpurun_out/fuzz_frame_world_r4e.log 2>&1; tail -4 gpurun_out/fuzz_frame_world_r4e.log | head -1
