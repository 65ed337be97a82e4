cd $GRAFT_REPO_ROOT
( time timeout 500 python tools/fuzz_extractor.py 7000 220 ) > gpurun_out/fuzz_r4c_plain.log 2>&1; tail -4 gpurun_out/fuzz_r4c_plain.log
( time timeout 500 python tools/fuzz_extractor.py 8000 220 --variants ) > gpurun_out/fuzz_r4c_variants.log 2>&1; tail -4 gpurun_out/fuzz_r4c_variants.log
( time timeout 600 python tools/fuzz_worlds.py 400 14 ) > gpurun_out/fuzz_worlds_r4c.log 2>&1; tail -5 gpurun_out/fuzz_worlds_r4c.log
( time timeout 400 python tools/fuzz_frame_world.py 30 12 ) > gpurun_out/fuzz_frame_world_r4c.log 2>&1; tail -5 gpurun_out/fuzz_frame_world_r4c.log
