cd $GRAFT_REPO_ROOT
timeout 300 python tools/fuzz_extractor.py 9000 80 2>&1 | tail -4
timeout 300 python tools/fuzz_extractor.py 9500 200 2>&1 | tail -2
ORBX_FAST_DMA=3 timeout 300 python tools/fuzz_extractor.py 9000 80 2>&1 | tail -2
