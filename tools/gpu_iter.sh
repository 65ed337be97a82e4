cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_extractor.py -x -q -m gpu -k "staging_variants" 2>&1 | tail -5
