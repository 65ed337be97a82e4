cd $GRAFT_REPO_ROOT
timeout 240 python tools/rccl_two_ranks_one_gpu.py 2>&1 | tail -25
