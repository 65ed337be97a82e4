cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_replay.py tests/test_bench_contract.py -m gpu -x -q 2>&1 | grep -E "passed|failed|FAILED|^E " | head -20
ORBX_REPLAY_ALTERNATE=0 timeout 900 python -m pytest tests/test_gpu_replay.py -m gpu -x -q -k "not alternate" 2>&1 | grep -E "passed|failed|FAILED|^E " | head
