cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_bench_contract.py -x -q -m gpu -k two_ranks 2>&1 | tail -30
