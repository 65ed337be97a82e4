cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_bench_contract.py -m gpu -x -q 2>&1 | grep -E "^E|assert|Error" | head -30
