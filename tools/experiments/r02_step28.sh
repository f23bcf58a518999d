set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -m pytest tests/test_gpu_batch.py tests/test_abi.py -x -q -k "comm or scheduling" 2>&1 | tail -4
