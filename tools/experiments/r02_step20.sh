set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 240 python -m pytest tests/test_gpu_batch.py -x -q -m gpu -k "scheduling" 2>&1 | tail -5
