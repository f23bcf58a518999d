set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -m gpu -x -q --durations=8 > gpurun_out/r02_gpu_tests6.log 2>&1 || { tail -60 gpurun_out/r02_gpu_tests6.log; exit 1; }
tail -14 gpurun_out/r02_gpu_tests6.log
