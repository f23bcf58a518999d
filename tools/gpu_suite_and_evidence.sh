set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/r02_gpu_tests.log 2>&1 || { tail -30 gpurun_out/r02_gpu_tests.log; exit 1; }
tail -2 gpurun_out/r02_gpu_tests.log
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
bash tools/collect_evidence.sh
