set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
hipcc -O3 --offload-arch=gfx950 tools/ubench_store6.hip -o /tmp/ubench_store6 2>/dev/null && timeout -k 10 200 /tmp/ubench_store6 > gpurun_out/r02_ubench_store6.txt 2>&1; cat gpurun_out/r02_ubench_store6.txt
