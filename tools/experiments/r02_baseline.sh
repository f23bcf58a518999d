set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
hipcc -O3 --offload-arch=gfx950 tools/ubench_alu.hip -o /tmp/ubench_alu && timeout -k 10 120 /tmp/ubench_alu > gpurun_out/r02_ubench_alu.txt 2>&1; cat gpurun_out/r02_ubench_alu.txt
timeout -k 10 400 python bench.py --steps 20 --warmup 5 > gpurun_out/r02_bench_driverflags.log 2>&1; tail -1 gpurun_out/r02_bench_driverflags.log | cut -c1-600
