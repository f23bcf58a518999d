set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for cfg in cfg2 cfg5 cfg1; do
timeout -k 10 300 python bench.py --no-cpu-baseline --steps 20 --warmup 5 --config $cfg > gpurun_out/r02_bench_$cfg.log 2>&1 || { tail -20 gpurun_out/r02_bench_$cfg.log; exit 1; }
tail -1 gpurun_out/r02_bench_$cfg.log | cut -c1-160
done
timeout -k 10 400 python bench.py --no-cpu-baseline --steps 100 --warmup 5 --calib 0 > gpurun_out/r02_bench_soak.log 2>&1 || { tail -20 gpurun_out/r02_bench_soak.log; exit 1; }
tail -1 gpurun_out/r02_bench_soak.log | cut -c1-160
for i in 1 2 3; do timeout -k 10 300 python bench.py --no-cpu-baseline --steps 20 --warmup 5 --calib 0 > gpurun_out/r02_bench_s.log 2>&1; tail -1 gpurun_out/r02_bench_s.log | cut -c1-110; done
for i in 1 2; do timeout -k 10 300 python bench.py --no-cpu-baseline --steps 20 --warmup 5 --calib 0 --hash gl > gpurun_out/r02_bench_s.log 2>&1; tail -1 gpurun_out/r02_bench_s.log | cut -c1-110; done
bash tools/rehearse_two_ranks.sh 2>&1 | tail -4 | cut -c1-200
