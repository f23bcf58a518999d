set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_batch.py -m gpu -x -q > gpurun_out/r02_gpu_tests7.log 2>&1 || { tail -40 gpurun_out/r02_gpu_tests7.log; exit 1; }
tail -2 gpurun_out/r02_gpu_tests7.log
for args in "" "--hash gl"; do
timeout -k 10 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline $args > gpurun_out/r02_bench_s.log 2>&1 || { tail -20 gpurun_out/r02_bench_s.log; exit 1; }
python3 -c "
import json;d=json.loads(open('gpurun_out/r02_bench_s.log').read().strip().splitlines()[-1]);print('$args', 'G', round(d['value']/1e9,1), 'ms/launch', round(d['ms_per_step']/12,3), 'iso', {k:round(v,2) for k,v in d['kernel_ms_isolated'].items()}, 'timed', {k:round(v,2) for k,v in d['kernel_ms_timed_region'].items()})"
done
