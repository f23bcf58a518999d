set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_batch.py -m gpu -x -q > gpurun_out/r02_gpu_tests3.log 2>&1 || { tail -40 gpurun_out/r02_gpu_tests3.log; exit 1; }
tail -2 gpurun_out/r02_gpu_tests3.log
for args in "--streams 1 --calib 3 --steps 6 --warmup 3" "" "--hash gl --steps 12 --warmup 6"; do
timeout -k 10 300 python bench.py --steps 24 --warmup 12 --no-cpu-baseline $args > gpurun_out/r02_bench_c.log 2>&1 || { tail -20 gpurun_out/r02_bench_c.log; exit 1; }
python3 -c "
import json;d=json.loads(open('gpurun_out/r02_bench_c.log').read().strip().splitlines()[-1]);print('$args', 'G', round(d['value']/1e9,1), 'ms', round(d['ms_per_step'],3), 'iso', {k:round(v,2) for k,v in d['kernel_ms_isolated'].items()}, 'timed', {k:round(v,2) for k,v in d['kernel_ms_timed_region'].items()}, 'exp', round(d['roofline']['achieved']), d['roofline']['achieved_back_to_back'])"
done
