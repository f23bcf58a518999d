set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_batch.py tests/test_gpu_variants.py tests/test_gpu_chips.py -x -q -m gpu 2>&1 | tail -3
run() { timeout -k 10 300 python bench.py --no-cpu-baseline --steps 20 --warmup 5 --calib 3 $1 > gpurun_out/r02_bench_s.log 2>&1 || { tail -20 gpurun_out/r02_bench_s.log; exit 1; }
python3 -c "
import json;d=json.loads(open('gpurun_out/r02_bench_s.log').read().strip().splitlines()[-1]);print('$1', 'G', round(d['value']/1e9,1), 'timed', {k:round(v,2) for k,v in d['kernel_ms_timed_region'].items()}, 'iso', {k:round(v,2) for k,v in d['kernel_ms_isolated'].items()}, 'b2b', round(d['roofline']['expand_back_to_back_GBps']))"; }
run ""
run "--hash gl"
run "--batch 32 --steps 2 --warmup 1"
