set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r02_gpu_tests5.log 2>&1 || { tail -40 gpurun_out/r02_gpu_tests5.log; exit 1; }
tail -2 gpurun_out/r02_gpu_tests5.log
timeout -k 10 400 python bench.py --steps 20 --warmup 5 > gpurun_out/r02_bench_driverflags2.log 2>&1 || { tail -20 gpurun_out/r02_bench_driverflags2.log; exit 1; }
python3 -c "
import json;d=json.loads(open('gpurun_out/r02_bench_driverflags2.log').read().strip().splitlines()[-1]);print('G', round(d['value']/1e9,1), 'ms', round(d['ms_per_step'],2), d['roofline']['kernel'], round(d['roofline']['frac'],3), {k:(round(v['ms_isolated'],2), round(v.get('frac',0),3)) for k,v in d['roofline']['kernels'].items()}, 'whole', round(d['roofline']['whole_job_frac'],3), 'cpu', round(d['cpu_baseline']['value']/1e6,1), d['cpu_baseline_all_cores']['cores'], round(d['cpu_baseline_all_cores']['value']/1e6,1))"
timeout -k 10 300 python bench.py --gpus 2 --share-gpu0 --backend gloo --steps 4 --warmup 1 --batch 8 --no-cpu-baseline --calib 1 > gpurun_out/r02_bench_2rank_weak.log 2>&1 || { tail -20 gpurun_out/r02_bench_2rank_weak.log; exit 1; }
tail -1 gpurun_out/r02_bench_2rank_weak.log | cut -c1-300
timeout -k 10 300 python bench.py --gpus 2 --share-gpu0 --backend gloo --config cfg5 --steps 4 --warmup 1 --batch 4 --no-cpu-baseline --calib 1 > gpurun_out/r02_bench_2rank_cfg5.log 2>&1 || { tail -20 gpurun_out/r02_bench_2rank_cfg5.log; exit 1; }
tail -1 gpurun_out/r02_bench_2rank_cfg5.log | cut -c1-300
