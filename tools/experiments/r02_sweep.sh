set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
: > gpurun_out/r02_sweep.txt
for args in "--streams 6" "--streams 6 --no-fork" "--streams 3" "--streams 3 --no-fork" "--streams 4" "--streams 8" "--streams 8 --no-fork" "--batch 64 --streams 3" "--batch 16 --streams 8"; do
timeout -k 10 200 python bench.py --steps 36 --warmup 12 --no-cpu-baseline --calib 0 $args > gpurun_out/r02_bench_s.log 2>&1 || { tail -20 gpurun_out/r02_bench_s.log; exit 1; }
python3 -c "
import json;d=json.loads(open('gpurun_out/r02_bench_s.log').read().strip().splitlines()[-1]);print('$args', 'G', round(d['value']/1e9,1), 'ms', round(d['ms_per_step'],3), 'timed', {k:round(v,2) for k,v in d['kernel_ms_timed_region'].items()})" | tee -a gpurun_out/r02_sweep.txt
done
