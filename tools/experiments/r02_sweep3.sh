set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
: > gpurun_out/r02_sweep3.txt
for args in "--batch 64 --streams 3" "--batch 48 --streams 4" "--batch 40 --streams 5" "--batch 32 --streams 6" "--batch 56 --streams 3" "--batch 80 --streams 2" "--batch 64 --streams 3 --proofs random"; do
timeout -k 10 200 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --calib 0 $args > gpurun_out/r02_bench_s.log 2>&1 || { tail -20 gpurun_out/r02_bench_s.log; exit 1; }
python3 -c "
import json;d=json.loads(open('gpurun_out/r02_bench_s.log').read().strip().splitlines()[-1]);print('$args', 'G', round(d['value']/1e9,1), 'ms/launch', round(d['ms_per_step']/12,3), 'timed', {k:round(v,2) for k,v in d['kernel_ms_timed_region'].items()})" | tee -a gpurun_out/r02_sweep3.txt
done
