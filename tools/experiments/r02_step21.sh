set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for args in "--batch 32 --steps 2 --warmup 1 --calib 3 --pair-chains 1" "--batch 32 --steps 2 --warmup 1 --calib 3 --pair-chains 0" "--steps 20 --warmup 5 --calib 3 --pair-chains 1" "--steps 20 --warmup 5 --calib 3 --pair-chains 0" "--steps 20 --warmup 5 --calib 0 --pair-chains 1" ; do
timeout -k 10 300 python bench.py --no-cpu-baseline $args > gpurun_out/r02_bench_s.log 2>&1 || { tail -20 gpurun_out/r02_bench_s.log; exit 1; }
python3 -c "
import json;d=json.loads(open('gpurun_out/r02_bench_s.log').read().strip().splitlines()[-1]);print('$args', 'G', round(d['value']/1e9,1), 'ms/launch', round(d['ms_per_step']/12,3), 'timed', {k:round(v,2) for k,v in d['kernel_ms_timed_region'].items()}, 'isolated', {k:round(v,2) for k,v in (d.get('kernel_ms_isolated') or {}).items()})"
done
