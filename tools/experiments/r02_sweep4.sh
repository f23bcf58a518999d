set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
: > gpurun_out/r02_sweep4.txt
for args in "--hash gl --batch 3 --streams 4" "--hash gl --batch 2 --streams 6" "--hash gl --batch 4 --streams 3" "--hash gl --batch 6 --streams 2" "--hash gl --batch 4 --streams 4 --advice-cap-gb 240"; do
timeout -k 10 200 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --calib 0 $args > gpurun_out/r02_bench_s.log 2>&1 || { tail -20 gpurun_out/r02_bench_s.log; exit 1; }
python3 -c "
import json;d=json.loads(open('gpurun_out/r02_bench_s.log').read().strip().splitlines()[-1]);print('$args', 'G', round(d['value']/1e9,1), 'ms/launch', round(d['ms_per_step']/12,3), 'S', d['config']['launches_in_flight'], 'timed', {k:round(v,2) for k,v in d['kernel_ms_timed_region'].items()})" | tee -a gpurun_out/r02_sweep4.txt
done
