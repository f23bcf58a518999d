set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for args in "" "--serial-expand 0" "--hash gl" "--hash gl --serial-expand 1"; do
timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --calib 0 $args > gpurun_out/r02_bench_s.log 2>&1 || { tail -20 gpurun_out/r02_bench_s.log; exit 1; }
python3 -c "
import json;d=json.loads(open('gpurun_out/r02_bench_s.log').read().strip().splitlines()[-1]);print('$args', 'G', round(d['value']/1e9,1), 'ms/launch', round(d['ms_per_step']/12,3), 'timed', {k:round(v,2) for k,v in d['kernel_ms_timed_region'].items()}, d['expand_schedule_timed_region'])"
done
