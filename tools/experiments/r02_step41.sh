set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() { timeout -k 10 300 python bench.py --no-cpu-baseline --steps 20 --warmup 5 --calib 0 --hash gl $1 > gpurun_out/r02_bench_s.log 2>&1 || { tail -20 gpurun_out/r02_bench_s.log; exit 1; }
python3 -c "
import json;d=json.loads(open('gpurun_out/r02_bench_s.log').read().strip().splitlines()[-1]);print('$1', 'G', round(d['value']/1e9,1), 'timed', {k:round(v,2) for k,v in d['kernel_ms_timed_region'].items()}, d['expand_schedule_timed_region']['expand_end_to_next_expand_start_ms'])"; }
for rep in 1 2; do
run ""
run "--batch 3 --streams 5"
run "--batch 2 --streams 7"
run "--batch 3 --streams 4"
done
